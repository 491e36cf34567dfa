"""A BASELINE config (worldgen.PRESETS: cfg2, cfg3, cfg4, cfg5) on one GPU: per-kernel HIP-event times and the step rate.
Run under `rocprofv3 --kernel-trace --stats` for the committed kernel-stats sets of the configs that are not the headline.

    python tools/cfg_probe.py cfg5 [--robots N] [--steps K]

cfg4 runs one GPU's share (8192 robots unless --robots says otherwise; its 200 social-force pedestrians ignore the robots and
stay inside libpedsim's 10 m root square, as in tests/test_gpu_parity.py)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img_env_amd import worldgen  # noqa: E402
from img_env_amd.world import World  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("cfg", choices=sorted(worldgen.PRESETS))
ap.add_argument("--robots", type=int, default=None)
ap.add_argument("--steps", type=int, default=100)
args = ap.parse_args()
c = dict(worldgen.PRESETS[args.cfg])
R = args.robots or min(c["n_robots"], 8192)
P = c["n_peds"]
over = {}
clearance = c["clearance"]
if args.cfg == "cfg4":
    over["relation_ped_robo"] = 0
    clearance = 0.5
grid = worldgen.make_grid(c["grid"], 0)
layout = worldgen.make_layout(grid, c["res"], R, P, seed=100, clearance=clearance)
if args.cfg == "cfg4":
    rng = np.random.default_rng(13)
    layout.ped_pose[:, :2] = rng.uniform(0.5, 9.5, (P, 2))
    layout.ped_traj[:, :, :2] = rng.uniform(0.5, 9.5, layout.ped_traj[:, :, :2].shape)
    layout.ped_goal[:] = rng.uniform(0.5, 9.5, (P, 2))
# (time_max far away: the probe never resets)
w = World(worldgen.make_params(R, P, res=c["res"], view_cells=c["view_cells"], beams=c["beams"], scene=c["scene"], time_max=10 ** 7, **over), grid)
w.reset(layout)
a = torch.zeros(R, 3, device="cuda")
a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
for s in range(300):  # clock ramp
    w.step(a, actions_ready=True)  # (pre-generated, resident in HBM)
w.timing(1)
prev = w.timing_read()
samples = {k: [] for k in prev}
for s in range(20):
    w.step(a, actions_ready=True)  # (pre-generated, resident in HBM)
    cur = w.timing_read()
    for k in cur:
        if cur[k][1] > prev[k][1]:
            samples[k].append(1e3 * (cur[k][0] - prev[k][0]) / (cur[k][1] - prev[k][1]))
    prev = cur
w.timing(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(args.steps):
    w.step(a, actions_ready=True)  # (pre-generated, resident in HBM)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
print(json.dumps(dict(config=args.cfg, robots=R, peds=P, grid=c["grid"], resolution=c["res"], view=c["view_cells"], beams=c["beams"],
                      scene=c["scene"], policy="active (v = 0)", us_per_step=1e6 * dt, value=R / dt, unit="robot-steps/s",
                      kernel_us={k: round(float(np.median(v)), 2) for k, v in samples.items() if v})))
w.close()
