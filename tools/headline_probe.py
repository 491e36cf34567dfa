"""The headline world (cfg-3: 8192 robots, 200 ORCA pedestrians, 400x400 grid @0.25 m, 48x48 views, 360 beams) with creation flags,
e.g. 2 / 4 = class layer composed / stamped whatever the size rule says.   usage (GPU box): python tools/headline_probe.py [flags] [plain]
(`plain`: stream-ordered actions, i.e. imgenv_step without IMGENV_STEP_ACTIONS_READY -- what a trainer whose policy writes them on the stream calls)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img_env_amd import worldgen  # noqa: E402
from img_env_amd.world import World  # noqa: E402

FLAGS = int(sys.argv[1]) if len(sys.argv) > 1 else 0
READY = not (len(sys.argv) > 2 and sys.argv[2] == "plain")
R, P = 8192, 200
grid = worldgen.make_grid(400, 0)
layout = worldgen.make_layout(grid, 0.25, R, P, seed=100, clearance=0.7)
# (time_max far away: the probe never resets)
w = World(dict(worldgen.make_params(R, P, res=0.25, view_cells=48, beams=360, scene="rvoscene", time_max=10 ** 7), flags=FLAGS), grid)
w.reset(layout)
a = torch.zeros(R, 3, device="cuda")
a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
for s in range(2000):
    w.step(a, actions_ready=READY)  # (pre-generated, resident in HBM)
w.timing(1)
for s in range(40):
    w.step(a, actions_ready=READY)  # (pre-generated, resident in HBM)
tm = w.timing_read()
w.timing(0)
print(" ".join("%s %.1f" % (k[2:], 1e3 * ms / n) for k, (ms, n) in tm.items() if n))
res = []
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(200):
        w.step(a, actions_ready=READY)  # (pre-generated, resident in HBM)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 200)
dt = float(np.median(res))
print("flags %d%s: %.1f us/step, %.2f M robot-steps/s" % (FLAGS, "" if READY else " plain", 1e6 * dt, R / dt / 1e6))
