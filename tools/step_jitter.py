"""Distribution of step time in windows of 20 steps over a long run (looks for intermittent slow phases).
usage: python tools/step_jitter.py [active|episode] [windows]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from img_env_amd import worldgen  # noqa: E402
from img_env_amd.world import World  # noqa: E402

policy = sys.argv[1] if len(sys.argv) > 1 else "active"
nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 150
SYNC_RESET = len(sys.argv) > 3 and sys.argv[3] == "sync"
RL, P = bench.ROBOTS_PER_GPU, bench.N_PEDS
grid = worldgen.make_grid(bench.grid_cells(1), 0)
layouts = [worldgen.make_layout(grid, bench.RES, RL, P, seed=100 + s, clearance=bench.CLEARANCE) for s in range(3)]
w = World(worldgen.make_params(RL, P, res=bench.RES, view_cells=48, beams=360, scene="rvoscene", time_max=bench.TIME_MAX), grid, device=0)
a = torch.zeros(RL, 3, device="cuda")
a[:, 1] = torch.rand(RL, device="cuda") * 1.8 - 0.9
if policy == "episode":
    a[:, 0] = torch.rand(RL, device="cuda") * 0.6
w.reset(layouts[0])
el, ep, out = 0, 0, []
reset_ms = []
for win in range(nwin):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    resets = 0
    for s in range(20):
        w.step(a)
        el += 1
        if el > bench.TIME_MAX:
            ep += 1
            if SYNC_RESET:
                torch.cuda.synchronize()
            tr0 = time.perf_counter()
            w.reset(layouts[ep % 3])
            if SYNC_RESET:
                torch.cuda.synchronize()
            reset_ms.append(1e3 * (time.perf_counter() - tr0))
            el = 0
            resets += 1
    torch.cuda.synchronize()
    out.append((1e6 * (time.perf_counter() - t0) / 20, resets))
us = np.array([o[0] for o in out])
print("us/step per 20-step window: min %.1f p50 %.1f p90 %.1f max %.1f" % (us.min(), np.median(us), np.percentile(us, 90), us.max()))
print("windows > 1.3 x median: " + " ".join("%d:%.0f%s" % (i, u, "R" if out[i][1] else "") for i, u in enumerate(us) if u > 1.3 * np.median(us)))
print("reset ms: p50 %.2f max %.2f; slow ones at reset # %s" % (np.median(reset_ms), max(reset_ms), [i for i, r in enumerate(reset_ms) if r > 3]))
