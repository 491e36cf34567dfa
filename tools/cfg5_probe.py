"""BASELINE cfg-5 geometry on one GPU: 800x800 grid @0.125 m, 96x96 view, 720 beams, 1000 ERVO pedestrians.
usage (GPU box): python tools/cfg5_probe.py [n_robots]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img_env_amd import worldgen  # noqa: E402
from img_env_amd.world import World  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 32: views through the tiled kernels, 64: through k_view
SHORT = len(sys.argv) > 3 and sys.argv[3] == "short"  # counter passes: a handful of steps (the profiler serialises every dispatch)
P = 1000
grid = worldgen.make_grid(800, 0)
layout = worldgen.make_layout(grid, 0.125, R, P, seed=100, clearance=0.7)
w = World(dict(worldgen.make_params(R, P, res=0.125, view_cells=96, beams=720, scene="ervoscene"), flags=FLAGS), grid)
w.reset(layout)
a = torch.zeros(R, 3, device="cuda")
a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
for s in range(3 if SHORT else 50):
    w.step(a, actions_ready=True)  # (pre-generated, resident in HBM)
w.timing(1)
prev = w.timing_read()
samples = {k: [] for k in prev}
for s in range(2 if SHORT else 20):
    w.step(a, actions_ready=True)  # (pre-generated, resident in HBM)
    cur = w.timing_read()
    for k in cur:
        if cur[k][1] > prev[k][1]:
            samples[k].append(1e3 * (cur[k][0] - prev[k][0]) / (cur[k][1] - prev[k][1]))
    prev = cur
w.timing(0)
print(" ".join("%s %.1f" % (k[2:], np.median(v)) for k, v in samples.items() if v))
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 4 if SHORT else 100
for s in range(N):
    w.step(a, actions_ready=True)  # (pre-generated, resident in HBM)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print("%d robots: %.1f us/step, %.2f M robot-steps/s" % (R, 1e6 * dt, R / dt / 1e6))
