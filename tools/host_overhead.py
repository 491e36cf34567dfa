"""Where does wall time go besides the kernels?  Times world.reset(), the host cost of issuing one step
(no sync) and the synchronous step loop at the bench workload.  usage: python tools/host_overhead.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from img_env_amd import worldgen  # noqa: E402
from img_env_amd.world import World  # noqa: E402

RL, P = bench.ROBOTS_PER_GPU, bench.N_PEDS
side = bench.grid_cells(1)
grid = worldgen.make_grid(side, 0)
layouts = [worldgen.make_layout(grid, bench.RES, RL, P, seed=100 + s, clearance=bench.CLEARANCE) for s in range(2)]
params = worldgen.make_params(RL, P, res=bench.RES, view_cells=48, beams=360, scene="rvoscene", time_max=bench.TIME_MAX)
w = World(params, grid, device=0)
dev = torch.device("cuda", 0)
a = torch.zeros(RL, 3, device=dev)
a[:, 1] = 0.3
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w.reset(layouts[i % 2])
    torch.cuda.synchronize(); print("reset %d: %.3f ms" % (i, 1e3 * (time.perf_counter() - t0)))
for i in range(20):
    w.step(a)
torch.cuda.synchronize()
for n in (50, 50, 50):
    t0 = time.perf_counter()
    for i in range(n):
        w.step(a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%d steps: issue %.1f us/step, total %.1f us/step" % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))

# drift check: per-kernel time (HIP events) in windows of 25 steps, no reset in between
w.reset(layouts[0])
w.timing(1)
prev = w.timing_read()
for win in range(10):
    for i in range(25):
        w.step(a)
    cur = w.timing_read()
    print("steps %3d-%3d " % (25 * win, 25 * win + 24) + " ".join(
        "%s %.1f" % (k[2:], 1e3 * (cur[k][0] - prev[k][0]) / max(cur[k][1] - prev[k][1], 1)) for k in cur))
    prev = cur
w.timing(0)

# run-to-run variance inside one process: 200-step bursts, with and without the event pair around k_view
for mode in (0, 2, 0, 2, 1, 0):
    w.reset(layouts[0])
    for i in range(10):
        w.step(a)
    w.timing(mode, 5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(90):
        w.step(a)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    w.timing_read()
    w.timing(0)
    print("timing mode %d: %.1f us/step" % (mode, 1e6 * dt / 90))
