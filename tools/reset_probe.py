import os, sys, time, gc
import ctypes as C
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from img_env_amd import worldgen, _cabi
from img_env_amd.world import World
RL, P = bench.ROBOTS_PER_GPU, bench.N_PEDS
grid = worldgen.make_grid(bench.grid_cells(1), 0)
layouts = [worldgen.make_layout(grid, bench.RES, RL, P, seed=100 + s, clearance=bench.CLEARANCE) for s in range(3)]
w = World(worldgen.make_params(RL, P, res=bench.RES, view_cells=48, beams=360, scene="rvoscene", time_max=bench.TIME_MAX), grid, device=0)
a = torch.zeros(RL, 3, device="cuda"); a[:, 1] = 0.3
w.reset(layouts[0])
rows = []
for ep in range(60):
    for s in range(101):
        w.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lay = layouts[ep % 3]
    b, keep = _cabi.make_reset_batch(lay.as_batch(), w.n_robots, w.n_peds)
    t1 = time.perf_counter()
    rc = w.lib.imgenv_reset(w.h, C.byref(b), w._stream())
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rows.append((1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)))
r = np.array(rows)
print("make_batch ms: p50 %.2f max %.2f | imgenv_reset call ms: p50 %.2f max %.2f | sync after ms: p50 %.2f max %.2f" % (
    np.median(r[:,0]), r[:,0].max(), np.median(r[:,1]), r[:,1].max(), np.median(r[:,2]), r[:,2].max()))
print("slow resets:", [(i, [round(x,2) for x in row]) for i, row in enumerate(rows) if sum(row) > 3])
