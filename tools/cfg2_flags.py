import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from img_env_amd import worldgen
from img_env_amd.world import World
flags = int(sys.argv[1]); R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
c = dict(worldgen.PRESETS["cfg2"])
grid = worldgen.make_grid(c["grid"], 0)
layout = worldgen.make_layout(grid, c["res"], R, 0, seed=100, clearance=c["clearance"])
w = World(dict(worldgen.make_params(R, 0, res=c["res"], view_cells=48, beams=360, scene=""), flags=flags), grid)
w.reset(layout)
a = torch.zeros(R, 3, device="cuda"); a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
for s in range(500): w.step(a)
w.timing(1); prev = w.timing_read(); samples = {k: [] for k in prev}
for s in range(20):
    w.step(a); cur = w.timing_read()
    for k in cur:
        if cur[k][1] > prev[k][1]: samples[k].append(1e3 * (cur[k][0] - prev[k][0]) / (cur[k][1] - prev[k][1]))
    prev = cur
w.timing(0); torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(500): w.step(a)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
print("flags", flags, "R", R, "%.1f us/step %.2f M" % (1e6 * dt, R / dt / 1e6), {k: round(float(np.median(v)), 1) for k, v in samples.items() if v}, "launches", w.launches())
