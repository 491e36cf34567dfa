import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from img_env_amd import worldgen
from img_env_amd.world import World
for R in (2048, 4096, 8192, 16384):
    side = int(400 * (R / 8192) ** 0.5 / 4) * 4
    grid = worldgen.make_grid(side, 0)
    layout = worldgen.make_layout(grid, 0.25, R, 200, seed=100, clearance=0.7)
    w = World(worldgen.make_params(R, 200, res=0.25, scene="rvoscene"), grid)
    w.reset(layout)
    a = torch.zeros(R, 3, device="cuda"); a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
    for s in range(100): w.step(a)
    w.timing(1); prev = w.timing_read(); samples = {k: [] for k in prev}
    for s in range(30):
        w.step(a); cur = w.timing_read()
        for k in cur:
            if cur[k][1] > prev[k][1]: samples[k].append(1e3 * (cur[k][0] - prev[k][0]) / (cur[k][1] - prev[k][1]))
        prev = cur
    print(R, " ".join("%s %.1f" % (k[2:], np.median(v)) for k, v in samples.items() if v))
    w.close()
