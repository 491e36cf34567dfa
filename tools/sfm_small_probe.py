import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from scenarios import small_world
from img_env_amd.world import World
for P in (8, 30):
    grid, params, layout = small_world(8, P, seed=14, scene="pedscene", grid_size=88, n_obstacles=3, clearance=0.45)
    w = World(params, grid)
    w.reset(layout)
    a = torch.zeros(8, 3, device="cuda"); a[:, 1] = 0.3
    for s in range(20): w.step(a)
    w.timing(1); prev = w.timing_read(); samples = {k: [] for k in prev}
    for s in range(30):
        w.step(a); cur = w.timing_read()
        for k in cur:
            if cur[k][1] > prev[k][1]: samples[k].append(1e3 * (cur[k][0] - prev[k][0]) / (cur[k][1] - prev[k][1]))
        prev = cur
    print(P, " ".join("%s %.1f" % (k[2:], np.median(v)) for k, v in samples.items() if v))
    w.close()
