import sys, time, json, os, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
from img_env_amd import worldgen
from img_env_amd.vec_env import VecImageEnv
mode = sys.argv[1]
grid = worldgen.make_grid(200, 2)
cfg = worldgen.make_yaml_cfg(4, 3, grid, time_max=100, n_obstacles=2, seed=5)
env = VecImageEnv(cfg, env_num=1024, seed=5, native_spawn=True, device_reset=(mode == "device"))
n = len(env)
acts = torch.zeros(16, n, 3, device="cuda"); acts[:, :, 0] = 0.3; acts[:, :, 1] = torch.rand(16, n, device="cuda") - 0.5
env.reset()
for s in range(150): env.step(acts[s % 16])
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(300): env.step(acts[s % 16])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(mode, "host issue %.1f us/step, end to end %.1f us/step" % (1e6 * (t1 - t0) / 300, 1e6 * (t2 - t0) / 300))
env.close()
