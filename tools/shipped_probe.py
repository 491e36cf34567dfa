"""Throughput on the geometry of the reference's shipped envs/cfg/test.yaml (733 x 733 grid from a 110 x 110 pixel map, 400 x 400
cell views shrunk to 48 x 48 with INTER_CUBIC, 1000 beams, 1 robot + 4 leg pedestrians + 4 obstacles per env), E envs in one
handle (VecImageEnv, native spawn).  BASELINE.md section 2 measured the reference's C++ core alone on this geometry at 194
robot-steps/s on one CPU core (no ROS, no Python post-processing).

    python tools/shipped_probe.py [--envs 64] [--steps 100]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=64)
ap.add_argument("--steps", type=int, default=100)
args = ap.parse_args()

import torch  # noqa: E402
from PIL import Image  # noqa: E402
from img_env_amd import worldgen  # noqa: E402
from img_env_amd.vec_env import VecImageEnv  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "spawn_ref.npz"))
sections = json.loads(str(z["test@1/cfg"]))
tmp = tempfile.mkdtemp()
m = np.full((110, 110), 255, np.uint8)
m[:5] = m[-5:] = 0
m[:, :5] = m[:, -5:] = 0
Image.fromarray(m).save(os.path.join(tmp, "room.png"))
cfg = worldgen.shipped_test_yaml_cfg("room.png", sections)
cfg.update(map_dir=tmp, seed=1)
E = args.envs
vec = VecImageEnv(cfg, env_num=E, seed=1, native_spawn=True)
vec.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.zeros(16, E, 3, device="cuda")
acts[:, :, 0] = torch.rand(16, E, generator=g, device="cuda") * 0.6
acts[:, :, 1] = torch.rand(16, E, generator=g, device="cuda") * 1.8 - 0.9
for s in range(20):
    vec.step(acts[s % 16])
w = vec.world
w.timing(1)
for s in range(10):
    vec.step(acts[s % 16])
torch.cuda.synchronize()
tm = w.timing_read()
w.timing(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
n_reset = 0
for s in range(args.steps):
    _, _, _, info = vec.step(acts[s % 16])
    n_reset += len(info["reset_envs"])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps(dict(config="shipped test.yaml geometry: 733x733 grid @0.015 m, 400x400 view -> 48x48 (INTER_CUBIC), 1000 beams, "
                             "1 robot + 4 leg peds + 4 obstacles per env", envs=E, steps=args.steps, env_resets=n_reset,
                      value=E * args.steps / dt, unit="robot-steps/s", us_per_step=1e6 * dt / args.steps,
                      reference_cpp_core_one_cpu_core=194,
                      kernel_us={k: round(1e3 * ms / n, 1) for k, (ms, n) in tm.items() if n})))
vec.close()
