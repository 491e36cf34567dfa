"""Long run of plain (stream-ordered) steps on the headline world: the observation's gate (k_gate) opens a few
hundred thousand times, with a full reset every 101 steps and promised steps mixed in; any device-side error (a gate that gave up,
an overflow flag) fails the next call.
usage (GPU box): python tools/gate_soak.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img_env_amd import worldgen  # noqa: E402
from img_env_amd.world import World  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
R, P = 8192, 200
grid = worldgen.make_grid(400, 0)
layouts = [worldgen.make_layout(grid, 0.25, R, P, seed=100 + k, clearance=0.7) for k in range(2)]
params = worldgen.make_params(R, P, res=0.25, view_cells=48, beams=360, scene="rvoscene", time_max=100)
a_w = World(params, grid)
g = torch.Generator(device="cuda")
g.manual_seed(3)
acts = torch.zeros(64, R, 3, device="cuda")
acts[:, :, 1] = torch.rand(64, R, generator=g, device="cuda") * 1.8 - 0.9
acts[:, :, 0] = torch.rand(64, R, generator=g, device="cuda") * 0.05
t0 = time.perf_counter()
a_w.reset(layouts[0])
for s in range(N):
    a = acts[s & 63]
    a_w.step(a, actions_ready=(s % 7 == 3))   # plain, now and then promised
    if s % 101 == 100:
        a_w.reset(layouts[(s // 101) & 1])
    if s % 50000 == 49999:
        torch.cuda.synchronize()
        print("step %d: %.1f us per step" % (s + 1, 1e6 * (time.perf_counter() - t0) / (s + 1)), flush=True)
torch.cuda.synchronize()
snap = a_w.snapshot()
print("gate soak ok: %d steps, frozen %.3f, counters %s" % (N, float(((snap["is_collisions"] != 0) | (snap["is_arrives"] != 0)).mean()), snap["counters"][:2]))
