"""Build a variant of the library with extra -D flags and print its per-kernel times on the bench world
(results of experiment builds are NOT parity-checked).  usage: python tools/experiment.py -DFLAG [...]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
out = [a for a in sys.argv[1:] if a.startswith("--out=")]
so = out[0][6:] if out else os.path.join(g.CSRC, "libimgenv_hip_exp.so")
subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + args + [os.path.join(g.CSRC, "imgenv_hip.hip"), "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
from img_env_amd import _cabi, worldgen  # noqa: E402
_cabi.library_path = lambda: so
import torch  # noqa: E402
from img_env_amd.world import World  # noqa: E402

R, P, res = 8192, 200, 0.25
grid = worldgen.make_grid(400, 0)
layout = worldgen.make_layout(grid, res, R, P, seed=100, clearance=0.7)
w = World(worldgen.make_params(R, P, res=res, scene="rvoscene"), grid)
w.reset(layout)
a = torch.zeros(R, 3, device="cuda")
a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
for s in range(300):
    w.step(a)
import time  # noqa: E402
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(300):
    w.step(a)
torch.cuda.synchronize()
print("ms/step %.4f" % ((time.perf_counter() - t0) / 300 * 1e3))
w.timing(1)
prev = w.timing_read()
samples = {k: [] for k in prev}
for s in range(40):
    w.step(a)
    cur = w.timing_read()
    for k in cur:
        if cur[k][1] > prev[k][1]:
            samples[k].append(1e3 * (cur[k][0] - prev[k][0]) / (cur[k][1] - prev[k][1]))
    prev = cur
print(" ".join("%s %.1f" % (k[2:], np.median(v)) for k, v in samples.items() if v))
