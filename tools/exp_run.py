"""Run the bench world for a few steps against a given build of the library (for rocprofv3 --pmc passes over
experiment builds made by tools/experiment.py --build-only).  usage: python3 tools/exp_run.py <lib.so> [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img_env_amd import _cabi, worldgen  # noqa: E402
so = os.path.abspath(sys.argv[1])
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
for s in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    w.step(a)
torch.cuda.synchronize()
