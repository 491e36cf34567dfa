"""Experiment build (-DIMGENV_EXP_RESOLVE_STATS): how many cells per robot the top beam leaves alone (agent.cpp:555-560) in the
bench world, and how long the per-cell ray lists behind them are.  Prints totals over the steps run."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
so = os.path.join(g.CSRC, "libimgenv_hip_exp.so")
subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + ["-DIMGENV_EXP_RESOLVE_STATS", os.path.join(g.CSRC, "imgenv_hip.hip"), "-o", so])
from img_env_amd import _cabi, worldgen  # noqa: E402
_cabi.library_path = lambda: so
import torch  # noqa: E402
from img_env_amd.world import World  # noqa: E402

R, P, res = 8192, 200, 0.25
grid = worldgen.make_grid(400, 0)
layout = worldgen.make_layout(grid, res, R, P, seed=100, clearance=0.7)
w = World(worldgen.make_params(R, P, res=res, scene="rvoscene"), grid)
w.lib.imgenv_debug_marks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
w.reset(layout)
a = torch.zeros(R, 3, device="cuda")
a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
steps = 20
for s in range(steps):
    w.step(a)
buf = (C.c_ulonglong * 32)()
w.lib.imgenv_debug_marks(w.h, buf)
v = list(buf)
launches = R * (steps + 1)
print("views %d, with skipped cells %d, entries/view %.1f, cells/view %.1f, mean list %.1f, max list %d" %
      (launches, v[16], v[17] / max(v[16], 1), v[18] / max(v[16], 1), v[19] / max(v[18], 1), v[20]))
print("cells by list length (bins of 8):", v[21:30])
print("walk ends at entry 1 / 2 / 3-4 / 5-8 / 9-16 / 17+ / never:", v[8:15], "cells that change value:", v[15])
print("entries walked: to a deciding beam %d, to the end of the list %d" % (v[31], v[30]))
print("chunks / view %.1f, result slots / view %.1f (after the filter)" % (v[6] / max(v[16], 1), v[7] / max(v[16], 1)))
