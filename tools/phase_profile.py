"""Per-phase cycle split of k_view / k_obs (needs a -DIMGENV_PHASE_PROFILE build: see the bottom).
usage (on the GPU box): python tools/phase_profile.py [active|episode]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

so = os.path.join(g.CSRC, "libimgenv_hip_prof.so")
subprocess.check_call(g.hip_command(so, ["-DIMGENV_PHASE_PROFILE"]))
from img_env_amd import _cabi, worldgen  # noqa: E402
_cabi.library_path = lambda: so
import torch  # noqa: E402
from img_env_amd.world import World  # noqa: E402

policy = sys.argv[1] if len(sys.argv) > 1 else "active"
R, P, res = int(sys.argv[2]) if len(sys.argv) > 2 else 8192, 200, 0.25
grid = worldgen.make_grid(400, 0)
layout = worldgen.make_layout(grid, res, R, P, seed=100, clearance=0.7)
w = World(worldgen.make_params(R, P, res=res, scene="rvoscene"), grid)
w.lib.imgenv_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
w.reset(layout)
a = torch.zeros(R, 3, device="cuda")
a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
if policy == "episode":
    a[:, 0] = torch.rand(R, device="cuda") * 0.6
buf = (C.c_ulonglong * 16)()
for s in range(10):
    w.step(a)
w.lib.imgenv_debug_phases(w.h, buf)
N = 20
for s in range(N):
    w.step(a)
w.lib.imgenv_debug_phases(w.h, buf)
v = list(buf)
names = ["view: collision", "view: crop", "view: hits", "view: compose+skip", "view: store"]
waves = N * R
tot = sum(v[:5]) or 1
for n, c in zip(names, v[:5]):
    print("  %-22s %9.0f cycles/wave  %5.1f %%" % (n, c / waves, 100.0 * c / tot))
print("  total %.0f cycles/wave" % (tot / waves))
names2 = ["obs: state + ped transform", "obs: bitonic sort", "obs: ped vector + raster", "obs: ped_map store"]
tot2 = sum(v[8:12]) or 1
for n, c in zip(names2, v[8:12]):
    print("  %-28s %9.0f cycles/wave  %5.1f %%" % (n, c / waves, 100.0 * c / tot2))
print("  total %.0f cycles/wave" % (tot2 / waves))
