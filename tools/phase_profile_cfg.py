"""Per-phase cycle split of k_view / k_obs on a BASELINE config (a -DIMGENV_PHASE_PROFILE build): tools/phase_profile.py for cfg5 / cfg3.
usage (GPU box): python tools/phase_profile_cfg.py cfg5 [robots]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

so = os.path.join(g.CSRC, "libimgenv_hip_prof.so")
subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + ["-DIMGENV_PHASE_PROFILE", os.path.join(g.CSRC, "imgenv_hip.hip"), "-o", so])
from img_env_amd import _cabi, worldgen  # noqa: E402
_cabi.library_path = lambda: so
import torch  # noqa: E402
from img_env_amd.world import World  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
c = dict(worldgen.PRESETS[what])
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
grid = worldgen.make_grid(c["grid"], 0)
layout = worldgen.make_layout(grid, c["res"], R, c["n_peds"], seed=100, clearance=c["clearance"])
w = World(worldgen.make_params(R, c["n_peds"], res=c["res"], view_cells=c["view_cells"], beams=c["beams"], scene=c["scene"]), grid)
w.lib.imgenv_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
w.reset(layout)
a = torch.zeros(R, 3, device="cuda")
a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
buf = (C.c_ulonglong * 16)()
for s in range(10):
    w.step(a)
w.lib.imgenv_debug_phases(w.h, buf)
N = 20
for s in range(N):
    w.step(a)
w.lib.imgenv_debug_phases(w.h, buf)
v = list(buf)
waves = N * R
for title, names, lo in (("k_view (per workgroup's first lane)", ["collision", "crop", "hits", "compose+skip", "store / resolve"], 0),
                         ("k_obs", ["state + ped transform", "sort", "ped vector", "ped_map"], 8)):
    tot = sum(v[lo:lo + len(names)]) or 1
    print(title)
    for n, cyc in zip(names, v[lo:lo + len(names)]):
        print("  %-24s %9.0f cycles/robot  %5.1f %%" % (n, cyc / waves, 100.0 * cyc / tot))
    print("  total %.0f cycles/robot" % (tot / waves))
