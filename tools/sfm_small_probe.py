"""k_sfm in a valid room-sized social-force world: per-kernel times and (profile build) wall-clock marks inside the kernel.
usage (GPU box): python tools/sfm_small_probe.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

so = os.path.join(g.CSRC, "libimgenv_hip_prof.so")
subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + ["-DIMGENV_PHASE_PROFILE", os.path.join(g.CSRC, "imgenv_hip.hip"), "-o", so])
from img_env_amd import _cabi  # noqa: E402
_cabi.library_path = lambda: so
import torch  # noqa: E402
from img_env_amd.world import World  # noqa: E402
from scenarios import small_world  # noqa: E402

names = ["neighbours (tree walk)", "desired force", "own angles", "pair terms", "sums + obstacle", "move", "tree moves (serial)"]
for P in (8, 30):
    grid, params, layout = small_world(8, P, seed=14, scene="pedscene", grid_size=88, n_obstacles=3, clearance=0.45)
    w = World(params, grid)
    w.lib.imgenv_debug_marks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    w.reset(layout)
    a = torch.zeros(8, 3, device="cuda")
    a[:, 1] = 0.3
    for s in range(20):
        w.step(a)
    buf = (C.c_ulonglong * 32)()
    w.lib.imgenv_debug_marks(w.h, buf)
    v = list(buf)
    print("%d peds + 8 robots: " % P + ", ".join("%s %.1f us" % (names[q], (v[q + 1] - v[q]) / 100.0) for q in range(6)))
    w.close()
