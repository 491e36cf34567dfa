"""Per-kernel times of the social-force (pedscene) world at the benchmark size and, with a -DIMGENV_PHASE_PROFILE
build, wall-clock marks inside k_sfm.  usage (GPU box): python tools/sfm_probe.py [n_peds]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

so = os.path.join(g.CSRC, "libimgenv_hip_prof.so")
subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + ["-DIMGENV_PHASE_PROFILE", os.path.join(g.CSRC, "imgenv_hip.hip"), "-o", so])
from img_env_amd import _cabi, worldgen  # noqa: E402
_cabi.library_path = lambda: so
import torch  # noqa: E402
from img_env_amd.world import World  # noqa: E402

R, P = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 200
grid = worldgen.make_grid(400, 0)
layout = worldgen.make_layout(grid, 0.25, R, P, seed=100, clearance=0.7)
rng = np.random.default_rng(13)  # the crowd stays inside libpedsim's 10 m x 10 m quadtree root (pedscene.h:17-20)
layout.ped_pose[:, :2] = rng.uniform(0.5, 9.5, (P, 2))
layout.ped_traj[:, :, :2] = rng.uniform(0.5, 9.5, layout.ped_traj[:, :, :2].shape)
layout.ped_goal[:] = rng.uniform(0.5, 9.5, (P, 2))
w = World(worldgen.make_params(R, P, res=0.25, scene="pedscene", relation_ped_robo=0), grid)
w.lib.imgenv_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
w.reset(layout)
a = torch.zeros(R, 3, device="cuda")
a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
for s in range(5):
    w.step(a)
buf = (C.c_ulonglong * 16)()
w.lib.imgenv_debug_phases(w.h, buf)
w.step(a)
w.lib.imgenv_debug_phases(w.h, buf)
v = list(buf)
names = ["neighbours (tree walk)", "desired + lookahead", "social", "obstacle", "move", "(barrier)", "tree moves (serial)"]
for q in range(6):
    print("  %-24s %10.1f us" % (names[q], (v[q + 1] - v[q]) / 100.0))
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(20):
    w.step(a)
torch.cuda.synchronize()
print("us/step", 1e6 * (time.perf_counter() - t0) / 20)
