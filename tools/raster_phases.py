"""Cycle marks inside k_raster on the shipped test.yaml geometry (-DIMGENV_PHASE_PROFILE -DIMGENV_PROFILE_RASTER build): a robot
block's pose / cached test, footprint samples into the LDS box, box -> stamps + list; every block's pedestrian.
usage (GPU box): python tools/raster_phases.py [envs]"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

so = os.path.join(g.CSRC, "libimgenv_hip_prof.so")
subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + ["-DIMGENV_PHASE_PROFILE", "-DIMGENV_PROFILE_RASTER", os.path.join(g.CSRC, "imgenv_hip.hip"), "-o", so])
from img_env_amd import _cabi, worldgen  # noqa: E402
_cabi.library_path = lambda: so
import torch  # noqa: E402
from PIL import Image  # noqa: E402
from img_env_amd.vec_env import VecImageEnv  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
z = np.load(os.path.join(ROOT, "tests", "golden", "spawn_ref.npz"))
tmp = tempfile.mkdtemp()
m = np.full((110, 110), 255, np.uint8)
m[:5] = m[-5:] = 0
m[:, :5] = m[:, -5:] = 0
Image.fromarray(m).save(os.path.join(tmp, "room.png"))
cfg = worldgen.shipped_test_yaml_cfg("room.png", json.loads(str(z["test@1/cfg"])))
cfg.update(map_dir=tmp, seed=1)
vec = VecImageEnv(cfg, env_num=E, seed=1, native_spawn=True)
vec.reset()
w = vec.world
w.lib.imgenv_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
a = torch.zeros(E, 3, device="cuda")
a[:, 0], a[:, 1] = torch.rand(E, device="cuda") * 0.6, torch.rand(E, device="cuda") * 1.8 - 0.9
buf = (C.c_ulonglong * 16)()
for s in range(6):
    w.step(a)
w.lib.imgenv_debug_phases(w.h, buf)
N = 10
for s in range(N):
    w.step(a)
w.lib.imgenv_debug_phases(w.h, buf)
v = list(buf)
P = 4 * E
for k, n, cnt in ((5, "robot: pose + cached-list test", E), (6, "robot: footprint samples -> LDS box", E), (7, "robot: box -> stamps + list", E),
                  (12, "pedestrian (two legs)", P)):
    print("  %-40s %9.0f cycles / block" % (n, v[k] / (N * cnt)))
vec.close()
