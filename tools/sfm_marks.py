import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import __graft_entry__ as g
so = os.path.join(g.CSRC, "libimgenv_hip_prof.so")
subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + ["-DIMGENV_PHASE_PROFILE", os.path.join(g.CSRC, "imgenv_hip.hip"), "-o", so])
from img_env_amd import _cabi, worldgen
_cabi.library_path = lambda: so
import torch
from img_env_amd.world import World
R, P = 1024, 200
grid = worldgen.make_grid(400, 0)
layout = worldgen.make_layout(grid, 0.5, R, P, seed=100, clearance=0.5)
rng = np.random.default_rng(13)
layout.ped_pose[:, :2] = rng.uniform(0.5, 9.5, (P, 2))
layout.ped_traj[:, :, :2] = rng.uniform(0.5, 9.5, layout.ped_traj[:, :, :2].shape)
layout.ped_goal[:] = rng.uniform(0.5, 9.5, (P, 2))
w = World(worldgen.make_params(R, P, res=0.5, scene="pedscene", relation_ped_robo=0), grid)
w.lib.imgenv_debug_marks.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
w.reset(layout)
a = torch.zeros(R, 3, device="cuda")
for s in range(30):
    w.step(a)
buf = (C.c_ulonglong * 32)()
for rep in range(3):
    w.step(a)
    w.lib.imgenv_debug_marks(w.h, buf)
    v = list(buf)[:16]
    print("first launch:", [round((v[b] - v[a]) / 100.0, 1) for a, b in ((12, 13), (13, 14), (14, 1), (1, 2), (2, 15))], "us: tree into LDS, who is in the tree, neighbour walk, desired force, angles")
    print([round((v[q + 1] - v[q]) / 100.0, 1) for q in range(2, 5)], "us: pairs (third launch's wait), sums+obstacle, move;  tree surgery:",
          [round((v[b] - v[a]) / 100.0, 1) for a, b in ((5, 7), (7, 8), (8, 9), (9, 6))], "us: who left + descents, counts + moves on quiet leaves, serial replay, tree back to HBM;",
          "agents replayed serially: %d%s" % (v[10], " (whole step: a tie or a stray entry)" if v[11] else ""))
