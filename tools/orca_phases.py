"""Per-phase cycle split of k_orca (a -DIMGENV_PHASE_PROFILE build) on the shipped test.yaml geometry (one robot, 4 leg pedestrians,
4 obstacles = 16 RVO segments per env) or on a BASELINE config.
usage (GPU box): python tools/orca_phases.py shipped [envs]   |   python tools/orca_phases.py cfg5|cfg3 [robots]"""
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
subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + ["-DIMGENV_PHASE_PROFILE", os.path.join(g.CSRC, "imgenv_hip.hip"), "-o", so])
from img_env_amd import _cabi, worldgen  # noqa: E402
_cabi.library_path = lambda: so
import torch  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "shipped"
N = 20
if what == "shipped":
    from PIL import Image
    from img_env_amd.vec_env import VecImageEnv
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 256
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
    a = torch.zeros(E, 3, device="cuda")
    a[:, 0], a[:, 1] = torch.rand(E, device="cuda") * 0.6, torch.rand(E, device="cuda") * 1.8 - 0.9
    waves_per_step = E  # 4 pedestrians = one group per env
else:
    from img_env_amd.world import World
    c = dict(worldgen.PRESETS[what])
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    grid = worldgen.make_grid(c["grid"], 0)
    layout = worldgen.make_layout(grid, c["res"], R, c["n_peds"], seed=100, clearance=c["clearance"])
    w = World(worldgen.make_params(R, c["n_peds"], res=c["res"], view_cells=c["view_cells"], beams=c["beams"], scene=c["scene"]), grid)
    w.reset(layout)
    a = torch.zeros(R, 3, device="cuda")
    a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
    waves_per_step = (c["n_peds"] + 3) // 4
w.lib.imgenv_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 16)()
for s in range(10):
    w.step(a)
w.lib.imgenv_debug_phases(w.h, buf)
for s in range(N):
    w.step(a)
w.lib.imgenv_debug_phases(w.h, buf)
v = list(buf)
names = {5: "staging + scalars + waypoint", 6: "neighbour scans", 7: "tree nodes (row)", 12: "traversal replay (home lane)",
         13: "rank sort + obstacle lines (row)", 14: "agent lines (row)", 15: "linear programs (home lane)"}
tot = sum(v[k] for k in names) or 1
for k, n in names.items():
    print("  %-34s %9.0f cycles/wave  %5.1f %%" % (n, v[k] / (N * waves_per_step), 100.0 * v[k] / tot))
print("  total %.0f cycles/wave (100 MHz counter ticks if s_memtime; see PHASE_MARK)" % (tot / (N * waves_per_step)))
