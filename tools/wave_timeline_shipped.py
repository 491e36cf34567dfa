"""Per-wave timeline of the robot blocks of k_raster (and of k_obs) on the shipped test.yaml geometry: tools/wave_timeline.py for
VecImageEnv's world (plain steps, no auto-reset).  usage (GPU box): python tools/wave_timeline_shipped.py [envs]"""
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

so = os.path.join(g.CSRC, "libimgenv_hip_tl.so")
subprocess.check_call([g.HIPCC] + g.HIP_FLAGS + ["-DIMGENV_WAVE_TIMELINE", os.path.join(g.CSRC, "imgenv_hip.hip"), "-o", so])
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
R = E
w.lib.imgenv_debug_waves.argtypes = [C.c_void_p, C.c_void_p]
a = torch.zeros(R, 3, device="cuda")
a[:, 0], a[:, 1] = torch.rand(R, device="cuda") * 0.6, torch.rand(R, device="cuda") * 1.8 - 0.9
for s in range(6):
    w.step(a)
torch.cuda.synchronize()
buf = np.zeros(12 * R, dtype=np.uint64)
w.lib.imgenv_debug_waves(w.h, buf.ctypes.data_as(C.c_void_p))
for name, rec in (("k_obs", buf[4 * R:8 * R].reshape(R, 4)), ("k_raster (robot blocks)", buf[8 * R:].reshape(R, 4))):
    t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    ok = (t1 > 0) & (t0 > t0.max() - 30000)
    if not ok.any():
        print(name, "no records")
        continue
    base = t0[ok].min()
    s, e = (t0[ok] - base) / 100.0, (t1[ok] - base) / 100.0  # 100 MHz wall clock -> us
    d = e - s
    print("%s: %d waves, span %.1f us; wave life us: min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
        name, ok.sum(), e.max(), d.min(), *np.percentile(d, [10, 50, 90, 99]), d.max()))
    print("   starts us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % (*np.percentile(s, [10, 50, 90]), s.max()))
    print("   starts histogram (5 us bins): " + " ".join(str(int(x)) for x in np.histogram(s, bins=np.arange(0, s.max() + 5, 5))[0]))
    ts = np.linspace(0, e.max(), 11)[1:-1]
    print("   resident waves at " + " ".join("%.0fus:%d" % (t, ((s <= t) & (e > t)).sum()) for t in ts))
vec.close()
