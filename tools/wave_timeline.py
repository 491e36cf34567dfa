"""Per-wave timeline of k_view / k_obs (needs the -DIMGENV_PHASE_PROFILE build): how long each robot's wavefront
lives, when it starts relative to the first one, and how many are resident at once.
usage (on the GPU box): python tools/wave_timeline.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

so = os.path.join(g.CSRC, "libimgenv_hip_tl.so")
subprocess.check_call(g.hip_command(so, ["-DIMGENV_WAVE_TIMELINE"] + os.environ.get("IMGENV_TL_FLAGS", "").split()))  # (IMGENV_TL_FLAGS: more -D switches)
from img_env_amd import _cabi, worldgen  # noqa: E402
_cabi.library_path = lambda: so
import torch  # noqa: E402
from img_env_amd.world import World  # noqa: E402

CFG = sys.argv[1] if len(sys.argv) > 1 else "cfg3"  # cfg3 (the headline) | cfg2 (1024 robots, no pedestrians: eight wavefronts per view)
R, P, res, scene, clearance = (1024, 0, 0.125, "", 1.0) if CFG == "cfg2" else (8192, 200, 0.25, "rvoscene", 0.7)
grid = worldgen.make_grid(400, 0)
layout = worldgen.make_layout(grid, res, R, P, seed=100, clearance=clearance)
w = World(dict(worldgen.make_params(R, P, res=res, scene=scene, time_max=10 ** 7), output_guard="none"), grid)  # (no guard: its checks synchronise)
w.lib.imgenv_debug_waves.argtypes = [C.c_void_p, C.c_void_p]
w.reset(layout)
a = torch.zeros(R, 3, device="cuda")
a[:, 1] = torch.rand(R, device="cuda") * 1.8 - 0.9
for s in range(400):  # (the host queues far ahead of the device, as in a timed loop: the records are the LAST step's, whose launches were all queued long before)
    w.step(a)
buf = np.zeros(12 * R, dtype=np.uint64)
w.lib.imgenv_debug_waves(w.h, buf.ctypes.data_as(C.c_void_p))
if os.path.isdir(os.path.join(ROOT, "gpurun_out")):  # the raw records, for a closer look offline
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "wave_timeline.npz"), rec=buf, robot_pose=np.asarray(layout.robot_pose), n_robots=R)
# the three kernels on ONE clock (wall_clock64 is the chip's): who holds the slots when, relative to the observation's first wavefront
def _span(rec):
    t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    ok = (t1 > 0) & (t0 > t0.max() - 8000)
    return t0[ok], t1[ok]
_k = {"k_obs": _span(buf[4 * R:8 * R].reshape(R, 4)), "k_raster": _span(buf[8 * R:12 * R].reshape(R, 4)), "k_view": _span(buf[:4 * R].reshape(R, 4))}
_k = {nm: v for nm, v in _k.items() if len(v[0])}
_base = min(v[0].min() for v in _k.values()) if "k_obs" not in _k else _k["k_obs"][0].min()
for nm, (a0, a1) in _k.items():
    print("%-12s first start %6.1f  p50 start %6.1f  last start %6.1f | first end %6.1f  p50 end %6.1f  last end %6.1f  (us after the first wavefront of k_obs, or of all)" % (
        nm, (a0.min() - _base) / 100, (np.median(a0) - _base) / 100, (a0.max() - _base) / 100, (a1.min() - _base) / 100, (np.median(a1) - _base) / 100, (a1.max() - _base) / 100))
print("resident wavefronts every 4 us: t | " + " ".join(_k))
for t in range(-8, 100, 4):
    tt = _base + t * 100
    print("  %4d | %s" % (t, " ".join("%6d" % (((a0 <= tt) & (a1 > tt)).sum()) for a0, a1 in _k.values())))
for name, rec in (("k_view", buf[:4 * R].reshape(R, 4)), ("k_obs", buf[4 * R:8 * R].reshape(R, 4)), ("k_raster", buf[8 * R:12 * R].reshape(R, 4))):
    t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    if not (t1 > 0).any():
        continue  # (not launched on this handle)
    ok = (t1 > 0) & (t0 > t0.max() - 6000)  # this launch only (60 us back from the last start): frozen robots keep an older record
    xc = (rec[:, 3] & 0xF).astype(np.int64)
    print("   per-XCC first start (ticks): " + " ".join(str(int(t0[ok & (xc == x)].min() - t0[ok].min())) for x in sorted(set(xc[ok].tolist()))))
    base = t0[ok].min()
    s, e = (t0[ok] - base) / 100.0, (t1[ok] - base) / 100.0  # 100 MHz wall clock -> us
    d = e - s
    print("%s: %d waves, span %.1f us; wave life us: min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
        name, ok.sum(), e.max(), d.min(), *np.percentile(d, [10, 50, 90, 99]), d.max()))
    print("   starts us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % (*np.percentile(s, [10, 50, 90]), s.max()))
    print("   starts histogram (5 us bins): " + " ".join(str(int(x)) for x in np.histogram(s, bins=np.arange(0, s.max() + 5, 5))[0]))
    ts = np.linspace(0, e.max(), 11)[1:-1]
    print("   resident waves at " + " ".join("%.0fus:%d" % (t, ((s <= t) & (e > t)).sum()) for t in ts))
    ph = np.stack([((rec[ok, 2] >> np.uint64(32)) & np.uint64(0xFFFF)), ((rec[ok, 2] >> np.uint64(48)) & np.uint64(0xFFFF)),
                   ((rec[ok, 3] >> np.uint64(32)) & np.uint64(0xFFFF)), ((rec[ok, 3] >> np.uint64(48)) & np.uint64(0xFFFF))], 1).astype(np.float64) / 100.0
    if ph.max() > 0:  # when each wavefront passed the kernel's phase marks (us since its start): all, and the slowest / fastest tenth
        n_ph = int((ph.max(axis=0) > 0).sum())
        seg = np.diff(np.concatenate([np.zeros((ph.shape[0], 1)), ph[:, :n_ph], d[:, None]], 1), axis=1)
        order = np.argsort(d)
        k10 = max(len(d) // 10, 1)
        for label, idx in (("all", order), ("fastest tenth", order[:k10]), ("slowest tenth", order[-k10:])):
            print("   phase lengths us (%s): %s | life %.1f" % (label, " ".join("%.1f" % x for x in seg[idx].mean(axis=0)), d[idx].mean()))
    hw = rec[ok, 2] & np.uint64(0xFFFFFFFF)
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    xcc = rec[ok, 3] & np.uint64(0xF)
    print("   distinct (xcc,se,cu): %d" % len(set(zip(xcc.tolist(), se.tolist(), cu.tolist()))))
