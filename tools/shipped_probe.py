"""Throughput on the geometry of the reference's shipped envs/cfg/test.yaml (733 x 733 grid from a 110 x 110 pixel map, 400 x 400
cell views shrunk to 48 x 48 with INTER_CUBIC, 1000 beams, 1 robot + 4 leg pedestrians + 4 obstacles per env), E envs in one
handle (VecImageEnv, native spawn).  BASELINE.md section 2 measured the reference's C++ core alone on this geometry at 194
robot-steps/s on one CPU core (no ROS, no Python post-processing).

    python tools/shipped_probe.py [--envs 64] [--steps 100] [--view-maps]

`measure()` is the `shipped` key of the bench line."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VIEW, IMAGE, BEAMS = 400, 48, 1000
# SURVEY 8(d)'s per-robot figure for this geometry: the view-cell gather, the sensor_map (float16) and the lasers; + the
# full-size view when it is an output
ALGORITHMIC_BYTES = {"k_crop_big": VIEW * VIEW, "k_view": 4 * BEAMS, "k_taps_big": 2 * IMAGE * IMAGE, "k_fullview_big": VIEW * VIEW}


def measure(envs=256, steps=100, view_maps=False, device=0, device_reset=True):
    import torch
    from PIL import Image
    from img_env_amd import worldgen
    from img_env_amd.vec_env import VecImageEnv
    z = np.load(os.path.join(ROOT, "tests", "golden", "spawn_ref.npz"))
    sections = json.loads(str(z["test@1/cfg"]))
    tmp = tempfile.mkdtemp()
    m = np.full((110, 110), 255, np.uint8)
    m[:5] = m[-5:] = 0
    m[:, :5] = m[:, -5:] = 0
    Image.fromarray(m).save(os.path.join(tmp, "room.png"))
    cfg = worldgen.shipped_test_yaml_cfg("room.png", sections)
    cfg.update(map_dir=tmp, seed=1, keep_view_maps=bool(view_maps), device=device)
    E = envs
    vec = VecImageEnv(cfg, env_num=E, seed=1, native_spawn=True, device_reset=device_reset)
    try:
        vec.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = torch.zeros(16, E, 3, device="cuda")
        acts[:, :, 0] = torch.rand(16, E, generator=g, device="cuda") * 0.6
        acts[:, :, 1] = torch.rand(16, E, generator=g, device="cuda") * 1.8 - 0.9
        for s in range(20):
            vec.step(acts[s % 16])
        w = vec.world
        w.timing(1)  # per-kernel times of the STEP's launches alone: plain steps, no auto-reset (its chain launches the same kernels
        for s in range(6):  # over a handful of envs); the few robots that finish meanwhile sit these steps out
            w.step(acts[s % 16].reshape(-1, 3))
            torch.cuda.synchronize()  # (one step at a time, as the auto-reset loop runs them: nothing of the next step underneath)
        tm = w.timing_read()
        w.timing(0)
        vec.reset()
        for s in range(80):  # (past the 64 calls during which the mirror's default output guard checksums and synchronises)
            vec.step(acts[s % 16])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_reset = 0
        if device_reset:
            worlds, first = w.autoreset_last()
            placed0 = first + len(worlds)
        for s in range(steps):
            _, _, _, info = vec.step(acts[s % 16])
            if info["reset_envs"] is not None:  # (device-side reset: nothing comes back to the host)
                n_reset += len(info["reset_envs"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if device_reset:  # placements handed out meanwhile
            worlds, first = w.autoreset_last()
            n_reset = first + len(worlds) - placed0
    finally:
        vec.close()
    kernel_us = {k: round(1e3 * ms / n, 1) for k, (ms, n) in tm.items() if n}
    dominant = max((k for k in kernel_us if k in ALGORITHMIC_BYTES), key=kernel_us.get)
    achieved = ALGORITHMIC_BYTES[dominant] * E / (kernel_us[dominant] * 1e-6) / 1e9
    return dict(config="shipped test.yaml geometry: 733x733 grid @0.015 m, 400x400 view -> 48x48 (INTER_CUBIC), 1000 beams, "
                       "1 robot + 4 leg peds + 4 obstacles per env%s; auto-reset %s" % ("" if view_maps else "; full-size view not materialised",
                                                                                         "on the device" if device_reset else "through the host"),
                envs=E, steps=steps, env_resets=n_reset, value=E * steps / dt, unit="robot-steps/s", us_per_step=1e6 * dt / steps,
                reference_cpp_core_one_cpu_core=194, kernel_us=kernel_us,
                roofline=dict(bound="hbm", kernel=dominant, achieved=achieved, peak=8000.0, unit="GB/s", frac=achieved / 8000.0,
                              algorithmic_bytes_per_robot=ALGORITHMIC_BYTES[dominant], kernel_avg_us=kernel_us[dominant],
                              units_per_launch=E))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--view-maps", action="store_true", help="materialise the 400 x 400 view_maps output as well")
    ap.add_argument("--host-reset", action="store_true", help="imgenv_step_autoreset (host in the loop) instead of the device-side reset")
    args = ap.parse_args()
    print(json.dumps(measure(args.envs, args.steps, args.view_maps, device_reset=not args.host_reset)))
