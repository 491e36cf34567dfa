#!/usr/bin/env python3
"""Soak + determinism: two identical VecImageEnv (native spawn, auto-reset, same seeds, same actions) stepped in lockstep for
many thousand steps; every output must stay bit-identical between the two (races show up as drift), finite, and the library
must not raise.    python tools/soak.py --steps 20000"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--robots", type=int, default=4)
    ap.add_argument("--peds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--time-max", type=int, default=40)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--device-reset", action="store_true", help="imgenv_step_autoreset_device: the host out of the loop")
    ap.add_argument("--shipped", action="store_true", help="the reference's shipped test.yaml geometry (big views, crop_map) instead")
    args = ap.parse_args()
    import torch
    from img_env_amd import worldgen
    from img_env_amd.vec_env import VecImageEnv
    grid = worldgen.make_grid(200, 2)
    envs = []
    for _ in range(2):
        if args.shipped:
            import json
            import tempfile
            import numpy as np
            from PIL import Image
            z = np.load(os.path.join(ROOT, "tests", "golden", "spawn_ref.npz"))
            tmp = tempfile.mkdtemp()
            m = np.full((110, 110), 255, np.uint8)
            m[:5] = m[-5:] = 0
            m[:, :5] = m[:, -5:] = 0
            Image.fromarray(m).save(os.path.join(tmp, "room.png"))
            cfg = worldgen.shipped_test_yaml_cfg("room.png", json.loads(str(z["test@1/cfg"])))
            cfg.update(map_dir=tmp, seed=5, keep_view_maps=False)
        else:
            cfg = worldgen.make_yaml_cfg(args.robots, args.peds, grid, time_max=args.time_max, n_obstacles=3, seed=5, flags=args.flags)
        envs.append(VecImageEnv(cfg, env_num=args.envs, seed=5, native_spawn=True, device_reset=args.device_reset))
    n = len(envs[0])
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.zeros(64, n, 3, device="cuda")
    acts[:, :, 0] = torch.rand(64, n, generator=g, device="cuda") * 0.6
    acts[:, :, 1] = torch.rand(64, n, generator=g, device="cuda") * 1.8 - 0.9
    for e in envs:
        e.reset()
    t0, resets = time.perf_counter(), 0
    names = ("view_maps", "sensor_maps", "lasers", "vector_states", "ped_maps", "ped_vector_states", "rewards", "dones",
             "dones_info", "robot_pose", "ped_state", "is_collisions", "is_arrives")
    for s in range(args.steps):
        infos = [e.step(acts[s % 64])[3] for e in envs]
        if not args.device_reset:  # (device-side reset: nothing comes back to the host; the outputs below say it all)
            assert infos[0]["reset_envs"] == infos[1]["reset_envs"], (s, "different envs ended")
            resets += len(infos[0]["reset_envs"])
        if s % 250 == 249 or s == args.steps - 1:
            a, b = envs[0].world.out, envs[1].world.out
            for f in names:
                if a[f] is None or (f == "view_maps" and args.shipped):
                    continue
                assert torch.equal(a[f], b[f]), (s, f, "the two runs drifted apart")
            assert bool(torch.isfinite(a["vector_states"]).all()) and bool(torch.isfinite(a["lasers"]).all()), (s, "non-finite output")
        if s % 5000 == 4999:
            print("step %d: %d env resets, %.1f us per pair of steps, %.2f GB in use" % (
                s + 1, resets, 1e6 * (time.perf_counter() - t0) / (s + 1), torch.cuda.memory_allocated() / 1e9), flush=True)
    if args.device_reset:
        worlds, first = envs[0].world.autoreset_last()
        worlds2, first2 = envs[1].world.autoreset_last()
        assert (list(worlds), first) == (list(worlds2), first2), "the two runs handed out different placements"
        resets = first + len(worlds)
    for e in envs:
        e.close()
    print("soak ok: %d steps, %d env resets" % (args.steps, resets))


if __name__ == "__main__":
    main()
