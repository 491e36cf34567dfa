#!/usr/bin/env python3
"""End-to-end rate of VecImageEnv (env_num reference envs in one handle, NeverStopWrapper-style auto-reset), with the episode
placements drawn by the Python EnvPos or inside the library (native_spawn).

    python tools/vec_env_probe.py --envs 1024 --robots 4 --peds 3"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(envs=1024, robots=4, peds=3, obstacles=2, steps=300, time_max=100, natives=(False, True, "device")):
    """robot-steps/s of VecImageEnv over `steps` steps, after the envs have drifted out of phase"""
    import torch
    from img_env_amd import worldgen
    from img_env_amd.vec_env import VecImageEnv
    grid = worldgen.make_grid(200, 2)
    out = {}
    for native in natives:
        cfg = worldgen.make_yaml_cfg(robots, peds, grid, time_max=time_max, n_obstacles=obstacles, seed=5)
        env = VecImageEnv(cfg, env_num=envs, seed=5, native_spawn=bool(native), device_reset=native == "device")
        n = len(env)
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = torch.zeros(16, n, 3, device="cuda")
        acts[:, :, 0] = torch.rand(16, n, generator=g, device="cuda") * 0.6
        acts[:, :, 1] = torch.rand(16, n, generator=g, device="cuda") * 1.8 - 0.9
        t0 = time.perf_counter()
        env.reset()
        torch.cuda.synchronize()
        t_reset = time.perf_counter() - t0
        for s in range(time_max + 20):  # past the first wave of time limits: the envs drift out of phase as robots collide
            env.step(acts[s % 16])
        torch.cuda.synchronize()
        placed0 = sum(env.world.autoreset_last()[::-1][0:1]) + len(env.world.autoreset_last()[0]) if native == "device" else 0
        resets, t0 = 0, time.perf_counter()
        for s in range(steps):
            _, _, _, info = env.step(acts[s % 16])
            if info["reset_envs"] is not None:  # (device-side reset: nothing comes back to the host)
                resets += len(info["reset_envs"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if native == "device":  # placements handed out meanwhile
            worlds, first = env.world.autoreset_last()
            resets = first + len(worlds) - placed0
        res = dict(robot_steps_per_s=n * steps / dt, us_per_step=1e6 * dt / steps, env_resets_per_step=resets / steps,
                   first_reset_ms=1e3 * t_reset)
        if native is True:  # the same steps without the reset half: what NeverStopWrapper costs on top of the step
            env.auto_reset = False
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(steps):
                env.step(acts[s % 16])
            torch.cuda.synchronize()
            res["us_per_step_without_resets"] = 1e6 * (time.perf_counter() - t0) / steps
        out["device_reset" if native == "device" else ("native_spawn" if native else "python_spawn")] = res
        env.close()
    return dict(envs=envs, robots_per_env=robots, peds_per_env=peds, **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--robots", type=int, default=4)
    ap.add_argument("--peds", type=int, default=3)
    ap.add_argument("--obstacles", type=int, default=2)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--time-max", type=int, default=100)
    ap.add_argument("--device-only", action="store_true", help="only the device-side auto-reset variant")
    args = ap.parse_args()
    natives = ("device",) if args.device_only else (False, True, "device")
    print(json.dumps(measure(args.envs, args.robots, args.peds, args.obstacles, args.steps, args.time_max, natives)))


if __name__ == "__main__":
    main()
