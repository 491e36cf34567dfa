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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--robots", type=int, default=4)
    ap.add_argument("--peds", type=int, default=3)
    ap.add_argument("--obstacles", type=int, default=2)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--time-max", type=int, default=100)
    args = ap.parse_args()
    import torch
    from img_env_amd import worldgen
    from img_env_amd.vec_env import VecImageEnv
    grid = worldgen.make_grid(200, 2)
    out = {}
    for native in (False, True):
        cfg = worldgen.make_yaml_cfg(args.robots, args.peds, grid, time_max=args.time_max, n_obstacles=args.obstacles, seed=5)
        env = VecImageEnv(cfg, env_num=args.envs, seed=5, native_spawn=native)
        n = len(env)
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = torch.zeros(16, n, 3, device="cuda")
        acts[:, :, 0] = torch.rand(16, n, generator=g, device="cuda") * 0.6
        acts[:, :, 1] = torch.rand(16, n, generator=g, device="cuda") * 1.8 - 0.9
        t0 = time.perf_counter()
        env.reset()
        torch.cuda.synchronize()
        t_reset = time.perf_counter() - t0
        for s in range(args.time_max + 20):  # past the first wave of time limits: the envs drift out of phase as robots collide
            env.step(acts[s % 16])
        torch.cuda.synchronize()
        resets, t0 = 0, time.perf_counter()
        for s in range(args.steps):
            _, _, _, info = env.step(acts[s % 16])
            resets += len(info["reset_envs"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res = dict(robot_steps_per_s=n * args.steps / dt, us_per_step=1e6 * dt / args.steps, env_resets_per_step=resets / args.steps,
                   first_reset_ms=1e3 * t_reset)
        if native:  # the same steps without the reset half: what NeverStopWrapper costs on top of the step
            env.auto_reset = False
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in range(args.steps):
                env.step(acts[s % 16])
            torch.cuda.synchronize()
            res["us_per_step_without_resets"] = 1e6 * (time.perf_counter() - t0) / args.steps
        out["native_spawn" if native else "python_spawn"] = res
        env.close()
    print(json.dumps(dict(envs=args.envs, robots_per_env=args.robots, peds_per_env=args.peds, **out)))


if __name__ == "__main__":
    main()
