#!/usr/bin/env python3
"""Throughput of E independent worlds in one handle (imgenv_cfg.n_worlds): the reference's env_num idiom, batched.

    python tools/multiworld_probe.py --worlds 64 --robots 128 --peds 16 --grid 200

Every world has its own obstacle map, crowd and time limit; world k is reset on its own (imgenv_reset_world) when its time
limit has run out, staggered so that a few worlds reset on every step.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stack_params(params, W):
    p = dict(params)
    for k in ("robot_shape", "robot_size", "robot_sensor_cfg", "robot_size_last", "ped_shape", "ped_size", "ped_max_speed"):
        p[k] = np.concatenate([np.asarray(p[k])] * W, axis=0)
    p["n_robots"], p["n_peds"], p["n_worlds"] = params["n_robots"] * W, params["n_peds"] * W, W
    return p


def measure(E, Rw, Pw, grid_cells, res=0.125, steps=300, warmup=100, time_max=100, policy="active", kernels=True, resets=True,
            clearance=None, device=0, flags=0, scene="rvoscene"):
    import torch
    from img_env_amd import worldgen
    from img_env_amd.world import World
    dev = torch.device("cuda", device)
    grid = worldgen.make_grid(grid_cells, 0)
    if clearance is None:
        clearance = 1.0 if Rw * 1.0 <= 0.25 * (grid_cells * res) ** 2 else 0.7
    params = worldgen.make_params(Rw, Pw, res=res, view_cells=48, beams=360, scene=scene, time_max=time_max, flags=flags)
    layouts = [worldgen.make_layout(grid, res, Rw, Pw, seed=500 + s, clearance=clearance) for s in range(8)]
    world = World(stack_params(params, E), grid, device=device)
    R = E * Rw
    prepared = [world.prepare_reset(lay) for lay in layouts]  # C-ABI batches, built once
    g = torch.Generator(device=dev).manual_seed(1)
    n_act = 16
    acts = torch.zeros(n_act, R, 3, device=dev)
    if policy == "episode":
        acts[:, :, 0] = torch.rand(n_act, R, generator=g, device=dev) * 0.6
    acts[:, :, 1] = torch.rand(n_act, R, generator=g, device=dev) * 1.8 - 0.9
    # world k starts its first episode "k (time_max+1) / E steps ago": resets spread evenly over the steps
    period = time_max + 1
    world.reset([layouts[k % len(layouts)] for k in range(E)])
    due = [[] for _ in range(period)]
    for k in range(E):
        due[(k * period) // E].append(k)
    st = dict(step=0, n_reset=0, reset_s=0.0)

    def do_step():
        s = st["step"]
        world.step(acts[s % n_act], actions_ready=True)  # (pre-generated, resident in HBM)
        st["step"] = s + 1
        if resets:
            ks = due[(s + 1) % period]
            if ks:  # every world whose time limit ran out on this step, in one call
                t0 = time.perf_counter()
                world.reset_worlds(ks, [prepared[(k + st["n_reset"]) % len(prepared)] for k in ks])
                st["n_reset"] += len(ks)
                st["reset_s"] += time.perf_counter() - t0

    for _ in range(warmup):
        do_step()
    per_kernel = None
    if kernels:
        world.timing(1)
        for _ in range(40):
            do_step()
        torch.cuda.synchronize()
        tm = world.timing_read()
        per_kernel = {k: round(1e3 * ms / n, 2) for k, (ms, n) in tm.items() if n}
        world.timing(0)
    torch.cuda.synchronize()
    st["n_reset"], st["reset_s"] = 0, 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        do_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frozen = float(((world.out["is_collisions"] != 0) | (world.out["is_arrives"] != 0)).float().mean().item())
    if world._trace is not None and world._trace[1]:
        print("python side: %.1f us per reset call inside the C call, %.1f us around it"
              % (1e6 * world._trace[0] / world._trace[1], 1e6 * (st["reset_s"] - world._trace[0]) / max(world._trace[1], 1)), file=sys.stderr)
    world.close()
    return dict(worlds=E, robots_per_world=Rw, peds_per_world=Pw, grid=grid_cells, resolution=res, policy=policy,
                value=R * steps / dt, unit="robot-steps/s", us_per_step=1e6 * dt / steps, steps=steps,
                world_resets=st["n_reset"], host_us_per_world_reset=(1e6 * st["reset_s"] / st["n_reset"]) if st["n_reset"] else None,
                frozen_now=frozen, kernel_us=per_kernel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, default=64)
    ap.add_argument("--robots", type=int, default=128, help="per world")
    ap.add_argument("--peds", type=int, default=16, help="per world")
    ap.add_argument("--grid", type=int, default=200)
    ap.add_argument("--res", type=float, default=0.125)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--policy", default="active")
    ap.add_argument("--no-resets", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="IMGENV_FLAG_*: 2 composed class layer, 4 stamped")
    ap.add_argument("--scene", default="rvoscene", help="rvoscene / ervoscene / pedscene")
    args = ap.parse_args()
    print(json.dumps(measure(args.worlds, args.robots, args.peds, args.grid, args.res, args.steps, args.warmup, policy=args.policy,
                             resets=not args.no_resets, flags=args.flags, scene=args.scene)))


if __name__ == "__main__":
    main()
