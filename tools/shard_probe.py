"""Rank 0 of the 8-GPU layout of a BASELINE config, timed on ONE GPU (no multi-GPU box was ever available to this project).

    python tools/shard_probe.py cfg4|cfg3 [--policy active|moving] [--steps K] [--time-max T] [--out profiles/x.json]

cfg4 (weak scaling): ONE world of 65 536 robots beside 200 social-force pedestrians, rank 0 owns robots [0, 8192).
cfg3 (strong scaling): the metric's 8192-robot world with 200 ORCA pedestrians, rank 0 owns [0, 1024), robots numbered along x as
bench.py numbers them for N > 1.

A whole-world handle runs the episode first and the other seven ranks' records of every step are kept (what the all-gather
delivers: 64 bytes per robot, the eighth double carrying the footprint bitmap).  The shard then replays the episode:
imgenv_step_begin, a device-to-device copy of that step's remote records standing in for the exchange's arrival (NOT its xGMI time:
DESIGN.md section 7 prices that separately), imgenv_step_end -- and must end on the whole world's outputs for its robots.  Reported:
us per step of the shard, its per-kernel HIP-event times, and the unsharded step it is held to on the same box."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from img_env_amd.world import World  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("cfg", choices=["cfg3", "cfg4"])
ap.add_argument("--policy", choices=["active", "moving"], default="active")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--time-max", type=int, default=100)
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--out", default=None)
args = ap.parse_args()

N = args.ranks
RL = 8192 if args.cfg == "cfg4" else 8192 // N
R = RL * N
P = 200
K = args.steps
dev = torch.device("cuda", 0)
n_lay = 2 + K // (args.time_max + 1)
grid, params, layouts = bench.make_workload(args.cfg, R, P, min(n_lay, 3), robot_begin=0, robot_end=R, sort_x=True)
params["time_max"] = args.time_max
gen = torch.Generator(device=dev).manual_seed(7)


def actions(n):
    a = torch.zeros(16, n, 3, device=dev)
    if args.policy == "moving":
        a[:, :, 0] = torch.rand(16, n, generator=gen, device=dev) * 0.6
    a[:, :, 1] = torch.rand(16, n, generator=gen, device=dev) * 1.8 - 0.9
    return a


acts = actions(R)
full = World(dict(params), grid)
build_id = full.lib.imgenv_build_id().decode()
events = []  # ("reset", layout index) | ("step", s)
saved = []
elapsed, episode = 0, 0
full.reset(layouts[0])
events.append(("reset", 0))
for s in range(K):
    full.step(acts[s % 16])
    saved.append(full.records[RL:].clone())
    events.append(("step", s))
    elapsed += 1
    if elapsed > args.time_max:
        episode += 1
        full.reset(layouts[episode % len(layouts)])
        events.append(("reset", episode % len(layouts)))
        elapsed = 0
want = full.snapshot() if events[-1][0] == "step" else None
frozen_full = float(((want["is_collisions"] != 0) | (want["is_arrives"] != 0)).mean()) if want else None
frozen_local = float(((want["is_collisions"][:RL] != 0) | (want["is_arrives"][:RL] != 0)).mean()) if want else None
full_mode = full.layer_mode()
full.close()

shard = World(dict(params, robot_begin=0, robot_end=RL), grid)


def replay():
    """the episode on the shard; returns the seconds its STEPS took (resets are not timed on either side)"""
    total, t0 = 0.0, None
    for kind, q in events:
        if kind == "reset":
            torch.cuda.synchronize()
            if t0 is not None:
                total += time.perf_counter() - t0
            shard.reset(layouts[q])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        else:
            shard.step_begin(acts[q % 16][:RL])
            shard.records[RL:].copy_(saved[q])
            shard.step_end()
    torch.cuda.synchronize()
    return total + time.perf_counter() - t0


replay()  # warm-up (and the clock ramp)
got = shard.snapshot()
ok = None
if want is not None:
    ok = all(np.array_equal(got[k], want[k][:RL], equal_nan=True) for k in ("view_maps", "sensor_maps", "lasers", "is_collisions", "is_arrives",
                                                                             "ped_vector_states", "ped_maps", "rewards", "dones"))
times = sorted(replay() for _ in range(5))
us_shard = 1e6 * times[2] / K
shard.timing(1)
replay()
tm = shard.timing_read()
shard.timing(0)
kernel_us = {k: round(1e3 * ms / n, 2) for k, (ms, n) in tm.items() if n}
shard_mode = shard.layer_mode()
launches = shard.launches()
shard.close()

# the unsharded step this is held to, on the same box: 8192 robots in a world of their own (one GPU's share of cfg-4; cfg-3's N = 1 point)
g1, p1, l1 = bench.make_workload(args.cfg, 8192, P, min(n_lay, 3), robot_begin=0, robot_end=8192, sort_x=False)
p1["time_max"] = args.time_max
one = World(dict(p1), g1)
a1 = actions(8192)


def run_one():
    one.reset(l1[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    el = 0
    for s in range(K):
        one.step(a1[s % 16])
        el += 1
        if el > args.time_max:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            one.reset(l1[1 % len(l1)])
            torch.cuda.synchronize()
            t0 += time.perf_counter() - t1  # (resets are not timed on either side)
            el = 0
    torch.cuda.synchronize()
    return time.perf_counter() - t0


run_one()
us_one = 1e6 * sorted(run_one() for _ in range(5))[2] / K
snap1 = one.snapshot()
frozen_one = float(((snap1["is_collisions"] != 0) | (snap1["is_arrives"] != 0)).mean())
one_mode = one.layer_mode()
one.close()
line = dict(probe="rank 0 of %d of %s on one GPU" % (N, args.cfg), robots_world=R, robots_local=RL, robots_remote=R - RL, peds=P, policy=args.policy,
            steps=K, time_max=args.time_max, us_per_step_shard=us_shard, us_per_step_unsharded_8192=us_one, ratio=us_shard / us_one,
            shard_matches_whole_world=ok, frozen_fraction_at_the_end=frozen_full, frozen_fraction_of_the_shards_robots=frozen_local,
            frozen_fraction_of_the_unsharded_world=frozen_one,
            note="a frozen robot (collided or arrived) skips its view: compare the two step times only beside their frozen fractions -- cfg-4's "
                 "65 536-robot world on 0.5 m cells is dense enough for most robots to stand in another robot's cells from the reset on", per_rank_kernel_us=kernel_us, launches_per_step=launches,
            shard_mode=shard_mode, whole_world_mode=full_mode, unsharded_mode=one_mode, exchange_bytes_per_step=64 * R,
            exchange_stand_in="device-to-device copy of the %d remote records between step_begin and step_end" % (R - RL),
            build_id=build_id)
print(json.dumps(line))
if args.out:
    with open(os.path.join(ROOT, args.out), "w") as fh:
        json.dump(line, fh, indent=1)
