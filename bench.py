#!/usr/bin/env python3
"""bench.py -- robot-steps/sec of the img_env step() path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json metric, configs[2] = SURVEY "cfg-3"): ONE world with 8192 robots per GPU and
200 ORCA pedestrians (rvoscene) on a 400x400 occupancy grid, 48x48 sensor_map + 3-channel ped_map,
360-beam laser.  A step is one env.step() of every robot: pedestrian advance, pose integrate, rasters,
collision + crop + laser + stamp, observation + reward/done.  Inputs (actions) are resident in HBM.
Episodes end by the time limit (time_max = 100) and are followed by a full reset inside the timed
region, like NeverStopWrapper does.

Multi-GPU: the world's robots are sharded contiguously over the ranks (weak scaling: 8192 robots per
GPU); the one exchange per step is an RCCL all-gather of the robot records between pose integration and
the robot raster.  Pedestrians are advanced redundantly on every rank.

Two action policies are timed, both on the same world:
  * "active"  (the reported `value`): v = 0, w ~ U(-0.9, 0.9).  Robots turn in place, so no robot
    collides or arrives and EVERY robot-step runs the full view path -- the conservative number.
  * "episode": the reference's random policy v ~ U(0, 0.6), w ~ U(-0.9, 0.9) (env_test.py:8-19).  In a
    shared world this freezes most robots within a few steps (collided robots take the early-out of
    agent.cpp:358-360), which makes steps cheaper; reported as `episode_value` with its frozen fraction.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ROBOTS_PER_GPU = 8192
N_PEDS = 200
GRID = 400
TIME_MAX = 100


RES = 0.25        # SURVEY 8(d) density rule for 8192 robots on 400x400: R*0.25 m^2 <= 0.5*(Hg*res)^2 -> 0.25 m
CLEARANCE = 0.7   # start-to-start distance: two r=0.17 m footprints never share a 0.25 m cell at reset


def grid_cells(n_gpus):
    """400x400 for one GPU (BASELINE).  A collision-free placement needs ~1.1 m^2 per robot at this
    resolution, so the multi-GPU weak-scaling world grows its area with the robot count (BASELINE's
    65536 robots on 400x400 cannot be placed without overlapping footprints at any resolution)."""
    side = int(np.ceil(GRID * np.sqrt(n_gpus) / 8.0)) * 8
    return side


def algorithmic_bytes(P, hv=48, wv=48, beams=360, max_ped=N_PEDS):
    """SURVEY 8(d) per robot-step (the variant that materialises the f16 copy of the sensor map, as this build
    does), split by the kernel that moves them"""
    view = hv * wv + hv * wv + 2 * hv * wv + 4 * beams   # grid window gather + sensor_map u8 + f16 copy + lasers f32
    obs = (20 * P + 3 * 48 * 48 * 4 + 4 * (1 + 7 * max_ped)) if P > 0 else 0
    state = 128
    return dict(k_view=view, k_obs=obs, k_tail=state, total=view + obs + state)


def cpu_baseline(params, grid, layout, seconds=12.0):
    """the CPU oracle (literal single-thread C restatement of the reference) on the same world"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import OracleWorld
    p = dict(params)
    p["robot_begin"], p["robot_end"] = 0, p["n_robots"]
    w = OracleWorld(p, grid)
    w.reset(layout)
    rng = np.random.default_rng(1)
    R = p["n_robots"]
    steps, t0 = 0, time.perf_counter()
    while True:
        a = np.stack([np.zeros(R), rng.uniform(-0.9, 0.9, R), np.zeros(R)], 1).astype(np.float32)
        w.step(a)
        steps += 1
        if time.perf_counter() - t0 > seconds or steps >= 50:
            break
    dt = time.perf_counter() - t0
    w.close()
    return dict(value=R * steps / dt, unit="robot-steps/s", cores=1, kind="port",
                sample="%d steps of the same %d-robot / %d-ped world, v=0 policy, single thread, %.1f s"
                       % (steps, R, p["n_peds"], dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--robots-per-gpu", type=int, default=ROBOTS_PER_GPU)
    ap.add_argument("--peds", type=int, default=N_PEDS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-episode", action="store_true")
    ap.add_argument("--no-multi-world", action="store_true", help="skip the secondary env_num-style measurement")
    ap.add_argument("--spinup", type=int, default=2000, help="untimed steps before the warm-up (clock ramp)")
    ap.add_argument("--timing-mode", type=int, default=2, help="diagnostic: 0 = no HIP events in the timed pass")
    ap.add_argument("--repeat", type=int, default=0, help="diagnostic: extra timed passes, printed to stderr")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) and use the "
                    "step_begin / all_gather / step_end path even with one rank (exercises the multi-GPU code on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from img_env_amd import worldgen
    from img_env_amd.world import World

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world_size))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world_size > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)

    RL = args.robots_per_gpu
    R = RL * world_size
    P = args.peds
    res, clearance = RES, CLEARANCE
    side = grid_cells(world_size)
    grid = worldgen.make_grid(side, 0)
    n_layouts = 2 + (args.steps + args.warmup) // (TIME_MAX + 1)
    layouts = [worldgen.make_layout(grid, res, R, P, seed=100 + s, clearance=clearance) for s in range(min(n_layouts, 4))]
    if world_size > 1:
        # robots are numbered along x, so that a rank's contiguous shard is a vertical strip of the map: each rank then
        # rasterises only the robots its own strip can see (the library clips to the shard's bounding box + view reach)
        for lay in layouts:
            order = np.argsort(lay.robot_pose[:, 0], kind="stable")
            lay.robot_pose = lay.robot_pose[order].copy()
            lay.robot_goal = lay.robot_goal[order].copy()
    params = worldgen.make_params(R, P, res=res, view_cells=48, beams=360, scene="rvoscene", time_max=TIME_MAX,
                                  robot_begin=rank * RL, robot_end=(rank + 1) * RL)
    world = World(params, grid, device=local_rank)
    r0, r1 = rank * RL, (rank + 1) * RL
    native = False
    if use_dist:
        try:  # preferred: the library runs ncclAllGather itself, on the step's stream
            world.init_comm(rank, world_size)
            native = True
        except Exception as e:  # fall back to torch.distributed's collective between step_begin / step_end
            if rank == 0:
                print("bench.py: native RCCL exchange unavailable (%s); using torch.distributed.all_gather" % e, file=sys.stderr)
        flags = [None] * world_size
        dist.all_gather_object(flags, native)
        native = all(flags)

    g = torch.Generator(device=dev).manual_seed(1 + rank)
    n_act = 16

    def make_actions(policy):
        # generated on the device: a pageable host-to-device copy here leaves a deferred un-pin behind in the HIP runtime
        # that was seen to stall kernel submission for ~30 ms a few dozen steps later
        a = torch.zeros(n_act, RL, 3, device=dev)
        if policy == "episode":
            a[:, :, 0] = torch.rand(n_act, RL, generator=g, device=dev) * 0.6
        a[:, :, 1] = torch.rand(n_act, RL, generator=g, device=dev) * 1.8 - 0.9
        return a

    state = dict(elapsed=0, episode=0)

    def do_reset():
        world.reset(layouts[state["episode"] % len(layouts)])
        state["episode"] += 1
        state["elapsed"] = 0

    def do_step(a):
        if use_dist and not native:
            world.step_begin(a)
            dist.all_gather_into_tensor(world.records, world.records[r0:r1])
            world.step_end()
        else:
            world.step(a)
        state["elapsed"] += 1
        if state["elapsed"] > TIME_MAX:  # TimeLimitWrapper has set done for every robot: NeverStopWrapper resets
            do_reset()

    def run(policy, steps, warmup, timing_mode=0, which=-1):
        acts = make_actions(policy)
        state["episode"] = 0
        do_reset()
        for s in range(warmup):
            do_step(acts[s % n_act])
        frozen0 = int(world.out["counters"][3].item())
        world.timing(timing_mode, which)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if os.environ.get("BENCH_TRACE"):  # diagnostic: where inside a pass does the time go (adds a sync every 20 steps)
            tw, marks = t0, []
            if os.environ.get("BENCH_TRACE") == "2":
                world.timing(1)
                prevk = world.timing_read()
            for s in range(steps):
                e0 = state["episode"]
                do_step(acts[s % n_act])
                if state["episode"] != e0:
                    marks.append(s)
                if s % 20 == 19:
                    torch.cuda.synchronize()
                    now = time.perf_counter()
                    print("  trace %s steps %d-%d: %.1f us/step%s" % (policy, s - 19, s, 1e6 * (now - tw) / 20,
                                                                  " reset@%s" % marks if marks else ""), file=sys.stderr)
                    if os.environ.get("BENCH_TRACE") == "2":
                        curk = world.timing_read()
                        print("      kernels: " + " ".join("%s %.0f" % (k[2:], 1e3 * (curk[k][0] - prevk[k][0]) / max(curk[k][1] - prevk[k][1], 1))
                                                         for k in curk), file=sys.stderr)
                        prevk = curk
                        now = time.perf_counter()
                    tw, marks = now, []
        else:
            for s in range(steps):
                do_step(acts[s % n_act])
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        tm = world.timing_read()
        world.timing(0)
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        frozen = int(world.out["counters"][3].item()) - frozen0
        return dt, tm, frozen / float(steps * RL)

    # pass 1 (not the headline): per-kernel breakdown, HIP events around every kernel, one sync per step;
    # the median over steps is robust against first-launch and host-submission hiccups
    def kernel_breakdown(steps=40):
        acts = make_actions("active")
        state["episode"] = 0
        do_reset()
        for s in range(5):
            do_step(acts[s % n_act])
        world.timing(1)
        prev = {k: (0.0, 0) for k in world.timing_read()}
        samples = {k: [] for k in prev}
        for s in range(steps):
            do_step(acts[s % n_act])
            cur = world.timing_read()
            for k, (ms, n) in cur.items():
                if n > prev[k][1]:
                    samples[k].append(1e3 * (ms - prev[k][0]) / (n - prev[k][1]))
            prev = cur
        world.timing(0)
        return {k: (float(np.median(v)) if v else 0.0) for k, v in samples.items()}

    per_kernel_us = kernel_breakdown()
    # dominant = the per-robot kernel with the longest launch.  k_orca is left out: it is 200 single-wave workgroups of
    # serial LP code that idle along on a side stream underneath k_view (latency, not work) and moves no per-robot bytes
    dominant = max((k for k in per_kernel_us if k != "k_orca"), key=per_kernel_us.get)
    dom_id = list(per_kernel_us).index(dominant)
    # untimed spin-up: the first ~second of sustained work after start-up runs at lower clocks than steady state
    if args.spinup > 0:
        run("active", args.spinup, 0)
    # pass 2: THE timed region; HIP events only around the dominant kernel
    dt, tm, frozen_active = run("active", args.steps, args.warmup, timing_mode=args.timing_mode, which=dom_id)
    dom_ms, dom_n = tm[dominant]
    value = R * args.steps / dt
    for q in range(args.repeat):
        for mode in (0, 2):
            d2, _, _ = run("active", args.steps, args.warmup, timing_mode=mode, which=dom_id)
            d3, _, _ = run("episode", args.steps, args.warmup, timing_mode=mode, which=dom_id)
            if rank == 0:
                print("repeat %d timing mode %d: active %.1f us/step, episode %.1f us/step"
                      % (q, mode, 1e6 * d2 / args.steps, 1e6 * d3 / args.steps), file=sys.stderr)
    episode = None
    if not args.no_episode:
        dte, _, frozen_ep = run("episode", args.steps, args.warmup)
        episode = dict(value=R * args.steps / dte, frozen_fraction=frozen_ep)

    launches_per_step = world.launches()
    multi_world = None
    if world_size == 1 and not args.no_multi_world and not args.force_dist:
        # secondary: the reference's own env_num idiom -- E independent worlds of R/E robots at
        # 0.125 m in ONE handle, every world with its own obstacle map, crowd and time limit, reset on its own when its
        # time limit runs out (imgenv_reset_worlds); no collective, so N GPUs simply hold N times the worlds
        world.close()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from multiworld_probe import measure
        try:
            multi_world = [measure(E, RL // E, pw, 200, res=0.125, steps=args.steps, warmup=300, time_max=TIME_MAX, kernels=False,
                                   device=local_rank) for E, pw in ((64, 16), (RL, 0))]
        except Exception as e:  # a secondary number must never cost the headline line
            multi_world = {"error": repr(e)}

    if rank == 0:
        ab = algorithmic_bytes(P)
        kernel_bytes = ab.get(dominant, ab["total"]) * RL
        dur_s = (dom_ms / dom_n) * 1e-3 if dom_n else float("nan")
        achieved = kernel_bytes / dur_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(dominant, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "robot-steps/sec (whole node) at 8192 robots, 48x48 maps, 360 lasers",
            "value": value, "unit": "robot-steps/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg-3: one world, %d robots/GPU x %d GPUs, %d ORCA peds (rvoscene), %dx%d grid "
                                   "@%.3f m, 48x48 sensor_map + 3ch ped_map, 360-beam laser, time_max %d, reset included"
                                   % (RL, world_size, P, side, side, res, TIME_MAX),
                       "robots": R, "peds": P, "grid": side, "resolution": res, "view": 48, "beams": 360,
                       "policy": "active: v=0, w~U(-0.9,0.9): every robot-step runs the full view path",
                       "parallelism": ("robot-sharded x%d, RCCL all-gather of robot records (%s)" % (
                           world_size, "ncclAllGather inside imgenv_step" if native else "torch.distributed between step_begin/step_end"))
                       if use_dist else "single GPU"},
            "frozen_fraction": frozen_active,
            "multi_world": multi_world,
            "episode_policy": episode,
            "kernel_us": per_kernel_us,
            "launches_per_step": launches_per_step,
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "algorithmic_bytes_per_robot_step": ab, "kernel_avg_us": dur_s * 1e6,
                         "units_per_launch": RL,
                         # the whole path against the same peak: all algorithmic bytes of a step over the whole step time
                         "path_achieved": ab["total"] * value / world_size / 1e9,
                         "path_frac": ab["total"] * value / world_size / 1e9 / 8000.0},
        }
        if world_size == 1 and not args.no_cpu_baseline:
            p1 = dict(params)
            try:
                out["cpu_baseline"] = cpu_baseline(p1, grid, layouts[0])
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "unit": "robot-steps/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    world.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
