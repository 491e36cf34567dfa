"""Two REAL processes, each holding a HIP handle on a shard of one world -- the multi-GPU protocol with the library itself as
every rank's compute (tests/test_sharded_gloo.py runs the same protocol with the oracle as a stand-in, on CPU).

    python tests/shard_ranks.py launch OUT_DIR      the launcher: never touches a GPU; starts rank 0, rank 1 and the unsharded
                                                    reference world as three child processes, waits, writes OUT_DIR/result.json
    python tests/shard_ranks.py rank K OUT_DIR PORT one rank (or K = -1: the whole world in one handle)

Every rank: imgenv_create(robot_begin, robot_end) on device 0 -> per step imgenv_step_begin (pedestrian advance + integrate of
the local robots), the robot records all-gathered across the two processes (staged through the host, gloo: the box has one
GPU, so RCCL between two processes on it is not available), imgenv_step_end (rasters + views).  The launcher is started by
tests/conftest.py at session start, BEFORE the pytest process makes its first GPU call (nothing is started from a process that
holds a GPU context); tests/test_gpu_shard_processes.py reads its verdict.  Reference: one node per env process,
/root/reference/create_launch.py:25-34; the robot-against-robot test that needs the exchange, img_env.cpp:620-629."""
import json
import os
import socket
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

N_ROBOTS, N_PEDS, STEPS, SEED = 3000, 40, 12, 23
PER_ROBOT = ("vector_states", "view_maps", "sensor_maps", "lasers", "ped_vector_states", "ped_maps", "is_collisions", "is_arrives",
             "step_ds", "ped_min_dists", "rewards", "dones", "dones_info", "robot_pose")


def _scenario():
    import numpy as np
    from img_env_amd import worldgen
    grid = worldgen.make_grid(400, SEED)
    params = worldgen.make_params(N_ROBOTS, N_PEDS, res=0.125, time_max=8)
    layouts = [worldgen.make_layout(grid, 0.125, N_ROBOTS, N_PEDS, seed=SEED + 1 + q, n_obstacles=3, clearance=0.6) for q in range(2)]
    rng = np.random.default_rng(SEED)
    acts = [np.stack([rng.uniform(0, 0.6, N_ROBOTS), rng.uniform(-0.9, 0.9, N_ROBOTS), np.zeros(N_ROBOTS)], 1).astype(np.float32)
            for _ in range(STEPS)]
    return grid, params, layouts, acts


def rank_main(rank, out_dir, port):
    import numpy as np
    import torch
    import torch.distributed as dist
    from img_env_amd.world import World
    grid, params, layouts, acts = _scenario()
    sharded = rank >= 0
    if sharded:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=2)
        rl = N_ROBOTS // 2
        r0, r1 = rank * rl, (rank + 1) * rl
        params = dict(params, robot_begin=r0, robot_end=r1)
    else:
        r0, r1 = 0, N_ROBOTS
    w = World(params, grid, device=0)
    assert w.lib.imgenv_backend() == b"hip-gfx950"
    snaps = []
    w.reset(layouts[0])
    snaps.append(w.snapshot())
    for s, a in enumerate(acts):
        if s == STEPS // 2:  # a reset in mid-flight: every rank knows every robot's start from the batch, no exchange
            w.reset(layouts[1])
            snaps.append(w.snapshot())
        if sharded:
            w.step_begin(a[r0:r1])
            torch.cuda.synchronize()
            mine = w.records[r0:r1].cpu().contiguous()
            full = torch.empty(N_ROBOTS, mine.shape[1], dtype=mine.dtype)
            dist.all_gather_into_tensor(full, mine)
            w.records.copy_(full.to(w.records.device))
            w.step_end()
        else:
            w.step(a)
        snaps.append(w.snapshot())
    np.savez(os.path.join(out_dir, "rank_%d.npz" % rank), **{"%s@%d" % (k, i): v for i, sn in enumerate(snaps) for k, v in sn.items()
                                                              if k in PER_ROBOT or k == "ped_state"})
    w.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


def launch(out_dir):
    import numpy as np
    os.makedirs(out_dir, exist_ok=True)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "rank", str(k), out_dir, str(port)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for k in (0, 1, -1)]
    logs, rcs = [], []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            pr.kill()
            out, _ = pr.communicate()
        logs.append(out.decode(errors="replace")[-2000:])
        rcs.append(pr.returncode)
    res = dict(ok=False, rcs=rcs, seconds=None, mismatches=[], logs=logs if any(rcs) else [])
    if not any(rcs):
        ref = np.load(os.path.join(out_dir, "rank_-1.npz"))
        rl = N_ROBOTS // 2
        n_snaps = STEPS + 2
        moved = False
        for rank in (0, 1):
            got = np.load(os.path.join(out_dir, "rank_%d.npz" % rank))
            for i in range(n_snaps):
                for k in PER_ROBOT:
                    a, b = got["%s@%d" % (k, i)], ref["%s@%d" % (k, i)][rank * rl:(rank + 1) * rl]
                    if not np.array_equal(a, b, equal_nan=True):
                        res["mismatches"].append([rank, i, k, int((np.asarray(a) != np.asarray(b)).sum())])
                if not np.array_equal(got["ped_state@%d" % i], ref["ped_state@%d" % i]):  # replicated crowd: identical on every rank
                    res["mismatches"].append([rank, i, "ped_state", -1])
        last = ref["is_collisions@%d" % (n_snaps - 1)]
        res["robot_robot_collisions"] = int((ref["is_collisions@%d" % (STEPS // 2)] == 3).sum())  # what needs the exchange
        res["collided_at_end"] = int((last != 0).sum())
        res["ok"] = not res["mismatches"]
        res["mismatches"] = res["mismatches"][:20]
    res["seconds"] = time.time() - t0
    res["robots"], res["steps"] = N_ROBOTS, STEPS
    with open(os.path.join(out_dir, "result.json.tmp"), "w") as f:
        json.dump(res, f)
    os.replace(os.path.join(out_dir, "result.json.tmp"), os.path.join(out_dir, "result.json"))


if __name__ == "__main__":
    if sys.argv[1] == "launch":
        launch(sys.argv[2])
    else:
        rank_main(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]))
