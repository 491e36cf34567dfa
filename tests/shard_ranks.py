"""Two REAL processes, each holding a HIP handle on a shard of one world -- the multi-GPU protocol with the library itself as
every rank's compute (tests/test_sharded_gloo.py runs the same protocol with the oracle as a stand-in, on CPU).

    python tests/shard_ranks.py launch OUT_DIR      the launcher: never touches a GPU; starts rank 0, rank 1 and the unsharded
                                                    reference world as three child processes, waits, writes OUT_DIR/result.json
    python tests/shard_ranks.py rank K OUT_DIR PORT one rank (or K = -1: the whole world in one handle)
    python tests/shard_ranks.py launch_rccl OUT_DIR N   the same with N ranks on N GPUs and the exchange INSIDE the library:
                                                    imgenv_comm_init on every rank, then plain imgenv_step (ncclAllGather over
                                                    xGMI on the step's stream); needs a box with N >= 2 devices
                                                    (tests/test_gpu_rccl_ranks.py, skipped on the one-GPU boxes of this pool)

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
N_ROBOTS_RCCL = 3360  # divisible by every rank count from 2 to 8
PER_ROBOT = ("vector_states", "view_maps", "sensor_maps", "lasers", "ped_vector_states", "ped_maps", "is_collisions", "is_arrives",
             "step_ds", "ped_min_dists", "rewards", "dones", "dones_info", "robot_pose")


def _scenario(n_robots=None):
    import numpy as np
    from img_env_amd import worldgen
    N_ROBOTS = n_robots or globals()["N_ROBOTS"]
    grid = worldgen.make_grid(400, SEED)
    params = worldgen.make_params(N_ROBOTS, N_PEDS, res=0.125, time_max=8)
    layouts = [worldgen.make_layout(grid, 0.125, N_ROBOTS, N_PEDS, seed=SEED + 1 + q, n_obstacles=3, clearance=0.6) for q in range(2)]
    rng = np.random.default_rng(SEED)
    acts = [np.stack([rng.uniform(0, 0.6, N_ROBOTS), rng.uniform(-0.9, 0.9, N_ROBOTS), np.zeros(N_ROBOTS)], 1).astype(np.float32)
            for _ in range(STEPS)]
    return grid, params, layouts, acts


def rank_main(rank, out_dir, port, n_ranks=2, rccl=False):
    import numpy as np
    import torch
    import torch.distributed as dist
    from img_env_amd.world import World
    N_ROBOTS = N_ROBOTS_RCCL if rccl else globals()["N_ROBOTS"]
    grid, params, layouts, acts = _scenario(N_ROBOTS)
    sharded = rank >= 0
    if sharded:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        if rccl:
            torch.cuda.set_device(rank)  # one process per GPU
        dist.init_process_group("nccl" if rccl else "gloo", rank=rank, world_size=n_ranks)
        rl = N_ROBOTS // n_ranks
        r0, r1 = rank * rl, (rank + 1) * rl
        params = dict(params, robot_begin=r0, robot_end=r1)
    else:
        r0, r1 = 0, N_ROBOTS
    w = World(params, grid, device=rank if (rccl and sharded) else 0)
    assert w.lib.imgenv_backend() == b"hip-gfx950"
    if rccl and sharded:  # the library's own communicator: the id travels through torch.distributed (plumbing), the exchange does not
        w.init_comm(rank, n_ranks)
        assert w.comm_info() == (n_ranks, rank), w.comm_info()
    snaps = []
    w.reset(layouts[0])
    snaps.append(w.snapshot())
    for s, a in enumerate(acts):
        if s == STEPS // 2:  # a reset in mid-flight: every rank knows every robot's start from the batch, no exchange
            w.reset(layouts[1])
            snaps.append(w.snapshot())
        if sharded and rccl:
            w.step(a[r0:r1])  # imgenv_step = step_begin; ncclAllGather of the robot records in place; step_end
        elif sharded:
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


def launch(out_dir, n_ranks=2, rccl=False):
    import numpy as np
    N_ROBOTS = N_ROBOTS_RCCL if rccl else globals()["N_ROBOTS"]
    os.makedirs(out_dir, exist_ok=True)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "rank", str(k), out_dir, str(port), str(n_ranks), "rccl" if rccl else "gloo"],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for k in list(range(n_ranks)) + [-1]]
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
        rl = N_ROBOTS // n_ranks
        n_snaps = STEPS + 2
        for rank in range(n_ranks):
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
    res["robots"], res["steps"], res["ranks"], res["exchange"] = N_ROBOTS, STEPS, n_ranks, "rccl (in-library)" if rccl else "gloo (caller)"
    with open(os.path.join(out_dir, "result.json.tmp"), "w") as f:
        json.dump(res, f)
    os.replace(os.path.join(out_dir, "result.json.tmp"), os.path.join(out_dir, "result.json"))


if __name__ == "__main__":
    if sys.argv[1] == "launch":
        launch(sys.argv[2])
    elif sys.argv[1] == "launch_rccl":
        launch(sys.argv[2], int(sys.argv[3]), rccl=True)
    else:
        rank_main(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 2,
                  len(sys.argv) > 6 and sys.argv[6] == "rccl")
