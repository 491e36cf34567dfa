"""The exchange of a robot-sharded world with MORE THAN ONE RCCL rank: N = min(device count, 8) processes, one per GPU, each a HIP
handle on its shard with the library's own communicator (imgenv_comm_init -> ncclAllGather of the robot records inside imgenv_step,
over xGMI), every rank's outputs bit for bit its slice of the unsharded world's.  Skipped on a one-GPU box (every box of this pool
so far): the test exists so that the first multi-GPU lease runs the N-rank path.  The job is started by conftest.py at session
start -- before this process touches a GPU -- like the two-process gloo job (tests/test_gpu_shard_processes.py).
Reference: the inter-robot raster that needs every robot's pose, /root/reference/src/img_env/src/img_env.cpp:620-629."""
import json
import os

import pytest

from conftest import RCCL_JOB

pytestmark = pytest.mark.gpu


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.timeout(1500)
@pytest.mark.skipif(_n_devices() < 2, reason="needs >= 2 GPUs: the in-library ncclAllGather with more than one rank")
def test_n_rccl_ranks_match_the_unsharded_world():
    assert RCCL_JOB.get("proc") is not None, "conftest.py did not start the RCCL job"
    rc = RCCL_JOB["proc"].wait(timeout=1400)
    path = os.path.join(RCCL_JOB["dir"], "result.json")
    assert rc == 0 and os.path.exists(path), ("launcher exited with", rc)
    res = json.load(open(path))
    assert res["ranks"] == RCCL_JOB["ranks"] and res["rcs"] == [0] * (res["ranks"] + 1), (res["rcs"], res["logs"])
    assert res["ok"], res["mismatches"]  # (every rank also asserted imgenv_comm_info() == (N, rank))
    assert res["robot_robot_collisions"] > 0  # robots of different shards met: the exchange mattered
    print("%d RCCL ranks: %d robots, %d steps, %.1f s, %d robot-robot collisions" % (res["ranks"], res["robots"], res["steps"], res["seconds"],
                                                                                  res["robot_robot_collisions"]))
