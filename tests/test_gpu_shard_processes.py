"""The robot-sharded world with the HIP library as every rank's compute, in two real processes (tests/shard_ranks.py):
imgenv_step_begin -> all-gather of the robot records across the processes -> imgenv_step_end, each rank's outputs bit for bit
those of its slice of the unsharded HIP world.  The job is started by conftest.py at session start (before this process touches
the GPU) and runs underneath the other tests; here its verdict is read."""
import json
import os

import pytest

from conftest import SHARD_JOB

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1200)
def test_two_hip_rank_processes_match_the_unsharded_world():
    assert SHARD_JOB.get("proc") is not None, "conftest.py did not start the shard job (no GPU visible at session start?)"
    rc = SHARD_JOB["proc"].wait(timeout=1000)
    path = os.path.join(SHARD_JOB["dir"], "result.json")
    assert rc == 0 and os.path.exists(path), ("launcher exited with", rc)
    res = json.load(open(path))
    assert res["rcs"] == [0, 0, 0], (res["rcs"], res["logs"])
    assert res["ok"], res["mismatches"]
    assert res["robot_robot_collisions"] > 0  # robots of different shards met: the exchange mattered
    print("two HIP rank processes: %d robots, %d steps, %.1f s, %d robot-robot collisions, %d collided at the end"
          % (res["robots"], res["steps"], res["seconds"], res["robot_robot_collisions"], res["collided_at_end"]))
