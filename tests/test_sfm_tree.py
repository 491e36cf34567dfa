"""libpedsim's quadtree surgery as k_sfm does it (img_env_amd/csrc/sfm.h: sfm_surgery_* -- moves that cannot split a leaf all at
once, the rest replayed in agent order) against the literal loop, on the CPU: tests/host/sfm_tree_check.cpp compiles the very
functions the kernel calls and runs them agent by agent (the operations the kernel does under a lock per leaf in shuffled order)
beside `Tagent::move` -> `scene->moveAgent(this)` one agent at a time (ped_agent.cpp:519-571, ped_tree.cpp:131-137), on crowds that
walk through / below / across the tree's 10 m square and on crowds snapped onto its centre lines; after every step the two trees
must be the same tree.  And the literal loop itself is held to the ORACLE's tree (oracle_sfm.c, pinned on the reference's own
pedsim build) on a recorded run of the oracle."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("sfm") / "sfm_tree_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host", "sfm_tree_check.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("mode,agents,steps", [
    (0, 200, 300),  # inside the tree's square: leaves split on ~50 of the steps
    (1, 230, 300),  # below it (cfg-4's crowd): every agent has left its leaf on every step
    (2, 200, 300),  # across its lower edge
    (3, 200, 200),  # with agents snapped onto the tree's centre lines: several children per insert, whole steps replayed literally, the tree outgrows the LDS mirror
    (0, 40, 400),   # a small crowd
    (1, 256, 150),  # the largest one k_sfm takes
])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_staged_surgery_equals_the_literal_loop(checker, mode, agents, steps, seed):
    out = subprocess.run([checker, str(seed), str(agents), str(steps), str(mode)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.splitlines()[-1].startswith("OK"), out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"splits on (\d+) steps", out.stdout)
    assert int(m.group(1)) >= (3 if agents <= 40 else 20), out.stdout  # the crowds do make leaves split


@pytest.mark.parametrize("where", ["below_root", "band1", "round_numbers"])
def test_literal_loop_reproduces_the_oracles_tree(checker, tmp_path, where):
    """a recorded run of the oracle (the crowds of tests/test_gpu_parity.py's quadtree test): the tree sfm.h's functions build from
    the recorded positions -- one agent at a time, a split seeing the agents behind the mover where the step found them -- is the
    oracle's tree after every step: node count, member entries, and who is in the tree"""
    from oracle_binding import OracleWorld, build_oracle, load_oracle, set_cr_atan2
    import test_gpu_parity as T
    build_oracle()
    set_cr_atan2(True)
    try:
        steps = 12
        grid, params, layout, rng = T._quadtree_world(where, 8)
        P = layout.ped_pose.shape[0]
        r = np.zeros(2 * P, np.int32)
        load_oracle().oracle_test_glibc_rand(1, 2 * P, r.ctypes.data_as(C.POINTER(C.c_int32)))  # PedScene's own start: rand() positions (pedscene.h:57-66)
        cpu = OracleWorld(params, grid)
        cpu.reset(layout)
        rec = [r.astype(np.float64) / 2147483647.0 * 10.0, cpu.snapshot()["ped_state"][:, :2].astype(np.float64).ravel().copy()]
        want = []
        for s in range(steps):
            cpu.step(T.random_actions(rng, 8))
            rec.append(cpu.snapshot()["ped_state"][:, :2].astype(np.float64).ravel().copy())
            want.append(cpu.sfm_tree())
        cpu.close()
    finally:
        set_cr_atan2(False)
    path = str(tmp_path / "rec.bin")
    np.concatenate(rec).tofile(path)
    out = subprocess.run([checker, "0", str(P), str(steps), "9", path], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    got = re.findall(r"step (\d+): (\d+) nodes, (\d+) member entries, in the tree: (\w+) (\w+) (\w+) (\w+)", out.stdout)
    assert len(got) == steps
    for s, g in enumerate(got):
        assert (int(g[1]), int(g[2])) == want[s][:2], (where, s, g, want[s])
        assert tuple(int(v, 16) for v in g[3:]) == want[s][4:], (where, s)
    assert want[-1][0] > want[0][0] or where != "band1"  # (the band's leaves split during the recorded steps)
