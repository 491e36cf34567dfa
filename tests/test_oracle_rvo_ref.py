"""Oracle RVO2/ERVO restatement vs (a) the reference's own RVO2 sources compiled unmodified into
oracle/_ref/librvo_ref.so and (b) a committed trace generated from that library."""
import os

import numpy as np
import pytest

from rvo_harness import OracleRvo, RefRvo, run_scenario

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rvo_ref_traces.npz")
CASES = {
    "plain": dict(),
    "ervo": dict(ervo=True),
    "dense": dict(n_agents=120, world=6.0, neighbor_dist=2.0, radius=0.2),
    "obstacles": dict(n_obs=8, n_agents=40, world=8.0, neighbor_dist=3.0, radius=0.15),
}


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_matches_committed_reference_trace(oracle_lib, case):
    z = np.load(GOLD)
    for seed in (0, 1):
        got = run_scenario(OracleRvo(0.25), seed, **CASES[case])
        assert np.array_equal(got, z["%s_%d" % (case, seed)])          # bit-exact float32
        brute = run_scenario(OracleRvo(0.25), seed, brute=True, **CASES[case])
        assert np.array_equal(brute, got)


@pytest.mark.skipif(not os.path.exists(RefRvo.path()), reason="oracle/_ref not built (no reference tree)")
@pytest.mark.parametrize("case", list(CASES))
def test_oracle_matches_reference_build(oracle_lib, case):
    for seed in (2, 3, 4):
        ref = run_scenario(RefRvo(0.25), seed, **CASES[case])
        got = run_scenario(OracleRvo(0.25), seed, **CASES[case])
        assert np.array_equal(got, ref)
        assert np.abs(ref[-1, :, :2] - ref[0, :, :2]).max() > 0.5       # agents really moved
