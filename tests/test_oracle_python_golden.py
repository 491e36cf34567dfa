"""Oracle (Python half) vs golden vectors produced by the reference's own Python
(tests/golden/gen_python_golden.py ran envs.make_env + ImageEnv + the wrapper stack unmodified)."""
import ast
import glob
import os

import numpy as np
import pytest

from oracle_binding import OracleWorld
from scenarios import clip_actions, golden_scenario

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "python_post_*.npz")))


def test_fixtures_present():
    assert len(FIXTURES) >= 4


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_python_half_matches_reference_python(oracle_lib, path):
    z = np.load(path)
    meta = ast.literal_eval(str(z["meta"]))
    grid, params, layout = golden_scenario(meta)
    w = OracleWorld(params, grid)
    try:
        outs = [w.reset(layout)]
        snaps = [w.snapshot()]
        pinfo = [w.pedinfo()]
        for s in range(meta["steps"]):
            w.step(clip_actions(z["actions"][s]))
            snaps.append(w.snapshot())
            pinfo.append(w.pedinfo())
    finally:
        w.close()
    T = meta["steps"] + 1
    # 1. the C++-half inputs the fixture was generated from are reproduced (else: regenerate)
    for t in range(T):
        for k in ("vector_states", "lasers_raw", "view_maps", "is_collisions", "is_arrives"):
            assert np.array_equal(snaps[t][k], z["in_" + k][t]), (k, t)
        assert np.array_equal(pinfo[t], z["in_pedinfo"][t])
    # 2. what the reference Python made of them
    for t in range(T):
        s = snaps[t]
        assert np.array_equal(s["vector_states"].astype(np.float64), z["exp_vector_states"][t])
        assert np.array_equal(s["sensor_maps"], z["exp_sensor_maps"][t])          # float16, bit-exact
        assert np.array_equal(s["is_collisions"], z["exp_is_collisions"][t])
        assert np.array_equal(s["is_arrives"].astype(bool), z["exp_is_arrives"][t])
        if s["lasers"].size and z["exp_lasers"][t].size:
            assert np.array_equal(s["lasers"], z["exp_lasers"][t])
        assert np.array_equal(s["ped_vector_states"], z["exp_ped_vector_states"][t])
        assert np.array_equal(s["ped_maps"], z["exp_ped_maps"][t])
        assert np.array_equal(s["step_ds"], z["exp_step_ds"][t])
        assert np.array_equal(s["ped_min_dists"], z["exp_ped_min_dists"][t].astype(np.float64))
    for t in range(1, T):
        s = snaps[t]
        assert np.array_equal(s["rewards"], z["exp_rewards"][t - 1]), t
        assert np.array_equal(s["dones"], z["exp_dones"][t - 1])
        assert np.array_equal(s["dones_info"], z["exp_dones_info"][t - 1])
        assert np.array_equal(s["is_clean"].astype(bool), z["exp_is_clean"][t - 1])
        # _step_req: alive = (dones == 0) from the previous ImageEnv.step (yaml_env.py:319-331)
        assert np.array_equal(snaps[t - 1]["base_dones"] == 0, z["in_alive"][t - 1])
