"""The Python mirror (make_env / ImageEnv / wrapper stack) against the golden vectors recorded from the
reference's own make_env + wrapper stack (tests/golden/gen_python_golden.py)."""
import ast
import glob
import os

import numpy as np
import pytest

from scenarios import golden_scenario

pytestmark = pytest.mark.gpu
FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "python_post_*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_make_env_stack_matches_reference(path):
    import torch
    from img_env_amd import make_env, worldgen
    z = np.load(path)
    meta = ast.literal_eval(str(z["meta"]))
    grid, params, layout = golden_scenario(meta)
    cfg = worldgen.make_yaml_cfg(meta["n_robots"], meta["n_peds"], grid, time_max=meta["time_max"],
                                 ped_shape=meta["ped_shape"], state_dim=meta["state_dim"], n_obstacles=meta["n_obstacles"])
    env = make_env(cfg)   # VelAction, TimeLimit, SensorsPaperReward, InfoLog, MultiRobotClean
    try:
        st = env.reset(layout=layout)
        assert len(st) == meta["n_robots"] == env.robot_total
        assert np.array_equal(st.numpy().sensor_maps, z["exp_sensor_maps"][0])
        for s in range(meta["steps"]):
            act = torch.as_tensor(z["actions"][s], dtype=torch.float32, device="cuda")
            st, rew, done, info = env.step(act)
            h = st.numpy()
            assert np.array_equal(h.is_collisions, z["exp_is_collisions"][s + 1])
            assert np.array_equal(h.is_arrives, z["exp_is_arrives"][s + 1])
            assert h.vector_states.dtype == np.float64 and h.sensor_maps.dtype == np.float16
            assert np.abs(h.vector_states - z["exp_vector_states"][s + 1]).max() <= 1e-4
            assert np.abs(h.ped_maps - z["exp_ped_maps"][s + 1]).max() <= 1e-4
            assert np.abs(rew.cpu().numpy() - z["exp_rewards"][s]).max() <= 1e-4, s
            assert np.array_equal(done.cpu().numpy(), z["exp_dones"][s]), s
            assert np.array_equal(info["dones_info"].cpu().numpy(), z["exp_dones_info"][s]), s
            assert np.array_equal(info["is_clean"].cpu().numpy(), z["exp_is_clean"][s]), s
            assert np.array_equal(info["all_down"].cpu().numpy(), z["exp_all_down"][s]), s
            assert np.abs(info["speeds"].cpu().numpy() - z["exp_speeds"][s]).max() <= 1e-6, s
    finally:
        env.close()


def test_reference_action_objects_and_random_spawn():
    """List[ContinuousAction] in, random EnvPos-style spawn, NeverStopWrapper auto-reset"""
    from img_env_amd import ContinuousAction, make_env, worldgen
    grid = worldgen.make_grid(200, 1)
    cfg = worldgen.make_yaml_cfg(6, 4, grid, time_max=5, n_obstacles=2, seed=3,
                                 wrappers=["VelActionWrapper", "TimeLimitWrapper", "SensorsPaperRewardWrapper",
                                           "InfoLogWrapper", "MultiRobotCleanWrapper", "StateBatchWrapper",
                                           "ObsLaserStateTmp", "NeverStopWrapper"], image_batch=2, state_batch=3)
    env = make_env(cfg)
    try:
        obs = env.reset()
        assert obs[0].shape == (6, 1, 360) and obs[1].shape == (6, 9) and obs[2].shape == (6, 3, 48, 48)
        resets = 0
        for s in range(14):
            obs, rew, done, info = env.step([[0.3, 0.1]] * 6)
            resets += int(bool(info["all_down"][0]))
        assert resets == 2            # time_max 5 -> every 6th step ends the episode
        assert cfg["node_id"] == 1    # make_env post-increments node_id (envs/__init__.py:32)
    finally:
        env.close()
