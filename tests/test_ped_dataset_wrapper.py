"""PedTrajectoryDatasetWrapper (the feeder of the ``dataset`` pedestrian scene) against what the reference's own class made of the
same recorded-trajectory file: tests/golden/ped_dataset_ref.npz, written by tests/golden/gen_ped_dataset_golden.py from the
imported reference (envs/wrapper/evaluation_wrapper/PedTrajectoryDatasetWrapper.py:15-291).  Host logic only: no GPU."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ped_dataset_ref.npz")


class _Env:
    def __init__(self):
        self.resets, self.speeds = [], None

    def step(self, action):
        return None, None, None, {"speeds": [self.speeds]}

    def reset(self, **kw):
        self.resets.append(kw)
        return "state"


def _wrapper(tmp_path, z, **over):
    from img_env_amd.envs import PedTrajectoryDatasetWrapper
    path = str(tmp_path / "world.csv")
    np.savetxt(path, z["csv"], delimiter=",", fmt="%.17g")
    cfg = dict(control_hz=0.4, ped_traj_dataset=path, repeated_time_per_env=2, ped_dataset_worlds=z["worlds"].tolist(), ped_sim=dict(total=0),
               node_id=0, output_file=str(tmp_path / "log.txt"), offset=z["offset"].tolist(), swapxy=True, fps=15, start_t=0, max_time=20,
               scale_x=1, scale_y=1, spawn_delay_s=0)
    cfg.update(over)
    env = _Env()
    return PedTrajectoryDatasetWrapper(env, cfg), env, cfg


def test_series_worlds_and_episode_records_match_the_reference(tmp_path):
    z = np.load(GOLDEN)
    w, env, cfg = _wrapper(tmp_path, z)
    assert cfg["ped_sim"]["total"] == int(z["total0"])  # the first world's pedestrian count goes into the config
    w.reset()
    n = sum(1 for k in z.files if k.startswith("series_"))
    for e in range(n):
        got = np.array(env.resets[-1]["cur_ped_pos_v_datas"], dtype=np.float64)
        assert np.array_equal(got, z["series_%d" % e]), e       # [pedestrian][step][x, y, theta, vx, vy], bit for bit
        assert w.cur_world == int(z["world_%d" % e])             # a new world after repeated_time_per_env episodes
        for v, ww in z["cmds_%d" % e]:
            env.speeds = (float(v), float(ww))
            w.step(None)
        code = int(z["code_%d" % e])
        if e + 1 < n:
            w.reset(dones_info=[code])
        else:
            w.out2logfile([code])
    assert open(cfg["output_file"]).read() == str(z["log"])      # arrive / pedestrian collision / stuck + the path figures, line by line
    with pytest.raises(SystemExit):                               # every world done: the reference exits
        w.reset()


def test_a_track_holds_its_first_pose_until_the_pedestrian_appears(tmp_path):
    z = np.load(GOLDEN)
    w, env, _ = _wrapper(tmp_path, z)
    series = np.array(w.change_world())
    assert series.ndim == 3 and series.shape[2] == 5
    assert (series[:, 0, 3:] == 0).all()                          # everybody starts at rest
    late = [s for s in series if (s[:3, :2] == s[0, :2]).all()]   # somebody appears after the world's first pedestrian: waits in place
    assert late
    assert np.isfinite(series).all()


def test_make_env_knows_the_wrapper():
    from img_env_amd.envs import wrapper_dict
    assert "PedTrajectoryDatasetWrapper" in wrapper_dict
