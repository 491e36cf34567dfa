"""Seeded scenarios shared by the oracle tests, the golden generators and the GPU parity tests."""
import numpy as np

from img_env_amd import config, worldgen


def apply_cfg_over(cfg, over):
    """``over``: {"key" or "section.key": value} laid over a YAML-schema config dict (sections are copied, not shared)"""
    for k, v in (over or {}).items():
        if "." in k:
            sec, sub = k.split(".", 1)
            cfg[sec] = dict(cfg[sec])
            cfg[sec][sub] = v
        else:
            cfg[k] = v
    return cfg


def golden_cfg(meta, grid):
    """the YAML-schema config of a golden fixture (what tests/golden/gen_python_golden.py::run() handed to make_env)"""
    cfg = worldgen.make_yaml_cfg(meta["n_robots"], meta["n_peds"], grid, time_max=meta["time_max"],
                                 ped_shape=meta["ped_shape"], state_dim=meta["state_dim"],
                                 n_obstacles=meta["n_obstacles"])
    return apply_cfg_over(cfg, meta.get("cfg_over"))


def golden_scenario(meta):
    """rebuilds exactly what tests/golden/gen_python_golden.py::run() set up"""
    grid = worldgen.make_grid(200, meta["seed"])
    cfg = golden_cfg(meta, grid)
    layout = worldgen.make_layout(grid, 0.125, meta["n_robots"], meta["n_peds"], seed=meta["seed"] + 100,
                                  n_obstacles=meta["n_obstacles"])
    if meta.get("near_goals"):
        yaw = 2.0 * np.arctan2(layout.robot_pose[:, 2], layout.robot_pose[:, 3])
        layout.robot_goal = layout.robot_pose[:, :2] + 0.9 * np.stack([np.cos(yaw), np.sin(yaw)], 1)
    return grid, config.params_from_cfg(cfg), layout


def clip_actions(act):
    """VelActionWrapper with continuous_actions [[0,0.6],[-0.9,0.9]] (base.py:45-53), then the float32
    wire rounding of Agent.msg v / w / v_y"""
    a = np.zeros((len(act), 3), np.float32)
    a[:, 0] = np.clip(act[:, 0], 0, 0.6)
    a[:, 1] = np.clip(act[:, 1], -0.9, 0.9)
    return a


def random_actions(rng, n):
    """env_test.py:8-19 RandomPolicy4Nav"""
    return np.stack([rng.uniform(0, 0.6, n), rng.uniform(-0.9, 0.9, n), np.zeros(n)], 1).astype(np.float32)


def small_world(n_robots, n_peds, seed=0, grid_size=200, res=0.125, n_obstacles=2, clearance=1.0, **kw):
    grid = worldgen.make_grid(grid_size, seed)
    params = worldgen.make_params(n_robots, n_peds, res=res, **kw)
    layout = worldgen.make_layout(grid, res, n_robots, n_peds, seed=seed + 1, n_obstacles=n_obstacles,
                                  clearance=clearance)
    return grid, params, layout
