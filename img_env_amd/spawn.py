"""Random spawn of obstacles / robots / pedestrians per episode (the reference's ``EnvPos``, envs/utils/reset_helper.py:104-345).

The placement itself runs inside the library (``imgenv_spawn``, csrc/spawn_host.h: the reference's rules -- starts more than
the clearance apart and clear of the obstacles, targets more than ``target_min_dist`` from their start, ``range_view`` targets
in the 4 m box around the start but outside its 2.5 m box, circle starts on the episode's circle with Gaussian noise, the
"50 failed draws" fall-backs -- with the library's own random stream and a hash grid for the distance tests); this module
turns the YAML's spawn sections into an ``imgenv_spawn_cfg`` and hands out ``ResetLayout`` objects.  Pose types: ``fix``,
``rand_angle``, ``range`` (4 or 6 numbers), ``range_multi``, ``range_circle`` / ``range_circle_fix`` starts; ``range``,
``range_multi``, ``range_view``, ``range_circle`` and ``circle_fix`` targets.  The same spawn cfg drives
``imgenv_reset_worlds_spawn`` / ``imgenv_step_autoreset`` (``VecImageEnv(native_spawn=True)``).

A replica of the reference's sampler that reproduces its episodes draw for draw from Python's ``random`` stream lives with the
tests (tests/spawn_replica.py) as the checker of this one.
"""
import math
import os

import numpy as np

from . import _cabi
from .worldgen import ResetLayout


def _module_size(size, shape):
    """reset_helper.py:167-186"""
    if shape == "circle":
        return size[2]
    if shape == "rectangle":
        return math.sqrt(size[0] ** 2 + size[2] ** 2)
    if shape == "leg":
        return size[-1] + size[-2]
    raise ValueError("unsupported shape %r" % shape)


class EnvPos:
    """``EnvPos(cfg, seed).reset() -> ResetLayout``: a fresh placement per call.  Episode k of an ``EnvPos`` seeded with s
    is the library's placement number ``mix(s) + k``; ``seed=None`` draws the stream's origin from the OS."""

    def __init__(self, cfg, seed=None):
        self.cfg = cfg
        self._spawn_cfg = make_spawn_cfg(cfg)
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        self._seed0 = (0x9E3779B97F4A7C15 * (1 + int(seed))) & 0xFFFFFFFFFFFFFFFF
        self.episodes = 0

    def reset(self, extent=None):
        """One episode's placement (``extent`` is accepted for compatibility and unused)."""
        layout = native_spawn(self.cfg, self._seed0 + self.episodes, self._spawn_cfg)
        self.episodes += 1
        return layout


def init_ped_dataset(layout, ped_pos_v_datas):
    """EnvPos.init_ped_dataset (reset_helper.py:417-432): ``ped_pos_v_datas[i]`` holds one row (x, y, yaw, vx, vy) per
    step for pedestrian i; the trajectory becomes the recorded poses, ``trajectory_v`` the recorded velocities and the
    initial pose the first row.  Used with ``ped_sim.type: dataset`` (img_env.cpp:294-296)."""
    d = np.asarray(ped_pos_v_datas, np.float64)
    P, T = d.shape[0], d.shape[1]
    layout.ped_traj = np.ascontiguousarray(d[:, :, :3])
    layout.ped_traj_v = np.ascontiguousarray(d[:, :, 3:5])
    layout.ped_traj_len = np.full(P, T, np.int32)
    pose = np.zeros((P, 4))
    pose[:, 0], pose[:, 1] = d[:, 0, 0], d[:, 0, 1]
    pose[:, 2], pose[:, 3] = np.sin(d[:, 0, 2] / 2.0), np.cos(d[:, 0, 2] / 2.0)
    layout.ped_pose = pose
    return layout


# ------------------------------------------------------------------------------------------------ native spawn
def _pose_type(t, values, target):
    """pose type string of the reference YAML (reset_helper.py:187-300) -> IMGENV_POSE_*"""
    if "plus" in t:
        raise NotImplementedError("pose type %r: random_view_plus does not exist in the reference either" % t)
    if t == "fix":
        return _cabi.POSE_FIX
    if t == "rand_angle":
        return _cabi.POSE_RAND_ANGLE
    if "circle" in t:
        if "range" not in t:
            if target and "circle_fix" in t:
                return _cabi.POSE_CIRCLE_FIX
            raise ValueError("pose type %r has no effect in the reference (only range_* starts are drawn)" % t)
        return _cabi.POSE_RANGE_CIRCLE_FIX if "fix" in t else _cabi.POSE_RANGE_CIRCLE
    if "multi" in t:
        return _cabi.POSE_RANGE_MULTI
    if target and "view" in t:
        return _cabi.POSE_RANGE_VIEW
    return _cabi.POSE_RANGE if len(values) == 4 else _cabi.POSE_RANGE_YAW


def _multi_boxes(values):
    """[[4 or 6 numbers], ...] -> [n][6] (a 4-number box carries the default yaw range)"""
    out = np.zeros((len(values), 6))
    for k, r in enumerate(values):
        out[k] = list(r) + ([-3.14, 3.14] if len(r) == 4 else [])
    return out


def make_spawn_cfg(cfg):
    """The reference YAML's spawn section as an ``imgenv_spawn_cfg`` (one env's cast): ``(struct, keepalive)``.
    The same fields ``EnvPos`` reads; the placement then runs inside the library (``csrc/spawn_host.h``)."""
    import ctypes as C
    nr, npd = int(cfg["robot"]["total"]), int(cfg["ped_sim"]["total"])
    n = nr + npd
    btype = cfg["robot"]["begin_poses_type"][:nr] + cfg["ped_sim"]["begin_poses_type"][:npd]
    ttype = cfg["robot"]["target_poses_type"][:nr] + cfg["ped_sim"]["target_poses_type"][:npd]
    bpose = cfg["robot"]["begin_poses"][:nr] + cfg["ped_sim"]["begin_poses"][:npd]
    tpose = cfg["robot"]["target_poses"][:nr] + cfg["ped_sim"]["target_poses"][:npd]
    sizes = cfg["robot"]["size"][:nr] + cfg["ped_sim"]["size"][:npd]
    shapes = cfg["robot"]["shape"][:nr] + cfg["ped_sim"]["shape"][:npd]
    agents = (_cabi.SpawnAgent * max(n, 1))()
    keep_multi = []
    for i in range(n):
        a = agents[i]
        a.begin_type, a.target_type = _pose_type(btype[i], bpose[i], False), _pose_type(ttype[i], tpose[i], True)
        for which, typ, vals in (("begin", a.begin_type, bpose[i]), ("target", a.target_type, tpose[i])):
            if typ == _cabi.POSE_RANGE_MULTI:
                boxes = _multi_boxes(vals)
                keep_multi.append(boxes)
                setattr(a, which + "_multi", boxes.ctypes.data_as(C.POINTER(C.c_double)))
                setattr(a, "n_%s_multi" % which, len(boxes))
            else:
                dst = getattr(a, which)
                for k, v in enumerate(vals[:6]):
                    dst[k] = float(v)
        a.module_size = 2 * _module_size(sizes[i], shapes[i])
    o = cfg["object"]
    nob = int(o["total"])
    obstacles = (_cabi.SpawnObstacle * max(nob, 1))()
    for i in range(nob):
        q = obstacles[i]
        q.shape = _cabi.SHAPE_CIRCLE if o["shape"][i] == "circle" else _cabi.SHAPE_RECTANGLE
        pr = o["poses"][i]
        if o["poses_type"][i] == "fix":
            q.pose_type = _cabi.POSE_FIX
            pr = list(pr) + [0] if len(pr) == 2 else list(pr)
        else:
            q.pose_type = _cabi.POSE_RANGE if len(pr) == 4 else _cabi.POSE_RANGE_YAW
        for k, v in enumerate(o["size_range"][i][:4]):
            q.size_range[k] = float(v)
        for k, v in enumerate(pr[:6]):
            q.pose[k] = float(v)
    c = _cabi.SpawnCfg()
    c.struct_size = C.sizeof(_cabi.SpawnCfg)
    c.n_robots, c.n_peds, c.n_obstacles = nr, npd, nob
    c.agents = C.cast(agents, C.POINTER(_cabi.SpawnAgent))
    c.obstacles = C.cast(obstacles, C.POINTER(_cabi.SpawnObstacle))
    c.clearance = float(cfg.get("spawn_clearance", 1.0))
    c.target_min_dist = float(cfg["target_min_dist"])
    cr = cfg.get("circle_ranges", [0.0, 0.0])
    c.circle_ranges[0], c.circle_ranges[1] = float(cr[0]), float(cr[1])
    c.go_back = {"yes": 1, "random": 2}.get(cfg["ped_sim"].get("go_back", "yes"), 0)
    c.ignore_obstacle = int(bool(cfg["ped_sim"].get("ignore_obstacle", False)))
    return c, (agents, obstacles, keep_multi)


def native_spawn(cfg, seed, spawn_cfg=None):
    """One placement by the library's own spawn (``imgenv_spawn``; no device needed) as a ``ResetLayout``."""
    import ctypes as C
    lib = _cabi.load_library()
    c, keep = spawn_cfg if spawn_cfg is not None else make_spawn_cfg(cfg)
    R, P, O = c.n_robots, c.n_peds, c.n_obstacles
    out = dict(robot_pose=np.zeros((R, 4)), robot_goal=np.zeros((R, 2)), ped_pose=np.zeros((P, 4)), ped_goal=np.zeros((P, 2)),
               ped_traj=np.zeros((P, 2, 3)), ped_traj_len=np.zeros(P, np.int32), obs_shape=np.zeros(O, np.int32),
               obs_size=np.zeros((O, 4), np.float32), obs_pose=np.zeros((O, 4)))
    rc = lib.imgenv_spawn(C.byref(c), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), *[a.ctypes.data for a in out.values()])
    if rc != 0:
        raise RuntimeError("imgenv_spawn failed (%d): %s" % (rc, lib.imgenv_last_error().decode()))
    return ResetLayout(ignore_obstacle=bool(c.ignore_obstacle), **out)
