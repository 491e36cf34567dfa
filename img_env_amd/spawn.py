"""Random spawn of obstacles / robots / pedestrians per episode: the role of the reference's ``EnvPos``
(envs/utils/reset_helper.py:104-345), for the pose types the shipped robot_nav configs use on plain
ranges: ``fix``, ``rand_angle``, ``range`` starts and ``range`` / ``range_view`` targets.

The rejection rules are the reference's (reset_helper.py:35-55, 62-82, 245-300): starts keep
``> 1.0 m`` to every other start and ``> r + r_obs`` to obstacles, targets keep ``> target_min_dist``
to their own start, ``> 1.0 m`` to other targets and clear obstacles, ``range_view`` targets lie in the
4 m box around the start but outside its 2.5 m box.  Distances are checked through a hash grid instead of
the reference's O(n^2) list scans so that 10^4 robots place in about a second; the ``*_circle*`` layouts
(reset_helper.py:231-237, 260-273) are not implemented yet.
"""
import math
import random

import numpy as np

from . import _cabi
from .worldgen import ResetLayout, _HashGrid, yaw_to_pose


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
    def __init__(self, cfg, seed=None):
        self.cfg = cfg
        self.rng = random.Random(seed)
        self.clearance = float(cfg.get("spawn_clearance", 1.0))  # free_check_robo_ped d=1.0

    def _rand_pose(self, r):
        if len(r) == 4:
            return [self.rng.uniform(r[0], r[1]), self.rng.uniform(r[2], r[3]), self.rng.uniform(-3.14, 3.14)]
        return [self.rng.uniform(r[0], r[1]), self.rng.uniform(r[2], r[3]), self.rng.uniform(r[4], r[5])]

    def reset_obs(self):
        """reset_helper.py:122-165"""
        o = self.cfg["object"]
        n = int(o["total"])
        shape = np.zeros(n, np.int32)
        size = np.zeros((n, 4), np.float32)
        pose = np.zeros((n, 3))
        self.obs_range = []
        for i in range(n):
            sr, pr = o["size_range"][i], o["poses"][i]
            if o["shape"][i] == "circle":
                radius = self.rng.uniform(sr[0], sr[1])
                shape[i], size[i, :3] = _cabi.SHAPE_CIRCLE, (0, 0, radius)
            else:
                radius = math.sqrt(sr[0] ** 2 + sr[2] ** 2)
                shape[i], size[i] = _cabi.SHAPE_RECTANGLE, sr[:4]
            if o["poses_type"][i] == "fix":
                p = list(pr) + [0] if len(pr) == 2 else list(pr)
            else:
                p = self._rand_pose(pr)
            pose[i] = p[:3]
            self.obs_range.append(list(p[:3]) + [radius])
        return shape, size, pose

    def _free_obj(self, x, y, r):
        """free_check_obj (reset_helper.py:46-55)"""
        for p in self.obs_range:
            if p[-1] == 0.0:
                continue
            if math.sqrt((x - p[0]) ** 2 + (y - p[1]) ** 2) <= r + p[-1]:
                return False
        return True

    def reset(self, extent):
        cfg = self.cfg
        oshape, osize, opose = self.reset_obs()
        nr, npd = int(cfg["robot"]["total"]), int(cfg["ped_sim"]["total"])
        n = nr + npd
        btype = cfg["robot"]["begin_poses_type"][:nr] + cfg["ped_sim"]["begin_poses_type"][:npd]
        ttype = cfg["robot"]["target_poses_type"][:nr] + cfg["ped_sim"]["target_poses_type"][:npd]
        bpose = cfg["robot"]["begin_poses"][:nr] + cfg["ped_sim"]["begin_poses"][:npd]
        tpose = cfg["robot"]["target_poses"][:nr] + cfg["ped_sim"]["target_poses"][:npd]
        sizes = cfg["robot"]["size"][:nr] + cfg["ped_sim"]["size"][:npd]
        shapes = cfg["robot"]["shape"][:nr] + cfg["ped_sim"]["shape"][:npd]
        starts, goals = _HashGrid(extent, self.clearance), _HashGrid(extent, self.clearance)
        init, target = [None] * n, [None] * n
        for i in range(n):
            for t in (btype[i], ttype[i]):
                if "circle" in t or "multi" in t or "plus" in t:
                    raise NotImplementedError("pose type %r (reset_helper.py circle/multi layouts) is not implemented" % t)
            if btype[i] == "fix":
                init[i] = list(bpose[i])
            elif btype[i] == "rand_angle":
                init[i] = [bpose[i][0], bpose[i][1], self.rng.uniform(bpose[i][2], bpose[i][3])]
            if ttype[i] == "fix":
                target[i] = list(tpose[i])
            elif ttype[i] == "rand_angle":
                target[i] = [tpose[i][0], tpose[i][1], self.rng.uniform(tpose[i][2], tpose[i][3])]
            if init[i] is not None:
                starts.add(init[i][0], init[i][1])
            if target[i] is not None:
                goals.add(target[i][0], target[i][1])
        tmin2 = cfg["target_min_dist"] ** 2
        for i in range(n):
            msize = 2 * _module_size(sizes[i], shapes[i])
            fixed_start, fixed_target = init[i] is not None, target[i] is not None
            while True:
                start = init[i]
                if not fixed_start:
                    for _ in range(10000):
                        p = self._rand_pose(bpose[i])
                        if starts.ok(p[0], p[1], self.clearance) and self._free_obj(p[0], p[1], msize):
                            start = p
                            break
                    else:
                        raise RuntimeError("could not place start %d" % i)
                if fixed_target:
                    break
                goal = None
                for _ in range(51):  # goal_fail > 50 re-draws the start (reset_helper.py:296-300)
                    if "view" in ttype[i]:  # random_view (reset_helper.py:62-82)
                        r = tpose[i]
                        while True:
                            p = self._rand_pose([start[0] - 4.0, start[0] + 4.0, start[1] - 4.0, start[1] + 4.0])
                            if abs(p[0] - start[0]) <= 2.5 and abs(p[1] - start[1]) <= 2.5:
                                continue
                            if r[0] <= p[0] <= r[1] and r[2] <= p[1] <= r[3]:
                                break
                    else:
                        p = self._rand_pose(tpose[i])
                    if (start[0] - p[0]) ** 2 + (start[1] - p[1]) ** 2 > tmin2 and \
                            goals.ok(p[0], p[1], self.clearance) and self._free_obj(p[0], p[1], msize):
                        goal = p
                        break
                if goal is not None:
                    target[i] = goal
                    break
                if fixed_start:
                    raise RuntimeError("no admissible target for fixed start %d" % i)
            init[i] = start
            if not fixed_start:
                starts.add(start[0], start[1])
            if not fixed_target:
                goals.add(target[i][0], target[i][1])
        init, target = np.array(init, float), np.array(target, float)
        poses = yaw_to_pose(init[:, :2], init[:, 2])
        go_back = cfg["ped_sim"].get("go_back", "yes")
        cap = 2
        traj = np.zeros((npd, cap, 3))
        tlen = np.ones(npd, np.int32)
        for j in range(npd):
            traj[j, 0, :2] = target[nr + j, :2]
            if go_back == "yes" or (go_back == "random" and self.rng.random() > 0.5):  # reset_helper.py:337-342
                traj[j, 1, :2] = init[nr + j, :2]
                tlen[j] = 2
        return ResetLayout(robot_pose=poses[:nr], robot_goal=target[:nr, :2], ped_pose=poses[nr:],
                           ped_goal=target[nr:, :2], ped_traj=traj, ped_traj_len=tlen, obs_shape=oshape, obs_size=osize,
                           obs_pose=yaw_to_pose(opose[:, :2], opose[:, 2]) if len(opose) else np.zeros((0, 4)),
                           ignore_obstacle=bool(cfg["ped_sim"].get("ignore_obstacle", False)))


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
    if "circle" in t or "multi" in t or "plus" in t:
        raise NotImplementedError("pose type %r (reset_helper.py circle/multi layouts) is not implemented" % t)
    if t == "fix":
        return _cabi.POSE_FIX
    if t == "rand_angle":
        return _cabi.POSE_RAND_ANGLE
    if target and "view" in t:
        return _cabi.POSE_RANGE_VIEW
    return _cabi.POSE_RANGE if len(values) == 4 else _cabi.POSE_RANGE_YAW


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
    for i in range(n):
        a = agents[i]
        a.begin_type, a.target_type = _pose_type(btype[i], bpose[i], False), _pose_type(ttype[i], tpose[i], True)
        for k, v in enumerate(bpose[i][:6]):
            a.begin[k] = float(v)
        for k, v in enumerate(tpose[i][:6]):
            a.target[k] = float(v)
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
    c.go_back = {"yes": 1, "random": 2}.get(cfg["ped_sim"].get("go_back", "yes"), 0)
    c.ignore_obstacle = int(bool(cfg["ped_sim"].get("ignore_obstacle", False)))
    return c, (agents, obstacles)


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
