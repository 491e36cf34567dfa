"""TEST-SIDE replica of the reference's ``EnvPos`` (envs/utils/reset_helper.py:104-345), call for call: the same control
flow, the same rejection tests in the same order and the same draws from Python's ``random`` generator, so that
``ReferenceEnvPos(cfg, seed=s)`` produces exactly the episode the reference produces after ``random.seed(s)``.  It is pinned bit
for bit on 15 episodes the reference's own ``EnvPos`` placed (tests/golden/spawn_ref.npz, gen_spawn_golden.py) and serves as
the CHECKER of the product's spawn (``img_env_amd.spawn.EnvPos`` = the library's own sampler, csrc/spawn_host.h): the product
is tested for the reference's rules and, statistically, against this replica (tests/test_host_logic.py).  Nothing under
img_env_amd/ imports it.

It includes the reference's quirks: a start that is re-drawn after 50 failed targets must also keep its distance to the start
it replaces (the old pose is still in the list, reset_helper.py:245-247), a failing circle start clears every circle agent
and the whole pass is repeated (249-255), the circle radius is drawn once per episode whether or not anybody stands on a
circle (202).  What differs is the cost of the distance tests: a hash grid instead of list scans (same booleans).
``range_view_plus`` is not implemented by the reference either (reset_helper.py:278-280 leaves ``rand_pose`` unset) and raises.
"""
import math
import random

import numpy as np

from img_env_amd import _cabi
from img_env_amd.spawn import _module_size
from img_env_amd.worldgen import ResetLayout


class _PoseSet:
    """``free_check_robo_ped`` (reset_helper.py:35-43) over a list of poses with ``None`` holes, answered through a hash grid:
    False iff some stored pose lies within ``d`` (``sqrt(dx^2 + dy^2) <= d``, the reference's own expression)."""

    def __init__(self, n, cell):
        self.pose = [None] * n
        self.cell = max(float(cell), 1e-6)
        self.buckets = {}

    def _key(self, x, y):
        return int(math.floor(x / self.cell)), int(math.floor(y / self.cell))

    def __getitem__(self, i):
        return self.pose[i]

    def set(self, i, p):
        old = self.pose[i]
        if old is not None:
            self.buckets[self._key(old[0], old[1])].remove(i)
        self.pose[i] = p
        if p is not None:
            self.buckets.setdefault(self._key(p[0], p[1]), []).append(i)

    def free(self, x, y, d=1.0):
        kx, ky = self._key(x, y)
        r = int(math.ceil(d / self.cell))
        for i in range(kx - r, kx + r + 1):
            for j in range(ky - r, ky + r + 1):
                for q in self.buckets.get((i, j), ()):
                    p = self.pose[q]
                    if math.sqrt((x - p[0]) * (x - p[0]) + (y - p[1]) * (y - p[1])) <= d:
                        return False
        return True


class ReferenceEnvPos:
    def __init__(self, cfg, seed=None):
        self.cfg = cfg
        self.rng = random.Random(seed)
        self.clearance = float(cfg.get("spawn_clearance", 1.0))  # free_check_robo_ped d=1.0

    def _random_pose(self, x, y, sita):
        """reset_helper.py:58-59"""
        return [self.rng.uniform(x[0], x[1]), self.rng.uniform(y[0], y[1]), self.rng.uniform(sita[0], sita[1])]

    def _random_noise(self, pose):
        """reset_helper.py:30-32"""
        pose[0] += self.rng.gauss(0, 0.5)
        pose[1] += self.rng.gauss(0, 0.5)

    def _random_view(self, init_pose, pose_range):
        """reset_helper.py:62-82"""
        task_view = [2.5, 4.0, 2.5, 4.0]
        while True:
            rand_pose = self._random_pose([init_pose[0] - task_view[1], init_pose[0] + task_view[1]],
                                          [init_pose[1] - task_view[3], init_pose[1] + task_view[3]], [-3.14, 3.14])
            if init_pose[0] - task_view[0] <= rand_pose[0] <= init_pose[0] + task_view[0] and \
                    init_pose[1] - task_view[2] <= rand_pose[1] <= init_pose[1] + task_view[2]:
                continue
            if pose_range[0] <= rand_pose[0] <= pose_range[1] and pose_range[2] <= rand_pose[1] <= pose_range[3]:
                break
        return rand_pose

    def reset_obs(self):
        """reset_helper.py:122-165"""
        o = self.cfg["object"]
        n = int(o["total"])
        shape = np.zeros(n, np.int32)
        size = np.zeros((n, 4), np.float32)
        pose = np.zeros((n, 3))
        self.obs_range = []
        for i in range(n):
            size_range, pose_range = o["size_range"][i], list(o["poses"][i])
            if o["shape"][i] == "circle":
                model_radius = self.rng.uniform(size_range[0], size_range[1])
            elif o["shape"][i] == "rectangle":
                model_radius = math.sqrt(size_range[0] ** 2 + size_range[2] ** 2)
            else:
                raise ValueError("unsupported obstacle shape %r" % o["shape"][i])
            if o["poses_type"][i] == "fix":
                self.obs_range.append(pose_range + ([0, model_radius] if len(pose_range) == 2 else [model_radius]))
            elif o["poses_type"][i] == "range":
                if len(pose_range) == 4:
                    rand_pose = self._random_pose(pose_range[:2], pose_range[2:4], [-3.14, 3.14])
                else:
                    rand_pose = self._random_pose(pose_range[:2], pose_range[2:4], pose_range[4:6])
                self.obs_range.append(rand_pose + [model_radius])
            else:
                raise ValueError("unsupported obstacle pose type %r" % o["poses_type"][i])
            if o["shape"][i] == "circle":
                shape[i], size[i, :3] = _cabi.SHAPE_CIRCLE, (0, 0, self.obs_range[i][-1])
            else:
                shape[i], size[i] = _cabi.SHAPE_RECTANGLE, size_range[:4]
            pose[i] = self.obs_range[i][:3]
        return shape, size, pose

    def _free_obj(self, target_pose):
        """free_check_obj (reset_helper.py:46-55)"""
        for p in self.obs_range:
            if p[-1] == 0.0:
                continue
            d = target_pose[-1] + p[-1]
            if math.sqrt((target_pose[0] - p[0]) ** 2 + (target_pose[1] - p[1]) ** 2) <= d:
                return False
        return True

    def _reset_robot_ped(self):
        """reset_helper.py:189-345, statement for statement"""
        cfg, rng = self.cfg, self.rng
        nr, npd = int(cfg["robot"]["total"]), int(cfg["ped_sim"]["total"])
        n = nr + npd
        btype = list(cfg["robot"]["begin_poses_type"][:nr]) + list(cfg["ped_sim"]["begin_poses_type"][:npd])
        ttype = list(cfg["robot"]["target_poses_type"][:nr]) + list(cfg["ped_sim"]["target_poses_type"][:npd])
        bpose = list(cfg["robot"]["begin_poses"][:nr]) + list(cfg["ped_sim"]["begin_poses"][:npd])
        tpose = list(cfg["robot"]["target_poses"][:nr]) + list(cfg["ped_sim"]["target_poses"][:npd])
        sizes = list(cfg["robot"]["size"][:nr]) + list(cfg["ped_sim"]["size"][:npd])
        shapes = list(cfg["robot"]["shape"][:nr]) + list(cfg["ped_sim"]["shape"][:npd])
        module = [_module_size(sizes[i], shapes[i]) for i in range(n)]
        for t in btype + ttype:
            if "plus" in t:
                raise NotImplementedError("pose type %r: random_view_plus does not exist in the reference either "
                                          "(reset_helper.py:278-280)" % t)
        for i in range(n):
            if btype[i] in ("fix", "rand_angle") and ttype[i] not in ("fix", "rand_angle"):
                raise ValueError("agent %d: start %r with target %r -- the reference never leaves its `while reset_init` loop for "
                                 "a fixed start with a random target (reset_helper.py:218-300)" % (i, btype[i], ttype[i]))
        init, target = _PoseSet(n, self.clearance), _PoseSet(n, self.clearance)
        circle_range = rng.uniform(cfg["circle_ranges"][0], cfg["circle_ranges"][1])
        self.circle_range = circle_range
        for i in range(n):
            if btype[i] == "fix":
                init.set(i, list(bpose[i]))
            if ttype[i] == "fix":
                target.set(i, list(tpose[i]))
            if btype[i] == "rand_angle":
                init.set(i, [bpose[i][0], bpose[i][1], rng.uniform(bpose[i][2], bpose[i][3])])
            if ttype[i] == "rand_angle":
                target.set(i, [tpose[i][0], tpose[i][1], rng.uniform(tpose[i][2], tpose[i][3])])
        tmin2 = cfg["target_min_dist"] ** 2
        d = self.clearance
        circle_ok = False
        while not circle_ok:
            circle_ok = True
            for i in range(n):
                if init[i] is not None and target[i] is not None:
                    continue
                reset_init = True
                while reset_init:
                    goal_fail = 0
                    circle_fail = 0
                    if "range" in btype[i]:
                        while reset_init:
                            pose_range = bpose[i]
                            if "circle" in btype[i]:
                                angle_range = rng.uniform(-3.14, 3.14)
                                if "fix" in btype[i]:
                                    angle_range = -3.14 + (6.28 / n) * i
                                rand_pose = [circle_range * math.cos(angle_range) + pose_range[0],
                                             circle_range * math.sin(angle_range) + pose_range[1], angle_range + 3.14]
                                self._random_noise(rand_pose)
                            else:
                                if "multi" in btype[i]:
                                    pose_range = pose_range[rng.randint(0, len(pose_range) - 1)]
                                if len(pose_range) == 4:
                                    rand_pose = self._random_pose(pose_range[:2], pose_range[2:4], [-3.14, 3.14])
                                elif len(pose_range) == 6:
                                    rand_pose = self._random_pose(pose_range[:2], pose_range[2:4], pose_range[4:6])
                                else:
                                    raise ValueError("agent %d: begin pose range %r has neither 4 nor 6 numbers" % (i, pose_range))
                            if init.free(rand_pose[0], rand_pose[1], d) and \
                                    self._free_obj([rand_pose[0], rand_pose[1], module[i] * 2]):
                                init.set(i, rand_pose[:])
                                reset_init = False
                                break
                            if "circle" in btype[i]:
                                circle_fail += 1
                                if circle_fail > 50:
                                    circle_ok = False
                                    for j in range(n):
                                        if "circle" in btype[j]:
                                            init.set(j, None)
                                            target.set(j, None)
                    if "circle_fix" in ttype[i] and init[i] is not None:
                        pose_range = tpose[i]
                        angle = init[i][2]
                        target.set(i, [circle_range * math.cos(angle) + pose_range[0],
                                       circle_range * math.sin(angle) + pose_range[1], angle - 3.14])
                    if "range" in ttype[i]:
                        while True:
                            pose_range = tpose[i]
                            if "circle" in ttype[i] and init[i] is not None:
                                angle = init[i][2]
                                rand_pose = [circle_range * math.cos(angle) + pose_range[0],
                                             circle_range * math.sin(angle) + pose_range[1], angle - 3.14]
                                self._random_noise(rand_pose)
                            if "multi" in ttype[i]:
                                pose_range = pose_range[rng.randint(0, len(pose_range) - 1)]
                            if "view" in ttype[i]:
                                rand_pose = self._random_view(init[i], pose_range)
                            elif len(pose_range) == 4:
                                rand_pose = self._random_pose(pose_range[:2], pose_range[2:4], [-3.14, 3.14])
                            elif len(pose_range) == 6:
                                rand_pose = self._random_pose(pose_range[:2], pose_range[2:4], pose_range[4:6])
                            if (init[i][0] - rand_pose[0]) ** 2 + (init[i][1] - rand_pose[1]) ** 2 > tmin2 \
                                    and target.free(rand_pose[0], rand_pose[1], d) and \
                                    self._free_obj([rand_pose[0], rand_pose[1], module[i] * 2]):
                                target.set(i, rand_pose[:])
                                break
                            goal_fail += 1
                            if goal_fail > 50:
                                reset_init = True
                                break
        self.init_poses, self.target_poses = init.pose, target.pose
        flag = all(p is not None for p in init.pose) and all(p is not None for p in target.pose)
        return flag

    def reset(self, extent=None):
        """One episode's placement as a ``ResetLayout`` (``extent`` is accepted for compatibility and unused)."""
        cfg = self.cfg
        oshape, osize, opose = self.reset_obs()
        while not self._reset_robot_ped():  # reset_helper.py:117-119
            pass
        nr, npd = int(cfg["robot"]["total"]), int(cfg["ped_sim"]["total"])
        init = np.array([p[:3] for p in self.init_poses], float).reshape(nr + npd, 3)
        target = np.array([p[:3] for p in self.target_poses], float).reshape(nr + npd, 3)
        poses = _poses_math(init)
        go_back = cfg["ped_sim"].get("go_back", "yes")
        assert go_back in ("yes", "no", "random")
        traj = np.zeros((npd, 2, 3))
        tlen = np.ones(npd, np.int32)
        for j in range(npd):
            traj[j, 0, :2] = target[nr + j, :2]
            if go_back == "yes" or (go_back == "random" and self.rng.random() > 0.5):  # reset_helper.py:337-342
                traj[j, 1, :2] = init[nr + j, :2]
                tlen[j] = 2
        return ResetLayout(robot_pose=poses[:nr], robot_goal=target[:nr, :2], ped_pose=poses[nr:],
                           ped_goal=target[nr:, :2], ped_traj=traj, ped_traj_len=tlen, obs_shape=oshape, obs_size=osize,
                           obs_pose=_poses_math(opose) if len(opose) else np.zeros((0, 4)),
                           ignore_obstacle=bool(cfg["ped_sim"].get("ignore_obstacle", False)))


def _poses_math(xyyaw):
    """(x, y, qz, qw) rows with the quaternion of ros_utils.rpy_to_q([0, 0, yaw]) = tf.transformations.quaternion_from_euler:
    libm's sin / cos of the half angle (numpy's vectorised sin / cos may differ from libm in the last bit)"""
    out = np.zeros((len(xyyaw), 4))
    for k, (x, y, yaw) in enumerate(xyyaw):
        out[k] = (x, y, math.sin(yaw / 2.0), math.cos(yaw / 2.0))
    return out


