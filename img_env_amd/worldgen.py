"""Synthetic worlds and reset layouts for tests and benchmarks.

All scalar parameters are float32-exact (0.125, 0.25, 6.0, ...) so the float32 wire rounding of
the reference's ROS messages is moot.  Layout rules follow the reference's ``EnvPos`` spawn
sampler (envs/utils/reset_helper.py:35-82, 189-345) -- uniform starts with a clearance, goals by
the ``random_view`` 2.5-4 m box rule, pedestrians ping-pong between start and goal
(``go_back: yes``) -- re-implemented with a hash grid so 10^4..10^5 agents place in seconds.
"""
import math
from dataclasses import dataclass, field

import numpy as np

from . import _cabi

ROBOT_RADIUS = 0.17


def make_grid(size, seed=0, border=8, wall_thick=4):
    """uint8 occupancy grid: free 255, wall 0; border wall + size/50 interior wall segments."""
    rng = np.random.default_rng(seed)
    g = np.full((size, size), 255, np.uint8)
    g[:border] = 0
    g[-border:] = 0
    g[:, :border] = 0
    g[:, -border:] = 0
    n_walls, length = size // 50, size // 4
    for _ in range(n_walls):
        horizontal = bool(rng.integers(0, 2))
        a = int(rng.integers(border + 8, size - border - 8 - length))
        b = int(rng.integers(border + 8, size - border - 8 - wall_thick))
        if horizontal:
            g[b:b + wall_thick, a:a + length] = 0
        else:
            g[a:a + length, b:b + wall_thick] = 0
    return g


def _clear_mask(grid, res, clearance):
    """cells whose centre is farther than ``clearance`` metres from every occupied cell"""
    from scipy import ndimage
    dist = ndimage.distance_transform_edt(grid >= 250) * res
    return dist > clearance


class _HashGrid:
    def __init__(self, extent, cell):
        self.cell = cell
        self.n = int(math.ceil(extent / cell)) + 2
        self.buckets = {}

    def _key(self, x, y):
        return int(x / self.cell), int(y / self.cell)

    def ok(self, x, y, d):
        kx, ky = self._key(x, y)
        r = int(math.ceil(d / self.cell))
        d2 = d * d
        for i in range(kx - r, kx + r + 1):
            for j in range(ky - r, ky + r + 1):
                for (px, py) in self.buckets.get((i, j), ()):
                    if (px - x) ** 2 + (py - y) ** 2 <= d2:
                        return False
        return True

    def add(self, x, y):
        self.buckets.setdefault(self._key(x, y), []).append((x, y))


def _sample(rng, n, mask, res, clearance, hashgrid, near=None, box=(2.5, 4.0), min_dist=1.0, max_tries=4000):
    """n points (x, y) on free cells of ``mask`` keeping ``clearance`` to the points in hashgrid.
    With ``near`` ([n,2] starts) the ``random_view`` rule of reset_helper.py:62-82 is applied."""
    H, W = mask.shape
    out = np.zeros((n, 2))
    for i in range(n):
        for t in range(max_tries):
            if near is None:
                x, y = rng.uniform(0, H * res), rng.uniform(0, W * res)
            else:
                x = rng.uniform(near[i, 0] - box[1], near[i, 0] + box[1])
                y = rng.uniform(near[i, 1] - box[1], near[i, 1] + box[1])
                if abs(x - near[i, 0]) <= box[0] and abs(y - near[i, 1]) <= box[0]:
                    continue
                if (x - near[i, 0]) ** 2 + (y - near[i, 1]) ** 2 <= min_dist ** 2:
                    continue
            m, k = int(round(x / res)), int(round(y / res))
            if m < 0 or m >= H or k < 0 or k >= W or not mask[m, k]:
                continue
            if not hashgrid.ok(x, y, clearance):
                continue
            hashgrid.add(x, y)
            out[i] = (x, y)
            break
        else:
            raise RuntimeError("worldgen: could not place agent %d (clearance %.2f too large for this map)"
                               % (i, clearance))
    return out


def yaw_to_pose(xy, yaw):
    """(x, y, qz, qw): tf.transformations.quaternion_from_euler(0, 0, yaw) (ros_utils.py:22-23)"""
    p = np.zeros((len(xy), 4))
    p[:, :2] = xy
    p[:, 2] = np.sin(yaw / 2.0)
    p[:, 3] = np.cos(yaw / 2.0)
    return p


@dataclass
class ResetLayout:
    robot_pose: np.ndarray
    robot_goal: np.ndarray
    ped_pose: np.ndarray
    ped_goal: np.ndarray
    ped_traj: np.ndarray
    ped_traj_len: np.ndarray
    obs_shape: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    obs_size: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float32))
    obs_pose: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))
    ignore_obstacle: bool = False
    ped_traj_v: np.ndarray = None  # [P, cap, 2] recorded (vx, vy) beside ped_traj: the "dataset" pedestrian scene

    def as_batch(self):
        return dict(ped_traj_v=self.ped_traj_v, robot_pose=self.robot_pose, robot_goal=self.robot_goal, ped_pose=self.ped_pose,
                    ped_goal=self.ped_goal, ped_traj=self.ped_traj, ped_traj_len=self.ped_traj_len,
                    ped_traj_cap=self.ped_traj.shape[1] if self.ped_traj.ndim == 3 else 2,
                    obs_shape=self.obs_shape, obs_size=self.obs_size, obs_pose=self.obs_pose,
                    ignore_obstacle=self.ignore_obstacle)


def make_layout(grid, res, n_robots, n_peds, seed=0, clearance=1.0, wall_clearance=None, n_obstacles=0):
    """One reset: obstacles (test.yaml mix: circle r=0.3 / rectangle +-0.15), robots, peds."""
    rng = np.random.default_rng(seed)
    if wall_clearance is None:
        wall_clearance = min(clearance, 1.0)
    mask = _clear_mask(grid, res, wall_clearance)
    extent = max(grid.shape) * res
    obs_shape = np.zeros(n_obstacles, np.int32)
    obs_size = np.zeros((n_obstacles, 4), np.float32)
    obs_pose = np.zeros((n_obstacles, 4))
    starts, goals = _HashGrid(extent, clearance), _HashGrid(extent, clearance)
    if n_obstacles:
        oxy = _sample(rng, n_obstacles, mask, res, 1.5, starts)
        for q in range(n_obstacles):
            goals.add(*oxy[q])
            if q % 2 == 0:
                obs_shape[q], obs_size[q] = _cabi.SHAPE_CIRCLE, (0.0, 0.0, 0.3, 0.0)
            else:
                obs_shape[q], obs_size[q] = _cabi.SHAPE_RECTANGLE, (-0.15, 0.15, -0.15, 0.15)
        obs_pose = yaw_to_pose(oxy, rng.uniform(-3.14, 3.14, n_obstacles))
    n = n_robots + n_peds
    sxy = _sample(rng, n, mask, res, clearance, starts)
    gxy = _sample(rng, n, mask, res, clearance, goals, near=sxy)
    yaw = rng.uniform(-3.14, 3.14, n)
    poses = yaw_to_pose(sxy, yaw)
    traj = np.zeros((n_peds, 2, 3))
    traj[:, 0, :2] = gxy[n_robots:]
    traj[:, 1, :2] = sxy[n_robots:]
    return ResetLayout(robot_pose=poses[:n_robots], robot_goal=gxy[:n_robots], ped_pose=poses[n_robots:],
                       ped_goal=gxy[n_robots:], ped_traj=traj, ped_traj_len=np.full(n_peds, 2, np.int32),
                       obs_shape=obs_shape, obs_size=obs_size, obs_pose=obs_pose)


def make_params(n_robots, n_peds, res=0.125, view_cells=48, beams=360, scene="rvoscene", ped_shape="circle",
                dt=0.25, state_dim=3, time_max=100, relation_ped_robo=1, max_ped=None, robot_ktype="diff",
                use_laser=True, ped_max_speed=0.5, **over):
    """Parameter dict for ``_cabi.make_cfg`` in the geometry of SURVEY.md section 8(d)."""
    P = n_peds
    if ped_shape == "leg":
        # EnvPos.init_ped mirrors the left leg [x, y, r] into the right one (reset_helper.py:400-404)
        ps = np.tile(np.array([0, 0.1, 0.1, 0, -0.1, 0.1], np.float32), (P, 1))
        pshape = np.full(P, _cabi.SHAPE_LEG, np.int32)
    else:
        ps = np.tile(np.array([0, 0, ROBOT_RADIUS, 0, 0, 0], np.float32), (P, 1))
        pshape = np.full(P, _cabi.SHAPE_CIRCLE, np.int32)
    p = dict(
        view_resolution=res, global_resolution=res, view_width=view_cells * res, view_height=view_cells * res,
        step_hz=dt, state_dim=state_dim, use_laser=int(use_laser), range_total=beams,
        view_angle_begin=-1.570795, view_angle_end=1.570795, view_min_dist=0.0, view_max_dist=10.0,
        relation_ped_robo=relation_ped_robo, ped_scene_type=_cabi.SCENES.get(scene, _cabi.SCENE_EMPTY) if P else 0,
        robot_ktype=_cabi.KTYPES[robot_ktype], n_robots=n_robots, n_peds=P,
        robot_shape=np.full(n_robots, _cabi.SHAPE_CIRCLE, np.int32),
        robot_size=np.tile(np.array([0, 0, ROBOT_RADIUS, 0], np.float32), (n_robots, 1)),
        robot_sensor_cfg=np.zeros((n_robots, 2), np.float32),
        robot_size_last=np.full(n_robots, ROBOT_RADIUS),
        ped_shape=pshape, ped_size=ps, ped_max_speed=np.full(P, ped_max_speed, np.float32),
        image_size=(view_cells, view_cells), ped_image_size=(48, 48),
        max_ped=max(P, 1) if max_ped is None else max_ped, ped_vec_dim=7, ped_image_r=0.3, laser_max=6.0,
        laser_norm=True, ped_safety_space=0.7, time_max=time_max,
    )
    p.update(over)
    return p


#: BASELINE.json configs: grid cells, resolution by the density rule, clearance
PRESETS = {
    "cfg1": dict(n_robots=1, n_peds=0, grid=200, res=0.125, view_cells=48, beams=360, scene="", clearance=1.0),
    "cfg2": dict(n_robots=1024, n_peds=0, grid=400, res=0.125, view_cells=48, beams=360, scene="", clearance=1.0),
    "cfg3": dict(n_robots=8192, n_peds=200, grid=400, res=0.25, view_cells=48, beams=360, scene="rvoscene",
                 clearance=0.7),
    "cfg4": dict(n_robots=65536, n_peds=200, grid=400, res=0.5, view_cells=48, beams=360, scene="pedscene",
                 clearance=0.55),
    "cfg5": dict(n_robots=8192, n_peds=1000, grid=800, res=0.125, view_cells=96, beams=720, scene="ervoscene",
                 clearance=0.7),
}


def make_yaml_cfg(n_robots, n_peds, grid, res=0.125, view_cells=48, beams=360, scene="rvoscene",
                  ped_shape="circle", dt=0.25, time_max=100, state_dim=3, wrappers=None, n_obstacles=0,
                  relation_ped_robo=1, robot_type="diff", max_ped=None, **over):
    """A config dict in the reference's YAML schema (envs/cfg/test.yaml) for a synthetic world.
    ``grid`` goes in as ``global_map.map_array``; spawn ranges span the map interior."""
    H = grid.shape[0] * res
    lo, hi = 1.5, H - 1.5
    rng4 = [lo, hi, lo, hi]
    if wrappers is None:
        wrappers = ["VelActionWrapper", "TimeLimitWrapper", "SensorsPaperRewardWrapper", "InfoLogWrapper",
                    "MultiRobotCleanWrapper"]
    psz = [0, 0.1, 0.1] if ped_shape == "leg" else [0, 0, ROBOT_RADIUS]
    cfg = dict(
        test=False, env_type="robot_nav", robot_type=robot_type, env_num=1, agent_num_per_env=n_robots,
        batch_num_per_env=1, env_id=0, env_name="synthetic", cfg_name="synthetic", cfg_type="yaml",
        control_hz=dt, time_max=time_max, robot_radius=ROBOT_RADIUS, ped_leg_radius=0.1, ped_safety_space=0.7,
        laser_max=6.0, image_batch=1, image_size=[view_cells, view_cells], ped_image_size=[48, 48],
        state_batch=1, state_dim=state_dim, state_normalize=False, laser_batch=0, act_dim=2,
        discrete_action=False, discrete_actions=[[0.0, 0.0]], continuous_actions=[[0, 0.6], [-0.9, 0.9]],
        circle_ranges=[1.8, 2.0], max_ped=max(n_peds, 1) if max_ped is None else max_ped, ped_vec_dim=7,
        ped_image_r=0.3, show_gui=False, sleep_t=0.0, window_height=500, show_image_height=125,
        is_draw_step=False, step_draw=3, use_laser=True, range_total=beams, view_angle_begin=-1.570795,
        view_angle_end=1.570795, view_min_dist=0.0, view_max_dist=10.0, beep_r=1.0, ped_ca_p=1.0,
        relation_ped_robo=relation_ped_robo, wrapper=list(wrappers),
        global_map=dict(resolution=res, map_file="synthetic.png", map_array=grid),
        view_map=dict(resolution=res, width=view_cells * res, height=view_cells * res),
        robot=dict(total=n_robots, shape=["circle"] * n_robots, size=[[0, 0, ROBOT_RADIUS]] * n_robots,
                   begin_poses_type=["range"] * n_robots, begin_poses=[rng4] * n_robots,
                   target_poses_type=["range_view"] * n_robots, target_poses=[rng4] * n_robots),
        object=dict(total=n_obstacles, shape=["circle", "rectangle"] * ((n_obstacles + 1) // 2),
                    size_range=[[0.3, 0.3], [-0.15, 0.15, -0.15, 0.15]] * ((n_obstacles + 1) // 2),
                    poses_type=["range"] * n_obstacles, poses=[rng4] * n_obstacles),
        ped_sim=dict(total=n_peds, type=scene, max_speed=[0.5] * n_peds, shape=[ped_shape] * n_peds,
                     size=[psz] * n_peds, begin_poses_type=["range"] * n_peds, begin_poses=[rng4] * n_peds,
                     target_poses_type=["range_view"] * n_peds, target_poses=[rng4] * n_peds, go_back="yes"),
        target_min_dist=1.0, node_id=0,
    )
    cfg.update(over)
    return cfg


def shipped_test_yaml_cfg(map_file, spawn_sections, env_num=1, wrappers=None):
    """A config dict with the numbers of the reference's shipped ``envs/cfg/test.yaml`` (lines 7-160): a 110 x 110 pixel map at
    0.1 m, grid and view cells of 0.015 m, a 6 m x 6 m view (400 x 400 cells) shrunk to 48 x 48, 1000 beams over +-90 degrees,
    dt 0.4 s, discrete actions, the wrapper list with TestEpisodeWrapper.  ``spawn_sections``: the ``robot`` / ``ped_sim`` /
    ``object`` / ``circle_ranges`` / ``target_min_dist`` keys (tests/golden/spawn_ref.npz carries the shipped file's)."""
    table = [[0.0, -0.9], [0.0, -0.6], [0.0, -0.3], [0.0, 0.05], [0.0, 0.3], [0.0, 0.6], [0.0, 0.9],
             [0.2, -0.9], [0.2, -0.6], [0.2, -0.3], [0.2, 0], [0.2, 0.3], [0.2, 0.6], [0.2, 0.9],
             [0.4, -0.9], [0.4, -0.6], [0.4, -0.3], [0.4, 0], [0.4, 0.3], [0.4, 0.6], [0.4, 0.9],
             [0.6, -0.9], [0.6, -0.6], [0.6, -0.3], [0.6, 0], [0.6, 0.3], [0.6, 0.6], [0.6, 0.9]]
    if wrappers is None:
        wrappers = ["VelActionWrapper", "TimeLimitWrapper", "SensorsPaperRewardWrapper", "InfoLogWrapper", "MultiRobotCleanWrapper",
                    "TestEpisodeWrapper", "StateBatchWrapper", "ObsLaserStateTmp", "NeverStopWrapper"]
    cfg = dict(
        test=False, env_type="robot_nav", robot_type="diff", env_num=env_num, agent_num_per_env=1, batch_num_per_env=1, env_id=0,
        env_name="image_ped_circle", cfg_name="test", cfg_type="yaml", control_hz=0.4, time_max=200, robot_radius=0.17,
        ped_leg_radius=0.1, ped_safety_space=0.7, laser_max=6.0, image_batch=1, image_size=[48, 48], ped_image_size=[48, 48],
        state_batch=3, state_dim=3, state_normalize=False, laser_batch=0, act_dim=2, discrete_action=True, discrete_actions=table,
        continuous_actions=[[0, 0.6], [-0.9, 0.9]], max_ped=10, ped_vec_dim=7, ped_image_r=0.3, show_gui=False, sleep_t=0.0,
        window_height=500, show_image_height=125, is_draw_step=False, step_draw=3, use_laser=True, range_total=1000,
        view_angle_begin=-1.570795, view_angle_end=1.570795, view_min_dist=0.0, view_max_dist=10.0, beep_r=1.0, ped_ca_p=1.0,
        relation_ped_robo=1, init_pose_bag_episodes=100, wrapper=list(wrappers),
        global_map=dict(resolution=0.1, map_file=map_file), view_map=dict(resolution=0.015, width=6, height=6), node_id=0,
    )
    cfg.update(spawn_sections)
    return cfg
