"""Hand-derived known answers for the CPU oracle's restatement of agent.cpp / img_env.cpp (rows whose reference cannot
be built here), plus its internal consistency: the shared min / max owner layers against the literal per-robot grids of
img_env.cpp:620-629."""
import numpy as np
import pytest

from img_env_amd import _cabi, worldgen
from oracle_binding import OracleWorld
from scenarios import random_actions, small_world

RES = 0.125


def _open_world(n_robots=1, n_peds=0, **kw):
    grid = np.full((200, 200), 255, np.uint8)
    grid[:8] = grid[-8:] = 0
    grid[:, :8] = grid[:, -8:] = 0
    params = worldgen.make_params(n_robots, n_peds, res=RES, **kw)
    return grid, params


def _layout(robot_xyyaw, goals, ped_xy=()):
    P = len(ped_xy)
    pose = np.array([[x, y, np.sin(t / 2), np.cos(t / 2)] for x, y, t in robot_xyyaw])
    pp = np.array([[x, y, 0.0, 1.0] for x, y in ped_xy]).reshape(P, 4)
    traj = np.zeros((P, 2, 3))
    traj[:, 0, :2] = traj[:, 1, :2] = pp[:, :2] if P else 0
    return worldgen.ResetLayout(robot_pose=pose, robot_goal=np.array(goals, float), ped_pose=pp, ped_goal=pp[:, :2].copy(),
                                ped_traj=traj, ped_traj_len=np.full(P, 2, np.int32))


def test_wall_two_metres_ahead(oracle_lib):
    """robot at cell (100, 100) looking along +x (rows), wall across row 116: the straight-ahead beam must read exactly
    16 cells = 2.0 m, the map must show free / hit / unknown bands and the robot's own footprint"""
    grid, params = _open_world()
    grid[116:120, :] = 0
    w = OracleWorld(params, grid)
    try:
        w.reset(_layout([(12.5, 12.5, 0.0)], [(20.0, 12.5)]))
        s = w.snapshot()
        lasers = s["lasers"][0] * 6.0                                  # laser_norm: / laser_max
        assert abs(lasers[180] - 2.0) < 1e-6                           # beam 180 of 360 over +-pi/2 points straight ahead
        assert lasers.min() >= 2.0 - 1e-6 and abs(lasers.min() - 2.0) < 1e-6
        assert lasers[0] == pytest.approx(6.0) and lasers[359] == pytest.approx(6.0)  # sideways beams see nothing: default 6.0
        vm = s["view_maps"][0]
        assert vm.shape == (48, 48)
        assert vm[24, 24] == 100                                       # own footprint (agent.cpp:503)
        assert vm[8, 24] == 0                                          # the hit: 16 cells ahead of the centre row 24
        assert (vm[9:23, 24] == 255).all()                             # free and seen between robot and wall
        assert (vm[:8, 24] == 200).all()                               # behind the wall: unknown
        assert (vm[30:, :] == 200).all()                               # behind the robot: outside the field of view
        sm = s["sensor_maps"][0].astype(np.float32)
        assert sm[8, 24] == 0.0 and sm[12, 24] == 1.0 and abs(sm[24, 24] - np.float16(100 / 255)) < 1e-6
        assert s["is_collisions"][0] == 0 and s["is_arrives"][0] == 0
        vs = s["vector_states"][0]
        assert abs(vs[0] - 7.5) < 1e-6 and abs(vs[1]) < 1e-6 and abs(vs[2]) < 1e-6   # goal 7.5 m straight ahead
    finally:
        w.close()


def test_collision_codes_and_arrival(oracle_lib):
    """is_collision_ = 1 on a static cell, 2 on a pedestrian, 3 on another robot (agent.cpp:294-326); arrive within 0.3 m"""
    grid, params = _open_world(n_robots=4, n_peds=1, scene="rvoscene")
    grid[60:64, 40:60] = 0
    w = OracleWorld(params, grid)
    try:
        lay = _layout([(7.55, 6.0, 0.0),      # robot 0 touches the wall at rows 60-63
                       (15.0, 15.0, 1.0),     # robot 1 shares cells with the pedestrian
                       (20.0, 8.0, 0.5),      # robots 2 and 3 overlap each other
                       (20.1, 8.05, -2.0)],
                      [(7.55, 9.0), (18.0, 15.0), (20.0, 12.0), (20.1, 12.0)],
                      ped_xy=[(15.1, 15.05)])
        w.reset(lay)
        s = w.snapshot()
        assert list(s["is_collisions"]) == [1, 2, 3, 3]
        assert s["is_arrives"][3] == 0                                 # arrival is decided by cmd(), not at reset
        w.step(np.zeros((4, 3), np.float32))
        s = w.snapshot()
        assert list(s["is_collisions"]) == [1, 2, 3, 3]                # frozen robots keep their code (agent.cpp:358-360)
        assert list(s["dones"]) == [1, 1, 1, 1] and list(s["dones_info"]) == [1, 2, 3, 3]
        assert (s["rewards"] == -500.0).all()
    finally:
        w.close()
    grid, params = _open_world(n_robots=1)
    w = OracleWorld(params, grid)
    try:
        w.reset(_layout([(10.0, 10.0, 0.0)], [(10.4, 10.0)]))          # 0.4 m ahead: one 0.25 s step at 0.6 m/s gets within 0.3 m
        w.step(np.array([[0.6, 0.0, 0.0]], np.float32))
        s = w.snapshot()
        assert s["is_arrives"][0] == 1 and s["dones"][0] == 1 and s["dones_info"][0] == 5 and s["rewards"][0] == 500.0
        assert abs(s["robot_pose"][0, 0] - (10.0 + float(np.float32(0.6)) * 0.25)) < 1e-12  # v is float32 on the wire (Agent.msg)
    finally:
        w.close()


def test_pedestrian_observation_known_values(oracle_lib):
    """one pedestrian 2 m ahead, 1 m to the left: PedInfo in the robot frame, the ped vector and the ped_map disc"""
    grid, params = _open_world(n_robots=1, n_peds=1, scene="rvoscene")
    w = OracleWorld(params, grid)
    try:
        w.reset(_layout([(10.0, 10.0, 0.0)], [(16.0, 10.0)], ped_xy=[(12.0, 11.0)]))
        s = w.snapshot()
        pv = s["ped_vector_states"][0]
        assert pv[0] == 1.0                                            # number of pedestrians
        assert abs(pv[1] - 2.0) < 1e-6 and abs(pv[2] - 1.0) < 1e-6     # (x, y) in the robot frame
        assert abs(pv[5] - 0.17) < 1e-6 and abs(pv[6] - 0.34) < 1e-6   # r, r + robot_size
        assert abs(pv[7] - np.sqrt(5.0)) < 1e-6                        # distance
        assert abs(s["ped_min_dists"][0] - (np.sqrt(5.0) - 0.34)) < 1e-6
        pm = s["ped_maps"][0]
        occ = np.argwhere(pm[0] == 1.0)
        # yaml_env.py:409-427: cell = (3 - p) // (6 / 48): x -> (3 - 2) / 0.125 = 8, y -> (3 - 1) / 0.125 = 16
        assert len(occ) > 0 and abs(occ[:, 0].mean() - 7.5) < 1.0 and abs(occ[:, 1].mean() - 15.5) < 1.0
        assert pm[0].sum() == len(occ) and 10 <= len(occ) <= 24        # a disc of radius 0.3 m on 0.125 m cells
    finally:
        w.close()


@pytest.mark.parametrize("seed", [3, 4])
def test_shared_owner_layers_equal_the_literal_private_grids(oracle_lib, seed):
    """view_robot stamps every OTHER robot into a private copy of the grid per robot (img_env.cpp:620-629); the shared
    min / max owner layers must give every robot exactly the same view"""
    n = 40
    grid, params, layout = small_world(n, 8, seed=seed, grid_size=100, clearance=0.45, n_obstacles=2)
    a, b = OracleWorld(params, grid), OracleWorld(dict(params, flags=_cabi.FLAG_PRIVATE_GRIDS), grid)
    try:
        a.reset(layout)
        b.reset(layout)
        rng = np.random.default_rng(seed)
        for s in range(25):
            act = random_actions(rng, n)
            a.step(act)
            b.step(act)
            sa, sb = a.snapshot(), b.snapshot()
            for k in ("view_maps", "lasers", "is_collisions", "sensor_maps", "rewards", "dones_info"):
                assert np.array_equal(sa[k], sb[k]), (s, k)
        assert (a.snapshot()["is_collisions"] == 3).any()              # the scenario really had robots on top of each other
    finally:
        a.close()
        b.close()
