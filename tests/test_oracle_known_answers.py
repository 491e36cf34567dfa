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


# ---------------------------------------------------------------------------------------------------------------------
# Agent::bresenhamLine (agent.cpp:511-624), one ray at a time on hand-made maps.  SENT marks cells the ray must not touch.
SENT = 77


def _ray(lib, src, x1, y1, x2, y2, res=0.5):
    import ctypes as C
    lib.oracle_test_bresenham.restype = C.c_double
    lib.oracle_test_bresenham.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double]
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.full_like(src, SENT)
    hit = lib.oracle_test_bresenham(x1, y1, x2, y2, src.ctypes.data, dst.ctypes.data, src.shape[0], src.shape[1], res)
    return hit, dst


def test_bresenham_axis_ray_leaves_the_hit_row_alone(oracle_lib):
    """a ray along +x in column 2 that hits at (4, 2): 255 before, 0 at the hit, and the cells BEHIND it share the hit's column
    (y == end_y), so `(cur.x != end_x) && (cur.y != end_y)` is false and they are left alone (agent.cpp:555-560)"""
    src = np.full((8, 8), 255, np.uint8)
    src[4, 2] = 0
    hit, dst = _ray(oracle_lib, src, 1, 2, 7, 2)
    assert hit == 3 * 0.5                                             # distance between cell centres (map2world), 3 cells
    assert list(dst[1:4, 2]) == [255, 255, 255] and dst[4, 2] == 0
    assert list(dst[5:8, 2]) == [SENT, SENT, SENT]                    # same column as the hit: untouched; (7, 2) is the excluded endpoint
    assert (dst == SENT).sum() == 64 - 4                              # nothing else was written


def test_bresenham_diagonal_ray_marks_200_behind_the_hit(oracle_lib):
    """w == h takes the y-stepping branch (`w > h` is false): cells (1,1) .. (6,6), endpoint excluded; behind the hit at (3,3)
    every cell differs from it in row AND column -> 200"""
    src = np.full((8, 8), 255, np.uint8)
    src[3, 3] = 0
    src[5, 5] = 0                                                     # a second obstacle behind the first is not a hit
    hit, dst = _ray(oracle_lib, src, 1, 1, 7, 7)
    assert hit == pytest.approx(2 * np.sqrt(2.0) * 0.5, abs=1e-15)
    assert dst[1, 1] == 255 and dst[2, 2] == 255 and dst[3, 3] == 0
    assert dst[4, 4] == 200 and dst[5, 5] == 200 and dst[6, 6] == 200 and dst[7, 7] == SENT
    assert (dst != SENT).sum() == 6


def test_bresenham_shallow_ray_mixes_alone_and_200(oracle_lib):
    """(0,0) -> (6,2): f = 2h - w = -2, delta1 = 4, delta2 = -8 gives the cells (0,0) (1,0) (2,1) (3,1) (4,1) (5,2).  Hit at
    (2,1): (3,1) and (4,1) share its column -> left alone; (5,2) shares neither -> 200"""
    src = np.full((8, 8), 255, np.uint8)
    src[2, 1] = 0
    hit, dst = _ray(oracle_lib, src, 0, 0, 6, 2)
    assert hit == pytest.approx(np.sqrt(5.0) * 0.5, abs=1e-15)
    assert dst[0, 0] == 255 and dst[1, 0] == 255 and dst[2, 1] == 0
    assert dst[3, 1] == SENT and dst[4, 1] == SENT and dst[5, 2] == 200
    assert (dst != SENT).sum() == 4
    # the same ray mirrored (dx = dy = -1): (6,2) -> (0,0) visits (6,2) (5,2) (4,1) (3,1) (2,1) (1,0)
    src = np.full((8, 8), 255, np.uint8)
    src[4, 1] = 0
    hit, dst = _ray(oracle_lib, src, 6, 2, 0, 0)
    assert hit == pytest.approx(np.sqrt(5.0) * 0.5, abs=1e-15)
    assert dst[6, 2] == 255 and dst[5, 2] == 255 and dst[4, 1] == 0
    assert dst[3, 1] == SENT and dst[2, 1] == SENT and dst[1, 0] == 200 and dst[0, 0] == SENT


def test_bresenham_edge_cases(oracle_lib):
    free = np.full((8, 8), 255, np.uint8)
    hit, dst = _ray(oracle_lib, free, 3, 3, 3, 3)                     # start == end: `y != y2` fails at once, no cell is touched
    assert hit == 6.0 and (dst == SENT).all()                          # 6 is the hard-coded "no hit" (agent.cpp:513)
    hit, dst = _ray(oracle_lib, free, 5, 5, 12, 5)                    # the ray leaves the view: returns at the first outside cell
    assert hit == 6.0 and list(dst[5:8, 5]) == [255, 255, 255] and (dst != SENT).sum() == 3
    unknown = np.full((8, 8), 200, np.uint8)                          # 200 (outside the field of view) counts as free: `cur_data != 0`
    unknown[6, 5] = 0
    hit, dst = _ray(oracle_lib, unknown, 2, 5, 7, 5)
    assert hit == 4 * 0.5 and list(dst[2:7, 5]) == [255, 255, 255, 255, 0]
    start_on_obstacle = free.copy()
    start_on_obstacle[2, 2] = 0                                       # the sensor's own cell is on the path: distance 0
    hit, dst = _ray(oracle_lib, start_on_obstacle, 2, 2, 2, 6)
    assert hit == 0.0 and dst[2, 2] == 0 and list(dst[2, 3:6]) == [SENT, SENT, SENT]  # same row as the hit: left alone


# ---------------------------------------------------------------------------------------------------------------------
# SpeedLimiter::limit (speed_limit.cpp:92-173) and Agent::cmd (agent.cpp:186-283) through one-robot worlds; state_dim = 4
# puts the limited (v, w) into the vector state (agent.cpp:172-175).  Every number below is a dyadic fraction, so the
# expected values are exact.
def _one_robot(oracle_lib, goal=(90.0, 90.0), yaw=0.0, **kw):
    grid = np.full((800, 800), 255, np.uint8)
    params = worldgen.make_params(1, 0, res=RES, state_dim=4, **kw)
    w = OracleWorld(params, grid)
    w.reset(_layout([(10.0, 10.0, yaw)], [goal]))
    return w


def _cmd(w, v, wv, v_y=0.0):
    w.step(np.array([[v, wv, v_y]], np.float32))
    s = w.snapshot()
    return float(s["vector_states"][0, 2]), float(s["vector_states"][0, 3]), s["robot_pose"][0].copy()


def test_acceleration_limiter_known_sequence(oracle_lib):
    """limit_acceleration with min / max acceleration -+0.5 and dt = 0.25: |dv| <= 0.125 per step, and the zero-crossing
    branch (`v_sign + v0_sign == 0`) first brakes to a stop"""
    lim = dict(has_acceleration_limits=True, min_acceleration=-0.5, max_acceleration=0.5)
    w = _one_robot(oracle_lib, limiter_v=lim)
    try:
        assert _cmd(w, 0.5, 0.0)[0] == 0.125        # from rest: dv = +0.5 -> clamp(|dv|, -0.125, 0.125) = 0.125
        assert _cmd(w, 0.5, 0.0)[0] == 0.25         # v0 = 0.125, again +0.125
        assert _cmd(w, 0.0, 0.0)[0] == 0.125        # braking: dv_sign = -1 matches neither sign -> -|clamp(-0.25, ...)| = -0.125
        assert _cmd(w, -0.5, 0.0)[0] == 0.0         # signs oppose: zero_dt = |0.125 / -0.5| = 0.25 >= dt -> v0 - 0.5 * 0.25 = 0
        assert _cmd(w, -0.5, 0.0)[0] == -0.125      # from rest backwards
        v, _, _ = _cmd(w, 0.5, 0.0)                 # signs oppose again: -(0.125 - 0.125) = -0.0
        assert v == 0.0
    finally:
        w.close()


def test_velocity_and_jerk_limiters_known_values(oracle_lib):
    """limit_velocity clamps last; limit_jerk works on the last two commands with da in [min_jerk, max_jerk] * 2 dt^2 where the
    node's constructor copies msg.min_jerk into max_jerk and never sets min_jerk (speed_limit.cpp:56-65; 0 here)"""
    w = _one_robot(oracle_lib, limiter_v=dict(has_velocity_limits=True, min_velocity=0.0, max_velocity=0.25),
                   limiter_w=dict(has_velocity_limits=True, min_velocity=-0.5, max_velocity=0.5))
    try:
        v, wv, _ = _cmd(w, 0.5, 0.875)
        assert (v, wv) == (0.25, 0.5)
        v, wv, _ = _cmd(w, 0.125, -0.75)
        assert (v, wv) == (0.125, -0.5)
    finally:
        w.close()
    w = _one_robot(oracle_lib, limiter_v=dict(has_jerk_limits=True, min_jerk=1.0, max_jerk=-123.0))  # msg.max_jerk is ignored
    try:
        assert _cmd(w, 0.5, 0.0)[0] == 0.125        # dv = 0.5, dv0 = 0: da = clamp(0.5, 0, 1 * 0.125) = 0.125
        assert _cmd(w, 0.5, 0.0)[0] == 0.375        # v0 = 0.125, v1 = 0: dv0 = 0.125, da = clamp(0.25, 0, 0.125) -> 0.125 + 0.125 + 0.125
        assert _cmd(w, 0.0, 0.0)[0] == 0.625        # dv - dv0 = -0.625 clamps to min_jerk * dt2 = 0: v0 + dv0 = 0.375 + 0.25
    finally:
        w.close()


def test_diff_drive_pose_update_known_values(oracle_lib):
    """the exact arc of agent.cpp:221-236: w == 0 straight, else r = v / w about the instantaneous centre; theta is not wrapped"""
    w = _one_robot(oracle_lib)
    try:
        _, _, p = _cmd(w, 0.5, 0.0)
        assert list(p) == [10.125, 10.0, 0.0]                                    # x += v dt cos(0)
        assert list(w.records[0, 3:5]) == [0.5, 0.0]                             # Agent::vx, vy of the last 0.05 s sub-step
        _, _, p = _cmd(w, 0.5, 0.5)                                              # r = 1: x += sin(0.125), y += 1 - cos(0.125)
        assert p[0] == pytest.approx(10.125 + np.sin(0.125), abs=2e-15) and p[1] == pytest.approx(10.0 + (1.0 - np.cos(0.125)), abs=2e-15)
        assert p[2] == 0.125
        for _ in range(60):
            _, _, p = _cmd(w, 0.0, 0.875)
        assert p[2] == pytest.approx(0.125 + 60 * 0.875 * 0.25, abs=1e-12) and p[2] > 4 * np.pi   # never wrapped (agent.cpp:233)
    finally:
        w.close()
    w = _one_robot(oracle_lib, yaw=np.pi / 2)
    try:
        _, _, p = _cmd(w, 0.5, 0.0)
        assert p[0] == pytest.approx(10.0, abs=1e-15) and p[1] == pytest.approx(10.125, abs=1e-15)
    finally:
        w.close()


def test_omni_pose_update_known_values(oracle_lib):
    """omni adds the lateral terms (agent.cpp:238-274) and, unlike diff, never updates Agent::vx / vy"""
    w = _one_robot(oracle_lib, robot_ktype="omni")
    try:
        _, _, p = _cmd(w, 0.5, 0.0, 0.25)
        assert list(p) == [10.125, 10.0625, 0.0]                                 # x += v dt, y += v_y dt at theta = 0
        assert list(w.records[0, 3:5]) == [0.0, 0.0]                             # vx, vy untouched
        _, _, p = _cmd(w, 0.5, 0.5, 0.25)                                        # v / w = 1, v_y / w = 0.5, theta: 0 -> 0.125
        s, c = np.sin(0.125), np.cos(0.125)
        assert p[0] == pytest.approx(10.125 + s + (-0.5 + 0.5 * c), abs=2e-15)
        assert p[1] == pytest.approx(10.0625 + (1.0 - c) + 0.5 * s, abs=2e-15)
        assert p[2] == 0.125
    finally:
        w.close()


def test_substep_loop_runs_six_times_for_a_quarter_second(oracle_lib):
    """`while (cur_control <= step_hz_)` with cur_control += 0.05 reaches exactly 0.25 on the fifth addition, so the loop body
    runs SIX times: odom_pose_ travels 0.3 s worth (0.18 m at 0.6 m/s) although the pose itself advances 0.25 s (0.15 m).  A goal
    0.47 m ahead is therefore reached by the sub-step test (0.29 <= 0.3) but not by the final one (0.32); at 0.49 m by neither."""
    for goal_x, arrives in ((10.47, 1), (10.49, 0)):
        w = _one_robot(oracle_lib, goal=(goal_x, 10.0))
        try:
            w.step(np.array([[0.6, 0.0, 0.0]], np.float32))
            s = w.snapshot()
            assert s["is_arrives"][0] == arrives, goal_x
            assert s["robot_pose"][0, 0] == pytest.approx(10.0 + float(np.float32(0.6)) * 0.25, abs=1e-15)
        finally:
            w.close()


def test_lottery_stream_is_glibc_rand(oracle_lib):
    """the oracle's restatement of glibc's TYPE_3 rand() (the beep lottery's stream, img_env.cpp:327) against the real
    libc of this container, default seed and another one"""
    import ctypes as C
    libc = C.CDLL("libc.so.6")
    for seed in (1, 20240611):
        out = (C.c_int32 * 4000)()
        oracle_lib.oracle_test_glibc_rand(C.c_uint(seed), 4000, out)
        libc.srand(seed)
        assert list(out) == [libc.rand() for _ in range(4000)], seed
    assert out[0] >= 0 and max(out) <= 2147483647


def test_beep_pushes_an_ervo_pedestrian_away(oracle_lib):
    """one robot, one standing ERVO pedestrian 0.8 m to its left (its goal is where it stands): a beep of radius 1.5 with
    ped_ca_p = 1 adds the unit vector robot -> pedestrian to its new velocity (ervo_ros Agent.cpp:63-69, unclamped), the
    same request without beep (v_y = 0), with ped_ca_p = 0 or out of range leaves it standing"""
    for beep_r, p, v_y, pushed in ((1.5, 1.0, 0.2, True), (1.5, 1.0, 0.0, False), (1.5, 0.0, 0.2, False), (0.5, 1.0, 0.2, False)):
        grid, params = _open_world(n_robots=1, n_peds=1, scene="ervoscene", relation_ped_robo=0, beep_r=beep_r, ped_ca_p=p)
        w = OracleWorld(params, grid)
        try:
            w.reset(_layout([(10.0, 10.0, 0.0)], [(20.0, 10.0)], ped_xy=[(10.0, 10.8)]))
            w.step(np.array([[0.0, 0.0, v_y]], np.float32))
            ps = w.snapshot()["ped_state"][0]
            if pushed:  # velocity (0, 1) for one step of 0.25 s; float32 arithmetic
                assert abs(ps[2]) < 1e-6 and abs(ps[3] - 1.0) < 1e-6 and abs(ps[1] - 11.05) < 1e-5
            else:
                assert abs(ps[2]) < 1e-6 and abs(ps[3]) < 1e-6 and abs(ps[1] - 10.8) < 1e-6
        finally:
            w.close()
