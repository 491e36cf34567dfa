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


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: the parts of the unpinned core that had no hand-derived answer yet.  Every expected value below is
# derived in the docstring from the reference's source alone; worlds use 0.5 m cells so that a footprint falls into cells that can
# be named on paper: cell index = round(world / 0.5) (grid_map.cpp:40-44), i.e. cell k covers [0.5 k - 0.25, 0.5 k + 0.25).
def _coarse_world(n_robots, n_peds, cells=60, **kw):
    grid = np.full((cells, cells), 255, np.uint8)
    params = worldgen.make_params(n_robots, n_peds, res=0.5, view_cells=8, beams=8, **kw)
    return grid, params


def test_leg_raster_left_leg_skips_obstacles_right_leg_overwrites_them(oracle_lib):
    """PedAgent::draw_leg (agent.cpp:737-774): left-leg samples skip cells that hold 0 (`== 0 -> is_collision`), right-leg samples
    skip cells that hold 1 ONLY -- so a right leg standing on an obstacle turns the obstacle cell into a pedestrian cell.

    Legs [lx, ly, lr, rx, ry, rr] = [0, 0.15, 0.1, 0, -0.15, 0.1], pedestrian at (x, 10.25), yaw 0, gait state 0 (legs at their
    sizes, agent.cpp:705-712).  Left leg: disc of radius 0.1 around (x, 10.40): y in [10.30, 10.50] -> y / 0.5 in [20.6, 21.0]
    -> column 21; right leg around (x, 10.10): y in [10.00, 10.20] -> [20.0, 20.4] -> column 20; x in [x - 0.1, x + 0.1] -> row
    2 x for x = 10, 15, 20.  Pedestrian A (row 20) stands on two obstacle cells, B (row 30) on free floor, C (row 40) with its
    left leg on a cell of value 100 and its right leg on a literal 2."""
    grid, params = _coarse_world(1, 3, scene="rvoscene", ped_shape="leg", relation_ped_robo=0,
                                 ped_size=np.tile(np.array([0, 0.15, 0.1, 0, -0.15, 0.1], np.float32), (3, 1)))
    grid[20, 21] = 0
    grid[20, 20] = 0
    grid[40, 21] = 100
    grid[40, 20] = 2
    w = OracleWorld(params, grid)
    try:
        w.reset(_layout([(25.0, 25.0, 0.0)], [(27.0, 25.0)], ped_xy=[(10.0, 10.25), (15.0, 10.25), (20.0, 10.25)]))
        w.step(np.zeros((1, 3), np.float32))  # update_bbox has run: state 0, legs at their sizes; nobody moved (goal = own position)
        obs, ped = w.grids()
        assert ped[20, 21] == 0        # A's left leg: the obstacle stays (agent.cpp:751)
        assert ped[20, 20] == 1        # A's right leg: the obstacle BECOMES a pedestrian cell (agent.cpp:767 tests == 1 only)
        assert ped[30, 21] == 1 and ped[30, 20] == 1   # B on free floor
        assert ped[40, 21] == 1        # C's left leg over a 100: anything but 0 is overwritten
        assert ped[40, 20] == 1        # C's right leg over a literal 2: anything but 1 is overwritten
        diff = np.argwhere(ped != obs)
        assert sorted(map(tuple, diff)) == [(20, 20), (30, 20), (30, 21), (40, 20), (40, 21)]  # and nothing else was touched
    finally:
        w.close()


def test_view_ped_and_view_robot_never_overwrite_classes_0_1_2(oracle_lib):
    """view_ped draws circle pedestrians with Agent::draw(peds_map, 1) and view_robot every OTHER robot with draw(map_i, 2)
    (img_env.cpp:594-629); draw only writes where the cell holds none of 0 / 1 / 2 (agent.cpp:313-320).

    Discs of radius 0.17 centred on cell centres (0.5 k) cover x, y in [0.5 k - 0.17, 0.5 k + 0.17] -> exactly cell k.  Pedestrians on
    cells holding 0, 1, 2, 100, 255 leave 0, 1, 2, 1, 1.  Robots: R0 on free floor (20, 20), R1 on a 100 at (30, 30), R2 on the cell
    of the pedestrian that stands on 255, R3 on an obstacle.  Robot i's private map shows every other robot as 2 where the cell held
    anything but 0 / 1 / 2, and never itself."""
    grid, params = _coarse_world(4, 5, scene="rvoscene", relation_ped_robo=0)
    cells = {(10, 10): 0, (10, 12): 1, (10, 14): 2, (10, 16): 100, (10, 18): 255, (30, 30): 100, (44, 44): 0}
    for (m, n), v in cells.items():
        grid[m, n] = v
    w = OracleWorld(params, grid)
    try:
        w.reset(_layout([(10.0, 10.0, 0.3), (15.0, 15.0, -1.0), (5.0, 9.0, 2.0), (22.0, 22.0, 0.0)],
                        [(12.0, 10.0), (17.0, 15.0), (5.0, 11.0), (24.0, 22.0)],
                        ped_xy=[(5.0, 5.0), (5.0, 6.0), (5.0, 7.0), (5.0, 8.0), (5.0, 9.0)]))
        obs, ped = w.grids()
        assert [int(ped[10, n]) for n in (10, 12, 14, 16, 18)] == [0, 1, 2, 1, 1]
        assert sorted(map(tuple, np.argwhere(ped != obs))) == [(10, 16), (10, 18)]
        g0, g1 = w.private_grid(0), w.private_grid(1)
        assert g0[20, 20] == 255 and g1[20, 20] == 2      # R0: invisible to itself, a 2 for R1
        assert g0[30, 30] == 2 and g1[30, 30] == 100      # R1 on a 100: a 2 for the others, the bare map for itself
        assert g0[10, 18] == 1 and g1[10, 18] == 1        # R2 stands on a pedestrian cell: the 1 stays
        assert g0[44, 44] == 0 and g1[44, 44] == 0        # R3 stands on an obstacle: the 0 stays
        expect0 = ped.copy()
        expect0[30, 30] = 2
        assert np.array_equal(g0, expect0)                # nothing else differs from peds_map
        s = w.snapshot()
        assert list(s["is_collisions"]) == [0, 0, 2, 1]   # R2 on the pedestrian: code 2; R3 on the obstacle: code 1
    finally:
        w.close()


def test_collision_code_is_that_of_the_last_footprint_sample_that_hits(oracle_lib):
    """Agent::draw overwrites is_collision with every occupied cell a sample falls on and returns the LAST value (agent.cpp:294-326);
    the samples of a disc are generated x-major, ascending (agent.cpp:18-30), the very last one being (+0.17, 0).

    0.25 m cells, robot centre at x = 10.125 = the boundary between rows 40 and 41 (10.125 / 0.25 = 40.5 rounds away from zero to
    41), y = 10.0 (column 40).  Row 40 column 40 holds a literal 1 ("pedestrian", code 2), row 41 column 40 a 0 (code 1).  Heading
    0: the last sample lies at world (10.295, 10.0) -> row 41 -> code 1.  Heading pi (qz = 1, qw = 0: the rotation is exactly -I): it
    lies at (9.955, 10.0) -> row 40 -> code 2, although the footprint covers the same cells."""
    for (qz, qw), code in (((0.0, 1.0), 1), ((1.0, 0.0), 2)):
        grid = np.full((80, 80), 255, np.uint8)
        grid[40, 40] = 1
        grid[41, 40] = 0
        params = worldgen.make_params(1, 0, res=0.25)
        w = OracleWorld(params, grid)
        try:
            lay = _layout([(10.125, 10.0, 0.0)], [(15.0, 10.0)])
            lay.robot_pose[0, 2:] = (qz, qw)
            w.reset(lay)
            assert w.snapshot()["is_collisions"][0] == code, (qz, qw)
        finally:
            w.close()


def test_crop_gate_of_a_seven_by_seven_view_cell_by_cell(oracle_lib):
    """The crop of Agent::view (agent.cpp:366-404) with view_angle -+0.9 rad and view_min_dist 0.4 on a 7 x 7 view of 0.5 m cells,
    no laser (the view_map is then the crop + the own footprint, agent.cpp:503).

    T_view->base has its origin at (1.75, 1.75) and yaw 3.14159 (agent.cpp:84-88), so view cell (i, j) lies at base
    (xb, yb) = (1.75 - 0.5 i, 1.75 - 0.5 j) up to 3e-6.  A cell passes iff -0.9 < atan2(yb, xb) < 0.9 (tan 0.9 = 1.26) and
    0.4 <= xb <= 10:  row 0 (xb 1.75): |yb| <= 1.75 -> all seven;  row 1 (xb 1.25): |yb| < 1.575 -> j = 1 .. 6 (yb = 1.75 fails);
    row 2 (xb 0.75): |yb| < 0.945 -> j = 2 .. 5;  rows 3 .. 6: xb <= 0.25 < 0.4 -> none.  Everything else keeps 200.
    Robot at (10.25, 10.25), heading 0: view cell (i, j) looks at world (12 - 0.5 i, 12 - 0.5 j) = grid cell (24 - i, 24 - j); a grid
    value < 250 gives 0, >= 250 gives 255 (agent.cpp:394-401).  Own footprint: base (x, y) -> view cell round(3.5 - 2 x), x in
    [-0.17, 0.17] -> 3 or 4: the four centre cells become 100 (they hold 200, not 0 / 1 / 2)."""
    grid = np.full((48, 48), 255, np.uint8)
    grid[24, 24] = 0      # view (0, 0): passes -> 0
    grid[23, 24] = 0      # view (1, 0): outside the angle gate -> stays 200
    grid[22, 21] = 0      # view (2, 3): passes -> 0
    grid[24, 20] = 100    # view (0, 4): < 250 -> 0
    grid[24, 19] = 250    # view (0, 5): >= 250 -> 255
    grid[24, 18] = 249    # view (0, 6): -> 0
    grid[20, 24] = 0      # view (4, 0): behind the min-distance gate -> 200
    grid[22, 18] = 0      # view (2, 6): outside the angle gate -> 200
    params = worldgen.make_params(1, 0, res=0.5, view_cells=7, beams=0, use_laser=False, view_angle_begin=-0.9, view_angle_end=0.9,
                                  view_min_dist=0.4)
    w = OracleWorld(params, grid)
    try:
        w.reset(_layout([(10.25, 10.25, 0.0)], [(14.0, 10.25)]))
        vm = w.snapshot()["view_maps"][0]
        U, F, O, S = 200, 255, 0, 100
        assert vm.tolist() == [[O, F, F, F, O, F, O],
                               [U, F, F, F, F, F, F],
                               [U, U, F, O, F, F, U],
                               [U, U, U, S, S, U, U],
                               [U, U, U, S, S, U, U],
                               [U, U, U, U, U, U, U],
                               [U, U, U, U, U, U, U]]
        assert w.snapshot()["is_collisions"][0] == 0
    finally:
        w.close()


def test_get_corners_of_both_shapes(oracle_lib):
    """Agent::get_corners (agent.cpp:626-651): circle [cx, cy, r] -> base corners (cx - r, cy - r), (cx + r, cy + r); rectangle
    [xmin, xmax, ymin, ymax] -> (xmin, ymin), (xmax, ymax); both through base -> world.
    Circle [0.1, -0.2, 0.3] at (5, 7) heading pi / 2 ((x, y) -> (-y, x)): pa = (5 + 0.5, 7 - 0.2), pb = (5 - 0.1, 7 + 0.4).
    Rectangle [-0.3, 0.5, -0.1, 0.2] at (1, 2) with the 3-4-5 rotation (cos 0.6, sin 0.8): pa = (1 - 0.18 + 0.08, 2 - 0.24 - 0.06),
    pb = (1 + 0.30 - 0.16, 2 + 0.40 + 0.12)."""
    import ctypes as C
    oracle_lib.oracle_test_corners.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]
    oracle_lib.oracle_test_corners.restype = None
    out = np.zeros(4)
    sizes = np.array([0.1, -0.2, 0.3, 0.0])
    oracle_lib.oracle_test_corners(_cabi.SHAPE_CIRCLE, sizes.ctypes.data, 5.0, 7.0, np.pi / 2, out.ctypes.data)
    assert out == pytest.approx([5.5, 6.8, 4.9, 7.4], abs=1e-12)
    sizes = np.array([-0.3, 0.5, -0.1, 0.2])
    oracle_lib.oracle_test_corners(_cabi.SHAPE_RECTANGLE, sizes.ctypes.data, 1.0, 2.0, np.arctan2(0.8, 0.6), out.ctypes.data)
    assert out == pytest.approx([0.9, 1.7, 1.14, 2.52], abs=1e-12)
    oracle_lib.oracle_test_corners(_cabi.SHAPE_RECTANGLE, sizes.ctypes.data, 1.0, 2.0, 0.0, out.ctypes.data)
    assert out.tolist() == [0.7, 1.9, 1.5, 2.2]


def test_leg_gait_over_two_cycles_of_its_state_machine(oracle_lib):
    """PedAgent::update_bbox (agent.cpp:696-735), step_len_ 0.3: state = int((move + remaining) / 0.3 + last), remaining carries
    what is left, state %= 7; legs: states 0 / 4 at their sizes, 1 / 3 left -0.15 right +0.15, 2 left -0.3 right +0.3, 5 left
    +0.15 right -0.15, 6 left +0.3 right -0.3 (only x changes; y keeps +-0.1).

    A recorded walk (dataset scene, img_env.cpp:361-386) along +x, 7 / 32 m per record, heading atan2(0, vx) = 0.  Step s uses record
    s - 1 (step_ counts from 0), so after step s the pedestrian has walked c = (s - 1) 7 / 32 and state = floor(c / 0.3) mod 7:
      s      1  2       3       4        5      6        7       8        9     10       11      12       13      14
      c      0  .21875  .4375   .65625   .875   1.09375  1.3125  1.53125  1.75  1.96875  2.1875  2.40625  2.625   2.84375
      state  0  0       1       2        2      3        4       5        5     6        0       1        1       2
    (no c is within 0.0125 of a multiple of 0.3, so rounding cannot move a state).  Checked through the raster: 1 / 16 m cells, the
    left leg's cells lie in columns > y / res, the right leg's below, and each leg's mean row is its centre (x + leg x) / res."""
    res, y0, x0, dx = 0.0625, 10.0, 10.0, 7.0 / 32.0
    grid = np.full((400, 400), 255, np.uint8)
    params = worldgen.make_params(1, 1, res=res, scene="dataset", ped_shape="leg", relation_ped_robo=0)
    T = 16
    data = np.zeros((1, T, 5))
    data[0, :, 0] = x0 + dx * np.arange(T)
    data[0, :, 1] = y0
    data[0, :, 3] = 0.5
    from img_env_amd import spawn
    lay = _layout([(20.0, 20.0, 0.0)], [(22.0, 20.0)], ped_xy=[(x0, y0)])
    spawn.init_ped_dataset(lay, data)
    expect = [0, 0, 1, 2, 2, 3, 4, 5, 5, 6, 0, 1, 1, 2]
    leg_x = {0: (0.0, 0.0), 4: (0.0, 0.0), 1: (-0.15, 0.15), 3: (-0.15, 0.15), 2: (-0.3, 0.3), 5: (0.15, -0.15), 6: (0.3, -0.3)}
    w = OracleWorld(params, grid)
    try:
        w.reset(lay)
        for s, st in enumerate(expect, start=1):
            w.step(np.zeros((1, 3), np.float32))
            _, ped = w.grids()
            cells = np.argwhere(ped == 1)
            x = x0 + dx * (s - 1)
            left, right = cells[cells[:, 1] > y0 / res], cells[cells[:, 1] < y0 / res]
            assert len(left) >= 6 and len(right) >= 6, s
            lx, rx = leg_x[st]
            assert abs(left[:, 0].mean() * res - (x + lx)) < 0.5 * res, (s, st, left[:, 0].mean() * res - x)
            assert abs(right[:, 0].mean() * res - (x + rx)) < 0.5 * res, (s, st, right[:, 0].mean() * res - x)
            assert abs(left[:, 1].mean() * res - (y0 + 0.1)) < 0.5 * res and abs(right[:, 1].mean() * res - (y0 - 0.1)) < 0.5 * res
    finally:
        w.close()
