"""GPU parity: the HIP step() path through the C ABI vs the CPU oracle on the same seeded inputs,
and vs the golden vectors of the reference's Python."""
import ast
import glob
import os

import numpy as np
import pytest

from parity import CLOSE, EXACT, compare, run_pair
from scenarios import clip_actions, golden_scenario, random_actions, small_world

pytestmark = pytest.mark.gpu

CASES = {
    "1robot_nopeds": dict(n_robots=1, n_peds=0, seed=0, n_obstacles=0),
    "cfg1_like": dict(n_robots=1, n_peds=0, seed=4, n_obstacles=4),
    "8robots_6peds_rvo": dict(n_robots=8, n_peds=6, seed=1, n_obstacles=4),
    "64robots_20peds_rvo": dict(n_robots=64, n_peds=20, seed=2, n_obstacles=4, clearance=0.8),
    "legs_ervo": dict(n_robots=12, n_peds=9, seed=3, n_obstacles=2, ped_shape="leg", scene="ervoscene"),
    "dense_collisions": dict(n_robots=96, n_peds=30, seed=5, grid_size=100, clearance=0.5, n_obstacles=2),
    "state5_norel": dict(n_robots=6, n_peds=5, seed=6, state_dim=5, relation_ped_robo=0),
    "res025": dict(n_robots=32, n_peds=12, seed=7, res=0.25, grid_size=120),
    "res010_not_pow2": dict(n_robots=24, n_peds=8, seed=13, res=0.1, grid_size=200),
    "pedscene_sfm_11m": dict(n_robots=4, n_peds=7, seed=14, scene="pedscene", grid_size=88, n_obstacles=3),
    "pedscene_sfm_16m_legs": dict(n_robots=6, n_peds=14, seed=15, scene="pedscene", grid_size=128, n_obstacles=2,
                                  ped_shape="leg", clearance=0.8),
    "pedscene_norel": dict(n_robots=12, n_peds=9, seed=16, scene="pedscene", grid_size=88, relation_ped_robo=0,
                           clearance=0.7),
    # pedestrian sort variants of k_obs: 64 * E register slots (E = 2, 4, 8, 16) and the LDS sort beyond 1024
    "peds_100_sort2": dict(n_robots=6, n_peds=100, seed=20, grid_size=240, clearance=0.6),
    "peds_200_sort4": dict(n_robots=6, n_peds=200, seed=21, grid_size=320, clearance=0.6),
    "peds_300_sort8": dict(n_robots=5, n_peds=300, seed=22, grid_size=400, clearance=0.6),
    "peds_600_sort16": dict(n_robots=4, n_peds=600, seed=23, grid_size=400, clearance=0.5),
    "peds_1100_lds_sort": dict(n_robots=3, n_peds=1100, seed=24, grid_size=480, clearance=0.5),
    # view widths that are not a multiple of 4 (generic k_view variants), power-of-two resolution and not
    "view_50_not_a4": dict(n_robots=16, n_peds=6, seed=25, view_cells=50),
    "view_37_odd_res010": dict(n_robots=16, n_peds=6, seed=26, view_cells=37, res=0.1, grid_size=200,
                               view_width=3.75, view_height=3.75),  # int(3.75 / 0.1) = 37 cells
    # BASELINE cfg-5 geometry: 96 x 96 view, 720 beams
    "view96_720beams": dict(n_robots=10, n_peds=6, seed=27, view_cells=96, beams=720, grid_size=240),
    # the tiled kernels (csrc/view_big.h) on views k_view could also take: full-size view and float16 sensor_map from
    # k_fullview_big, nothing shrunk (IMGENV_FLAG_VIEW_TILED)
    "view96_720beams_tiled": dict(n_robots=10, n_peds=6, seed=27, view_cells=96, beams=720, grid_size=240),
    "view_50_not_a4_tiled": dict(n_robots=16, n_peds=6, seed=25, view_cells=50),
    "no_laser_tiled": dict(n_robots=5, n_peds=3, seed=8, use_laser=False),
    "no_laser": dict(n_robots=5, n_peds=3, seed=8, use_laser=False),
    "time_limit": dict(n_robots=4, n_peds=2, seed=9, time_max=6),
    # k_orca's obstacle phases: more than 16 obstacle neighbours per pedestrian (several rounds of candidate lines per row, lines
    # of earlier rounds covering later segments), a pedestrian count that leaves rows of the last group empty; and an obstacle
    # table beyond the LDS staging area (280 segments: the whole solve on the home lane, out of HBM)
    "orca_many_obstacle_segments": dict(n_robots=5, n_peds=14, seed=28, n_obstacles=24, grid_size=112, clearance=0.6),
    "orca_obstacle_table_beyond_lds": dict(n_robots=6, n_peds=9, seed=29, n_obstacles=70, clearance=0.6, ped_shape="leg",
                                           scene="ervoscene"),
}


@pytest.fixture(scope="module")
def worlds():
    import torch
    assert torch.cuda.is_available()
    from img_env_amd.world import World
    from oracle_binding import OracleWorld, build_oracle
    build_oracle()
    return World, OracleWorld


# the same scenarios on the counting class layer (SUM mode, world.h), which small handles do not pick by themselves
SUM_CASES = ["8robots_6peds_rvo", "64robots_20peds_rvo", "legs_ervo", "dense_collisions", "res025", "res010_not_pow2", "pedscene_sfm_16m_legs",
             "view_50_not_a4", "time_limit", "orca_many_obstacle_segments", "1robot_nopeds"]


@pytest.mark.parametrize("case", list(CASES) + [c + "+counting_layer" for c in SUM_CASES])
def test_hip_matches_oracle(worlds, case):
    World, OracleWorld = worlds
    from img_env_amd import _cabi
    case, _, layer = case.partition("+")
    kw = dict(CASES[case])
    n = kw["n_robots"]
    grid, params, layout = small_world(**kw)
    if layer:
        params = dict(params, flags=int(params.get("flags", 0)) | _cabi.FLAG_LAYER_SUM)
    if case.endswith("_tiled"):
        params = dict(params, flags=int(params.get("flags", 0)) | _cabi.FLAG_VIEW_TILED)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        rng = np.random.default_rng(kw["seed"] + 50)
        acts = [random_actions(rng, n) for _ in range(40)]
        fails = run_pair(gpu, cpu, layout, acts)
        assert not fails, fails[:3]
        # the episode really exercised the path
        snap = cpu.snapshot()
        assert snap["counters"][0] == 40
    finally:
        gpu.close()
        cpu.close()


def _room_map(px, seed):
    """a room like the reference's room_10.png: white floor, black walls, a few black blocks"""
    rng = np.random.default_rng(seed)
    m = np.full((px, px), 255, np.uint8)
    m[:5] = m[-5:] = 0
    m[:, :5] = m[:, -5:] = 0
    for _ in range(3):
        a, b = rng.integers(15, px - 25, 2)
        m[a:a + rng.integers(3, 9), b:b + rng.integers(3, 9)] = 0
    return m


RESAMPLED = {
    # the geometry of the reference's envs/cfg/test.yaml: a 110 x 110 pixel map at 0.1 m resized (INTER_LINEAR) to 733 x 733
    # cells of 0.015 m, a 6 m = 400 x 400 cell view shrunk (INTER_CUBIC) to 48 x 48, 1000 beams, 4 leg pedestrians, 4 obstacles
    "test_yaml_geometry": dict(n_robots=1, n_peds=4, seed=61, map_px=110, global_res=0.1, res=0.015, view_m=6.0, beams=1000,
                               image=48, ped_shape="leg", dt=0.4, n_obstacles=4, steps=12),
    # the same with several robots that see each other (circle.yaml has two), pedestrians as discs, ERVO
    "shipped_view_three_robots": dict(n_robots=3, n_peds=5, seed=62, map_px=110, global_res=0.1, res=0.015, view_m=6.0, beams=1000,
                                      image=48, scene="ervoscene", dt=0.1, n_obstacles=3, steps=10, clearance=1.2),
    # only the sensor_map is shrunk (96 -> 48), power-of-two cells; and only the map is resized (0.25 m pixels -> 0.125 m cells)
    "shrink_96_to_48": dict(n_robots=10, n_peds=6, seed=63, map_px=200, global_res=0.125, res=0.125, view_m=12.0, beams=360,
                            image=48, n_obstacles=2, steps=15),
    "map_upscaled_twice": dict(n_robots=12, n_peds=6, seed=64, map_px=120, global_res=0.25, res=0.125, view_m=6.0, beams=360,
                               image=48, n_obstacles=2, steps=15),
    # a view that is not square in cells after the shrink target, odd sizes, no laser
    "shrink_to_84_no_laser": dict(n_robots=4, n_peds=3, seed=65, map_px=150, global_res=0.1, res=0.05, view_m=10.0, beams=0,
                                  image=84, use_laser=False, n_obstacles=2, steps=10),
}


@pytest.mark.parametrize("view_maps", [True, False], ids=["full_view", "no_view_maps"])
@pytest.mark.parametrize("case", list(RESAMPLED))
def test_resampled_maps_and_large_views_match_oracle(worlds, case, view_maps):
    """cv::resize INTER_LINEAR at map load (grid_map.cpp:28-38), views of up to 400 x 400 cells, and the
    cv2.resize INTER_CUBIC shrink of the sensor_map (yaml_env.py:431-438) -- the kernels of csrc/view_big.h and the host
    resize against the oracle; once with the full-size view as an output, once without (IMGENV_FLAG_NO_VIEW_MAPS: only the
    view cells the shrink reads are evaluated)"""
    World, OracleWorld = worlds
    from img_env_amd import _cabi, worldgen
    kw = dict(RESAMPLED[case])
    n, P, steps = kw.pop("n_robots"), kw.pop("n_peds"), kw.pop("steps")
    src = _room_map(kw.pop("map_px"), kw["seed"])
    gres, res, view_m, image = kw.pop("global_res"), kw.pop("res"), kw.pop("view_m"), kw.pop("image")
    seed, n_obs, clearance = kw.pop("seed"), kw.pop("n_obstacles"), kw.pop("clearance", 1.0)
    params = worldgen.make_params(n, P, res=res, view_cells=1, **kw)
    params.update(global_resolution=gres, view_width=view_m, view_height=view_m, image_size=(image, image))
    layout = worldgen.make_layout(src, gres, n, P, seed=seed, n_obstacles=n_obs, clearance=clearance)
    gpu = World(dict(params, flags=int(params.get("flags", 0)) | (0 if view_maps else _cabi.FLAG_NO_VIEW_MAPS)), src)
    cpu = OracleWorld(params, src)
    try:
        o = cpu.out
        assert o["sensor_maps"].shape == (n, image, image) and o["view_maps"].shape[1] == int(float(np.float32(view_m)) / float(np.float32(res)))  # agent.cpp:81-83, doubles
        rng = np.random.default_rng(seed)
        fields = tuple(f for f in EXACT + CLOSE if view_maps or f != "view_maps")
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(steps)], fields=fields)
        assert not fails, fails[:3]
        if not view_maps and image * image != o["view_maps"][0].size:
            assert not gpu.snapshot()["view_maps"].any()  # never written
        sm = cpu.snapshot()["sensor_maps"].astype(np.float32)
        assert sm.min() >= 0.0 and sm.max() <= 1.0 and len(np.unique(sm)) > 3  # the shrink really interpolated
    finally:
        gpu.close()
        cpu.close()


def test_orca_pedestrians_inside_an_obstacle_lattice(worlds):
    """38 small obstacles on a 1.1 m lattice around the pedestrians: 50-90 obstacle segments within a pedestrian's 3 m range, i.e.
    several 16-wide rounds of candidate ORCA lines per row of k_orca, lines of earlier rounds covering later segments, ties in the
    distance order (two segments meeting at the nearest corner) settled by the tree's visiting order (Agent.cpp:437-671, 813-838;
    KdTree.cpp:310-353)"""
    World, OracleWorld = worlds
    from img_env_amd import _cabi
    n, P = 4, 10
    grid, params, layout = small_world(n, P, seed=33, grid_size=96, n_obstacles=0, clearance=0.7)
    pts = np.vstack([layout.robot_pose[:, :2], layout.ped_pose[:, :2], layout.robot_goal, layout.ped_goal])
    obs = []
    for a in np.arange(2.6, 9.6, 1.1):
        for b in np.arange(2.6, 9.6, 1.1):
            if np.linalg.norm(pts - (a, b), axis=1).min() > 0.55:
                obs.append((a, b))
    obs = np.array(obs[:60])
    assert 30 <= len(obs) <= 64  # <= 256 segments: staged in LDS
    rng = np.random.default_rng(33)
    yaw = rng.uniform(-3.14, 3.14, len(obs))
    layout.obs_shape = np.where(np.arange(len(obs)) % 2 == 0, _cabi.SHAPE_CIRCLE, _cabi.SHAPE_RECTANGLE).astype(np.int32)
    layout.obs_size = np.where((np.arange(len(obs)) % 2 == 0)[:, None], np.float32([0, 0, 0.12, 0]), np.float32([-0.1, 0.1, -0.08, 0.08])).astype(np.float32)
    layout.obs_pose = np.zeros((len(obs), 4))
    layout.obs_pose[:, :2], layout.obs_pose[:, 2], layout.obs_pose[:, 3] = obs, np.sin(yaw / 2), np.cos(yaw / 2)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(40)])
        assert not fails, fails[:3]
        assert np.abs(cpu.snapshot()["ped_state"][:, 2:]).max() > 0.02  # the crowd moves between the obstacles
    finally:
        gpu.close()
        cpu.close()


def test_second_reset_reuses_handle(worlds):
    World, OracleWorld = worlds
    grid, params, layout = small_world(10, 5, seed=11)
    _, _, layout2 = small_world(10, 5, seed=12)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        rng = np.random.default_rng(1)
        assert not run_pair(gpu, cpu, layout, [random_actions(rng, 10) for _ in range(10)])
        assert not run_pair(gpu, cpu, layout2, [random_actions(rng, 10) for _ in range(10)])
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("case", ["small_view", "shipped_geometry", "view_50_not_a4"])
def test_agent_state_extras_match_oracle(worlds, case):
    """AgentState.hits_x / hits_y / angular_map (AgentState.msg:4-6, agent.cpp:405-438, img_env.cpp:558-560): produced on request
    (IMGENV_FLAG_AGENT_STATE_EXTRAS) by k_view and by the big-view kernels, against the oracle's literal beam loop"""
    World, OracleWorld = worlds
    from img_env_amd import _cabi, worldgen
    from parity import EXTRAS
    if case == "shipped_geometry":
        kw = dict(RESAMPLED["test_yaml_geometry"])
        n, P, steps = kw.pop("n_robots"), kw.pop("n_peds"), kw.pop("steps")
        grid = _room_map(kw.pop("map_px"), kw["seed"])
        gres, res, view_m, image = kw.pop("global_res"), kw.pop("res"), kw.pop("view_m"), kw.pop("image")
        seed, n_obs = kw.pop("seed"), kw.pop("n_obstacles")
        params = worldgen.make_params(n, P, res=res, view_cells=1, **kw)
        params.update(global_resolution=gres, view_width=view_m, view_height=view_m, image_size=(image, image))
        layout = worldgen.make_layout(grid, gres, n, P, seed=seed, n_obstacles=n_obs)
    else:
        n, steps = 12, 25
        grid, params, layout = small_world(n, 5, seed=41, n_obstacles=3, **({"view_cells": 50} if case == "view_50_not_a4" else {}))
    params = dict(params, flags=int(params.get("flags", 0)) | _cabi.FLAG_AGENT_STATE_EXTRAS)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        assert all(k in gpu.out for k in EXTRAS) and gpu.out["angular_map"].shape == (n, 72)
        rng = np.random.default_rng(3)
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(steps)], fields=EXACT + CLOSE + EXTRAS)
        assert not fails, fails[:3]
        s = cpu.snapshot()
        assert (s["angular_map"] < s["angular_map"].max()).any() and np.abs(s["hits_x"]).max() > 0.5  # something was hit
    finally:
        gpu.close()
        cpu.close()


def test_aborted_step_is_recovered_by_reset(worlds):
    """a step that was begun and never ended (the caller's exchange failed between imgenv_step_begin and imgenv_step_end) leaves
    the fused tails' hand-over words half filled; a reset recovers the handle and the next episode matches the oracle"""
    World, OracleWorld = worlds
    grid, params, layout = small_world(70, 9, seed=31)   # more than one group of 64 robots
    _, _, layout2 = small_world(70, 9, seed=32)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        rng = np.random.default_rng(5)
        assert not run_pair(gpu, cpu, layout, [random_actions(rng, 70) for _ in range(4)])
        gpu.step_begin(random_actions(rng, 70))  # k_integrate + k_obs are in flight; imgenv_step_end never comes
        with pytest.raises(RuntimeError):
            gpu.step_begin(random_actions(rng, 70))
        assert not run_pair(gpu, cpu, layout2, [random_actions(rng, 70) for _ in range(8)])
    finally:
        gpu.close()
        cpu.close()


FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "python_post_*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_hip_matches_reference_python_golden(worlds, path):
    """HIP outputs vs what the reference's own Python produced (tests/golden/gen_python_golden.py)"""
    World, _ = worlds
    z = np.load(path)
    meta = ast.literal_eval(str(z["meta"]))
    grid, params, layout = golden_scenario(meta)
    gpu = World(params, grid)
    try:
        gpu.reset(layout)
        snaps = [gpu.snapshot()]
        for s in range(meta["steps"]):
            gpu.step(clip_actions(z["actions"][s]))
            snaps.append(gpu.snapshot())
    finally:
        gpu.close()
    for t, s in enumerate(snaps):
        assert np.array_equal(s["is_collisions"], z["exp_is_collisions"][t]), t
        assert np.array_equal(s["is_arrives"].astype(bool), z["exp_is_arrives"][t]), t
        assert np.array_equal(s["sensor_maps"], z["exp_sensor_maps"][t]), t
        assert np.abs(s["vector_states"] - z["exp_vector_states"][t]).max() <= 1e-4
        if z["exp_lasers"][t].size:
            assert np.abs(s["lasers"] - z["exp_lasers"][t]).max() <= 1e-4
        assert np.abs(s["ped_maps"] - z["exp_ped_maps"][t]).max() <= 1e-4
        assert np.abs(s["ped_vector_states"] - z["exp_ped_vector_states"][t]).max() <= 1e-4
        assert np.abs(s["step_ds"] - z["exp_step_ds"][t]).max() <= 1e-4
        if t > 0:
            assert np.abs(s["rewards"] - z["exp_rewards"][t - 1]).max() <= 1e-4
            assert np.array_equal(s["dones"], z["exp_dones"][t - 1])
            assert np.array_equal(s["dones_info"], z["exp_dones_info"][t - 1])


def test_step_before_reset_is_an_error(worlds):
    World, _ = worlds
    import torch
    grid, params, layout = small_world(2, 0, seed=1)
    gpu = World(params, grid)
    try:
        with pytest.raises(RuntimeError, match="step before reset"):
            gpu.step(torch.zeros(2, 3, device="cuda"))
    finally:
        gpu.close()


def _exchange_by_hand(ranks, bounds):
    """what the all-gather of a robot-sharded world delivers: every rank's own slice of the records, to every other rank"""
    import torch
    torch.cuda.synchronize()
    for r, w in enumerate(ranks):
        for q, o in enumerate(ranks):
            if q != r:
                w.records[bounds[q]:bounds[q + 1]].copy_(o.records[bounds[q]:bounds[q + 1]])


SHARD_FIELDS = ("vector_states", "view_maps", "sensor_maps", "lasers", "ped_vector_states", "ped_maps", "is_collisions", "is_arrives", "rewards",
                "dones", "dones_info", "step_ds", "ped_min_dists")


@pytest.mark.parametrize("layer", ["counting", "composed"])
@pytest.mark.parametrize("by_x", [False, True], ids=["interleaved", "spatial_shards"])
def test_two_sharded_handles_match_single_world(worlds, by_x, layer):
    """Two handles on one GPU play ranks 0 / 1 of a robot-sharded world (caller-owned exchange between step_begin and
    step_end).  Every shard must equal the same robots of the unsharded world bit for bit -- with the counting layer (the
    other rank's robots arrive as bitmaps in their records, k_remote) and with the composed owner layers (each rank rasterises
    the other's robots itself, clipped to what its own robots can see: spatially separated shards)."""
    from img_env_amd import _cabi
    World, _ = worlds
    n, n_peds, steps = 24, 10, 30
    grid, params, layout = small_world(n, n_peds, seed=31, grid_size=320, clearance=0.8)
    if layer == "composed":
        params = dict(params, flags=int(params.get("flags", 0)) | _cabi.FLAG_COMPOSE_DENSE)
    if by_x:  # contiguous index ranges = vertical strips of the map
        order = np.argsort(layout.robot_pose[:, 0], kind="stable")
        layout.robot_pose = layout.robot_pose[order].copy()
        layout.robot_goal = layout.robot_goal[order].copy()
    full = World(params, grid)
    ranks = [World(dict(params, robot_begin=r * n // 2, robot_end=(r + 1) * n // 2), grid) for r in range(2)]
    try:
        for w in ranks:
            mode = w.layer_mode()
            assert mode["layer"] == layer and mode["shard_bitmaps"] == (layer == "counting") and mode["early_observation"], mode
        full.reset(layout)
        for w in ranks:
            w.reset(layout)
        rng = np.random.default_rng(5)
        for s in range(steps):
            a = random_actions(rng, n)
            full.step(a)
            for r, w in enumerate(ranks):
                w.step_begin(a[r * n // 2:(r + 1) * n // 2])
            _exchange_by_hand(ranks, [0, n // 2, n])
            for w in ranks:
                w.step_end()
            want = full.snapshot()
            for r, w in enumerate(ranks):
                got = w.snapshot()
                sl = slice(r * n // 2, (r + 1) * n // 2)
                for k in SHARD_FIELDS:
                    assert np.array_equal(got[k], want[k][sl]), (s, r, k)
    finally:
        full.close()
        for w in ranks:
            w.close()


@pytest.mark.parametrize("case", ["dense_circles", "rectangles", "mixed_classes", "no_pedestrians", "fine_grid_falls_back"])
def test_four_shards_through_a_reset_match_single_world(worlds, case):
    """ranks 0..3 of a crowded world -- robots brush past each other and collide, so footprints of different ranks share cells
    and leave them again -- over 25 steps, a reset onto another layout, and 15 more steps.  `rectangles`: a 7 x 7-cell footprint
    box; `mixed_classes`: three robot classes with different boxes (the bitmap's radius read per robot) and sensor offsets;
    `no_pedestrians`: no side streams at all; `fine_grid_falls_back`: 0.05 m cells, where a footprint's box no longer fits the
    record's 64-bit bitmap and the shards keep the composed owner layers."""
    from img_env_amd import worldgen
    World, _ = worlds
    n, n_peds, nr = 256, (0 if case == "no_pedestrians" else 12), 4
    res = 0.05 if case == "fine_grid_falls_back" else 0.125
    size = 400 if case == "fine_grid_falls_back" else 168
    grid = worldgen.make_grid(size, 2)
    params = worldgen.make_params(n, n_peds, res=res, view_cells=48 if res > 0.1 else 60)
    if case == "rectangles":
        _rect(params, n)
    if case == "mixed_classes":
        _mixed(params, n)
    layouts = [worldgen.make_layout(grid, res, n, n_peds, seed=41 + q, n_obstacles=2, clearance=0.6 if case == "mixed_classes" else 0.45) for q in range(2)]
    full = World(params, grid)
    bounds = [q * n // nr for q in range(nr + 1)]
    ranks = [World(dict(params, robot_begin=bounds[r], robot_end=bounds[r + 1]), grid) for r in range(nr)]
    try:
        want_bitmaps = case != "fine_grid_falls_back"
        for w in ranks:
            assert w.layer_mode()["shard_bitmaps"] == want_bitmaps, w.layer_mode()
        rng = np.random.default_rng(7)
        collided = 0
        for episode, steps in enumerate((25, 15)):
            full.reset(layouts[episode])
            for w in ranks:
                w.reset(layouts[episode])
            for s in range(-1, steps):
                if s >= 0:
                    a = random_actions(rng, n)
                    full.step(a)
                    for r, w in enumerate(ranks):
                        w.step_begin(a[bounds[r]:bounds[r + 1]])
                    _exchange_by_hand(ranks, bounds)
                    for w in ranks:
                        w.step_end()
                want = full.snapshot()
                for r, w in enumerate(ranks):
                    got = w.snapshot()
                    for k in SHARD_FIELDS:
                        if n_peds == 0 and k in ("ped_vector_states", "ped_maps", "ped_min_dists"):
                            continue
                        assert np.array_equal(got[k], want[k][bounds[r]:bounds[r + 1]]), (case, episode, s, r, k)
            collided += int((want["is_collisions"] == 3).sum())
        assert collided > 0, "no robot ever ran into another one: the case does not exercise the inter-robot layer"
    finally:
        full.close()
        for w in ranks:
            w.close()


def test_counting_layer_reports_a_cell_with_more_robots_than_its_count_field_holds(worlds):
    """The counting layer's word has room for at least 63 robots on one cell (255 on the headline shape: 3 + 8 pedestrian + 8 count
    + 13 index bits) and reset poses are the caller's: 300 robots placed on ONE spot must raise the device flag at the next call
    instead of carrying the count into the index sum silently; with the composed owner layers the same world runs."""
    import torch
    from img_env_amd import _cabi, worldgen
    World, _ = worlds
    n, n_peds = 8192, 200
    grid = worldgen.make_grid(400, 0)
    params = worldgen.make_params(n, n_peds, res=0.25, view_cells=48, beams=360, scene="rvoscene")
    layout = worldgen.make_layout(grid, 0.25, n, n_peds, seed=100, clearance=0.7)
    layout.robot_pose[:300] = layout.robot_pose[0]
    a = torch.zeros(n, 3, device="cuda")
    w = World(dict(params), grid)
    try:
        assert w.layer_mode()["layer"] == "counting"
        w.reset(layout)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="count field"):
            w.step(a)
    finally:
        w.close()
    w = World(dict(params, flags=int(params.get("flags", 0)) | _cabi.FLAG_COMPOSE_DENSE), grid)
    try:
        w.reset(layout)
        w.step(a)
        snap = w.snapshot()
        assert (snap["is_collisions"][:300] == 3).all()  # every one of them stands in another robot
    finally:
        w.close()


def test_in_library_rccl_exchange_matches_plain_step(worlds):
    """imgenv_comm_init + imgenv_step on one rank: the library's own ncclAllGather (RCCL, in place, on the step's stream) sits
    between the integrate and raster stages and must change nothing; RCCL itself reports the communicator's size and rank"""
    World, _ = worlds
    n = 24
    grid, params, layout = small_world(n, 10, seed=31, grid_size=320, clearance=0.8)
    plain, comm = World(params, grid), World(params, grid)
    try:
        comm.init_comm(0, 1)
        assert comm.comm_info() == (1, 0)
        plain.reset(layout)
        comm.reset(layout)
        rng = np.random.default_rng(6)
        for s in range(20):
            a = random_actions(rng, n)
            plain.step(a)
            comm.step(a)
            want, got = plain.snapshot(), comm.snapshot()
            for k in want:
                assert np.array_equal(got[k], want[k], equal_nan=True) if want[k].dtype.kind == "f" else np.array_equal(got[k], want[k]), (s, k)
    finally:
        plain.close()
        comm.close()


def _rect(params, n):
    from img_env_amd import _cabi
    params["robot_shape"] = np.full(n, _cabi.SHAPE_RECTANGLE, np.int32)
    params["robot_size"] = np.tile(np.array([-0.25, 0.15, -0.12, 0.12], np.float32), (n, 1))  # x_min, x_max, y_min, y_max
    params["robot_size_last"] = np.full(n, 0.12)


def _mixed(params, n):
    from img_env_amd import _cabi
    shape = np.full(n, _cabi.SHAPE_CIRCLE, np.int32)
    size = np.tile(np.array([0, 0, 0.17, 0], np.float32), (n, 1))
    size[1::3] = (0.02, -0.01, 0.24, 0)                     # a larger, off-centre disc
    shape[2::3] = _cabi.SHAPE_RECTANGLE
    size[2::3] = (-0.2, 0.2, -0.1, 0.1)
    sensor = np.zeros((n, 2), np.float32)
    sensor[::2] = (0.08, -0.03)                             # laser not at the base origin
    params.update(robot_shape=shape, robot_size=size, robot_sensor_cfg=sensor,
                  robot_size_last=np.where(shape == _cabi.SHAPE_CIRCLE, size[:, 2], 0.1).astype(np.float64))


def _limiter(params, n):
    params["limiter_v"] = dict(has_velocity_limits=True, has_acceleration_limits=True, has_jerk_limits=True,
                               min_velocity=-0.1, max_velocity=0.5, min_acceleration=-0.8, max_acceleration=0.6,
                               min_jerk=-2.0, max_jerk=2.0)
    params["limiter_w"] = dict(has_velocity_limits=True, has_acceleration_limits=True, min_velocity=-0.7, max_velocity=0.7,
                               min_acceleration=-1.5, max_acceleration=1.5)


VARIANTS = {
    "omni_with_lateral_speed": (dict(robot_ktype="omni"), None, True),
    "rectangle_robots": ({}, _rect, False),
    "mixed_classes_sensor_offset": ({}, _mixed, False),
    "speed_limiters": ({}, _limiter, False),
    "state4_raw_laser": (dict(state_dim=4, laser_norm=False), None, False),
    "narrow_fov_min_dist": (dict(view_angle_begin=-0.9, view_angle_end=0.6, view_min_dist=0.4, view_max_dist=2.5), None, False),
    "dt_040_nine_substeps": (dict(dt=0.4), None, False),
    "dt_100_many_substeps": (dict(dt=1.0), None, False),
    "dt_200_serial_integrate": (dict(dt=2.0), None, False),
}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_hip_matches_oracle_variants(worlds, name):
    """feature variants of the same path: kinematics, footprint classes, limiter, state layout, FOV gate, step length"""
    World, OracleWorld = worlds
    kw, mutate, lateral = VARIANTS[name]
    n = 12
    grid, params, layout = small_world(n, 5, seed=40 + len(name), n_obstacles=3, **kw)
    if mutate:
        mutate(params, n)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        rng = np.random.default_rng(len(name))
        acts = []
        for _ in range(30):
            a = random_actions(rng, n)
            if lateral:
                a[:, 2] = rng.uniform(-0.3, 0.3, n).astype(np.float32)
            acts.append(a)
        fails = run_pair(gpu, cpu, layout, acts)
        assert not fails, fails[:3]
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("ped_ca_p, ktype", [(1.0, "diff"), (0.6, "diff"), (0.35, "omni")])
def test_beep_lottery_and_ervo_evacuation_match_oracle(worlds, ped_ca_p, ktype):
    """the beep lottery (img_env.cpp:323-342: one glibc rand() per robot per step, `< ped_ca_p`, action beep > 0)
    and ERVO's evacuation term (ervo_ros Agent.cpp:63-69) through the C ABI.  The crowd must really have been pushed around:
    the same episode with ped_ca_p = 0 ends elsewhere."""
    World, OracleWorld = worlds
    n, P, steps = 40, 30, 25
    kw = dict(seed=91, grid_size=120, clearance=0.6, n_obstacles=2, scene="ervoscene", robot_ktype=ktype)
    grid, params, layout = small_world(n, P, beep_r=1.5, ped_ca_p=ped_ca_p, **kw)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    quiet = OracleWorld(dict(params, ped_ca_p=0.0), grid)
    try:
        rng = np.random.default_rng(17)
        acts = []
        for _ in range(steps):
            a = random_actions(rng, n)
            a[:, 2] = np.where(rng.random(n) < 0.5, rng.uniform(0.05, 0.3, n), rng.uniform(-0.3, 0.0, n)).astype(np.float32)
            acts.append(a)
        fails = run_pair(gpu, cpu, layout, acts)
        assert not fails, fails[:3]
        quiet.reset(layout)
        for a in acts:
            quiet.step(a)
        moved = np.abs(cpu.snapshot()["ped_state"] - quiet.snapshot()["ped_state"]).max()
        assert moved > 0.05, moved
        # a second episode on the same handles: the rand() stream carries on, it is not restarted by a reset
        fails = run_pair(gpu, cpu, layout, acts[:8])
        assert not fails, ("second episode", fails[:3])
    finally:
        gpu.close()
        cpu.close()
        quiet.close()


def test_beep_lottery_is_rejected_in_a_robot_shard(worlds):
    World, _ = worlds
    grid, params, layout = small_world(8, 4, seed=5, scene="ervoscene", beep_r=1.0, ped_ca_p=0.5)
    with pytest.raises(ValueError, match="beep"):
        World(dict(params, robot_begin=0, robot_end=4), grid)


def _dataset_world(n_robots, n_peds, steps, seed, ped_shape="circle"):
    """pedestrians replay a recorded random walk: rows (x, y, yaw, vx, vy) per step (reset_helper.py:417-432)"""
    from img_env_amd import spawn
    grid, params, layout = small_world(n_robots, n_peds, seed=seed, n_obstacles=2, scene="dataset", ped_shape=ped_shape)
    rng = np.random.default_rng(seed)
    T = steps - 7  # shorter than the episode: the last record is held (img_env.cpp:365-368)
    data = np.zeros((n_peds, T, 5))
    pos = layout.ped_pose[:, :2].copy()
    for t in range(T):
        v = rng.uniform(-0.5, 0.5, (n_peds, 2))
        if t % 5 == 0:
            v[0] = 0.0  # atan2(0, 0) and a standing pedestrian
        data[:, t, :2] = pos
        data[:, t, 2] = rng.uniform(-3, 3, n_peds)
        data[:, t, 3:] = v
        pos = pos + 0.25 * v
    spawn.init_ped_dataset(layout, data)
    return grid, params, layout


@pytest.mark.parametrize("ped_shape", ["circle", "leg"])
def test_dataset_pedestrians_match_oracle(worlds, ped_shape):
    """ped_sim type "dataset" (img_env.cpp:294-296, 361-386)"""
    World, OracleWorld = worlds
    n, steps = 10, 30
    grid, params, layout = _dataset_world(n, 7, steps, seed=77, ped_shape=ped_shape)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        rng = np.random.default_rng(3)
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(steps)])
        assert not fails, fails[:3]
        moved = np.abs(cpu.snapshot()["ped_state"][:, :2] - layout.ped_pose[:, :2]).max()
        assert moved > 0.5  # the crowd really followed the record
    finally:
        gpu.close()
        cpu.close()


def test_two_thousand_robots_match_oracle(worlds):
    """a crowd at the density of the benchmark world (0.25 m cells, 0.7 m clearance): every wavefront slot of a few
    dozen CUs is busy, robots see dozens of each other, collisions and arrivals happen"""
    World, OracleWorld = worlds
    n = 2048
    grid, params, layout = small_world(n, 60, seed=51, grid_size=200, res=0.25, clearance=0.7, n_obstacles=0)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        rng = np.random.default_rng(9)
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(6)])
        assert not fails, fails[:3]
        assert (cpu.snapshot()["is_collisions"] != 0).any()
    finally:
        gpu.close()
        cpu.close()


def test_headline_world_matches_oracle(worlds):
    """BASELINE cfg-3 at full size, the world `bench.py` times: 8192 robots and 200 ORCA pedestrians on the 400 x 400 map at
    0.25 m, every output of every robot against the oracle for a few steps (the oracle needs ~0.6 s per step)"""
    World, OracleWorld = worlds
    from img_env_amd import worldgen
    n, P = 8192, 200
    grid = worldgen.make_grid(400, 0)
    params = worldgen.make_params(n, P, res=0.25, view_cells=48, beams=360, scene="rvoscene", time_max=100)
    layout = worldgen.make_layout(grid, 0.25, n, P, seed=100, clearance=0.7)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        rng = np.random.default_rng(10)
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(5)])
        assert not fails, fails[:3]
        snap = cpu.snapshot()
        assert (snap["is_collisions"] != 0).sum() > 50 and snap["counters"][0] == 5
    finally:
        gpu.close()
        cpu.close()


def test_cfg5_world_matches_oracle(worlds):
    """BASELINE cfg-5 at full size (one GPU's worth of it): 8192 robots, 1000 ERVO pedestrians, 800 x 800 map at 0.125 m,
    96 x 96 views, 720 beams -- the 16-slot register sort of k_obs, 36 KB of LDS per view"""
    World, OracleWorld = worlds
    from img_env_amd import worldgen
    n, P = 8192, 1000
    grid = worldgen.make_grid(800, 0)
    params = worldgen.make_params(n, P, res=0.125, view_cells=96, beams=720, scene="ervoscene", time_max=100)
    layout = worldgen.make_layout(grid, 0.125, n, P, seed=100, clearance=0.7)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        rng = np.random.default_rng(11)
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(2)])
        assert not fails, fails[:3]
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("layer", ["default", "composed", "counting"])
def test_cfg2_world_matches_oracle(worlds, layer):
    """BASELINE cfg-2 at full size: 1024 robots, no pedestrians, 400 x 400 map at 0.125 m -- with the class layer the library
    picks for it (stamped: a launch-bound handle, csrc/imgenv_hip.hip) and with the composed one (`k_compose`)"""
    World, OracleWorld = worlds
    from img_env_amd import _cabi, worldgen
    n = 1024
    grid = worldgen.make_grid(400, 0)
    params = worldgen.make_params(n, 0, res=0.125, view_cells=48, beams=360, scene="", time_max=100)
    layout = worldgen.make_layout(grid, 0.125, n, 0, seed=100, clearance=1.0)
    gpu = World(dict(params, flags={"composed": _cabi.FLAG_COMPOSE_DENSE, "counting": _cabi.FLAG_LAYER_SUM}.get(layer, 0)), grid)
    cpu = OracleWorld(params, grid)
    try:
        rng = np.random.default_rng(12)
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(12)])
        assert not fails, fails[:3]
    finally:
        gpu.close()
        cpu.close()


def _cfg4_world(n, seed=100):
    """BASELINE cfg-4: n robots at 0.5 m cells on the 400 x 400 map beside a 200-agent social-force crowd.  The crowd ignores
    the robots (with relation_ped_robo = 1 the reference node itself recurses forever beyond 8 robots) and stays inside
    libpedsim's 10 m x 10 m quadtree root (pedscene.h:17-20), the only region the reference's PedScene can simulate."""
    from img_env_amd import worldgen
    P = 200
    grid = worldgen.make_grid(400, 0)
    params = worldgen.make_params(n, P, res=0.5, view_cells=48, beams=360, scene="pedscene", time_max=100, relation_ped_robo=0)
    layout = worldgen.make_layout(grid, 0.5, n, P, seed=seed, clearance=0.5)
    rng = np.random.default_rng(13)
    layout.ped_pose[:, :2] = rng.uniform(0.5, 9.5, (P, 2))
    layout.ped_traj[:, :, :2] = rng.uniform(0.5, 9.5, layout.ped_traj[:, :, :2].shape)
    layout.ped_goal[:] = rng.uniform(0.5, 9.5, (P, 2))
    return grid, params, layout, rng


@pytest.fixture
def cr_atan2_oracle():
    """oracle with every atan2 of the social-force model correctly rounded (libquadmath), like the device's cr_atan2"""
    from oracle_binding import set_cr_atan2
    set_cr_atan2(True)
    yield
    set_cr_atan2(False)


def test_cfg4_share_matches_oracle(worlds, cr_atan2_oracle):
    """one GPU's share of BASELINE cfg-4 (8192 robots, 200 social-force pedestrians, 0.5 m cells): EVERY field on EVERY one
    of 12 steps, usual bars, nothing excluded.

    Tagent::socialForce switches a full-size force term on sign(theta), theta being the difference of two atan2 of NEARLY
    PARALLEL vectors -- parallel up to rounding while the whole crowd stands still, i.e. on the first step of a handle's first
    episode (velocities persist across resets, pedscene.h:34-36).  The device's atan2 is correctly rounded (cr_atan2.h), so
    this test runs the oracle with its atan2 correctly rounded too (by an independent route: libquadmath's atan2q rounded
    once); what glibc's own atan2 does to that step is counted in test_cfg4_glibc_atan2_coin below."""
    World, OracleWorld = worlds
    n = 8192
    grid, params, layout, rng = _cfg4_world(n)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(12)])
        assert not fails, fails[:3]
        snap = cpu.snapshot()
        assert np.abs(snap["ped_state"][:, 2:]).max() > 0.05  # the crowd moved
        # ... and on through a reset between two steps: the crowd is stepped a step AHEAD here (it ignores the robots) and the
        # observation starts beside the move -- the reset drops what was computed ahead, velocities and tree persist (pedscene.h:34-46)
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(5)])
        assert not fails, ("behind the reset", fails[:3])
    finally:
        gpu.close()
        cpu.close()


def test_cfg4_glibc_atan2_coin(worlds):
    """the same first step against the oracle on the host libm's atan2 (what the reference would link): glibc 2.35 misrounds
    ~0.05 % of its inputs by one ulp, and each of the 40 000 pair terms of a crowd at rest then has a small chance to come out
    with the other sign.  Reported, not hidden: everything that does not carry a pedestrian velocity is still exact, nobody has
    moved yet, and only about one pedestrian in twelve carries a flipped term."""
    World, OracleWorld = worlds
    n = 1024
    grid, params, layout, rng = _cfg4_world(n)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        gpu.reset(layout)
        cpu.reset(layout)
        assert not compare(gpu.snapshot(), cpu.snapshot())
        a = random_actions(rng, n)
        gpu.step(a)
        cpu.step(a)
        g, c = gpu.snapshot(), cpu.snapshot()
        crowd_velocity = ("ped_state", "ped_vector_states", "ped_maps")  # the fields that carry pedestrian velocities
        bad = compare(g, c, tuple(f for f in EXACT + CLOSE if f not in crowd_velocity))
        assert not bad, bad
        assert np.abs(g["ped_state"][:, :2] - c["ped_state"][:, :2]).max() <= 1e-9  # nobody has moved yet (v was 0)
        dv = np.abs(g["ped_state"][:, 2:] - c["ped_state"][:, 2:]).max(axis=1)
        print("pedestrians whose first-step velocity differs under glibc atan2: %d of 200 (max %.3g m/s)" % ((dv > 1e-9).sum(), dv.max()))
        assert (dv > 1e-9).sum() <= 0.15 * 200 and dv.max() < 0.2, (int((dv > 1e-9).sum()), float(dv.max()))
    finally:
        gpu.close()
        cpu.close()


def test_cfg4_full_size_and_shard_match_oracle(worlds, cr_atan2_oracle):
    """BASELINE cfg-4 at its real size: ONE world of 65 536 robots and 200 social-force pedestrians on the 400 x 400 map at
    0.5 m.  The whole world in one handle, and rank 0's shard [0, 8192) of the 8-GPU layout in another (fed the other ranks'
    records from the whole-world handle, which is what the all-gather delivers), both against the oracle's 65 536 robots:
    every field, every step, usual bars.  (The oracle needs ~45 s for the reset and ~5 s per step at this size.)"""
    import torch
    World, OracleWorld = worlds
    n, nl = 65536, 8192
    grid, params, layout, rng = _cfg4_world(n)
    full, cpu = World(params, grid), OracleWorld(params, grid)
    shard = World(dict(params, robot_begin=0, robot_end=nl), grid)
    try:
        full.reset(layout)
        shard.reset(layout)
        cpu.reset(layout)
        want = cpu.snapshot()
        assert not compare(full.snapshot(), want)
        per_robot = [k for k in EXACT + CLOSE if k not in ("counters", "ped_state")]

        def cut(snap):
            return {k: (v[:nl] if k in per_robot else v) for k, v in snap.items()}
        assert not compare(shard.snapshot(), cut(want), tuple(per_robot) + ("ped_state",))
        for s in range(3):
            a = random_actions(rng, n)
            full.step(a)
            shard.step_begin(a[:nl])
            torch.cuda.synchronize()
            shard.records[nl:].copy_(full.records[nl:])  # the other seven ranks' records
            shard.step_end()
            cpu.step(a)
            want = cpu.snapshot()
            bad = compare(full.snapshot(), want)
            assert not bad, (s, bad)
            bad = compare(shard.snapshot(), cut(want), tuple(per_robot) + ("ped_state",))
            assert not bad, (s, "shard", bad)
        assert (want["is_collisions"] != 0).sum() > 1000 and want["counters"][0] == 3
    finally:
        full.close()
        shard.close()
        cpu.close()


def _sfm_tree(gpu, world=0):
    """the library's quadtree of one social-force crowd, digested as the oracle's sfm_tree() digests its own"""
    import ctypes as C
    out = np.zeros(8, np.uint64)  # (the last four: one bit per agent that is in the tree)
    gpu.lib.imgenv_debug_sfm_tree.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    gpu._check(gpu.lib.imgenv_debug_sfm_tree(gpu.h, world, out.ctypes.data), "imgenv_debug_sfm_tree")
    return tuple(int(v) for v in out)


def _quadtree_world(where, n=8):
    """crowds for the quadtree test: 200 m world at 0.5 m cells, the robots anywhere, the crowd (which ignores them) placed by `where`"""
    from img_env_amd import worldgen
    band = int(where[4:]) if where.startswith("band") else -1
    rng = np.random.default_rng(50 + band if band >= 0 else {"below_root": 6, "round_numbers": 7, "far_corner": 8}[where])
    P = int(rng.integers(100, 200)) if band >= 0 else {"below_root": 200, "round_numbers": 49, "far_corner": 66}[where]
    grid = worldgen.make_grid(400, 0)
    params = worldgen.make_params(n, P, res=0.5, view_cells=48, beams=360, scene="pedscene", time_max=1000, relation_ped_robo=0)
    layout = worldgen.make_layout(grid, 0.5, n, P, seed=31, clearance=0.5)
    shape = layout.ped_traj[:, :, :2].shape
    if band >= 0:                 # a dense band inside the tree's square (x in [0, 10], y in [10, 20]) marching to the far side: it keeps entering leaves that are still coarse
        c = rng.uniform(1.5, 8.5, 2)
        g = 10.0 - c
        layout.ped_pose[:, 0] = np.clip(c[0] + rng.uniform(-1.5, 1.5, P), 0.2, 9.8)
        layout.ped_pose[:, 1] = 10 + np.clip(c[1] + rng.uniform(-1.5, 1.5, P), 0.2, 9.8)
        layout.ped_traj[:, :, 0] = np.clip(g[0] + rng.uniform(-1.0, 1.0, shape[:2]), 0.2, 9.8)
        layout.ped_traj[:, :, 1] = 10 + np.clip(g[1] + rng.uniform(-1.0, 1.0, shape[:2]), 0.2, 9.8)
        layout.ped_goal[:] = layout.ped_traj[:, -1, :2]
    elif where == "below_root":   # cfg-4's crowd: y < 10, outside the tree's rectangle -- every agent has left its leaf on every step
        layout.ped_pose[:, :2] = rng.uniform(0.5, 9.5, (P, 2))
        layout.ped_traj[:, :, :2] = rng.uniform(3.0, 7.0, shape)
        layout.ped_goal[:] = rng.uniform(3.0, 7.0, (P, 2))
    elif where == "far_corner":   # 60 inside the square, 6 strolling 15-20 m beyond its far corner: their 40 m squares reach PART of the tree (the partial neighbour walk)
        far = np.arange(P) >= 60
        layout.ped_pose[:, 0] = np.where(far, rng.uniform(24.0, 29.5, P), rng.uniform(0.5, 9.5, P))
        layout.ped_pose[:, 1] = np.where(far, rng.uniform(33.0, 39.5, P), rng.uniform(10.5, 19.5, P))
        layout.ped_traj[:, :, 0] = np.where(far[:, None], rng.uniform(24.0, 29.5, shape[:2]), rng.uniform(0.5, 9.5, shape[:2]))
        layout.ped_traj[:, :, 1] = np.where(far[:, None], rng.uniform(33.0, 39.5, shape[:2]), rng.uniform(10.5, 19.5, shape[:2]))
        layout.ped_goal[:] = layout.ped_traj[:, -1, :2]
    else:                         # on the tree's own centre lines (5, 2.5, 1.25 ...): the reference inserts such an agent into SEVERAL children
        lat = np.arange(1, 8) * 1.25
        gx, gy = np.meshgrid(lat, 10.0 + lat)
        layout.ped_pose[:, 0], layout.ped_pose[:, 1] = gx.ravel(), gy.ravel()
        pick = rng.permutation(P)
        layout.ped_goal[:, 0], layout.ped_goal[:, 1] = gx.ravel()[pick], gy.ravel()[pick]
        layout.ped_traj[:, :, 0] = layout.ped_goal[:, None, 0]
        layout.ped_traj[:, :, 1] = layout.ped_goal[:, None, 1]
    return grid, params, layout, rng


@pytest.mark.parametrize("where", ["band0", "band1", "band2", "band3", "below_root", "round_numbers", "far_corner"])
def test_social_force_quadtree_matches_oracle_step_by_step(worlds, cr_atan2_oracle, where):
    """libpedsim's quadtree is behaviour, not an accelerator (sfm.h): who is in it decides who exerts forces, and Ttree::moveAgent's
    insert-from-the-root-then-erase (ped_tree.cpp:131-137) loses agents.  k_sfm does the moves that cannot split a leaf all at
    once and replays the rest in agent order; here the TREE ITSELF -- node count, every leaf's rectangle and members, every agent's
    treehash entry -- is held to the oracle's after every step (a digest that does not depend on node numbering), next to the
    outputs: dense bands of 100-200 agents marching through the tree's square (leaves split on every other step: the oracle alone,
    when the test was written, split on 8-13 of the first 25 steps and grew from ~120 to 160-290 nodes), cfg-4's crowd below the
    square, pedestrians that start exactly on the tree's centre lines (inserted into several children: the literal replay), and a
    crowd with a few members 15-20 m beyond the tree's corner, whose neighbour squares reach only part of it.
    The bands' outputs are compared over the first ten steps only: a packed crowd amplifies the last-bit differences between the
    device's and the host's exp / sqrt by a digit every few steps (1e-3 m after 40), which says nothing about the tree."""
    World, OracleWorld = worlds
    n = 8
    grid, params, layout, rng = _quadtree_world(where, n)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    dense = where.startswith("band")
    try:
        gpu.reset(layout)
        cpu.reset(layout)
        first = cpu.sfm_tree()
        assert _sfm_tree(gpu) == first
        steps, split_steps, worst = (25 if dense else 80), 0, 0.0
        nodes = first[0]
        for s in range(steps):
            if where in ("far_corner", "below_root") and s in (20, 21, 47):
                # a reset between two steps (twice in a row, too): positions and waypoints are set, velocities and the tree stay
                # (pedscene.h:34-46) -- and the state the library had computed a step ahead for this crowd is dropped
                gpu.reset(layout)
                cpu.reset(layout)
                assert _sfm_tree(gpu) == cpu.sfm_tree(), (where, s, "reset")
            a = random_actions(rng, n)
            gpu.step(a)
            cpu.step(a)
            want, got = cpu.sfm_tree(), _sfm_tree(gpu)
            if got != want:
                bits = lambda t: {64 * k + b for k in range(4) for b in range(64) if (t[4 + k] >> b) & 1}
                dp = np.abs(gpu.snapshot()["ped_state"] - cpu.snapshot()["ped_state"]).max()
                raise AssertionError((where, s, got[:4], want[:4], "only in the library's tree", sorted(bits(got) - bits(want)),
                                      "only in the oracle's", sorted(bits(want) - bits(got)), "pedestrian states differ by", float(dp)))
            split_steps += want[0] != nodes
            nodes = want[0]
            g, c = gpu.snapshot(), cpu.snapshot()
            worst = max(worst, float(np.abs(g["ped_state"] - c["ped_state"]).max()))
            if (s % 10 == 9 and not dense) or s < 3 or (dense and s in (5, 9)):
                bad = compare(g, c)
                assert not bad, (where, s, bad)
        last = cpu.sfm_tree()
        print("%s: quadtree %d -> %d nodes (splits on %d of %d steps), %d -> %d member entries; pedestrian states within %.2g of the oracle's throughout"
              % (where, first[0], last[0], split_steps, steps, first[1], last[1], worst))
        if dense:
            assert split_steps >= 6
    finally:
        gpu.close()
        cpu.close()


def test_quadtree_overflow_is_reported(worlds):
    """more than 8 social-force agents outside the 10 m x 10 m root square of libpedsim's quadtree: the reference recurses
    forever (ped_tree.cpp:65-96); the library raises the device flag and the next call fails loudly"""
    import torch
    World, _ = worlds
    grid, params, layout = small_world(4, 40, seed=61, grid_size=400, scene="pedscene", relation_ped_robo=0, clearance=0.8)
    w = World(params, grid)
    try:
        w.reset(layout)
        a = np.zeros((4, 3), np.float32)
        with pytest.raises(RuntimeError, match="quadtree overflowed"):
            for s in range(6):
                w.step(a)
                torch.cuda.synchronize()
    finally:
        w.close()


def test_ragged_trajectories_and_padded_ped_vector(worlds):
    """pedestrian waypoint lists of different lengths (1..4, img_env.cpp:306-319 cycles through them) and a ped vector
    padded to max_ped > n_peds (yaml_env.py:397-408)"""
    World, OracleWorld = worlds
    n, P = 8, 9
    grid, params, layout = small_world(n, P, seed=71, n_obstacles=2, max_ped=14)
    rng = np.random.default_rng(71)
    cap = 4
    traj = np.zeros((P, cap, 3))
    lens = rng.integers(1, cap + 1, P).astype(np.int32)
    for j in range(P):
        for q in range(lens[j]):
            traj[j, q, :2] = layout.ped_pose[j, :2] + rng.uniform(-2.0, 2.0, 2)
    layout.ped_traj, layout.ped_traj_len = traj, lens
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        acts = [random_actions(rng, n) for _ in range(60)]
        fails = run_pair(gpu, cpu, layout, acts)
        assert not fails, fails[:3]
        pv = cpu.snapshot()["ped_vector_states"]
        assert pv.shape[1] == 1 + 7 * 14 and (pv[:, 1 + 7 * P:] == 0).all() and (pv[:, 0] == P).all()
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("n_peds", [13, 40, 150])
def test_robots_inside_a_crowd_get_the_reference_ped_map(worlds, n_peds):
    """yaml_env.py:409-427 stamps the pedestrians inside the +-3 m box nearest first, so that farther discs overwrite nearer ones.
    With more than a dozen of them k_obs stamps all discs at once (every cell takes its highest-ranked disc: rank plane +
    atomicMax); a crowd packed into 3 m x 3 m around the robots -- every pedestrian in every robot's box, discs overlapping by the
    dozen -- must give the oracle's map bit for bit, every step, also right behind the threshold between the two ways (13) and
    with several rounds of 64 discs (150)."""
    from img_env_amd import worldgen
    World, OracleWorld = worlds
    n = 6
    grid = worldgen.make_grid(120, 5)
    params = worldgen.make_params(n, n_peds, res=0.125)
    layout = worldgen.make_layout(grid, 0.125, n, n_peds, seed=77, clearance=0.25)
    rng = np.random.default_rng(9)
    # everybody inside the same 3 m square: every robot has every pedestrian in its box
    layout.robot_pose[:, :2] = rng.uniform(6.0, 9.0, (n, 2))
    layout.ped_pose[:, :2] = rng.uniform(6.0, 9.0, (n_peds, 2))
    layout.ped_traj[:, 1, :2] = layout.ped_pose[:, :2]
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(10)])
        assert not fails, fails[:2]
        snap = cpu.snapshot()
        inside = (np.abs(snap["ped_vector_states"][:, 1:].reshape(n, -1, 7)[:, :n_peds, :2]) <= 3).all(axis=2).sum(axis=1)
        assert inside.max() > 12, inside  # the all-at-once path ran
        assert (snap["ped_maps"][:, 0] > 0).sum() > 100
    finally:
        gpu.close()
        cpu.close()


def test_equidistant_pedestrians_keep_index_order(worlds):
    """two (and three) pedestrians at exactly the same distance from a robot: the sort key ties, and the reference's stable
    Python sort keeps them in index order (yaml_env.py:451) -- the register sort's key-only fast path must notice and fall back"""
    from test_oracle_known_answers import _layout, _open_world
    from parity import compare
    World, OracleWorld = worlds
    grid, params = _open_world(n_robots=2, n_peds=5, scene="rvoscene")
    lay = _layout([(10.0, 10.0, 0.0), (14.0, 14.0, 1.5707963267948966)], [(16.0, 10.0), (14.0, 20.0)],
                  ped_xy=[(12.0, 11.0), (12.0, 9.0), (8.0, 11.0), (16.0, 15.0), (12.0, 13.0)])
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        gpu.reset(lay)
        cpu.reset(lay)
        a, b = gpu.snapshot(), cpu.snapshot()
        assert not compare(a, b)
        pv = b["ped_vector_states"][0]
        d = pv[1:].reshape(5, 7)[:, 6]
        assert d[0] == d[1] == d[2]                                    # a three-way tie in robot 0's list
        assert np.allclose(pv[1:3], (2.0, 1.0)) and np.allclose(pv[8:10], (2.0, -1.0))  # ... kept in index order
        for s in range(3):
            act = np.zeros((2, 3), np.float32)
            gpu.step(act)
            cpu.step(act)
            assert not compare(gpu.snapshot(), cpu.snapshot())
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("n_peds", [5, 70, 300, 700])
def test_pedestrian_distances_that_differ_below_float32_keep_the_float64_order(worlds, n_peds):
    """the register sort of k_obs orders float32(key) << 32 | index and repairs runs of equal surrogates with the exact comparator:
    pedestrians whose squared distances (float64, yaml_env.py:451) differ by less than a float32 ulp -- 1 + 2^-26 against 1, and a
    run of three -- must come out in float64 order, against their index order; plus exact ties inside the same run"""
    from test_oracle_known_answers import _layout, _open_world
    from parity import compare
    World, OracleWorld = worlds
    grid, params = _open_world(n_robots=2, n_peds=n_peds, scene="rvoscene")
    e = 2.0 ** -13
    # robot 0 at (10, 10) heading 0: its frame is the world shifted.  Squared distances: 1 + 4 e^2, 1 + e^2, 1, 1 (tie), then far ones
    ped_xy = [(11.0, 10.0 + 2 * e), (11.0, 10.0 - e), (11.0, 10.0), (9.0, 10.0), (13.0, 11.0)]
    rng = np.random.default_rng(5)
    while len(ped_xy) < n_peds:
        ped_xy.append((float(rng.uniform(3, 21)), float(rng.uniform(14, 21))))
    lay = _layout([(10.0, 10.0, 0.0), (14.0, 18.0, 1.5707963267948966)], [(16.0, 10.0), (14.0, 20.0)], ped_xy=ped_xy)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        gpu.reset(lay)
        cpu.reset(lay)
        a, b = gpu.snapshot(), cpu.snapshot()
        assert not compare(a, b)
        pv = b["ped_vector_states"][0][1:].reshape(n_peds, 7)
        assert np.allclose(pv[:4, :2], [(1.0, 0.0), (-1.0, 0.0), (1.0, -e), (1.0, 2 * e)], atol=1e-7)  # float64 order, ties by index
        assert np.array_equal(a["ped_vector_states"], b["ped_vector_states"])
        for s in range(2):
            act = np.zeros((2, 3), np.float32)
            gpu.step(act)
            cpu.step(act)
            assert not compare(gpu.snapshot(), cpu.snapshot())
    finally:
        gpu.close()
        cpu.close()


def _fuzz_case(seed):
    """a random but valid configuration of the same path: geometry, sensor, kinematics, crowd, footprints"""
    from img_env_amd import _cabi
    rng = np.random.default_rng(1000 + seed)
    res = float(rng.choice([0.125, 0.25, 0.1, 0.2]))
    view_cells = int(rng.choice([24, 32, 40, 48, 50, 64]))
    beams = int(rng.choice([90, 180, 360, 400, 719]))
    n = int(rng.integers(3, 40))
    scene = str(rng.choice(["rvoscene", "ervoscene", "rvoscene", ""]))
    P = int(rng.integers(1, 70)) if scene else 0
    kw = dict(res=res, view_cells=view_cells, beams=beams, scene=scene, dt=float(rng.choice([0.1, 0.25, 0.4])),
              state_dim=int(rng.choice([3, 4, 5])), relation_ped_robo=int(rng.integers(0, 2)),
              ped_shape=str(rng.choice(["circle", "leg"])), robot_ktype=str(rng.choice(["diff", "omni"])),
              time_max=int(rng.integers(8, 40)), use_laser=bool(rng.random() > 0.1), laser_norm=bool(rng.random() > 0.3),
              view_width=(view_cells + 0.5) * res, view_height=(view_cells + 0.5) * res)
    extent = max(30.0, 1.2 * np.sqrt((n + P) * 2.5))
    grid_size = int(np.ceil(extent / res / 4) * 4)
    grid, params, layout = small_world(n, P, seed=seed, grid_size=grid_size, n_obstacles=int(rng.integers(0, 5)),
                                       clearance=float(rng.choice([0.6, 0.8, 1.0])), **kw)
    if rng.random() < 0.5:  # mixed footprint classes
        shape = np.full(n, _cabi.SHAPE_CIRCLE, np.int32)
        size = np.tile(np.array([0, 0, 0.17, 0], np.float32), (n, 1))
        size[1::2] = (0.01, 0.02, float(rng.uniform(0.12, 0.3)), 0)
        if rng.random() < 0.5:
            shape[::3] = _cabi.SHAPE_RECTANGLE
            size[::3] = (-0.22, 0.18, -0.12, 0.1)
        params.update(robot_shape=shape, robot_size=size,
                      robot_size_last=np.where(shape == _cabi.SHAPE_CIRCLE, size[:, 2], 0.1).astype(np.float64))
    return grid, params, layout, n, bool(kw["robot_ktype"] == "omni")


@pytest.mark.parametrize("seed", range(int(os.environ.get("IMGENV_FUZZ_SEEDS", "16"))))  # more seeds for a bug hunt
def test_fuzzed_configurations_match_oracle(worlds, seed):
    World, OracleWorld = worlds
    grid, params, layout, n, lateral = _fuzz_case(seed)
    _, _, layout2 = small_world(n, params["n_peds"], seed=seed + 500, grid_size=grid.shape[0], res=params["view_resolution"],
                                n_obstacles=2, clearance=0.6)
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        rng = np.random.default_rng(seed)

        def acts(k):
            out = []
            for _ in range(k):
                a = random_actions(rng, n)
                if lateral:
                    a[:, 2] = rng.uniform(-0.3, 0.3, n).astype(np.float32)
                out.append(a)
            return out
        fails = run_pair(gpu, cpu, layout, acts(18))
        assert not fails, (seed, fails[:2])
        fails = run_pair(gpu, cpu, layout2, acts(10))  # a second episode on the same handles
        assert not fails, (seed, "second episode", fails[:2])
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("IMGENV_FUZZ_SEEDS_SFM", "8"))))
def test_fuzzed_social_force_rooms_match_oracle(worlds, seed):
    """room-sized social-force worlds (the envelope libpedsim's 10 m quadtree allows), random crowd / robot counts"""
    World, OracleWorld = worlds
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.integers(1, 8))
    P = int(rng.integers(2, 16))
    rel = int(rng.integers(0, 2))
    grid_size = int(rng.choice([88, 96, 128]))
    grid, params, layout = small_world(n, P, seed=200 + seed, scene="pedscene", grid_size=grid_size, relation_ped_robo=rel,
                                       n_obstacles=int(rng.integers(0, 4)), clearance=0.45,
                                       ped_shape=str(rng.choice(["circle", "leg"])))
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    try:
        fails = run_pair(gpu, cpu, layout, [random_actions(rng, n) for _ in range(40)])
        assert not fails, (seed, fails[:2])
    finally:
        gpu.close()
        cpu.close()


@pytest.mark.parametrize("n,grid_size", [(512, 200), (2048, 400), (8192, 400)])
def test_step_waits_for_actions_written_on_the_callers_stream(worlds, n, grid_size):
    """A trainer's policy writes the actions on the caller's stream right in front of `imgenv_step`, without synchronising.  Every
    kernel of the step that reads them -- the early-launched observation on its side stream included -- has to
    run behind that write: two handles, one fed actions that are still being produced (behind ~ms of matrix products on the
    same stream), one fed the finished values, stay bit-identical -- also when promised (`IMGENV_STEP_ACTIONS_READY`) and plain steps alternate."""
    import torch
    World, _ = worlds
    # (512 robots: the move inside the raster launch, k_obs behind it; 2048: early-observation steps on one side stream; 8192: on two)
    grid, params, layout = small_world(n, 40, seed=91, grid_size=grid_size, res=0.25, clearance=0.6, n_obstacles=2)
    params = dict(params, output_guard="none")  # (World's default guard synchronises during a handle's first 64 calls)
    a, b = World(params, grid), World(params, grid)
    try:
        a.reset(layout)
        b.reset(layout)
        rng = np.random.default_rng(17)
        dev = a.device
        big = torch.randn(2048, 2048, device=dev)
        late = torch.zeros(n, 3, device=dev)
        for s in range(12):
            want = torch.as_tensor(random_actions(rng, n), device=dev)
            torch.cuda.synchronize()
            b.step(want.clone(), actions_ready=(s % 4 == 1))  # (b: finished values, now and then with the promise -- the two kinds of
                                                               # early step hand over differently: an event behind the views / the gate)
            if s % 3 == 2:           # a: every third step with finished values and the promise ...
                late.copy_(want)
                torch.cuda.synchronize()
                a.step(late, actions_ready=True)
            else:                    # ... the others with actions that are still being produced
                y = big
                for _ in range(6):
                    y = (y @ big) * 1e-3   # keeps the stream busy: the actions below exist only when this is through
                late.copy_(want + 0.0 * y[:1, :3].nan_to_num(0.0, 0.0, 0.0))  # (depends on the products, changes nothing)
                a.step(late)             # no synchronisation in between
            ga, gb = a.snapshot(), b.snapshot()
            for k in EXACT + CLOSE:
                assert np.array_equal(ga[k], gb[k], equal_nan=True), (s, k)
    finally:
        a.close()
        b.close()
