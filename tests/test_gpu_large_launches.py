"""The kernel variants only LARGE launches select, against the oracle.

The library picks kernel variants by launch size (csrc/imgenv_hip.hip, launch_views): one wavefront per robot / pedestrian
instead of four in `k_raster` above 1024 blocks (robots and pedestrians in blocks of their own up to 8192 blocks, one block for a
robot AND a pedestrian beyond), eight / two / one wavefronts per `k_view` up to 1024 / up to 4096 / more robots (four where the
view's LDS bounds the occupancy: cfg-5), the step's move
inside the raster launch (`k_move_raster`) for pedestrian-free handles up to 4096 robots, 32 / 64 instead of 8 tiles per
`k_crop_big` wavefront from 48 / 1024 robots on, strided instead of one-to-one grids in the device-side reset chain.  The other GPU suites stay below those
thresholds (a dozen robots per oracle); here the handles are big enough to cross them, with one oracle per world as the checker:

* STAMP-mode many-world handles above 1024 robots (what `bench.py`'s `multi_world` / `vec_env` figures run);
* the reference's shipped ``envs/cfg/test.yaml`` geometry (733 x 733 cells of 0.015 m, 400 x 400-cell views shrunk to 48 x 48,
  1000 beams, one robot + 4 leg pedestrians + 4 obstacles per env: test.yaml:54,126-132) at 64 and 1024 envs (`shipped`);
* ``VecImageEnv(device_reset=True)`` at that geometry (the non-power-of-two reset kernels, the crop_map across resets;
  img_env.cpp:162-292), and a host reset with longer waypoint lists in between device-side steps.
"""
import copy
import json
import os

import numpy as np
import pytest

from parity import compare
from scenarios import random_actions
from test_gpu_multiworld import PER_ROBOT, _run, _stack_params, _world_slice

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def worlds():
    import torch
    assert torch.cuda.is_available()
    from img_env_amd.world import World
    from oracle_binding import OracleWorld, build_oracle
    build_oracle()
    return World, OracleWorld


@pytest.mark.parametrize("res,legs", [(0.125, False), (0.1, True)], ids=["pow2_cells", "tenth_cells_legs"])
def test_stamped_worlds_above_1024_robots_match_one_oracle_each(worlds, res, legs):
    """300 worlds x (4 robots + 3 pedestrians) = 1200 robots, STAMP mode: the step's rasters are the one-wavefront variant with
    robots and pedestrians in blocks of their own (`k_raster<.., true, 1>`, split), its views the two-wavefront one
    (`k_view<.., true, 2>`), the resets of a few worlds in mid-flight the four-wavefront ones, the reset of 250 worlds at once
    (1000 robots + ...) sits just below the threshold, the one of all 300 above it"""
    World, OracleWorld = worlds
    W = 300
    resets = {1: [7, 150, 299], 3: list(range(0, W, 7)), 4: [1], 5: list(range(250)), 7: list(range(W))}
    kw = dict(ped_shape="leg") if legs else {}
    fails, snap, _ = _run(World, OracleWorld, W, 4, 3, 10, resets, seed=81, n_obstacles=2, grid_size=120, res=res, flags=4,
                          time_max=6, clearance=0.8, view_cells=48, **kw)
    assert not fails, fails[:3]
    assert snap["is_collisions"].shape[0] == 1200


def test_stamped_worlds_above_4096_robots_match_one_oracle_each(worlds):
    """1400 worlds x (4 robots + 3 pedestrians) = 5600 robots + 4200 pedestrians, STAMP mode: beyond 4096 robots the views are the
    one-wavefront variant (`k_view<.., true, 1>`), beyond 8192 raster blocks a block draws a robot AND a pedestrian; the reset of
    900 worlds at once (3600 robots) takes the two-wavefront views and the split rasters"""
    World, OracleWorld = worlds
    W = 1400
    resets = {1: [3, 700, 1399], 3: list(range(900)), 4: list(range(0, W, 5))}
    fails, snap, _ = _run(World, OracleWorld, W, 4, 3, 6, resets, seed=87, n_obstacles=2, grid_size=120, res=0.125, flags=4,
                          time_max=4, clearance=0.8, view_cells=48)
    assert not fails, fails[:3]
    assert snap["is_collisions"].shape[0] == 5600


def test_stamped_worlds_without_pedestrians_above_1024_robots(worlds):
    """1100 one-robot worlds (the `8192 x 1` row of the many-worlds table in small): no side streams, the step's move and
    get_state inside the raster launch (`k_move_raster<.., true, 1>`), `k_state` after a reset, the views are `k_view<.., true, 2>`"""
    World, OracleWorld = worlds
    W = 1100
    fails, _, _ = _run(World, OracleWorld, W, 1, 0, 7, {2: [0, 500, 1099], 4: list(range(0, W, 3))}, seed=83, n_obstacles=2,
                       grid_size=100, flags=4, time_max=5, scene="")
    assert not fails, fails[:3]


# ---------------------------------------------------------------------------------------------- the shipped geometry
def _shipped(n_envs, seed):
    """parameters, map and one layout per env of the reference's test.yaml cast (1 robot, 4 leg pedestrians, 4 obstacles)"""
    from img_env_amd import worldgen
    from test_gpu_parity import _room_map
    src = _room_map(110, seed)
    params = worldgen.make_params(1, 4, res=0.015, view_cells=1, beams=1000, ped_shape="leg", dt=0.4)
    params.update(global_resolution=0.1, view_width=6.0, view_height=6.0, image_size=(48, 48))
    layouts = [worldgen.make_layout(src, 0.1, 1, 4, seed=seed + 13 * k, n_obstacles=4) for k in range(n_envs)]
    return src, params, layouts


def _fields(view_maps):
    return tuple(f for f in PER_ROBOT if view_maps or f != "view_maps")


@pytest.mark.parametrize("view_maps", [False, True], ids=["no_view_maps", "full_view"])
def test_shipped_geometry_64_envs_with_resets_matches_one_oracle_each(worlds, view_maps):
    """64 envs of the shipped geometry in one handle: `k_crop_big` with 32 tiles per wavefront (>= 48 robots), resets of 3 worlds
    (8 tiles per wavefront) and of 60 worlds (32) in mid-flight; with and without the full-size view as an output"""
    World, OracleWorld = worlds
    from img_env_amd import _cabi, worldgen
    E = 64
    src, params, layouts = _shipped(E, 61)
    flags = int(params.get("flags", 0)) | (0 if view_maps else _cabi.FLAG_NO_VIEW_MAPS)
    gpu = World(_stack_params(dict(params, flags=flags), E), src)
    cpus = [OracleWorld(params, src) for _ in range(E)]
    fields = _fields(view_maps)
    fails = []

    def check(where):
        snap = gpu.snapshot()
        for k, c in enumerate(cpus):
            bad = compare(_world_slice(snap, k, 1, 4), c.snapshot(), fields + ("ped_state",))
            if bad:
                fails.append((where, k, bad))

    try:
        gpu.reset(layouts)
        for c, lay in zip(cpus, layouts):
            c.reset(lay)
        check("reset")
        rng = np.random.default_rng(9)
        for s in range(8):
            a = random_actions(rng, E)
            gpu.step(a)
            for k, c in enumerate(cpus):
                c.step(a[k:k + 1])
            check(s)
            ks = {2: [5, 17, 63], 4: list(range(2, 62))}.get(s, [])
            if ks:
                lays = [worldgen.make_layout(src, 0.1, 1, 4, seed=1000 * s + k, n_obstacles=4) for k in ks]
                gpu.reset_worlds(ks, lays)
                for k, lay in zip(ks, lays):
                    cpus[k].reset(lay)
                check((s, "reset"))
            assert len(fails) < 4, fails[:3]
        assert not fails, fails[:3]
    finally:
        gpu.close()
        for c in cpus:
            c.close()


def test_shipped_geometry_1024_envs_matches_one_oracle_each(worlds):
    """1024 envs of the shipped geometry, as `bench.py`'s `shipped` figure runs them (no full-size view): `k_crop_big` with 64 tiles
    per wavefront, `k_raster<false, true, 1>` (4096 leg pedestrians), `k_beams_big` / `k_taps_big` over 1024 robots.  The device runs
    first and its outputs are recorded; then one oracle at a time replays its env (1024 live oracles would hold 2 GB of maps)."""
    World, OracleWorld = worlds
    from img_env_amd import _cabi
    E, steps = 1024, 3
    src, params, layouts = _shipped(E, 71)
    gpu = World(_stack_params(dict(params, flags=int(params.get("flags", 0)) | _cabi.FLAG_NO_VIEW_MAPS), E), src)
    fields = _fields(False)
    keep = fields + ("ped_state",)
    try:
        rng = np.random.default_rng(10)
        acts = [random_actions(rng, E) for _ in range(steps)]
        gpu.reset(layouts)
        rec = [{f: v for f, v in gpu.snapshot().items() if f in keep}]
        for a in acts:
            gpu.step(a)
            rec.append({f: v for f, v in gpu.snapshot().items() if f in keep})
    finally:
        gpu.close()
    fails = []
    moved = 0.0
    for k in range(E):
        c = OracleWorld(params, src)
        try:
            c.reset(layouts[k])
            for s in range(steps + 1):
                if s:
                    c.step(acts[s - 1][k:k + 1])
                mine = {f: rec[s][f][k:k + 1] for f in fields}
                mine["ped_state"] = rec[s]["ped_state"][4 * k:4 * k + 4]
                bad = compare(mine, c.snapshot(), keep)
                if bad:
                    fails.append((s - 1, k, bad))
            moved = max(moved, float(np.abs(c.snapshot()["ped_state"][:, 2:]).max()))
        finally:
            c.close()
        assert len(fails) < 4, fails[:3]
    assert not fails, fails[:3]
    assert moved > 0.05  # the crowds walked


def _shipped_vec_cfg(tmp_path, time_max):
    from PIL import Image
    from img_env_amd import worldgen
    m = np.full((110, 110), 255, np.uint8)
    m[:5] = m[-5:] = 0
    m[:, :5] = m[:, -5:] = 0
    Image.fromarray(m).save(str(tmp_path / "room.png"))
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "spawn_ref.npz"))
    cfg = worldgen.shipped_test_yaml_cfg("room.png", json.loads(str(z["test@1/cfg"])))
    cfg.update(map_dir=str(tmp_path), seed=3, time_max=time_max)
    return cfg


@pytest.mark.parametrize("E,keep_view_maps,steps", [(24, False, 14), (70, False, 14), (9, True, 14), (1100, False, 7)])
def test_device_side_reset_at_the_shipped_geometry_matches_oracles_fed_the_same_placements(tmp_path, E, keep_view_maps, steps):
    """``VecImageEnv(device_reset=True)`` on the shipped test.yaml cast and geometry with a short time limit: the finished envs are
    placed and reset by kernels alone -- `k_restore_maps_dev<false>` / `k_reset_obstacles<false>` (cells of 0.015 m are no power of
    two), the crop_map kept up to date across resets, the big-view chain over the listed worlds.  One oracle per env, fed the
    placement each world really received (imgenv_world_placement), checks every field after every step; a host-side reset of two
    envs in between.  1100 envs: ~270 worlds finish per step, so the reset chain itself runs the large-launch variants (one
    wavefront per pedestrian in its raster, 64 tiles per crop wavefront), as in `bench.py`'s `shipped` figure at 2048 envs"""
    import torch
    from img_env_amd import spawn
    from img_env_amd.vec_env import VecImageEnv
    from oracle_binding import OracleWorld, build_oracle
    build_oracle()
    cfg = _shipped_vec_cfg(tmp_path, time_max=4)
    cfg["keep_view_maps"] = keep_view_maps
    n_obs = int(cfg["object"]["total"])
    R, P = 1, int(cfg["ped_sim"]["total"])
    vec = VecImageEnv(copy.deepcopy(cfg), env_num=E, seed=3, device_reset=True)
    cpus = [OracleWorld(vec.params, vec.grid) for _ in range(E)]
    fields = tuple(f for f in ("is_collisions", "is_arrives", "view_maps", "sensor_maps", "vector_states", "lasers", "ped_maps",
                               "ped_vector_states", "rewards", "dones", "dones_info", "robot_pose") if keep_view_maps or f != "view_maps")

    def check(where):
        snap = vec.world.snapshot()
        for k, c in enumerate(cpus):
            mine = {f: snap[f][k * R:(k + 1) * R] for f in fields}
            bad = compare(mine, c.snapshot(), fields)
            assert not bad, (where, k, bad)

    try:
        seed0 = vec._spawn_seed
        vec.reset()
        for k in range(E):
            cpus[k].reset(spawn.native_spawn(cfg, seed0 + k))
        check("reset")
        rng = np.random.default_rng(2)
        resets, expect_serial = 0, 0
        for s in range(steps):
            a = np.zeros((E * R, 3), np.float32)
            a[:, 0], a[:, 1] = rng.uniform(0, 0.6, E * R), rng.uniform(-0.9, 0.9, E * R)
            _, rew, done, info = vec.step(torch.as_tensor(a, device="cuda"))
            worlds, first = vec.world.autoreset_last()
            rew, done = rew.cpu().numpy(), done.cpu().numpy()
            assert first == expect_serial, s
            for k, c in enumerate(cpus):
                c.step(a[k * R:(k + 1) * R])
                ref = c.snapshot()
                assert np.array_equal(rew[k * R:(k + 1) * R], ref["rewards"]), (s, k)
                assert np.array_equal(done[k * R:(k + 1) * R], ref["dones"]), (s, k)
            for q, k in enumerate(worlds):
                lay, serial = vec.world.world_placement(k, n_obs)
                assert serial == first + q, (s, k)
                lay.ignore_obstacle = bool(cfg["ped_sim"].get("ignore_obstacle", False))
                cpus[k].reset(lay)
            expect_serial += len(worlds)
            resets += len(worlds)
            check(s)
            if s == 5:  # two envs start over through the host (their obstacles are news to the device-side map restore)
                ep = vec._episodes
                vec.reset_envs([1, E - 1])
                cpus[1].reset(spawn.native_spawn(cfg, seed0 + ep))
                cpus[E - 1].reset(spawn.native_spawn(cfg, seed0 + ep + 1))
                check("host reset")
        assert resets >= (2 * E if steps >= 14 else E)
    finally:
        vec.close()
        for c in cpus:
            c.close()


def test_host_reset_with_longer_waypoint_lists_between_device_side_steps():
    """a host-side reset that brings longer pedestrian waypoint lists than any before re-lays the trajectory table out
    (stage_world); the device-side reset's copy of that table's address and stride must follow, or the worlds it resets afterwards
    get their pedestrians' paths written into the old table with the old stride"""
    import torch
    from img_env_amd import spawn, worldgen
    from img_env_amd.vec_env import VecImageEnv
    from oracle_binding import OracleWorld, build_oracle
    build_oracle()
    E, R, P, n_obs = 6, 2, 3, 2
    grid = worldgen.make_grid(200, 3)
    cfg = worldgen.make_yaml_cfg(R, P, grid, time_max=4, n_obstacles=n_obs, seed=9)
    vec = VecImageEnv(copy.deepcopy(cfg), env_num=E, seed=9, device_reset=True)
    cpus = [OracleWorld(vec.params, vec.grid) for _ in range(E)]
    fields = ("is_collisions", "is_arrives", "view_maps", "vector_states", "lasers", "ped_maps", "ped_vector_states", "rewards",
              "dones", "robot_pose", "ped_state")

    def check(where):
        snap = vec.world.snapshot()
        for k, c in enumerate(cpus):
            mine = {f: snap[f][k * R:(k + 1) * R] for f in fields if f != "ped_state"}
            mine["ped_state"] = snap["ped_state"][k * P:(k + 1) * P]
            bad = compare(mine, c.snapshot(), fields)
            assert not bad, (where, k, bad)

    try:
        vec.reset()
        for k in range(E):
            cpus[k].reset(spawn.native_spawn(cfg, vec._spawn_seed + k))
        rng = np.random.default_rng(4)
        n_reset = 0
        for s in range(16):
            a = np.zeros((E * R, 3), np.float32)
            a[:, 0], a[:, 1] = rng.uniform(0, 0.6, E * R), rng.uniform(-0.9, 0.9, E * R)
            vec.step(torch.as_tensor(a, device="cuda"))
            worlds, _ = vec.world.autoreset_last()
            for k, c in enumerate(cpus):
                c.step(a[k * R:(k + 1) * R])
            for k in worlds:
                lay, _ = vec.world.world_placement(k, n_obs)
                cpus[k].reset(lay)
            n_reset += len(worlds)
            check(s)
            if s == 5:  # env 2 by the host, five waypoints per pedestrian where every reset so far had two
                lay = worldgen.make_layout(grid, 0.125, R, P, seed=77, n_obstacles=n_obs)
                traj = np.zeros((P, 5, 3))
                for j in range(P):
                    for q in range(5):
                        traj[j, q, :2] = lay.ped_pose[j, :2] + rng.uniform(-1.5, 1.5, 2)
                lay.ped_traj, lay.ped_traj_len = traj, np.full(P, 5, np.int32)
                vec.reset_envs([2], layouts=[lay])
                cpus[2].reset(lay)
                check("host reset, longer paths")
        assert n_reset >= 2 * E
    finally:
        vec.close()
        for c in cpus:
            c.close()


def _device_reset_against_oracles(cfg, E, steps, fields, seed=9, min_resets=None):
    """``VecImageEnv(device_reset=True)`` from ``cfg`` with one oracle per env fed the placements the device reports; every field
    of ``fields`` after every step, the step's own rewards / dones too"""
    import torch
    from img_env_amd import spawn
    from img_env_amd.vec_env import VecImageEnv
    from oracle_binding import OracleWorld, build_oracle
    build_oracle()
    R, P, n_obs = int(cfg["robot"]["total"]), int(cfg["ped_sim"]["total"]), int(cfg["object"]["total"])
    vec = VecImageEnv(copy.deepcopy(cfg), env_num=E, seed=seed, device_reset=True)
    cpus = [OracleWorld(vec.params, vec.grid) for _ in range(E)]

    def check(where):
        snap = vec.world.snapshot()
        for k, c in enumerate(cpus):
            mine = {f: (snap[f][k * P:(k + 1) * P] if f == "ped_state" else snap[f][k * R:(k + 1) * R]) for f in fields}
            bad = compare(mine, c.snapshot(), fields)
            assert not bad, (where, k, bad)

    try:
        vec.reset()
        for k in range(E):
            cpus[k].reset(spawn.native_spawn(cfg, vec._spawn_seed + k))
        check("reset")
        rng = np.random.default_rng(2)
        resets, expect_serial = 0, 0
        for s in range(steps):
            a = np.zeros((E * R, 3), np.float32)
            a[:, 0], a[:, 1] = rng.uniform(0, 0.6, E * R), rng.uniform(-0.9, 0.9, E * R)
            _, rew, done, info = vec.step(torch.as_tensor(a, device="cuda"))
            worlds, first = vec.world.autoreset_last()
            rew, done = rew.cpu().numpy(), done.cpu().numpy()
            assert first == expect_serial, s
            for k, c in enumerate(cpus):
                c.step(a[k * R:(k + 1) * R])
                ref = c.snapshot()
                assert np.array_equal(rew[k * R:(k + 1) * R], ref["rewards"]), (s, k)
                assert np.array_equal(done[k * R:(k + 1) * R], ref["dones"]), (s, k)
            for q, k in enumerate(worlds):
                lay, serial = vec.world.world_placement(k, n_obs)
                assert serial == first + q, (s, k)
                lay.ignore_obstacle = bool(cfg["ped_sim"].get("ignore_obstacle", False))
                cpus[k].reset(lay)
            expect_serial += len(worlds)
            resets += len(worlds)
            check(s)
        assert resets >= (2 * E if min_resets is None else min_resets), resets
    finally:
        vec.close()
        for c in cpus:
            c.close()


DEVICE_RESET_FIELDS = ("is_collisions", "is_arrives", "view_maps", "sensor_maps", "vector_states", "lasers", "ped_maps", "ped_vector_states",
                       "rewards", "dones", "dones_info", "robot_pose", "ped_state")


def test_device_side_reset_of_worlds_with_more_than_64_agents():
    """worlds of 20 robots + 60 pedestrians: the device-side sampler's distance tests take the agents placed so far 64 at a time
    (csrc/spawn_device.h: up to 256 agents per world); reset_helper.py:35-55, 189-345"""
    from img_env_amd import worldgen
    grid = worldgen.make_grid(200, 3)
    cfg = worldgen.make_yaml_cfg(20, 60, grid, time_max=3, n_obstacles=3, seed=9)
    _device_reset_against_oracles(cfg, 3, 10, DEVICE_RESET_FIELDS, min_resets=6)


def test_device_side_reset_of_pedscene_worlds():
    """social-force crowds (pedscene.h:17-91) reset by the device: a finished world's pedestrians get their new positions
    (velocities persist), their waypoint deques [goal, trajectory], the crowd its new obstacle segments; robots are crowd members
    (relation_ped_robo = 1).  One oracle per env fed the placements the device drew."""
    from img_env_amd import worldgen
    from oracle_binding import set_cr_atan2
    grid = worldgen.make_grid(88, 3)
    cfg = worldgen.make_yaml_cfg(2, 5, grid, time_max=4, n_obstacles=2, seed=9, scene="pedscene")
    set_cr_atan2(True)  # the oracle's atan2 correctly rounded, like the device's (tests/test_gpu_parity.py::cr_atan2_oracle)
    try:
        _device_reset_against_oracles(cfg, 5, 16, DEVICE_RESET_FIELDS)
    finally:
        set_cr_atan2(False)
