"""Several independent worlds in ONE handle (imgenv_cfg.n_worlds): the reference's ``env_num`` env processes, batched.

Each world of the handle must behave exactly like a world of its own: the checker is one oracle per world, reset and
stepped with that world's slice of the batch.  Covers per-world reset in mid-flight (the other worlds keep their state
and their time limits), the whole-handle reset with a shared obstacle list, a dataset crowd (its clock is per world) and
world sizes that do not divide a workgroup."""
import os

import numpy as np
import pytest

from parity import CLOSE, EXACT, compare
from scenarios import random_actions, small_world

pytestmark = pytest.mark.gpu

PER_ROBOT = tuple(f for f in EXACT + CLOSE if f not in ("counters", "ped_state"))


@pytest.fixture(scope="module")
def worlds():
    import torch
    assert torch.cuda.is_available()
    from img_env_amd.world import World
    from oracle_binding import OracleWorld, build_oracle
    build_oracle()
    return World, OracleWorld


def _stack_params(params, W):
    """the parameter dict of W copies of one world: per-robot / per-pedestrian rows repeat, world-major"""
    p = dict(params)
    for k in ("robot_shape", "robot_size", "robot_sensor_cfg", "robot_size_last", "ped_shape", "ped_size", "ped_max_speed"):
        a = np.asarray(p[k])
        p[k] = np.concatenate([a] * W, axis=0)
    p["n_robots"], p["n_peds"], p["n_worlds"] = params["n_robots"] * W, params["n_peds"] * W, W
    return p


def _world_slice(snap, k, Rw, Pw):
    out = {f: snap[f][k * Rw:(k + 1) * Rw] for f in PER_ROBOT}
    out["ped_state"] = snap["ped_state"][k * Pw:(k + 1) * Pw]
    return out


def _compare_all(gpu, cpus, Rw, Pw, where, fails):
    snap = gpu.snapshot()
    for k, cpu in enumerate(cpus):
        bad = compare(_world_slice(snap, k, Rw, Pw), cpu.snapshot(), PER_ROBOT + (("ped_state",) if Pw else ()))
        if bad:
            fails.append((where, k, bad))


def _run(World, OracleWorld, W, Rw, Pw, steps, resets, seed, whole_reset=False, beep=False, **kw):
    """``resets``: {step: [worlds reset after that step]}; ``beep``: half of the actions carry a positive beep (v_y)"""
    grid, params, lay0 = small_world(Rw, Pw, seed=seed, **kw)
    layouts = [lay0] + [small_world(Rw, Pw, seed=seed + 100 * k, **kw)[2] for k in range(1, W)]
    gpu = World(_stack_params(params, W), grid)
    cpus = [OracleWorld(params, grid) for _ in range(W)]
    fails = []
    try:
        if whole_reset:  # one batch of every robot / pedestrian, world 0's obstacles everywhere
            for lay in layouts[1:]:
                lay.obs_shape, lay.obs_size, lay.obs_pose = lay0.obs_shape, lay0.obs_size, lay0.obs_pose
            b = [lay.as_batch() for lay in layouts]
            big = dict(b[0])
            for f in ("robot_pose", "robot_goal", "ped_pose", "ped_goal", "ped_traj", "ped_traj_len"):
                big[f] = np.concatenate([x[f] for x in b], axis=0)
            gpu.reset(big)
        else:
            gpu.reset(layouts)
        for cpu, lay in zip(cpus, layouts):
            cpu.reset(lay)
        _compare_all(gpu, cpus, Rw, Pw, -1, fails)
        rng = np.random.default_rng(seed + 7)
        n_reset = 0
        for s in range(steps):
            a = random_actions(rng, W * Rw)
            if beep:
                a[:, 2] = np.where(rng.random(W * Rw) < 0.5, 0.2, 0.0).astype(np.float32)
            gpu.step(a)
            for k, cpu in enumerate(cpus):
                cpu.step(a[k * Rw:(k + 1) * Rw])
            _compare_all(gpu, cpus, Rw, Pw, s, fails)
            ks = list(resets.get(s, ()))
            if ks:  # several worlds end their episode on the same step: one imgenv_reset_worlds call
                lays = []
                for k in ks:
                    n_reset += 1
                    lays.append(small_world(Rw, Pw, seed=seed + 1000 * n_reset + k, **kw)[2])
                    cpus[k].reset(lays[-1])
                if len(ks) == 1:
                    gpu.reset_world(ks[0], lays[0])
                else:
                    gpu.reset_worlds(ks, lays)
                _compare_all(gpu, cpus, Rw, Pw, (s, "reset", tuple(ks)), fails)
            if len(fails) > 4:
                break
        return fails, gpu.snapshot(), [c.snapshot() for c in cpus]
    finally:
        gpu.close()
        for c in cpus:
            c.close()


def test_worlds_match_one_oracle_each(worlds):
    """3 worlds x (5 robots, 6 pedestrians, own obstacles): ORCA crowds that must not see each other, per-world resets at
    steps 6 and 9 and per-world time limits (time_max 12: worlds 0 / 2 run out at step 12, world 1 twelve steps after its
    own reset)"""
    World, OracleWorld = worlds
    fails, snap, cs = _run(World, OracleWorld, 3, 5, 6, 24, {6: [1], 9: [1], 14: [0, 2]}, seed=31, n_obstacles=3, time_max=12)
    assert not fails, fails[:3]


def test_whole_handle_reset_shares_one_obstacle_list(worlds):
    World, OracleWorld = worlds
    fails, _, _ = _run(World, OracleWorld, 4, 3, 4, 10, {4: [2]}, seed=32, whole_reset=True, n_obstacles=4)
    assert not fails, fails[:3]


def test_many_small_worlds_without_pedestrians(worlds):
    """40 worlds x 7 robots: a workgroup of the per-robot kernels straddles several worlds"""
    World, OracleWorld = worlds
    fails, _, _ = _run(World, OracleWorld, 40, 7, 0, 8, {3: [39, 0, 17], 4: [5], 5: list(range(40))}, seed=33, n_obstacles=2, grid_size=120)
    assert not fails, fails[:3]


def test_worlds_with_legs_ervo_and_odd_sizes(worlds):
    World, OracleWorld = worlds
    fails, _, _ = _run(World, OracleWorld, 5, 9, 11, 12, {5: [4], 6: [0]}, seed=34, n_obstacles=2, ped_shape="leg",
                       scene="ervoscene", res=0.1, view_cells=37, view_width=3.75, view_height=3.75, grid_size=150)
    assert not fails, fails[:3]


def test_worlds_with_their_own_beep_lotteries(worlds):
    """every world of a handle is one node process of the reference: its own rand() stream for the beep lottery
    (img_env.cpp:327), which a reset of that world does not restart"""
    World, OracleWorld = worlds
    fails, _, _ = _run(World, OracleWorld, 4, 10, 12, 16, {5: [3], 8: [0, 2]}, seed=35, n_obstacles=2, scene="ervoscene",
                       grid_size=80, clearance=0.6, beep=True, beep_r=1.5, ped_ca_p=0.7)
    assert not fails, fails[:3]


def test_dataset_clock_is_per_world(worlds):
    """a replayed crowd restarts its record when ITS world is reset (img_env.cpp:361-386 indexes by the env's step count)"""
    World, OracleWorld = worlds
    from img_env_amd import spawn
    W, Rw, Pw, steps = 3, 4, 5, 16
    grid, params, _ = small_world(Rw, Pw, seed=41, scene="dataset")

    def layout(seed):
        lay = small_world(Rw, Pw, seed=seed, n_obstacles=2, scene="dataset")[2]
        rng = np.random.default_rng(seed)
        T = 9
        data = np.zeros((Pw, T, 5))
        pos = lay.ped_pose[:, :2].copy()
        for t in range(T):
            v = rng.uniform(-0.5, 0.5, (Pw, 2))
            data[:, t, :2], data[:, t, 2], data[:, t, 3:] = pos, rng.uniform(-3, 3, Pw), v
            pos = pos + 0.25 * v
        spawn.init_ped_dataset(lay, data)
        return lay

    gpu = World(_stack_params(params, W), grid)
    cpus = [OracleWorld(params, grid) for _ in range(W)]
    fails = []
    try:
        lays = [layout(50 + k) for k in range(W)]
        gpu.reset(lays)
        for c, lay in zip(cpus, lays):
            c.reset(lay)
        rng = np.random.default_rng(5)
        for s in range(steps):
            a = random_actions(rng, W * Rw)
            gpu.step(a)
            for k, c in enumerate(cpus):
                c.step(a[k * Rw:(k + 1) * Rw])
            _compare_all(gpu, cpus, Rw, Pw, s, fails)
            if s == 5:
                lay = layout(99)
                gpu.reset_world(1, lay)
                cpus[1].reset(lay)
                _compare_all(gpu, cpus, Rw, Pw, (s, "reset"), fails)
        assert not fails, fails[:3]
    finally:
        gpu.close()
        for c in cpus:
            c.close()


def test_one_world_reset_grows_the_shared_tables(worlds):
    """a per-world reset that brings longer waypoint lists and more obstacles than any reset before it: the trajectory table
    and the obstacle slices are re-laid out, and the other worlds' contents must survive the move"""
    World, OracleWorld = worlds
    W, Rw, Pw = 3, 4, 5
    grid, params, _ = small_world(Rw, Pw, seed=61, max_ped=8)
    lays = [small_world(Rw, Pw, seed=61 + k, n_obstacles=1)[2] for k in range(W)]
    gpu = World(_stack_params(params, W), grid)
    cpus = [OracleWorld(params, grid) for _ in range(W)]
    fails = []
    try:
        gpu.reset(lays)
        for c, lay in zip(cpus, lays):
            c.reset(lay)
        rng = np.random.default_rng(61)
        for s in range(30):
            a = random_actions(rng, W * Rw)
            gpu.step(a)
            for k, c in enumerate(cpus):
                c.step(a[k * Rw:(k + 1) * Rw])
            _compare_all(gpu, cpus, Rw, Pw, s, fails)
            if s in (4, 11):
                k = 1 if s == 4 else 2
                lay = small_world(Rw, Pw, seed=200 + s, n_obstacles=6 if s == 4 else 14)[2]
                cap = 4 if s == 4 else 7
                traj = np.zeros((Pw, cap, 3))
                lens = rng.integers(1, cap + 1, Pw).astype(np.int32)
                lens[0] = cap
                for j in range(Pw):
                    for q in range(lens[j]):
                        traj[j, q, :2] = lay.ped_pose[j, :2] + rng.uniform(-2.0, 2.0, 2)
                lay.ped_traj, lay.ped_traj_len = traj, lens
                gpu.reset_world(k, lay)
                cpus[k].reset(lay)
                _compare_all(gpu, cpus, Rw, Pw, (s, "reset"), fails)
        assert not fails, fails[:3]
    finally:
        gpu.close()
        for c in cpus:
            c.close()


@pytest.mark.parametrize("flags", [2, 4, 512], ids=["composed_layer", "stamped_layer", "counting_layer"])
def test_compose_modes_give_the_same_worlds(worlds, flags):
    """the class layer composed every step from owner layers, stamped directly by the rasters (STAMP mode), or kept as counts
    by the agents themselves (SUM mode, world.h): crowded worlds, legs, per-world resets, a map width that is not a multiple of
    the tile"""
    World, OracleWorld = worlds
    fails, _, _ = _run(World, OracleWorld, 4, 12, 9, 30, {5: [1], 6: [1, 3], 20: [0, 1, 2, 3]}, seed=35, n_obstacles=3, ped_shape="leg",
                       grid_size=100, clearance=0.6, flags=flags)
    assert not fails, fails[:3]
    fails, _, _ = _run(World, OracleWorld, 1, 40, 12, 30, {}, seed=36, n_obstacles=3, grid_size=116, clearance=0.6, flags=flags)
    assert not fails, fails[:3]
    fails, _, _ = _run(World, OracleWorld, 3, 10, 5, 16, {7: [2]}, seed=37, n_obstacles=2, grid_size=117, clearance=0.6, flags=flags)
    assert not fails, fails[:3]


@pytest.mark.parametrize("seed", range(int(os.environ.get("IMGENV_FUZZ_SEEDS_WORLDS", "10"))))  # more seeds for a bug hunt
def test_fuzzed_world_batches_match_one_oracle_each(worlds, seed):
    """random world counts and sizes, scenes, geometry, compose mode and reset schedules (single worlds, subsets, all)"""
    World, OracleWorld = worlds
    rng = np.random.default_rng(4000 + seed)
    W, Rw = int(rng.integers(2, 9)), int(rng.integers(1, 12))
    scene = str(rng.choice(["rvoscene", "ervoscene", ""]))
    Pw = int(rng.integers(1, 14)) if scene else 0
    res = float(rng.choice([0.125, 0.25, 0.1]))
    view_cells = int(rng.choice([32, 48, 50]))
    steps = int(rng.integers(10, 22))
    resets = {}
    for s in range(steps - 1):
        if rng.random() < 0.4:
            resets[s] = sorted(int(k) for k in rng.choice(W, size=int(rng.integers(1, W + 1)), replace=False))
    extent = max(22.0, 1.5 * np.sqrt((Rw + Pw) * 2.5))
    kw = dict(res=res, view_cells=view_cells, view_width=(view_cells + 0.5) * res, view_height=(view_cells + 0.5) * res,
              beams=int(rng.choice([90, 360])), scene=scene, ped_shape=str(rng.choice(["circle", "leg"])),
              relation_ped_robo=int(rng.integers(0, 2)), time_max=int(rng.integers(5, 14)), dt=float(rng.choice([0.25, 0.4])),
              grid_size=int(np.ceil(extent / res)) | int(rng.integers(0, 2)), n_obstacles=int(rng.integers(0, 4)),
              clearance=float(rng.choice([0.6, 0.8])), flags=int(rng.choice([0, 2, 4, 512])))
    fails, _, _ = _run(World, OracleWorld, W, Rw, Pw, steps, resets, seed=300 + seed, whole_reset=bool(rng.random() < 0.3), **kw)
    assert not fails, (seed, W, Rw, Pw, kw, fails[:2])


def test_bad_multi_world_configurations_are_rejected(worlds):
    World, _ = worlds
    grid, params, _ = small_world(4, 4, seed=1)
    p = _stack_params(params, 2)
    p["n_robots"] = 7  # not a multiple of n_worlds
    for k in ("robot_shape", "robot_size", "robot_sensor_cfg", "robot_size_last"):
        p[k] = np.asarray(p[k])[:7]
    with pytest.raises(ValueError, match="multiples of n_worlds"):
        World(p, grid)
    grid, params, _ = small_world(4, 4, seed=1)
    p = _stack_params(params, 2)
    p["robot_begin"], p["robot_end"] = 0, 4
    with pytest.raises(ValueError, match="shard"):
        World(p, grid)
    w = World(_stack_params(params, 2), grid)
    try:
        lay = small_world(4, 4, seed=2)[2]
        w.reset_world(0, lay)
        with pytest.raises(RuntimeError, match="before reset"):  # world 1 has never been reset
            w.step(np.zeros((8, 3), np.float32))
        with pytest.raises(RuntimeError, match="out of range"):
            w.reset_world(2, lay)
        with pytest.raises(RuntimeError, match="listed twice"):
            w.reset_worlds([1, 1], [lay, lay])
    finally:
        w.close()


def test_stamped_layer_survives_its_tag_coming_round(worlds):
    """STAMP mode tags every stamp with the step count modulo 255 and sweeps the layer when the tag wraps: 300 steps of two
    small worlds (with a per-world reset in between) stay on the oracles across the wrap"""
    World, OracleWorld = worlds
    fails, _, _ = _run(World, OracleWorld, 2, 3, 2, 300, {120: [1], 254: [0], 255: [1]}, seed=38, n_obstacles=2, grid_size=100,
                       clearance=0.6, time_max=1000, flags=4)
    assert not fails, fails[:3]


def test_counting_layer_over_long_episodes_with_standing_and_frozen_agents(worlds):
    """SUM mode keeps its counts by differences (an agent takes itself off the cells it leaves and adds itself to the ones it
    enters): 150 steps of crowded worlds -- robots that collide and freeze, pedestrians with legs that reach their goals and wait,
    per-world resets that redraw obstacles underneath standing agents -- must not drift off the oracles by a single cell"""
    World, OracleWorld = worlds
    fails, _, _ = _run(World, OracleWorld, 3, 14, 9, 150, {40: [1], 41: [1], 90: [0, 2], 120: [0, 1, 2]}, seed=39, n_obstacles=3, ped_shape="leg",
                       grid_size=96, clearance=0.6, time_max=1000, flags=512)
    assert not fails, fails[:3]
    fails, _, _ = _run(World, OracleWorld, 1, 60, 20, 120, {50: [0]}, seed=40, n_obstacles=3, grid_size=100, res=0.25, clearance=0.55,
                       time_max=1000, flags=512)
    assert not fails, fails[:3]


@pytest.mark.parametrize("relation", [1, 0])
def test_pedscene_worlds_match_one_oracle_each(worlds, relation):
    """a social-force crowd per world (the reference: one PedScene per env process, pedscene.h:17-91): 3 worlds x (2 robots that
    are crowd members, 7 pedestrians, own obstacle segments), per-world resets in mid-flight -- positions persist in the
    library's quadtree per world, velocities persist across resets per world.  relation 0: crowds that ignore the robots are
    stepped a step AHEAD on a stream of their own; a reset in between drops what was computed ahead."""
    World, OracleWorld = worlds
    fails, snap, _ = _run(World, OracleWorld, W=3, Rw=2, Pw=7, steps=14, resets={4: [1], 9: [0, 2], 10: [1]}, seed=71, scene="pedscene",
                          grid_size=88, n_obstacles=3, relation_ped_robo=relation)
    assert not fails, fails[:3]
    assert np.abs(snap["ped_state"][:, 2:]).max() > 0.05  # the crowds really moved


def test_pedscene_worlds_above_64_agents_use_the_spread_pair_terms(worlds):
    """two worlds of 70 social-force agents each: the n^2 pair terms run in their own launch, a share of workgroups per world"""
    World, OracleWorld = worlds
    fails, _, _ = _run(World, OracleWorld, W=2, Rw=1, Pw=69, steps=5, resets={2: [1]}, seed=73, scene="pedscene", grid_size=88,
                       n_obstacles=2, relation_ped_robo=1, clearance=0.45)
    assert not fails, fails[:3]
