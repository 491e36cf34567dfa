"""Host-side logic that needs neither the GPU nor the HIP library: synthetic worlds, the EnvPos spawn
rules, the YAML -> InitEnv parameter mapping and the float32 wire rounding of the C ABI structs."""
import numpy as np
import pytest

from img_env_amd import _cabi, config, spawn, worldgen


def _pairwise_min(xy):
    d = np.linalg.norm(xy[:, None, :] - xy[None, :, :], axis=-1)
    d[np.diag_indices(len(xy))] = np.inf
    return d.min()


def test_grid_is_deterministic_and_walled():
    a, b = worldgen.make_grid(200, 3), worldgen.make_grid(200, 3)
    assert a.dtype == np.uint8 and a.shape == (200, 200) and np.array_equal(a, b)
    assert not np.array_equal(a, worldgen.make_grid(200, 4))
    assert set(np.unique(a)) <= {0, 255}
    assert (a[:8] == 0).all() and (a[-8:] == 0).all() and (a[:, :8] == 0).all() and (a[:, -8:] == 0).all()
    assert (a[100] == 255).any()


def test_layout_respects_clearances():
    res, clearance = 0.125, 1.0
    grid = worldgen.make_grid(320, 1)
    lay = worldgen.make_layout(grid, res, 40, 12, seed=5, clearance=clearance, n_obstacles=4)
    starts = np.vstack([lay.robot_pose[:, :2], lay.ped_pose[:, :2]])
    assert _pairwise_min(starts) >= clearance - 1e-9
    cells = np.rint(starts / res).astype(int)
    assert (grid[cells[:, 0], cells[:, 1]] == 255).all()               # every start on a free cell
    d_goal = np.linalg.norm(lay.robot_goal - lay.robot_pose[:, :2], axis=1)
    assert (d_goal >= 1.0 - 1e-9).all() and (np.abs(lay.robot_goal - lay.robot_pose[:, :2]).max(1) <= 4.0 + 1e-9).all()
    q = lay.robot_pose[:, 2:]
    assert np.allclose((q ** 2).sum(1), 1.0)                           # (qz, qw) is a unit quaternion
    assert lay.ped_traj.shape == (12, 2, 3) and (lay.ped_traj_len == 2).all()  # go_back: [goal, start]
    assert np.allclose(lay.ped_traj[:, 1, :2], lay.ped_pose[:, :2])
    b = lay.as_batch()
    assert b["ped_traj_cap"] == 2 and len(b["obs_shape"]) == 4 and b["ped_traj_v"] is None
    again = worldgen.make_layout(grid, res, 40, 12, seed=5, clearance=clearance, n_obstacles=4)
    assert np.array_equal(again.robot_pose, lay.robot_pose) and np.array_equal(again.ped_traj, lay.ped_traj)


def test_yaml_cfg_maps_to_the_same_parameters_as_make_params():
    grid = worldgen.make_grid(200, 0)
    cfg = worldgen.make_yaml_cfg(6, 3, grid, time_max=40, ped_shape="leg", state_dim=5, n_obstacles=2)
    p = config.params_from_cfg(cfg)
    q = worldgen.make_params(6, 3, ped_shape="leg", state_dim=5, time_max=40)
    for k in ("view_resolution", "view_width", "view_height", "step_hz", "state_dim", "range_total", "view_angle_begin",
              "view_angle_end", "relation_ped_robo", "ped_scene_type", "n_robots", "n_peds", "time_max", "laser_max",
              "ped_image_r", "max_ped"):
        assert p[k] == q[k], k
    assert np.array_equal(np.asarray(p["ped_size"], np.float32), np.asarray(q["ped_size"], np.float32))
    assert np.array_equal(np.asarray(p["robot_size"], np.float32), np.asarray(q["robot_size"], np.float32))
    assert tuple(p["image_size"]) == (48, 48)


def test_cfg_struct_rounds_to_the_float32_wire():
    p = worldgen.make_params(2, 0, res=0.1)                            # 0.1 is not a float32
    c, keep = _cabi.make_cfg(p)
    assert c.view_resolution == np.float32(0.1) and c.view_resolution != 0.1
    assert c.struct_size == _cabi.C.sizeof(_cabi.Cfg) and c.abi_version == _cabi.ABI_VERSION
    assert c.robot_end == 2 and c.robot_begin == 0 and c.n_robots == 2
    assert keep["robot_size"].dtype == np.float32 and keep["robot_size_last"].dtype == np.float64


def test_reset_batch_keeps_its_buffers_alive_and_typed():
    grid = worldgen.make_grid(200, 0)
    lay = worldgen.make_layout(grid, 0.125, 3, 2, seed=1)
    b, keep = _cabi.make_reset_batch(lay.as_batch(), 3, 2)
    assert b.struct_size == _cabi.C.sizeof(_cabi.ResetBatch) and b.ped_traj_cap == 2 and not b.ped_traj_v
    assert keep["robot_pose"].shape == (3, 4) and keep["ped_traj"].shape == (2, 2, 3)
    assert b.robot_pose[0] == lay.robot_pose[0, 0]


def test_envpos_spawn_rules():
    """the product's EnvPos (the library's own sampler behind img_env_amd.spawn) keeps the reference's placement rules"""
    grid = worldgen.make_grid(320, 0)
    cfg = worldgen.make_yaml_cfg(16, 6, grid, n_obstacles=2)
    ep = spawn.EnvPos(cfg, seed=11)
    extent = 320 * 0.125
    lay = ep.reset(extent)
    starts = np.vstack([lay.robot_pose[:, :2], lay.ped_pose[:, :2]])
    assert _pairwise_min(starts) > 1.0 - 1e-9                          # free_check_robo_ped d = 1.0 (reset_helper.py:35-43)
    assert (np.linalg.norm(lay.robot_goal - lay.robot_pose[:, :2], axis=1) > float(cfg["target_min_dist"]) - 1e-9).all()
    module = 2 * 0.17                                                  # starts clear of the obstacles (reset_helper.py:46-55, 167-186)
    for o in range(len(lay.obs_shape)):
        rad = lay.obs_size[o, 2] if lay.obs_shape[o] == 0 else np.hypot(lay.obs_size[o, 0], lay.obs_size[o, 2])
        assert (np.linalg.norm(starts[:16] - lay.obs_pose[o, :2], axis=1) > rad + module - 1e-6).all()
    same = spawn.EnvPos(cfg, seed=11).reset(extent)
    assert np.array_equal(same.robot_pose, lay.robot_pose)
    assert not np.array_equal(spawn.EnvPos(cfg, seed=12).reset(extent).robot_pose, lay.robot_pose)
    second = ep.reset(extent)                                          # a new episode per call
    assert not np.array_equal(second.robot_pose, lay.robot_pose)


def test_envpos_matches_the_reference_sampler_statistically():
    """the product's sampler (its own random stream) against the test-side replica of the reference's EnvPos (Python's
    Mersenne Twister, pinned bit for bit on reference episodes): the same distributions of starts, goals and obstacle sizes
    over many episodes of a range / range_view cast and of a circle cast"""
    import json
    import os
    from spawn_replica import ReferenceEnvPos
    grid = worldgen.make_grid(320, 0)
    cfgs = [worldgen.make_yaml_cfg(6, 3, grid, n_obstacles=2)]
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "spawn_ref.npz"))
    cfgs.append(json.loads(str(z["test@1/cfg"])))  # the shipped test.yaml cast: range_circle starts and targets
    for cfg in cfgs:
        nr = int(cfg["robot"]["total"])
        ours, ref = spawn.EnvPos(cfg, seed=3), ReferenceEnvPos(cfg, seed=3)
        A = [ours.reset() for _ in range(400)]
        B = [ref.reset() for _ in range(400)]

        def stats(L):
            start = np.array([np.vstack([l.robot_pose[:, :2], l.ped_pose[:, :2]]) for l in L])   # [episode][agent][2]
            goal = np.array([np.vstack([l.robot_goal, l.ped_goal]) for l in L])
            d = np.linalg.norm(goal - start, axis=2)
            osz = np.array([l.obs_size[:, 2] for l in L]) if len(L[0].obs_shape) else np.zeros((len(L), 1))
            return start.mean(axis=(0, 1)), start.std(axis=(0, 1)), d.mean(), d.std(), osz.mean()
        sa, sb = stats(A), stats(B)
        span = max(sb[1].max(), 0.5)
        assert np.all(np.abs(sa[0] - sb[0]) < 0.12 * span), (sa[0], sb[0])      # where agents start: mean ...
        assert np.all(np.abs(sa[1] - sb[1]) < 0.12 * span), (sa[1], sb[1])      # ... and spread
        assert abs(sa[2] - sb[2]) < 0.1 * sb[2] and abs(sa[3] - sb[3]) < 0.15 * max(sb[3], 0.2), (sa[2:4], sb[2:4])  # start -> goal distances
        assert abs(sa[4] - sb[4]) < 0.02 + 0.05 * sb[4]                          # obstacle radii
        assert nr == A[0].robot_pose.shape[0]


def test_init_ped_dataset_shapes():
    grid = worldgen.make_grid(200, 0)
    lay = worldgen.make_layout(grid, 0.125, 2, 3, seed=1)
    data = np.arange(3 * 5 * 5, dtype=np.float64).reshape(3, 5, 5)
    spawn.init_ped_dataset(lay, data)
    assert lay.ped_traj.shape == (3, 5, 3) and lay.ped_traj_v.shape == (3, 5, 2) and (lay.ped_traj_len == 5).all()
    assert np.array_equal(lay.ped_pose[:, :2], data[:, 0, :2])
    assert np.allclose(lay.ped_pose[:, 2], np.sin(data[:, 0, 2] / 2)) and np.allclose(lay.ped_pose[:, 3], np.cos(data[:, 0, 2] / 2))
    b, keep = _cabi.make_reset_batch(lay.as_batch(), 2, 3)
    assert b.ped_traj_cap == 5 and bool(b.ped_traj_v) and keep["ped_traj_v"].shape == (3, 5, 2)


def _spawn_cases():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "spawn_ref.npz"))
    return z, sorted(k[:-5] for k in z.files if k.endswith("/seed"))


@pytest.mark.parametrize("case", _spawn_cases()[1])
def test_envpos_reproduces_the_reference_episodes(case):
    """the test-side replica `spawn_replica.ReferenceEnvPos(cfg, seed)` against episodes the REFERENCE's own EnvPos placed after `random.seed(seed)`
    (tests/golden/gen_spawn_golden.py): obstacles, starts, targets, trajectories -- bit for bit.  Covers the spawn sections of
    the shipped circle.yaml / test.yaml / 10obs_5ped_baseline.yaml (range_circle starts and targets), a synthetic cast with
    every other pose type, and a circle too small for its cast (the "50 failed circle starts" path)."""
    import json
    z, _ = _spawn_cases()
    g = lambda k: z["%s/%s" % (case, k)]  # noqa: E731
    cfg = json.loads(str(g("cfg")))
    from spawn_replica import ReferenceEnvPos
    ep = ReferenceEnvPos(cfg, seed=int(g("seed")))
    lay = ep.reset()
    nr = cfg["robot"]["total"]
    assert ep.circle_range == float(g("circle_range"))
    assert np.array_equal(np.array(ep.obs_range, float).reshape(-1, 4), g("obs_range"))
    assert np.array_equal(np.array([p[:3] for p in ep.init_poses], float).reshape(-1, 3), g("init"))
    assert np.array_equal(np.array([p[:3] for p in ep.target_poses], float).reshape(-1, 3), g("target"))
    # ... and what becomes of them in the reset batch (ResetEnv.srv: poses as quaternions, goals, trajectories)
    assert np.array_equal(lay.obs_shape, g("obs_shape"))
    assert np.array_equal(lay.obs_size, g("obs_size").astype(np.float32))
    assert np.array_equal(lay.obs_pose[:, :2], g("obs_range")[:, :2]) and np.array_equal(lay.obs_pose[:, 2:], g("obs_quat"))
    assert np.array_equal(lay.robot_pose[:, :2], g("init")[:nr, :2]) and np.array_equal(lay.robot_pose[:, 2:], g("robot_quat"))
    assert np.array_equal(lay.robot_goal, g("robot_goal"))
    assert np.array_equal(lay.ped_pose[:, :2], g("init")[nr:, :2])
    assert np.array_equal(lay.ped_traj_len, g("ped_traj_len"))
    for j, n in enumerate(lay.ped_traj_len):
        assert np.array_equal(lay.ped_traj[j, :n, :2], g("ped_traj")[j, :n])


def test_unsupported_spawn_layouts_fail_loudly():
    grid = worldgen.make_grid(200, 0)
    cfg = worldgen.make_yaml_cfg(2, 0, grid)
    cfg["robot"]["target_poses_type"] = ["range_view_plus", "range_view_plus"]  # random_view_plus does not exist in the reference
    with pytest.raises(NotImplementedError):
        spawn.EnvPos(cfg, seed=0).reset(25.0)
    cfg = worldgen.make_yaml_cfg(2, 0, grid)
    cfg["robot"]["begin_poses_type"] = ["fix", "range"]  # fixed start + random target: the reference loops forever
    cfg["robot"]["begin_poses"] = [[3.0, 3.0, 0.0], cfg["robot"]["begin_poses"][1]]
    with pytest.raises(RuntimeError, match="never leaves"):
        spawn.EnvPos(cfg, seed=0).reset(25.0)


def test_stack_params_repeats_one_env_world_major():
    """VecImageEnv's parameter dict: env_num copies of one env's per-robot / per-pedestrian rows, totals and n_worlds"""
    from img_env_amd import worldgen
    from img_env_amd.vec_env import stack_params
    p = worldgen.make_params(3, 2, ped_shape="leg")
    p["robot_size"][1, 2] = 0.25
    q = stack_params(p, 4)
    assert (q["n_robots"], q["n_peds"], q["n_worlds"]) == (12, 8, 4)
    assert q["robot_size"].shape == (12, 4) and q["ped_size"].shape == (8, 6) and q["robot_size_last"].shape == (12,)
    for k in range(4):
        assert np.array_equal(q["robot_size"][3 * k:3 * k + 3], p["robot_size"])
        assert np.array_equal(q["ped_max_speed"][2 * k:2 * k + 2], p["ped_max_speed"])
    assert p["n_robots"] == 3 and "n_worlds" not in p  # the input is left alone
    cfg, keep = __import__("img_env_amd._cabi", fromlist=["make_cfg"]).make_cfg(q)
    assert (cfg.n_worlds, cfg.n_robots, cfg.n_peds) == (4, 12, 8)


def test_native_spawn_follows_the_envpos_rules():
    """the library's own placement (csrc/spawn_host.h, no device needed) against the rules test_envpos_spawn_rules holds the
    Python EnvPos to (reset_helper.py:35-82, 245-300)"""
    grid = worldgen.make_grid(320, 0)
    cfg = worldgen.make_yaml_cfg(16, 6, grid, n_obstacles=2)
    sc = spawn.make_spawn_cfg(cfg)
    lay = spawn.native_spawn(cfg, 11, sc)
    starts = np.vstack([lay.robot_pose[:, :2], lay.ped_pose[:, :2]])
    goals = np.vstack([lay.robot_goal, lay.ped_goal])
    assert _pairwise_min(starts) > 1.0 - 1e-9 and _pairwise_min(goals) > 1.0 - 1e-9     # free_check_robo_ped d = 1.0
    assert (np.linalg.norm(goals - starts, axis=1) > float(cfg["target_min_dist"]) - 1e-9).all()
    for q in range(2):                                                                   # starts / goals clear of the obstacles
        radius = lay.obs_size[q, 2] if lay.obs_shape[q] == _cabi.SHAPE_CIRCLE else np.hypot(lay.obs_size[q, 0], lay.obs_size[q, 2])
        assert radius > 0
        for pts in (starts, goals):
            assert (np.linalg.norm(pts - lay.obs_pose[q, :2], axis=1) > radius).all()
    assert np.allclose(lay.robot_pose[:, 2] ** 2 + lay.robot_pose[:, 3] ** 2, 1.0)       # (x, y, qz, qw)
    assert (lay.ped_traj_len == 2).all() and np.array_equal(lay.ped_traj[:, 0, :2], lay.ped_goal)  # go_back: target, then start
    assert np.array_equal(lay.ped_traj[:, 1, :2], lay.ped_pose[:, :2])
    same = spawn.native_spawn(cfg, 11, sc)
    assert np.array_equal(same.robot_pose, lay.robot_pose) and np.array_equal(same.ped_goal, lay.ped_goal)
    assert not np.array_equal(spawn.native_spawn(cfg, 12, sc).robot_pose, lay.robot_pose)
    # range_view targets: inside the 4 m box around the start, outside its 2.5 m box (random_view, reset_helper.py:62-82)
    cfg["robot"]["target_poses_type"] = ["range_view"] * 16
    lay = spawn.native_spawn(cfg, 3)
    d = np.abs(lay.robot_goal - lay.robot_pose[:, :2])
    assert (d.max(axis=1) <= 4.0 + 1e-9).all() and (d.max(axis=1) > 2.5).all()
    cfg["robot"]["target_poses_type"] = ["range_view_plus"] * 16
    with pytest.raises(NotImplementedError):
        spawn.make_spawn_cfg(cfg)


@pytest.mark.parametrize("case", ["circle@1", "test@2", "10obs_5ped_baseline@7", "synthetic@3", "synthetic@5"])
def test_native_spawn_places_the_reference_casts(case):
    """the library's own placement on the spawn sections of the reference's shipped configs (range_circle starts / targets) and
    on the synthetic cast with every pose type: the rules of reset_helper.py, checked on the result"""
    import json
    z, _ = _spawn_cases()
    cfg = json.loads(str(z[case + "/cfg"]))
    nr, npd = cfg["robot"]["total"], cfg["ped_sim"]["total"]
    btype = cfg["robot"]["begin_poses_type"][:nr] + cfg["ped_sim"]["begin_poses_type"][:npd]
    ttype = cfg["robot"]["target_poses_type"][:nr] + cfg["ped_sim"]["target_poses_type"][:npd]
    bpose = cfg["robot"]["begin_poses"][:nr] + cfg["ped_sim"]["begin_poses"][:npd]
    tpose = cfg["robot"]["target_poses"][:nr] + cfg["ped_sim"]["target_poses"][:npd]
    sc = spawn.make_spawn_cfg(cfg)
    for seed in range(20):
        lay = spawn.native_spawn(cfg, 100 + seed, sc)
        starts = np.vstack([lay.robot_pose[:, :2], lay.ped_pose[:, :2]])
        goals = np.vstack([lay.robot_goal, lay.ped_goal])
        yaw = 2.0 * np.arctan2(np.concatenate([lay.robot_pose[:, 2], lay.ped_pose[:, 2]]), np.concatenate([lay.robot_pose[:, 3], lay.ped_pose[:, 3]]))
        drawn = [i for i in range(nr + npd) if btype[i] not in ("fix", "rand_angle")]
        if len(drawn) > 1:
            assert _pairwise_min(starts[drawn]) > 1.0 - 1e-9
        for i in range(nr + npd):
            if ttype[i] not in ("fix", "rand_angle", "circle_fix"):
                assert np.linalg.norm(goals[i] - starts[i]) > cfg["target_min_dist"] - 1e-9
            if "circle" in btype[i]:  # on the episode's circle (1.8 .. 3.2 m) give or take the noise (sigma 0.5), facing the centre
                r = np.linalg.norm(starts[i] - np.array(bpose[i][:2]))
                assert 0.0 < r < cfg["circle_ranges"][1] + 3.5
            if "circle" in ttype[i] and "circle" in btype[i]:
                # the target lies where the start looks: on the far side of the centre
                to_goal = goals[i] - np.array(tpose[i][:2])
                assert np.dot(to_goal, [np.cos(yaw[i]), np.sin(yaw[i])]) > -2.5
            if btype[i] == "range_multi":
                assert any(b[0] <= starts[i, 0] <= b[1] and b[2] <= starts[i, 1] <= b[3] for b in bpose[i])
            if btype[i] == "range":
                b = bpose[i]
                assert b[0] <= starts[i, 0] <= b[1] and b[2] <= starts[i, 1] <= b[3]
            if ttype[i] == "range_view":
                d = np.abs(goals[i] - starts[i])
                assert d.max() <= 4.0 + 1e-9 and d.max() > 2.5
        for q in range(len(lay.obs_shape)):
            radius = lay.obs_size[q, 2] if lay.obs_shape[q] == _cabi.SHAPE_CIRCLE else np.hypot(lay.obs_size[q, 0], lay.obs_size[q, 2])
            if len(drawn):
                assert (np.linalg.norm(starts[drawn] - lay.obs_pose[q, :2], axis=1) > radius).all()
    a, b = spawn.native_spawn(cfg, 5, sc), spawn.native_spawn(cfg, 5, sc)
    assert np.array_equal(a.robot_pose, b.robot_pose) and np.array_equal(a.ped_goal, b.ped_goal)


def test_episode_stats_accumulate_what_the_reference_computes_from_lists():
    """img_env_amd.envs.EpisodeStats (running sums for all robots at once) against the list-based formulas of the reference's
    evaluation helper (envs/wrapper/evaluation_wrapper/utils.py:60-129), restated here with numpy"""
    import torch
    from img_env_amd.envs import EpisodeStats
    rng = np.random.default_rng(0)
    R, T, dt = 5, 40, 0.4
    V, W = rng.uniform(0, 0.6, (T, R)), np.round(rng.uniform(-0.9, 0.9, (T, R)), 1)  # rounded: exact zeros occur
    st = EpisodeStats(R, dt, "cpu")
    for t in range(T):
        st.add(torch.tensor(V[t]), torch.tensor(W[t]))
    out = st.finish()
    for r in range(R):
        v, w = V[:, r], W[:, r]
        tmp = w_zero = 0
        for x in w:  # cal_w_zero
            if x == 0:
                if tmp != 0:
                    w_zero += 1
            elif (x > 0 and tmp < 0) or (x < 0 and tmp > 0):
                w_zero += 1
            tmp = x
        va, wa = np.diff(v) / dt, np.diff(w) / dt
        want = dict(w_variance=np.var(w), w_zero=w_zero, v_jerk=np.average(np.abs(np.diff(va) / dt)), w_jerk=np.average(np.abs(np.diff(wa) / dt)),
                    v_acc=np.average(np.abs(va)), w_acc=np.average(np.abs(wa)), v_avg=np.average(v), w_avg=np.average(np.abs(w)))
        for k, x in want.items():
            assert abs(float(out[k][r]) - x) < 1e-9, (k, r)
    assert float(st.n.sum()) == 0  # finish() starts the accumulators over


def test_bench_workloads_are_the_baseline_configs():
    """bench.py's --config table against BASELINE.json's configs as worldgen.PRESETS spells them, and the algorithmic bytes per
    robot-step against SURVEY.md section 8(d)'s own numbers (10 784 B for cfg-2 with the f16 copy, 48 036 for cfg-3, 97 092 +
    2 x 96^2 for cfg-5 with the f16 copy)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for name, wl in bench.WORKLOADS.items():
        pre = worldgen.PRESETS[name]
        assert wl["grid"] == pre["grid"] and wl["res"] == pre["res"] and wl["view"] == pre["view_cells"] and wl["beams"] == pre["beams"], name
        assert wl["peds"] == pre["n_peds"] and wl["scene"] == pre["scene"], name
        # robots: the world's for the configs that fit (or are strong-scaled over) one node, the per-GPU share of cfg-4's 65 536
        assert wl["robots"] == (pre["n_robots"] if name != "cfg4" else pre["n_robots"] // 8), name
    ab = bench.algorithmic_bytes
    assert ab(0)["total"] == 10784
    assert ab(200)["total"] == 48036
    assert ab(1000, hv=96, wv=96, beams=720)["total"] == 97092 + 2 * 96 * 96
    # every rank its own world, the same world strong-scaled, or one GPU's share weak-scaled: the three ways bench.py places robots
    assert {w["scaling"] for w in bench.WORKLOADS.values()} == {"replicas", "strong", "weak"}
