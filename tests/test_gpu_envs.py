"""The Python mirror (make_env / ImageEnv / wrapper stack) against the golden vectors recorded from the
reference's own make_env + wrapper stack (tests/golden/gen_python_golden.py)."""
import ast
import glob
import os

import numpy as np
import pytest

from scenarios import golden_cfg, golden_scenario

pytestmark = pytest.mark.gpu
FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "python_post_*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_make_env_stack_matches_reference(path):
    import torch
    from img_env_amd import make_env, worldgen
    z = np.load(path)
    meta = ast.literal_eval(str(z["meta"]))
    grid, params, layout = golden_scenario(meta)
    cfg = golden_cfg(meta, grid)
    env = make_env(cfg)   # VelAction, TimeLimit, SensorsPaperReward, InfoLog, MultiRobotClean
    try:
        st = env.reset(layout=layout)
        assert len(st) == meta["n_robots"] == env.robot_total
        assert np.array_equal(st.numpy().sensor_maps, z["exp_sensor_maps"][0])
        for s in range(meta["steps"]):
            act = torch.as_tensor(z["actions"][s], dtype=torch.float32, device="cuda")
            st, rew, done, info = env.step(act)
            h = st.numpy()
            assert np.array_equal(h.is_collisions, z["exp_is_collisions"][s + 1])
            assert np.array_equal(h.is_arrives, z["exp_is_arrives"][s + 1])
            assert h.vector_states.dtype == np.float64 and h.sensor_maps.dtype == np.float16
            assert np.abs(h.vector_states - z["exp_vector_states"][s + 1]).max() <= 1e-4
            assert np.abs(h.ped_maps - z["exp_ped_maps"][s + 1]).max() <= 1e-4
            assert np.abs(rew.cpu().numpy() - z["exp_rewards"][s]).max() <= 1e-4, s
            assert np.array_equal(done.cpu().numpy(), z["exp_dones"][s]), s
            assert np.array_equal(info["dones_info"].cpu().numpy(), z["exp_dones_info"][s]), s
            assert np.array_equal(info["is_clean"].cpu().numpy(), z["exp_is_clean"][s]), s
            assert np.array_equal(info["all_down"].cpu().numpy(), z["exp_all_down"][s]), s
            assert np.abs(info["speeds"].cpu().numpy() - z["exp_speeds"][s]).max() <= 1e-6, s
    finally:
        env.close()


STACKS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "python_stack_*.npz")))


@pytest.mark.parametrize("path", STACKS, ids=[os.path.basename(p) for p in STACKS])
def test_full_wrapper_stack_matches_reference(path):
    """StatePedVectorWrapper, StateBatchWrapper (frame stacks), the discrete VelActionWrapper table, ObsStateTmp /
    ObsLaserStateTmp and NeverStopWrapper against what the reference's own stack returned (gen_python_golden.py::run_stack),
    through its time-limit auto-resets"""
    import torch
    from img_env_amd import make_env, worldgen
    from img_env_amd.envs import ImageEnv
    z = np.load(path)
    meta = ast.literal_eval(str(z["meta"]))
    grid = worldgen.make_grid(200, meta["seed"])
    cfg = golden_cfg(meta, grid)
    layouts = [worldgen.make_layout(grid, 0.125, meta["n_robots"], meta["n_peds"], seed=meta["seed"] + 100 + k, n_obstacles=2)
               for k in range(meta["n_layouts"])]
    env = make_env(cfg)
    inner = env
    while not isinstance(inner, ImageEnv):
        inner = inner.env

    class Seq:  # every reset takes the next fixed layout, as the generator's reset service did
        n = 0

        def reset(self, extent=None):
            lay = layouts[Seq.n % len(layouts)]
            Seq.n += 1
            return lay
    inner.env_pose = Seq()

    def same_obs(obs, t):
        assert len(obs) == 3
        for k, o in enumerate(obs):
            want = z["exp_obs%d" % k][t]
            got = o.cpu().numpy()
            assert got.shape == want.shape, (t, k, got.shape, want.shape)
            if meta["obs_names"][k] == "sensor_maps":
                assert np.array_equal(got, want), (t, k)
            else:
                assert np.abs(got.astype(np.float64) - want).max() <= 1e-4, (t, k)
    try:
        same_obs(env.reset(), 0)
        for s in range(meta["steps"]):
            a = z["actions"][s]
            act = torch.as_tensor(a, device="cuda") if a.ndim == 1 else torch.as_tensor(a, dtype=torch.float32, device="cuda")
            obs, rew, done, info = env.step(act)
            same_obs(obs, s + 1)
            assert np.abs(rew.cpu().numpy() - z["exp_rewards"][s]).max() <= 1e-4, s
            assert np.array_equal(done.cpu().numpy(), z["exp_dones"][s]), s
            assert np.array_equal(info["dones_info"].cpu().numpy(), z["exp_dones_info"][s]), s
            assert np.array_equal(info["is_clean"].cpu().numpy(), z["exp_is_clean"][s]), s
            assert np.array_equal(info["all_down"].cpu().numpy(), z["exp_all_down"][s]), s
            assert np.abs(info["speeds"].cpu().numpy() - z["exp_speeds"][s]).max() <= 1e-6, s
            assert np.array_equal(info["arrive"].cpu().numpy().astype(bool), z["exp_arrive"][s].astype(bool)), s
            assert np.array_equal(info["collision"].cpu().numpy(), z["exp_collision"][s]), s
            if meta["n_peds"]:
                assert np.array_equal(info["bool_get_close_to_human"].cpu().numpy(), z["exp_close"][s]), s
        assert Seq.n == int(z["n_resets"]) >= 2
        if "te_counts" in z.files:  # TestEpisodeWrapper's statistics (TestEpisodeWrapper.py:36-80)
            te = env
            while type(te).__name__ != "TestEpisodeWrapper":
                te = te.env
            got = [te.cur_episode, te.arrive_num, te.static_coll_num, te.ped_coll_num, te.other_coll_num, te.steps, te.stuck_num,
                   te.speed_step]
            assert got == z["te_counts"].tolist() and te.cur_episode >= 3
            assert np.allclose([te.v_sum, te.w_sum], z["te_sums"], atol=1e-6)
            assert np.allclose(np.array([te.w_variance_array, te.v_jerk_array, te.w_jerk_array, te.w_zero_array], np.float64),
                               z["te_arrays"], atol=1e-4)
    finally:
        env.close()


def test_shipped_test_yaml_geometry_loads_through_make_env(tmp_path):
    """the numbers of the reference's envs/cfg/test.yaml through make_env: a 110 x 110 pixel PNG at 0.1 m (a synthetic room in
    place of the reference's room_10.png) resized to 733 x 733 cells, 400 x 400 cell views shrunk to 48 x 48, 1000 beams, the
    shipped cast (1 robot on a circle, 4 leg pedestrians, 4 obstacles: the spawn sections as the fixture carries them) and
    the shipped wrapper list, TestEpisodeWrapper included"""
    import json
    import torch
    from PIL import Image
    from img_env_amd import make_env, worldgen
    m = np.full((110, 110), 255, np.uint8)
    m[:5] = m[-5:] = 0
    m[:, :5] = m[:, -5:] = 0
    Image.fromarray(np.stack([m] * 3, -1)).save(str(tmp_path / "room.png"))  # grey in RGB, as the reference's maps are
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "spawn_ref.npz"))
    sections = json.loads(str(z["test@1/cfg"]))
    cfg = worldgen.shipped_test_yaml_cfg("room.png", sections)
    cfg.update(map_dir=str(tmp_path), seed=5, time_max=12, init_pose_bag_episodes=1000)
    env = make_env(cfg)
    try:
        w = env.world
        assert (w.n_robots, w.n_peds) == (1, 4)
        assert w.out["view_maps"].shape == (1, 400, 400) and w.out["sensor_maps"].shape == (1, 48, 48)
        assert w.out["lasers"].shape == (1, 1000)
        obs = env.reset()
        assert obs[0].shape == (1, 1, 1000) and obs[1].shape == (1, 9) and obs[2].shape == (1, 3, 48, 48)
        vm = w.out["view_maps"][0].cpu().numpy()
        assert set(np.unique(vm)) <= {0, 100, 200, 255} and (vm == 100).sum() > 300 and (vm == 255).sum() > 10000
        resets = 0
        rng = np.random.default_rng(1)
        for s in range(40):
            obs, rew, done, info = env.step(torch.as_tensor(rng.integers(0, 28, 1), device="cuda"))
            resets += int(bool(info["all_down"][0]))
            assert torch.isfinite(obs[0]).all() and float(obs[0].max()) <= 1.0
        assert resets >= 2
        te = env
        while type(te).__name__ != "TestEpisodeWrapper":
            te = te.env
        assert te.cur_episode == resets and te.speed_step > 0
    finally:
        env.close()


def test_reference_action_objects_and_random_spawn():
    """List[ContinuousAction] in, random EnvPos-style spawn, NeverStopWrapper auto-reset"""
    from img_env_amd import ContinuousAction, make_env, worldgen
    grid = worldgen.make_grid(200, 1)
    cfg = worldgen.make_yaml_cfg(6, 4, grid, time_max=5, n_obstacles=2, seed=3,
                                 wrappers=["VelActionWrapper", "TimeLimitWrapper", "SensorsPaperRewardWrapper",
                                           "InfoLogWrapper", "MultiRobotCleanWrapper", "StateBatchWrapper",
                                           "ObsLaserStateTmp", "NeverStopWrapper"], image_batch=2, state_batch=3)
    env = make_env(cfg)
    try:
        obs = env.reset()
        assert obs[0].shape == (6, 1, 360) and obs[1].shape == (6, 9) and obs[2].shape == (6, 3, 48, 48)
        resets = 0
        for s in range(14):
            obs, rew, done, info = env.step([[0.3, 0.1]] * 6)
            resets += int(bool(info["all_down"][0]))
        assert resets == 2            # time_max 5 -> every 6th step ends the episode
        assert cfg["node_id"] == 1    # make_env post-increments node_id (envs/__init__.py:32)
    finally:
        env.close()


def test_terminal_step_survives_the_auto_reset():
    """an episode that ends by COLLISION and one that ends by ARRIVAL (not by the time limit): NeverStopWrapper resets the
    env inside the same step() call (base.py:198-211), and the reset rewrites the library's done / collision / arrive buffers
    in place.  What step() returned for the terminal step must still be the terminal step's values, as with the reference's
    `deepcopy(self.dones)` (yaml_env.py:377)."""
    import torch
    from img_env_amd import make_env, worldgen
    from test_oracle_known_answers import _layout
    grid = worldgen.make_grid(200, 1)
    stack = ["VelActionWrapper", "TimeLimitWrapper", "SensorsPaperRewardWrapper", "InfoLogWrapper", "MultiRobotCleanWrapper",
             "NeverStopWrapper"]
    cfg = worldgen.make_yaml_cfg(1, 0, grid, time_max=50, n_obstacles=0, seed=3, wrappers=stack)
    env = make_env(cfg)
    try:
        # 0.45 m in front of the border wall (8 cells = 1.0 m thick), driving at it: collides within a few steps
        env.reset(layout=_layout([(1.62, 12.0, 3.14159)], [(20.0, 12.0)]))
        for s in range(10):
            st, rew, done, info = env.step(torch.tensor([[0.6, 0.0]], device="cuda"))
            if bool(info["all_down"][0]):
                break
        assert s < 9, "the robot never reached the wall"
        assert done.cpu().tolist() == [1] and info["collision"].cpu().tolist() == [1] and info["arrive"].cpu().tolist() == [0]
        assert info["dones_info"].cpu().tolist() == [1] and rew.cpu().tolist() == [-500.0]
        assert st.is_collisions.cpu().tolist() == [0]    # the state IS the new episode's first observation
        # arrival: the goal 0.4 m ahead
        env.reset(layout=_layout([(12.0, 12.0, 0.0)], [(12.4, 12.0)]))
        st, rew, done, info = env.step(torch.tensor([[0.6, 0.0]], device="cuda"))
        assert bool(info["all_down"][0])
        assert done.cpu().tolist() == [1] and info["arrive"].cpu().tolist() == [1] and info["collision"].cpu().tolist() == [0]
        assert info["dones_info"].cpu().tolist() == [5] and rew.cpu().tolist() == [500.0]
        assert st.is_arrives.cpu().tolist() == [0]
    finally:
        env.close()


def test_vec_env_matches_one_make_env_per_env():
    """env_num envs in one handle (VecImageEnv) against env_num separate make_env stacks with the same spawn seeds:
    same observations, rewards, dones and infos on every step, through time-limit auto-resets (all envs at once) and a
    manual reset that puts one env out of phase with the others"""
    import copy
    import torch
    from img_env_amd import make_env, worldgen
    from img_env_amd.vec_env import VecImageEnv
    E, R, P = 3, 4, 3
    grid = worldgen.make_grid(200, 2)
    cfg = worldgen.make_yaml_cfg(R, P, grid, time_max=6, n_obstacles=2, seed=21,
                                 wrappers=["VelActionWrapper", "TimeLimitWrapper", "SensorsPaperRewardWrapper",
                                           "InfoLogWrapper", "MultiRobotCleanWrapper", "NeverStopWrapper"])
    envs = []
    for k in range(E):
        c = copy.deepcopy(cfg)
        c["seed"] = 21 + k
        envs.append(make_env(c))
    vec = VecImageEnv(copy.deepcopy(cfg), env_num=E, seed=21)

    def same_state(sv, refs, where):
        v = sv.numpy()
        for k, sr in enumerate(refs):
            r, sl = sr.numpy(), slice(k * R, (k + 1) * R)
            assert np.array_equal(v.sensor_maps[sl], r.sensor_maps), (where, k)
            assert np.array_equal(v.is_collisions[sl], r.is_collisions), (where, k)
            assert np.array_equal(v.is_arrives[sl], r.is_arrives), (where, k)
            for f in ("vector_states", "lasers", "ped_vector_states", "ped_maps", "step_ds"):
                assert np.abs(getattr(v, f)[sl] - getattr(r, f)).max() <= 1e-4, (where, k, f)

    try:
        same_state(vec.reset(), [e.reset() for e in envs], "reset")
        rng = np.random.default_rng(5)
        n_auto = 0
        for s in range(30):
            a = np.zeros((E * R, 3), np.float32)
            a[:, 0], a[:, 1] = rng.uniform(0, 0.6, E * R), rng.uniform(-0.9, 0.9, E * R)
            at = torch.as_tensor(a, device="cuda")
            sv, rew, done, info = vec.step(at)
            refs = []
            for k, e in enumerate(envs):
                sr, rr, dr, ir = e.step(at[k * R:(k + 1) * R])
                refs.append(sr)
                sl = slice(k * R, (k + 1) * R)
                assert np.abs(rew[sl].cpu().numpy() - rr.cpu().numpy()).max() <= 1e-4, (s, k)
                assert np.array_equal(done[sl].cpu().numpy() > 0, dr.cpu().numpy() > 0), (s, k)
                assert np.array_equal(info["dones_info"][sl].cpu().numpy(), ir["dones_info"].cpu().numpy()), (s, k)
                assert np.array_equal(info["is_clean"][sl].cpu().numpy() > 0, ir["is_clean"].cpu().numpy() > 0), (s, k)
                assert bool(ir["all_down"][0]) == (k in info["reset_envs"]), (s, k)
            same_state(sv, refs, s)
            n_auto += len(info["reset_envs"])
            if s == 2:  # env 1 starts over on its own: from now on it times out three steps after the others
                refs[1] = envs[1].reset()
                same_state(vec.reset_envs([1]), refs, "manual reset")
        assert n_auto >= 3 * E
    finally:
        vec.close()
        for e in envs:
            e.close()


@pytest.mark.parametrize("E,R,P", [(5, 3, 4), (6, 1, 0), (3, 2, 1)])
def test_vec_env_with_native_spawn_matches_oracles_fed_the_same_placements(E, R, P):
    """VecImageEnv(native_spawn=True): every episode's placement is drawn inside the library (imgenv_reset_worlds_spawn).
    The same placements, re-drawn through imgenv_spawn with the seeds the env used, go to one oracle per env."""
    import copy
    import torch
    from img_env_amd import spawn, worldgen
    from img_env_amd.vec_env import VecImageEnv
    from oracle_binding import OracleWorld, build_oracle
    from parity import compare
    build_oracle()
    grid = worldgen.make_grid(200, 3)
    cfg = worldgen.make_yaml_cfg(R, P, grid, time_max=5, n_obstacles=3, seed=9)
    vec = VecImageEnv(copy.deepcopy(cfg), env_num=E, seed=9, native_spawn=True)
    cpus = [OracleWorld(vec.params, vec.grid) for _ in range(E)]
    fields = tuple(f for f in ("is_collisions", "is_arrives", "view_maps", "sensor_maps", "vector_states", "lasers", "ped_maps",
                               "ped_vector_states", "rewards", "dones", "dones_info", "robot_pose"))

    def check(where):
        snap = vec.world.snapshot()
        for k, c in enumerate(cpus):
            mine = {f: snap[f][k * R:(k + 1) * R] for f in fields}
            bad = compare(mine, c.snapshot(), fields)
            assert not bad, (where, k, bad)

    try:
        seed0, n_eps = vec._spawn_seed, 0
        vec.reset()
        for k in range(E):
            cpus[k].reset(spawn.native_spawn(cfg, seed0 + n_eps + k))
        n_eps += E
        check("reset")
        rng = np.random.default_rng(2)
        resets = 0
        for s in range(20):
            a = np.zeros((E * R, 3), np.float32)
            a[:, 0], a[:, 1] = rng.uniform(0, 0.6, E * R), rng.uniform(-0.9, 0.9, E * R)
            _, rew, done, info = vec.step(torch.as_tensor(a, device="cuda"))
            rew, done, dinfo = rew.cpu().numpy(), done.cpu().numpy(), info["dones_info"].cpu().numpy()
            down = info["all_down"].cpu().numpy()
            assert [k for k in range(E) if down[k * R:(k + 1) * R].all()] == list(info["reset_envs"]), s
            assert all(down[k * R:(k + 1) * R].all() or not down[k * R:(k + 1) * R].any() for k in range(E)), s
            for k, c in enumerate(cpus):
                c.step(a[k * R:(k + 1) * R])
                ref = c.snapshot()  # what the step itself returned, also for the envs the library has already reset
                assert np.array_equal(rew[k * R:(k + 1) * R], ref["rewards"]), (s, k)
                assert np.array_equal(done[k * R:(k + 1) * R], ref["dones"]), (s, k)
                assert np.array_equal(dinfo[k * R:(k + 1) * R], ref["dones_info"]), (s, k)
            for q, k in enumerate(info["reset_envs"]):  # the envs that ended: the same placements for their oracles
                cpus[k].reset(spawn.native_spawn(cfg, seed0 + n_eps + q))
            n_eps += len(info["reset_envs"])
            resets += len(info["reset_envs"])
            check(s)
            if s == 2:  # put env 1 out of phase with the others
                vec.reset_envs([1])
                cpus[1].reset(spawn.native_spawn(cfg, seed0 + n_eps))
                n_eps += 1
                check("manual reset")
        assert resets >= 2 * E
    finally:
        vec.close()
        for c in cpus:
            c.close()


@pytest.mark.parametrize("E,R,P,view", [(5, 3, 4, 48), (6, 1, 0, 48), (3, 2, 1, 48), (70, 2, 3, 48)])
def test_vec_env_with_device_side_reset_matches_oracles_fed_the_same_placements(E, R, P, view):
    """VecImageEnv(device_reset=True): imgenv_step_autoreset_device finds the finished envs, draws their placements (pool filled
    ahead on a side stream, csrc/spawn_device.h), builds their RVO obstacle trees and resets them without the host in the loop.
    One oracle per env, fed the placement each world really received (imgenv_world_placement), checks every step; the
    placements keep the reference's spawn rules; their numbers follow the documented order (seed0 + k, ascending world)."""
    import copy
    import torch
    from img_env_amd import spawn, worldgen
    from img_env_amd.vec_env import VecImageEnv
    from oracle_binding import OracleWorld, build_oracle
    from parity import compare
    build_oracle()
    grid = worldgen.make_grid(200, 3)
    n_obs = 3
    cfg = worldgen.make_yaml_cfg(R, P, grid, time_max=5, n_obstacles=n_obs, seed=9)
    vec = VecImageEnv(copy.deepcopy(cfg), env_num=E, seed=9, device_reset=True)
    cpus = [OracleWorld(vec.params, vec.grid) for _ in range(E)]
    fields = tuple(f for f in ("is_collisions", "is_arrives", "view_maps", "sensor_maps", "vector_states", "lasers", "ped_maps",
                               "ped_vector_states", "rewards", "dones", "dones_info", "robot_pose"))

    def check(where):
        snap = vec.world.snapshot()
        for k, c in enumerate(cpus):
            mine = {f: snap[f][k * R:(k + 1) * R] for f in fields}
            bad = compare(mine, c.snapshot(), fields)
            assert not bad, (where, k, bad)

    try:
        seed0 = vec._spawn_seed
        vec.reset()  # the first episodes: placed by the host-side library spawn (imgenv_reset_worlds_spawn)
        for k in range(E):
            cpus[k].reset(spawn.native_spawn(cfg, seed0 + k))
        check("reset")
        rng = np.random.default_rng(2)
        resets, expect_serial = 0, 0
        for s in range(16):
            a = np.zeros((E * R, 3), np.float32)
            a[:, 0], a[:, 1] = rng.uniform(0, 0.6, E * R), rng.uniform(-0.9, 0.9, E * R)
            _, rew, done, info = vec.step(torch.as_tensor(a, device="cuda"))
            assert info["reset_envs"] is None
            worlds, first = vec.world.autoreset_last()
            rew, done, dinfo = rew.cpu().numpy(), done.cpu().numpy(), info["dones_info"].cpu().numpy()
            down = info["all_down"].cpu().numpy()
            assert [k for k in range(E) if down[k * R:(k + 1) * R].all()] == worlds, s
            assert first == expect_serial, s
            for k, c in enumerate(cpus):
                c.step(a[k * R:(k + 1) * R])
                ref = c.snapshot()  # what the step itself returned, also for the envs the library has already reset
                assert np.array_equal(rew[k * R:(k + 1) * R], ref["rewards"]), (s, k)
                assert np.array_equal(done[k * R:(k + 1) * R], ref["dones"]), (s, k)
                assert np.array_equal(dinfo[k * R:(k + 1) * R], ref["dones_info"]), (s, k)
            for q, k in enumerate(worlds):  # the envs that ended: their oracles get the placements the device drew
                lay, serial = vec.world.world_placement(k, n_obs)
                assert serial == first + q, (s, k)
                lay.ignore_obstacle = bool(cfg["ped_sim"].get("ignore_obstacle", False))
                cpus[k].reset(lay)
                # the reference's placement rules (reset_helper.py:35-55, 289): starts apart and clear of the obstacles, goals away
                starts = np.vstack([lay.robot_pose[:, :2], lay.ped_pose[:, :2]])
                if len(starts) > 1:
                    d = np.linalg.norm(starts[:, None] - starts[None], axis=2) + 10 * np.eye(len(starts))
                    assert d.min() > 1.0 - 1e-9
                assert (np.linalg.norm(lay.robot_goal - lay.robot_pose[:, :2], axis=1) > float(cfg["target_min_dist"]) - 1e-9).all()
            expect_serial += len(worlds)
            resets += len(worlds)
            check(s)
            if s == 2 and E > 1:  # a reset by the HOST in between (its obstacles are news to the device-side map restore)
                ep = vec._episodes
                vec.reset_envs([1])
                cpus[1].reset(spawn.native_spawn(cfg, seed0 + ep))
                check("manual reset")
        assert resets >= 2 * E
    finally:
        vec.close()
        for c in cpus:
            c.close()


def test_device_side_placements_follow_the_host_sampler():
    """the device's sampler (same rules, same xoshiro stream, the device's libm) against the library's host sampler, placement
    by placement: equal up to the last bits wherever no rejection test was that close -- nearly always"""
    import copy
    import torch
    from img_env_amd import spawn, worldgen
    from img_env_amd.vec_env import VecImageEnv
    grid = worldgen.make_grid(200, 3)
    E, R, P, n_obs = 24, 2, 3, 3
    cfg = worldgen.make_yaml_cfg(R, P, grid, time_max=3, n_obstacles=n_obs, seed=4)
    vec = VecImageEnv(copy.deepcopy(cfg), env_num=E, seed=4, device_reset=True)
    try:
        vec.reset()
        a = torch.zeros(E * R, 3, device="cuda")
        seen = same = 0
        for s in range(12):
            vec.step(a)
            worlds, first = vec.world.autoreset_last()
            for q, k in enumerate(worlds):
                lay, serial = vec.world.world_placement(k, n_obs)
                ref = spawn.native_spawn(cfg, (vec._device_seed0 + serial) & 0xFFFFFFFFFFFFFFFF)  # the device's own stream (2^63 away from the host-side resets')
                seen += 1
                same += int(np.allclose(lay.robot_pose, ref.robot_pose, atol=1e-9) and np.allclose(lay.ped_pose, ref.ped_pose, atol=1e-9) and
                            np.allclose(lay.robot_goal, ref.robot_goal, atol=1e-9) and np.allclose(lay.obs_pose, ref.obs_pose, atol=1e-9) and
                            np.array_equal(lay.ped_traj_len, ref.ped_traj_len) and np.allclose(lay.obs_size, ref.obs_size))
        assert seen >= 3 * E and same >= 0.98 * seen, (seen, same)
    finally:
        vec.close()


def test_ped_trajectory_dataset_wrapper_feeds_the_dataset_scene(tmp_path):
    """make_env with ``ped_sim.type: dataset`` and PedTrajectoryDatasetWrapper in the wrapper list: every reset hands the env the
    current world's recorded tracks (cur_ped_pos_v_datas), the pedestrians replay them step by step (img_env.cpp:361-386), the next
    world is up after repeated_time_per_env episodes, and each finished episode leaves its line in the output file"""
    import torch
    from img_env_amd import make_env, worldgen
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ped_dataset_ref.npz"))
    path = str(tmp_path / "world.csv")
    np.savetxt(path, z["csv"], delimiter=",", fmt="%.17g")
    grid = worldgen.make_grid(200, 3)
    cfg = worldgen.make_yaml_cfg(1, 4, grid, scene="dataset", time_max=6, dt=0.4, n_obstacles=0, seed=4,
                                 wrappers=["VelActionWrapper", "TimeLimitWrapper", "SensorsPaperRewardWrapper", "InfoLogWrapper",
                                           "PedTrajectoryDatasetWrapper", "NeverStopWrapper"])
    cfg.update(ped_traj_dataset=path, repeated_time_per_env=2, ped_dataset_worlds=[[0, 3], [3, 6]], offset=[12.0, 12.0, 0.3], fps=15,
               output_file=str(tmp_path / "log.txt"), max_time=100)
    env = make_env(cfg)
    try:
        wr = env
        while type(wr).__name__ != "PedTrajectoryDatasetWrapper":
            wr = wr.env
        base = wr
        while not hasattr(base, "world"):
            base = base.env
        env.reset()
        series = np.array(wr.change_world())  # the world the reset just fed
        assert np.allclose(base.world.out["ped_state"][:, :2].cpu().numpy(), series[:, 0, :2])
        episodes, k = 0, 0
        for s in range(30):
            a = torch.tensor([[0.2, 0.1]], device="cuda")
            _, _, done, info = env.step(a)
            k += 1
            if bool(info["all_down"][0]):  # NeverStopWrapper has reset the env: a new episode, maybe a new world
                episodes += 1
                k = 0
                series = np.array(wr.change_world())
                assert np.allclose(base.world.out["ped_state"][:, :2].cpu().numpy(), series[:, 0, :2])
            else:
                want = series[:, min(k - 1, series.shape[1] - 1), :]
                got = base.world.out["ped_state"].cpu().numpy()
                assert np.allclose(got[:, :2], want[:, :2]) and np.allclose(got[:, 2:], want[:, 3:5]), (s, k)
            if episodes == 3:
                break
        assert episodes == 3 and wr.cur_world == 1  # two episodes in world 0, then world 1
        lines = open(cfg["output_file"]).read().strip().splitlines()
        assert len(lines) == 3 and lines[0].startswith("0, ") and len(lines[0].split(", ")) == 13
    finally:
        env.close()
