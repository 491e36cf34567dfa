"""Generate tests/golden/python_post_*.npz: golden vectors for the PYTHON half of the step() path.

Run in the build container only (needs /root/reference):  python tests/golden/gen_python_golden.py

The reference's own, unmodified Python runs here: ``envs.make_env(cfg)`` builds ``ImageEnv`` and its
wrapper stack (VelAction, TimeLimit, SensorsPaperReward, InfoLog, MultiRobotClean); ``reset()`` and
``step()`` execute ``_step_req``, ``_get_states``, ``_draw_ped_map``, reward / done logic exactly as
shipped (envs/env/yaml_env.py:296-481, envs/wrapper/base.py:37-254).  The only thing replaced is the
ROS service behind them: its part (the C++ node) is played by this repo's CPU oracle, whose
AgentState outputs (state, laser, view_map, is_collision, is_arrive, pedinfo -- float32 on the wire)
are recorded as the fixture's INPUTS.  The fixture's EXPECTED values are what the reference Python
made of those inputs.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from img_env_amd import config, worldgen  # noqa: E402
from oracle_binding import OracleWorld  # noqa: E402
from scenarios import golden_cfg  # noqa: E402


def _field(m, k, default=0):
    """a message field as the wire would carry it: what the reference assigned, else the ROS default (0 / "" / [])"""
    return m.__dict__.get(k, default)


def capture_init_request(req):
    """InitEnvRequest exactly as ImageEnv._init_req + EnvPos.init built it (yaml_env.py:183-209, reset_helper.py:348-412):
    every field of InitEnv.srv / Env.msg / Agent.msg / SpeedLimiter.msg that reaches the step() path"""
    def limiter(m):
        return {k: _field(m, k) for k in ("has_velocity_limits", "has_acceleration_limits", "has_jerk_limits", "min_velocity",
                                          "max_velocity", "min_acceleration", "max_acceleration", "min_jerk", "max_jerk")}
    env = req.env
    out = {k: _field(req, k) for k in ("view_resolution", "view_width", "view_height", "step_hz", "state_dim", "use_laser",
                                       "range_total", "view_angle_begin", "view_angle_end", "view_min_dist", "view_max_dist",
                                       "beep_r", "ped_ca_p", "relation_ped_robo", "sleep_t", "is_show_gui")}
    out["global_resolution"] = _field(env, "global_resolution")
    out["ped_scene_type"] = _field(env, "ped_scene_type", "")
    out["robots"] = [dict(ktype=_field(a, "ktype", ""), shape=_field(a, "shape", ""), size=[float(x) for x in _field(a, "size", [])],
                          sensor_cfg=[float(x) for x in _field(a, "sensor_cfg", [])],
                          speed_limiter_v=limiter(_field(a, "speed_limiter_v", ref_import.Msg())),
                          speed_limiter_w=limiter(_field(a, "speed_limiter_w", ref_import.Msg()))) for a in env.robots]
    out["peds"] = [dict(ktype=_field(a, "ktype", ""), shape=_field(a, "shape", ""), size=[float(x) for x in _field(a, "size", [])],
                        max_speed=float(_field(a, "max_speed", 0.0))) for a in env.peds]
    return out


def agent_states(world):
    """AgentState[] as the node would serialise them (float32 fields -> python floats)"""
    o = world.out
    pi = world.pedinfo()
    msgs = []
    for l in range(world.n_local):
        m = ref_import.Msg()
        m.state = tuple(float(x) for x in o["vector_states"][l])
        m.laser = tuple(float(x) for x in o["lasers_raw"][l]) if world.cfg.use_laser else ()
        m.view_map = o["view_maps"][l].copy()
        m.is_collision = int(o["is_collisions"][l])
        m.is_arrive = bool(o["is_arrives"][l])
        m.pedinfo = [ref_import.Msg(px=float(a[0]), py=float(a[1]), vx=float(a[2]), vy=float(a[3]), r_=float(a[4]))
                     for a in pi[l]]
        msgs.append(m)
    return msgs


def run(name, n_robots, n_peds, steps, seed, time_max, ped_shape="circle", state_dim=3, n_obstacles=2,
        near_goals=False, cfg_over=None):
    envs = ref_import.import_reference_envs()
    grid = worldgen.make_grid(200, seed)
    meta = dict(n_robots=n_robots, n_peds=n_peds, steps=steps, seed=seed, time_max=time_max, ped_shape=ped_shape,
                state_dim=state_dim, n_obstacles=n_obstacles, near_goals=near_goals)
    if cfg_over:
        meta["cfg_over"] = cfg_over
    cfg = golden_cfg(meta, grid)
    world = OracleWorld(config.params_from_cfg(cfg), grid)
    layout = worldgen.make_layout(grid, 0.125, n_robots, n_peds, seed=seed + 100, n_obstacles=n_obstacles)
    if near_goals:  # goals 0.9 m ahead of each robot so that arrivals (and the +500 reward) occur
        yaw = 2.0 * np.arctan2(layout.robot_pose[:, 2], layout.robot_pose[:, 3])
        layout.robot_goal = layout.robot_pose[:, :2] + 0.9 * np.stack([np.cos(yaw), np.sin(yaw)], 1)
    rec_in, rec_alive, rec_init = [], [], []

    def record_inputs():
        o = world.out
        rec_in.append(dict(vector_states=o["vector_states"].copy(), lasers_raw=o["lasers_raw"].copy(),
                           view_maps=o["view_maps"].copy(), is_collisions=o["is_collisions"].copy(),
                           is_arrives=o["is_arrives"].copy(), pedinfo=world.pedinfo()))

    def init_srv(req):
        rec_init.append(capture_init_request(req))
        return ref_import.Msg()

    def reset_srv(req):
        world.reset(layout)
        record_inputs()
        return ref_import.Msg(robot_states=agent_states(world))

    def step_srv(req):
        # Agent.msg v / w / v_y are float32 on the wire
        a = np.array([[np.float32(r.v), np.float32(r.w), np.float32(r.v_y)] for r in req.robots], np.float32)
        rec_alive.append(np.array([bool(r.alive) for r in req.robots]))
        world.step(a)
        record_inputs()
        return ref_import.Msg(robot_states=agent_states(world))

    ref_import.SERVICES.update(init_image_env=init_srv, reset_image_env=reset_srv, step_image_env=step_srv)
    ref_cfg = dict(cfg)
    ref_cfg["global_map"] = dict(cfg["global_map"], map_array=None)
    env = envs.make_env(ref_cfg)
    keys = ("vector_states", "sensor_maps", "is_collisions", "is_arrives", "lasers", "ped_vector_states", "ped_maps",
            "step_ds", "ped_min_dists")
    exp = {k: [] for k in keys}
    for k in ("rewards", "dones", "dones_info", "is_clean", "all_down", "speeds"):
        exp[k] = []
    ora = {k: [] for k in ("rewards", "dones", "dones_info", "is_clean", "step_ds", "ped_min_dists", "ped_maps",
                           "ped_vector_states", "lasers", "sensor_maps", "base_rewards", "base_dones")}

    def grab_state(st):
        for k in keys:
            exp[k].append(np.array(getattr(st, k)))

    def grab_oracle():
        for k in ora:
            ora[k].append(world.out[k].copy())

    st = env.reset()
    grab_state(st)
    grab_oracle()
    rng = np.random.default_rng(seed + 7)
    actions_all = []
    for s in range(steps):
        act = np.stack([rng.uniform(-0.1, 0.7, n_robots), rng.uniform(-1.0, 1.0, n_robots)], 1)
        if near_goals:
            act = np.stack([rng.uniform(0.2, 0.6, n_robots), rng.uniform(-0.2, 0.2, n_robots)], 1)
        actions_all.append(act)
        st, rew, done, info = env.step(act)
        grab_state(st)
        grab_oracle()
        exp["rewards"].append(np.array(rew, np.float64))
        exp["dones"].append(np.array(done))
        exp["dones_info"].append(np.array(info["dones_info"]))
        exp["is_clean"].append(np.array(info["is_clean"]))
        exp["all_down"].append(np.array(info["all_down"]))
        exp["speeds"].append(np.array(info["speeds"], np.float64))
    out = {"in_" + k: np.stack([r[k] for r in rec_in]) for k in rec_in[0]}
    out["in_alive"] = np.stack(rec_alive)
    out["actions"] = np.stack(actions_all)
    for k, v in exp.items():
        out["exp_" + k] = np.stack(v)
    # how far the oracle's own Python-half restatement is from the reference Python, for the log
    for k in ("step_ds", "ped_min_dists", "ped_maps", "ped_vector_states", "lasers", "sensor_maps"):
        a, b = np.stack(ora[k]).astype(np.float64), out["exp_" + k].astype(np.float64)
        fin = np.isfinite(a) & np.isfinite(b)
        print("  %-18s max|oracle-ref| = %.3g   (inf pattern equal: %s)" % (k, np.abs(a[fin] - b[fin]).max() if fin.any() else 0,
                                                                           np.array_equal(np.isfinite(a), np.isfinite(b))))
    for k in ("rewards", "dones", "dones_info", "is_clean"):
        a, b = np.stack(ora[k][1:]).astype(np.float64), out["exp_" + k].astype(np.float64)
        print("  %-18s max|oracle-ref| = %.3g" % (k, np.abs(a - b).max()))
    out["meta"] = np.array(repr(meta))
    out["init_req"] = np.array(json.dumps(rec_init[-1]))  # the InitEnvRequest the reference sent (test_oracle_python_golden.py)
    path = os.path.join(HERE, "python_post_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KiB) collisions=%s arrives=%s" % (path, os.path.getsize(path) / 1024,
                                                           out["in_is_collisions"][-1], out["in_is_arrives"][-1]))
    world.close()


def run_stack(name, n_robots, n_peds, steps, seed, time_max, wrappers, cfg_over, obs_names, n_layouts=3):
    """The wrappers `run` does not reach: StatePedVectorWrapper, StateBatchWrapper (frame stacks), the discrete VelActionWrapper
    table, ObsStateTmp / ObsLaserStateTmp and NeverStopWrapper (base.py:19-66, 97-150, 198-211; filter_states.py) -- the
    reference's own stack, auto-resets included (every reset service call gets the next of `n_layouts` fixed layouts)."""
    envs = ref_import.import_reference_envs()
    grid = worldgen.make_grid(200, seed)
    meta = dict(n_robots=n_robots, n_peds=n_peds, steps=steps, seed=seed, time_max=time_max, ped_shape="circle", state_dim=3,
                n_obstacles=2, near_goals=False, cfg_over=dict(cfg_over, wrapper=list(wrappers)), n_layouts=n_layouts,
                obs_names=list(obs_names))
    cfg = golden_cfg(meta, grid)
    world = OracleWorld(config.params_from_cfg(cfg), grid)
    layouts = [worldgen.make_layout(grid, 0.125, n_robots, n_peds, seed=seed + 100 + k, n_obstacles=2) for k in range(n_layouts)]
    n_resets = [0]

    def init_srv(req):
        return ref_import.Msg()

    def reset_srv(req):
        world.reset(layouts[n_resets[0] % n_layouts])
        n_resets[0] += 1
        return ref_import.Msg(robot_states=agent_states(world))

    def step_srv(req):
        a = np.array([[np.float32(r.v), np.float32(r.w), np.float32(r.v_y)] for r in req.robots], np.float32)
        world.step(a)
        return ref_import.Msg(robot_states=agent_states(world))

    ref_import.SERVICES.update(init_image_env=init_srv, reset_image_env=reset_srv, step_image_env=step_srv)
    ref_cfg = dict(cfg)
    ref_cfg["global_map"] = dict(cfg["global_map"], map_array=None)
    env = envs.make_env(ref_cfg)
    out = {}
    obs = env.reset()
    rec = {"obs%d" % k: [np.array(o)] for k, o in enumerate(obs)}
    for k in ("rewards", "dones", "dones_info", "is_clean", "all_down", "speeds", "arrive", "collision", "close"):
        rec[k] = []
    rng = np.random.default_rng(seed + 7)
    acts = []
    discrete = bool(cfg["discrete_action"])
    for s in range(steps):
        if discrete:
            act = rng.integers(0, len(cfg["discrete_actions"]), n_robots)
        else:
            act = np.stack([rng.uniform(-0.1, 0.7, n_robots), rng.uniform(-1.0, 1.0, n_robots)], 1)
        acts.append(act)
        obs, rew, done, info = env.step(act)
        for k, o in enumerate(obs):
            rec["obs%d" % k].append(np.array(o))
        rec["rewards"].append(np.array(rew, np.float64))
        rec["dones"].append(np.array(done))
        rec["dones_info"].append(np.array(info["dones_info"]))
        rec["is_clean"].append(np.array(info["is_clean"]))
        rec["all_down"].append(np.array(info["all_down"]))
        rec["speeds"].append(np.array(info["speeds"], np.float64))
        rec["arrive"].append(np.array(info["arrive"]))
        rec["collision"].append(np.array(info["collision"]))
        rec["close"].append(np.array(info.get("bool_get_close_to_human", np.zeros(n_robots))))
    for k, v in rec.items():
        out["exp_" + k] = np.stack(v)
    te = env
    while te is not None and type(te).__name__ != "TestEpisodeWrapper":
        te = getattr(te, "env", None)
    if te is not None:  # what the reference's episode statistics had counted by the end of the run
        out["te_counts"] = np.array([te.cur_episode, te.arrive_num, te.static_coll_num, te.ped_coll_num, te.other_coll_num, te.steps,
                                     te.stuck_num, te.speed_step], np.int64)
        out["te_sums"] = np.array([te.v_sum, te.w_sum], np.float64)
        out["te_arrays"] = np.array([te.w_variance_array, te.v_jerk_array, te.w_jerk_array, te.w_zero_array], np.float64)
    out["actions"] = np.stack(acts)
    out["n_resets"] = np.array(n_resets[0])
    out["meta"] = np.array(repr(meta))
    path = os.path.join(HERE, "python_stack_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KiB): %d service resets, obs shapes %s" % (path, os.path.getsize(path) / 1024, n_resets[0],
                                                                   [out["exp_obs%d" % k].shape for k in range(len(obs))]))
    world.close()


if __name__ == "__main__":
    run("a", n_robots=6, n_peds=5, steps=30, seed=3, time_max=24)
    run("b", n_robots=4, n_peds=7, steps=16, seed=11, time_max=100, ped_shape="leg", state_dim=5)
    run("c", n_robots=3, n_peds=0, steps=8, seed=5, time_max=100, state_dim=4, n_obstacles=0)
    run("d", n_robots=5, n_peds=3, steps=12, seed=21, time_max=100, near_goals=True)
    # every InitEnv field off its default: omni robots of mixed footprints with sensor offsets, both speed limiters, ERVO legs
    run("e", n_robots=4, n_peds=4, steps=10, seed=31, time_max=100, ped_shape="leg", state_dim=5, cfg_over={
        "robot_type": "omni", "relation_ped_robo": 0, "range_total": 180, "view_angle_begin": -1.2, "view_angle_end": 0.9,
        "view_min_dist": 0.25, "view_max_dist": 2.5, "control_hz": 0.4,
        "robot.shape": ["circle", "rectangle", "circle", "rectangle"],
        "robot.size": [[0, 0, 0.17], [-0.2, 0.2, -0.1, 0.1], [0.02, -0.01, 0.24], [-0.25, 0.15, -0.12, 0.12]],
        "robot.sensor_cfgs": [[0.0, 0.0], [0.08, -0.03], [0.05, 0.0], [0.0, 0.04]],
        "speed_limiter_v": {"has_velocity_limits": True, "has_acceleration_limits": True, "max_velocity": 0.5,
                            "min_acceleration": -0.8, "max_acceleration": 0.6},
        "speed_limiter_w": {"has_velocity_limits": True, "has_jerk_limits": True, "min_velocity": -0.7, "max_velocity": 0.7,
                            "min_jerk": -1.5},
        "ped_sim.type": "ervoscene", "ped_sim.max_speed": [0.4, 0.5, 0.6, 0.3]})
    table = [[0.0, -0.9], [0.0, 0.3], [0.2, -0.6], [0.2, 0.0], [0.4, 0.6], [0.6, -0.3], [0.6, 0.0, 1], [0.6, 0.9]]
    base = ["VelActionWrapper", "TimeLimitWrapper", "SensorsPaperRewardWrapper", "InfoLogWrapper", "MultiRobotCleanWrapper"]
    # frame stacks of 2 images / 3 vector states / 2 laser scans, discrete action table, laser observation, auto-reset by time limit
    run_stack("a", n_robots=3, n_peds=4, steps=20, seed=41, time_max=6,
              wrappers=base + ["StatePedVectorWrapper", "StateBatchWrapper", "ObsLaserStateTmp", "NeverStopWrapper"],
              cfg_over=dict(discrete_action=True, discrete_actions=table, image_batch=2, state_batch=3, laser_batch=2),
              obs_names=["lasers", "vector_states", "ped_maps"])
    # image observation, continuous actions, single frames, laser_batch 0 (a stack of one)
    run_stack("b", n_robots=2, n_peds=3, steps=14, seed=43, time_max=5,
              wrappers=base + ["StateBatchWrapper", "ObsStateTmp", "NeverStopWrapper"],
              cfg_over=dict(image_batch=1, state_batch=1, laser_batch=0), obs_names=["sensor_maps", "vector_states", "ped_maps"])
    # the wrapper list of the shipped test.yaml (one robot): TestEpisodeWrapper's statistics over a handful of short episodes
    run_stack("c", n_robots=1, n_peds=3, steps=44, seed=45, time_max=7,
              wrappers=base + ["TestEpisodeWrapper", "StateBatchWrapper", "ObsLaserStateTmp", "NeverStopWrapper"],
              cfg_over=dict(discrete_action=True, discrete_actions=table, image_batch=1, state_batch=3, laser_batch=0,
                            init_pose_bag_episodes=100), obs_names=["lasers", "vector_states", "ped_maps"], n_layouts=4)
