"""``VecImageEnv``: ``env_num`` copies of one reference YAML env in ONE library handle.

The reference trains on ``env_num`` env processes (create_launch.py:57-66 starts one C++ node per env, the trainer calls
``make_env(cfg)`` once per env and steps them side by side), each wrapped in the stack
``VelActionWrapper -> TimeLimitWrapper -> SensorsPaperRewardWrapper -> InfoLogWrapper -> MultiRobotCleanWrapper ->
NeverStopWrapper`` (envs/cfg/*.yaml ``wrapper:``).  Here all envs are worlds of one ``World`` (``imgenv_cfg.n_worlds``),
stepped by one set of kernel launches; the library's fused outputs ARE that wrapper stack (``rewards``, ``dones``,
``dones_info``, ``is_clean``), and the envs whose robots are all done are reset together in one
``imgenv_reset_worlds`` call (NeverStopWrapper, base.py:198-211), each from its own ``EnvPos`` spawn stream.

Robots are numbered env-major: env k owns rows ``[k * robot_total, (k + 1) * robot_total)`` of every tensor.
"""
import numpy as np

from . import _cabi, config, spawn
from .envs import ContinuousAction, ImageState

_PER_AGENT = ("robot_shape", "robot_size", "robot_sensor_cfg", "robot_size_last", "ped_shape", "ped_size", "ped_max_speed")


def stack_params(params, env_num):
    """the parameter dict of ``env_num`` copies of one world (per-robot / per-pedestrian rows repeat, env-major)"""
    p = dict(params)
    for k in _PER_AGENT:
        p[k] = np.concatenate([np.asarray(params[k])] * env_num, axis=0)
    p["n_robots"], p["n_peds"], p["n_worlds"] = params["n_robots"] * env_num, params["n_peds"] * env_num, env_num
    return p


class VecImageEnv:
    """``reset() -> ImageState``; ``step(actions) -> (ImageState, rewards, dones, info)`` over ``env_num * robot_total`` robots.

    ``info``: ``dones_info`` (InfoLogWrapper codes, 10 = time limit), ``is_clean`` (MultiRobotCleanWrapper), ``arrive``,
    ``collision``, ``all_down`` (per robot: its env is finished) and ``reset_envs`` (the envs that were reset after this step:
    their rows of the returned state are already the new episode's first observation, as with NeverStopWrapper).
    """

    def __init__(self, cfg, env_num=None, seed=None, auto_reset=True, native_spawn=False, device_reset=False):
        import torch
        from .world import World
        self.cfg = cfg
        self.env_num = int(env_num if env_num is not None else cfg.get("env_num", 1))
        self.params = config.params_from_cfg(cfg)
        self.grid = config.load_map(cfg)
        self.robot_total = self.params["n_robots"]  # per env, as in the reference
        self.ped_total = self.params["n_peds"]
        self.auto_reset = auto_reset
        seed = cfg.get("seed") if seed is None else seed
        self.env_poses = [spawn.EnvPos(cfg, seed=None if seed is None else seed + k) for k in range(self.env_num)]
        # native_spawn: the placements of an episode are drawn inside the library (csrc/spawn_host.h: EnvPos' rules, its own
        # random stream) -- the Python EnvPos costs 40-170 us per small env, twenty times the device's whole step
        # device_reset: NeverStopWrapper without the host in the loop (imgenv_step_autoreset_device): finished envs are found,
        # placed and reset by kernels alone, ``step`` returns without synchronising and ``info["reset_envs"]`` is None --
        # ``info["all_down"]`` (a device tensor, per robot) says which envs started over
        self.device_reset = bool(device_reset)
        self.native_spawn = bool(native_spawn) or self.device_reset
        native_spawn = self.native_spawn
        self._spawn_cfg = spawn.make_spawn_cfg(cfg) if native_spawn else None
        self._spawn_seed = (0x9E3779B97F4A7C15 * (1 + (seed or 0))) & 0xFFFFFFFFFFFFFFFF
        self._episodes = 0
        # device-side resets number their placements seed0 + k on their own: a stream 2^63 away from the host-side resets'
        # (_spawn_seed + episodes so far), so that an env reset by the host never replays an episode the device handed out
        self._device_seed0 = (self._spawn_seed + (1 << 63)) & 0xFFFFFFFFFFFFFFFF
        self._extent = max(self.grid.shape) * float(cfg["global_map"]["resolution"])
        if not cfg.get("keep_view_maps", False):
            # ImageState has no full-size view: where the view is shrunk into the sensor_map (the shipped 400 x 400 -> 48 x 48)
            # the library then only evaluates the view cells the shrink reads (IMGENV_FLAG_NO_VIEW_MAPS)
            self.params["flags"] = int(self.params.get("flags", 0)) | _cabi.FLAG_NO_VIEW_MAPS
        self.world = World(stack_params(self.params, self.env_num), self.grid, device=cfg.get("device", 0))
        self._all_down = self.world.out["step_all_down"].view(torch.bool) if native_spawn else None

    def __len__(self):
        return self.env_num * self.robot_total

    def _state(self):
        o = self.world.out
        return ImageState(o["vector_states"], o["sensor_maps"], o["is_collisions"], o["is_arrives"], o["lasers"],
                          o["ped_vector_states"], o["ped_maps"], o["step_ds"], o["ped_min_dists"])

    def reset(self, layouts=None):
        """every env starts a new episode (ImageEnv.reset per env, yaml_env.py:296-317)"""
        if layouts is None and self.native_spawn:
            return self.reset_envs(range(self.env_num))
        if layouts is None:
            layouts = [ep.reset(self._extent) for ep in self.env_poses]
        self.world.reset(list(layouts))
        return self._state()

    def reset_envs(self, envs, layouts=None):
        envs = [int(k) for k in envs]
        if layouts is None and self.native_spawn:
            seeds = [self._spawn_seed + self._episodes + q for q in range(len(envs))]
            self._episodes += len(envs)
            self.world.reset_worlds_spawn(envs, self._spawn_cfg, seeds)
            return self._state()
        if layouts is None:
            layouts = [self.env_poses[k].reset(self._extent) for k in envs]
        self.world.reset_worlds(envs, layouts)
        return self._state()

    def _actions(self, actions):
        if isinstance(actions, (list, tuple)) and len(actions) and isinstance(actions[0], ContinuousAction):
            actions = np.array([[a.v, a.w, a.beep] for a in actions], np.float32)  # float32 wire (Agent.msg:8-10)
        return actions

    def _step_native(self, actions):
        """step + NeverStopWrapper inside the library (``imgenv_step_autoreset``): the finished envs are found on the device,
        their placements drawn on the host in C++, the per-robot results of the step itself kept in ``out['step_*']`` --
        no tensor work on the Python side"""
        o, finished = self.world.step_autoreset(self._actions(actions), self._spawn_cfg, self._spawn_seed + self._episodes)
        self._episodes += len(finished)
        info = {"dones_info": o["step_dones_info"], "is_clean": o["step_is_clean"], "arrive": o["step_is_arrives"],
                "collision": o["step_is_collisions"], "all_down": self._all_down, "reset_envs": finished}
        return self._state(), o["step_rewards"], o["step_dones"], info

    def _step_device(self, actions):
        o = self.world.step_autoreset_device(self._actions(actions), self._spawn_cfg, self._device_seed0)
        info = {"dones_info": o["step_dones_info"], "is_clean": o["step_is_clean"], "arrive": o["step_is_arrives"],
                "collision": o["step_is_collisions"], "all_down": self._all_down, "reset_envs": None}
        return self._state(), o["step_rewards"], o["step_dones"], info

    def step(self, actions):
        import torch
        if self.device_reset and self.auto_reset:
            return self._step_device(actions)
        if self.native_spawn and self.auto_reset:
            return self._step_native(actions)
        o = self.world.step(self._actions(actions))
        E, R = self.env_num, self.robot_total
        all_down = (o["dones"].view(E, R) > 0).all(dim=1)
        info = {"dones_info": o["dones_info"], "is_clean": o["is_clean"], "arrive": o["is_arrives"],
                "collision": o["is_collisions"], "all_down": all_down.repeat_interleave(R), "reset_envs": []}
        rewards, dones = o["rewards"], o["dones"]
        if self.auto_reset:
            finished = torch.nonzero(all_down).flatten().tolist()  # the one host read per step (NeverStopWrapper reads it too)
            if finished:
                # the reset rewrites the finished envs' rows in place: hand out this step's values, not the new episode's
                rewards, dones = rewards.clone(), dones.clone()
                info["dones_info"], info["is_clean"] = o["dones_info"].clone(), o["is_clean"].clone()
                info["arrive"], info["collision"] = o["is_arrives"].clone(), o["is_collisions"].clone()
                self.reset_envs(finished)
                info["reset_envs"] = finished
        return self._state(), rewards, dones, info

    def end_ep(self, robot_res=None):
        return True

    def close(self):
        self.world.close()
