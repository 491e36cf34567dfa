"""Host-side mirror of the reference's Gym surface for the step() path, over the HIP library.

Same names, argument meaning and return layout as the reference:

  reference                                         here
  envs/__init__.py:21-33   make_env(cfg)            make_env(cfg)
  envs/env/yaml_env.py     ImageEnv                 ImageEnv   (.robot_total, .reset(**kw), .step(actions), .end_ep())
  envs/state/state.py      ImageState               ImageState (same 9 fields; values are device tensors)
  envs/action/action.py    ContinuousAction, ...    ContinuousAction, DiscreteActions
  envs/wrapper/base.py     *Wrapper(env, cfg)       same names, registered in wrapper_dict

Every array the reference builds with numpy / Python loops (`_get_states`, `_draw_ped_map`, rewards, dones,
infos) is produced by the HIP kernels; the wrappers only select the tensors that belong to their place in
the stack.  Tensors are views of library-owned HBM that the next step overwrites (clone() to keep one).
"""
import math
from collections import deque

import numpy as np

from . import _cabi, config, spawn
from .config import read_yaml  # noqa: F401  (re-export, envs/__init__.py:9-18)


class Action:
    pass


class ContinuousAction(Action):
    """envs/action/action.py:8-20"""

    def __init__(self, v, w, beep=0):
        self.v, self.w, self.beep = v, w, beep

    def reverse(self):
        return [self.v, self.w, self.beep]


class DiscreteActions:
    """envs/action/action.py:23-38"""

    def __init__(self, actions):
        self.actions = []
        for a in actions:
            assert a[0] >= 0 and len(a) in (2, 3)
            self.actions.append(ContinuousAction(a[0], a[1], 0) if len(a) == 2 else ContinuousAction(*a))

    def __len__(self):
        return len(self.actions)

    def __getitem__(self, i):
        return self.actions[i]


class ImageState:
    """envs/state/state.py:4-28 -- same constructor order and attributes"""

    def __init__(self, vector_states, sensor_maps, is_collisions, is_arrives, lasers, ped_vector_states, ped_maps,
                 step_ds, ped_min_dists):
        assert len(vector_states) == len(sensor_maps) == len(is_collisions) == len(is_arrives) == len(lasers) \
            == len(ped_vector_states) == len(ped_maps) == len(step_ds) == len(ped_min_dists)
        self.vector_states = vector_states
        self.sensor_maps = sensor_maps
        self.is_collisions = is_collisions
        self.is_arrives = is_arrives
        self.lasers = lasers
        self.ped_vector_states = ped_vector_states
        self.ped_maps = ped_maps
        self.ped_min_dists = ped_min_dists
        self.step_ds = step_ds

    def __len__(self):
        return len(self.vector_states)

    def numpy(self):
        """host copy with the reference's dtypes (yaml_env.py:472-481)"""
        c = lambda t: t.detach().cpu().numpy()  # noqa: E731
        return ImageState(c(self.vector_states).astype(np.float64), c(self.sensor_maps), c(self.is_collisions).astype(np.int64),
                          c(self.is_arrives).astype(bool), c(self.lasers), c(self.ped_vector_states), c(self.ped_maps),
                          c(self.step_ds), c(self.ped_min_dists))

    def __str__(self):
        return "Image State Info:\n" + "\n".join("        %s: %s" % (k, getattr(self, k)) for k in (
            "vector_states", "sensor_maps", "is_collisions", "is_arrives", "lasers", "ped_vector_states", "ped_maps",
            "ped_min_dists", "step_ds"))


class Env:
    """the slice of gym.Env the reference relies on"""
    metadata = {}

    def reset(self, **kwargs):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError


class Wrapper(Env):
    """gym.Wrapper delegation"""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)


class ObservationWrapper(Wrapper):
    def reset(self, **kwargs):
        return self.observation(self.env.reset(**kwargs))

    def step(self, action):
        o, r, d, i = self.env.step(action)
        return self.observation(o), r, d, i


class ImageEnv(Env):
    """envs/env/yaml_env.py:51-481 over the HIP library.

    ``step`` accepts the reference's ``List[ContinuousAction]`` or, on the fast path, a ``[R, 3]`` float32
    tensor / array of (v, w, beep).  It returns ``(ImageState, rewards, dones, {'dones_info': zeros})`` with
    ``rewards = is_arrives - is_collisions`` and ``dones`` as yaml_env.py:372-377.
    """

    def __init__(self, cfg):
        from .world import World
        self.cfg = cfg
        self.test = cfg.get("test", False)
        self.env_name = cfg.get("env_name", "")
        self.params = config.params_from_cfg(cfg)
        self.grid = config.load_map(cfg)
        self.robot_total = self.params["n_robots"]
        self.ped_total = self.params["n_peds"]
        self.control_hz = cfg["control_hz"]
        self.laser_max = cfg["laser_max"]
        self.image_size = tuple(cfg["image_size"])
        self.ped_image_size = tuple(cfg["ped_image_size"])
        self.world = World(self.params, self.grid, device=cfg.get("device", 0))
        self.env_pose = spawn.EnvPos(cfg, seed=cfg.get("seed"))
        self._extent = max(self.grid.shape) * float(cfg["global_map"]["resolution"])
        self.dones = None
        self._zeros_info = None

    def _state(self):
        o = self.world.out
        return ImageState(o["vector_states"], o["sensor_maps"], o["is_collisions"], o["is_arrives"], o["lasers"],
                          o["ped_vector_states"], o["ped_maps"], o["step_ds"], o["ped_min_dists"])

    def reset(self, layout=None, **kwargs):
        """yaml_env.py:296-317.  ``layout`` (a worldgen.ResetLayout) overrides the random spawn; other keyword arguments
        (NeverStopWrapper passes the last step's whole info dict) are ignored as in the reference, except
        ``cur_ped_pos_v_datas``."""
        if layout is None:
            layout = self.env_pose.reset(self._extent)
        data = kwargs.get("cur_ped_pos_v_datas")
        if data is not None:  # EnvPos.init_ped_dataset (reset_helper.py:417-432, yaml_env.py:246-247)
            spawn.init_ped_dataset(layout, np.asarray(data, np.float64))
        self.world.reset(layout)
        self.dones = self.world.out["base_dones"]
        return self._state()

    def _actions(self, actions):
        if isinstance(actions, (list, tuple)) and len(actions) and isinstance(actions[0], ContinuousAction):
            actions = np.array([[a.v, a.w, a.beep] for a in actions], np.float32)  # float32 wire (Agent.msg:8-10)
        return actions

    def step(self, actions):
        """yaml_env.py:360-377"""
        import torch
        self.world.step(self._actions(actions))
        o = self.world.out
        self.dones = o["base_dones"]
        if self._zeros_info is None:
            self._zeros_info = torch.zeros_like(o["dones_info"])
        # copies, like the reference's `deepcopy(self.dones)`: the library's buffers are rewritten in place by the next
        # step -- or by the reset NeverStopWrapper issues inside this very step() call
        return self._state(), o["base_rewards"].clone(), o["base_dones"].clone(), {"dones_info": self._zeros_info}

    def end_ep(self, robot_res=None):
        """yaml_env.py:379-390: episode recording is out of scope; kept for API compatibility"""
        return True

    def close(self):
        self.world.close()


# ------------------------------------------------------------------------------------------ wrappers
class VelActionWrapper(Wrapper):
    """envs/wrapper/base.py:37-66"""

    def __init__(self, env, cfg):
        super().__init__(env)
        self.discrete = bool(cfg["discrete_action"])
        if self.discrete:
            self.actions = DiscreteActions(cfg["discrete_actions"])
            self.table = np.array([a.reverse() for a in self.actions.actions], np.float32)
        else:
            self.clip = np.array(cfg["continuous_actions"], np.float32)

    def action(self, actions):
        import torch
        if isinstance(actions, torch.Tensor):
            a = actions
            if self.discrete and a.dim() == 1:
                return torch.as_tensor(self.table, device=a.device)[a.long()]
            out = torch.zeros(a.shape[0], 3, device=a.device, dtype=torch.float32)
            n = min(a.shape[1], len(self.clip)) if not self.discrete else a.shape[1]
            lo = torch.as_tensor(self.clip[:n, 0], device=a.device)
            hi = torch.as_tensor(self.clip[:n, 1], device=a.device)
            out[:, :n] = torch.minimum(torch.maximum(a[:, :n].float(), lo), hi)
            return out
        a = np.asarray(actions)
        if self.discrete and a.ndim == 1:
            return self.table[a.astype(np.int64)]
        out = np.zeros((a.shape[0], 3), np.float32)
        n = len(self.clip)
        out[:, :n] = np.clip(a[:, :n], self.clip[:, 0], self.clip[:, 1])
        return out

    def step(self, action):
        import torch
        a = self.action(action)
        if not isinstance(a, torch.Tensor):
            a = torch.as_tensor(np.ascontiguousarray(a, np.float32), device=self.world.device)
        state, reward, done, info = self.env.step(a)
        info["speeds"] = a[:, :2]
        return state, reward, done, info

    def reverse_action(self, actions):
        return actions


class TimeLimitWrapper(Wrapper):
    """envs/wrapper/base.py:215-231"""

    def __init__(self, env, cfg):
        super().__init__(env)
        self._max_episode_steps = cfg["time_max"]
        self._elapsed_steps = 0

    def step(self, ac):
        import torch
        state, reward, done, info = self.env.step(ac)
        self._elapsed_steps += 1
        if self._elapsed_steps > self._max_episode_steps:  # the same test k_obs applies to out["dones"]
            done = torch.ones_like(done)
            info["dones_info"] = torch.full_like(info["dones_info"], 10)
        return state, reward, done, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


class SensorsPaperRewardWrapper(Wrapper):
    """envs/wrapper/base.py:153-195 (reward computed by k_obs)"""

    def __init__(self, env, cfg):
        super().__init__(env)
        self.ped_safety_space = cfg["ped_safety_space"]

    def step(self, action):
        state, reward, done, info = self.env.step(action)
        return state, self.world.out["paper_rewards"].clone(), done, info


class InfoLogWrapper(Wrapper):
    """envs/wrapper/base.py:234-254"""

    def __init__(self, env, cfg):
        super().__init__(env)
        self.robot_total = cfg["robot"]["total"]
        self.ped = cfg["ped_sim"]["total"] > 0 and cfg["env_type"] == "robot_nav"

    def step(self, action):
        import torch
        state, reward, done, info = self.env.step(action)
        info["arrive"] = state.is_arrives.clone()      # (the state tensors are views of buffers a reset rewrites in place)
        info["collision"] = state.is_collisions.clone()
        di = info["dones_info"]
        di = torch.where(state.is_collisions > 0, state.is_collisions.to(di.dtype), di)
        di = torch.where(state.is_arrives == 1, torch.full_like(di, 5), di)
        info["dones_info"] = di
        info["all_down"] = ((done > 0).sum() == len(done)).expand(len(done))
        if self.ped:
            info["bool_get_close_to_human"] = (state.ped_min_dists < 1).to(torch.int64)
        return state, reward, done, info


class MultiRobotCleanWrapper(Wrapper):
    """envs/wrapper/base.py:69-93"""

    def __init__(self, env, cfg):
        super().__init__(env)
        self.is_clean = None

    def step(self, action):
        import torch
        state, reward, done, info = self.env.step(action)
        if self.is_clean is None:
            self.is_clean = torch.ones_like(done, dtype=torch.bool)
        info["is_clean"] = self.is_clean.clone()
        reward = torch.where(self.is_clean, reward, torch.zeros_like(reward))
        if "speeds" in info:
            info["speeds"] = info["speeds"] * self.is_clean.unsqueeze(1).to(info["speeds"].dtype)
        self.is_clean = torch.where(done > 0, torch.zeros_like(self.is_clean), self.is_clean)
        return state, reward, done, info

    def reset(self, **kwargs):
        state = self.env.reset(**kwargs)
        self.is_clean = None
        return state


class StateBatchWrapper(Wrapper):
    """envs/wrapper/base.py:97-150: frame stacking, on the device"""

    def __init__(self, env, cfg):
        super().__init__(env)
        self.q = {
            "sensor_maps": deque([], maxlen=cfg["image_batch"]) if cfg["image_batch"] > 0 else None,
            "vector_states": deque([], maxlen=cfg["state_batch"]) if cfg["state_batch"] > 0 else None,
            "lasers": deque([], maxlen=max(cfg["laser_batch"], 1)) if cfg["laser_batch"] >= 0 else None,
        }

    def _concate(self, name, t):
        import torch
        q = self.q[name]
        if q is None:
            return t
        t = t.unsqueeze(1).clone()
        while len(q) < q.maxlen:
            q.append(torch.zeros_like(t))
        q.append(t)
        return torch.cat(list(q), dim=1)

    def batch_state(self, state):
        state.sensor_maps = self._concate("sensor_maps", state.sensor_maps)
        v = self._concate("vector_states", state.vector_states)
        state.vector_states = v.reshape(v.shape[0], -1) if v.dim() == 3 else v
        state.lasers = self._concate("lasers", state.lasers)
        return state

    def step(self, action):
        state, reward, done, info = self.env.step(action)
        return self.batch_state(state), reward, done, info

    def reset(self, **kwargs):
        for q in self.q.values():
            if q is not None:
                q.clear()
        return self.batch_state(self.env.reset(**kwargs))


class NeverStopWrapper(Wrapper):
    """envs/wrapper/base.py:198-211: reset when every robot is done (reads one flag from the device)"""

    def __init__(self, env, cfg):
        super().__init__(env)

    def step(self, action):
        states, reward, done, info = self.env.step(action)
        if bool(info["all_down"][0]):
            states = self.env.reset(**info)  # the reference hands the whole info dict down (TestEpisodeWrapper reads dones_info)
        return states, reward, done, info


class TrajectoryPathHelper:
    """envs/wrapper/evaluation_wrapper/utils.py:5-129: path statistics of one episode's (v, w) commands"""

    def __init__(self, dt):
        self.dt = dt
        self.v_array, self.w_array = [], []
        self.w_zero = self.v_jerk = self.w_jerk = self.w_variance = self.w_avg = self.v_avg = self.w_acc = self.v_acc = 0

    def add_vw(self, v, w):
        self.v_array.append(v)
        self.w_array.append(w)

    def reset(self):
        """cal_w_variance, cal_w_zero, cal_jerk, cal_v_avg, cal_w_avg (utils.py:60-101, 123-129)"""
        self.w_variance = np.var(self.w_array)
        tmp = w_zero = 0
        for w in self.w_array:  # sign changes of w, a zero in between counts once
            if w == 0:
                if tmp != 0:
                    w_zero += 1
            elif (w > 0 and tmp < 0) or (w < 0 and tmp > 0):
                w_zero += 1
            tmp = w
        self.w_zero = w_zero
        v_acc = np.diff(self.v_array, axis=0) / self.dt
        self.v_jerk = np.average(np.abs(np.diff(v_acc, axis=0) / self.dt))
        self.v_acc = np.average(np.abs(v_acc))
        w_acc = np.diff(self.w_array, axis=0) / self.dt
        self.w_jerk = np.average(np.abs(np.diff(w_acc, axis=0) / self.dt))
        self.w_acc = np.average(np.abs(w_acc))
        self.v_avg = np.average(self.v_array)
        self.w_avg = np.average(np.abs(self.w_array))

    def clear_vw_array(self):
        self.v_array.clear()
        self.w_array.clear()
        self.w_zero = self.w_variance = self.w_avg = self.v_avg = self.w_jerk = self.v_jerk = self.v_acc = self.w_acc = 0


class TestEpisodeWrapper(Wrapper):
    """envs/wrapper/evaluation_wrapper/TestEpisodeWrapper.py:8-119: episode statistics of ONE robot (the first), printed after
    ``init_pose_bag_episodes`` episodes -- upon which the reference exits the process, and so does this."""
    __test__ = False  # (not a pytest class)

    def __init__(self, env, cfg):
        super().__init__(env)
        self.cur_episode = 0
        self.max_episodes = cfg["init_pose_bag_episodes"]
        self.dt = cfg["control_hz"]
        self.arrive_num = self.static_coll_num = self.ped_coll_num = self.other_coll_num = 0
        self.steps = self.tmp_steps = self.stuck_num = 0
        self.v_sum = self.w_sum = 0
        self.speed_step = 0
        self.w_variance_array, self.v_jerk_array, self.w_jerk_array, self.w_zero_array = [], [], [], []
        self.traj_helper = TrajectoryPathHelper(dt=self.dt)

    def step(self, action):
        states, reward, done, info = self.env.step(action)
        self.tmp_steps += 1
        speeds = [float(x) for x in info["speeds"][0][:2]]  # "suppose only one agent here"
        self.v_sum += speeds[0]
        self.w_sum += abs(speeds[1])
        self.traj_helper.add_vw(*speeds)
        return states, reward, done, info

    def reset(self, **kwargs):
        if self.tmp_steps > 3:  # two or three steps: too short to count
            self.cur_episode += 1
            self.dones_statistics(kwargs.get("dones_info"))
        if self.cur_episode == self.max_episodes:
            self.screen_out()
        self.tmp_steps = 0
        return self.env.reset(**kwargs)

    def dones_statistics(self, t):
        if t is None:
            return
        t = int(t[0])
        self.speed_step += self.tmp_steps
        if t == 5:
            self.arrive_num += 1
            self.steps += self.tmp_steps
        elif t == 10:
            self.stuck_num += 1
        elif t == 1:
            self.static_coll_num += 1
        elif t == 2:
            self.ped_coll_num += 1
        elif t == 3:
            self.other_coll_num += 1
        else:
            raise ValueError("[TestEpisodeWrapper]: No dones info: %r" % t)
        self.traj_helper.reset()  # path_statistics
        self.v_jerk_array.append(round(self.traj_helper.v_jerk, 4))
        self.w_jerk_array.append(round(self.traj_helper.w_jerk, 4))
        self.w_zero_array.append(self.traj_helper.w_zero)
        self.w_variance_array.append(round(self.traj_helper.w_variance, 4))
        self.traj_helper.clear_vw_array()

    def statistics(self):
        n = self.max_episodes
        return dict(arrive_rate=self.arrive_num / n, static_coll_rate=self.static_coll_num / n, ped_coll_rate=self.ped_coll_num / n,
                    other_coll_rate=self.other_coll_num / n, avg_arrive_steps=self.steps / max(1, self.arrive_num),
                    stuck_rate=self.stuck_num / n, avg_v=self.v_sum / self.speed_step, avg_w=self.w_sum / self.speed_step,
                    avg_w_variance=sum(self.w_variance_array) / n, avg_v_jerk=sum(self.v_jerk_array) / n,
                    avg_w_jerk=sum(self.w_jerk_array) / n, avg_w_zero=sum(self.w_zero_array) / n)

    def screen_out(self):
        import sys
        print("[TestEpisodeWrapper]: Have run max episodes %d, statistics number are in the following:" % self.max_episodes)
        for k, v in self.statistics().items():
            print("    %s: %s," % (k, v))
        print("[TestEpisodeWrapper]: Exit Progress!", flush=True)
        sys.exit()


class StatePedVectorWrapper(ObservationWrapper):
    """envs/wrapper/base.py:19-34"""
    avg = (0.0, 0.0, 0.0, 0.0, 0.25, 0.25, 0.0)
    std = (6.0, 6.0, 0.6, 0.9, 0.50, 0.5, 6.0)

    def __init__(self, env, cfg=None):
        super().__init__(env)

    def observation(self, state):
        import torch
        p = state.ped_vector_states.clone()
        n = int(p[0, 0].item()) if len(p) else 0
        if n:
            avg = torch.tensor(self.avg, device=p.device, dtype=p.dtype).repeat(n)
            std = torch.tensor(self.std, device=p.device, dtype=p.dtype).repeat(n)
            p[:, 1:1 + 7 * n] = (p[:, 1:1 + 7 * n] - avg) / std
        state.ped_vector_states = p
        return state


class ObsStateTmp(ObservationWrapper):
    """envs/wrapper/filter_states.py:6-12"""

    def __init__(self, env, cfg):
        super().__init__(env)

    def observation(self, states):
        return [states.sensor_maps, states.vector_states, states.ped_maps]


class ObsLaserStateTmp(ObservationWrapper):
    """envs/wrapper/filter_states.py:15-20"""

    def __init__(self, env, cfg):
        super().__init__(env)

    def observation(self, states):
        return [states.lasers, states.vector_states, states.ped_maps]


wrapper_dict = {
    "StatePedVectorWrapper": StatePedVectorWrapper,
    "VelActionWrapper": VelActionWrapper,
    "StateBatchWrapper": StateBatchWrapper,
    "SensorsPaperRewardWrapper": SensorsPaperRewardWrapper,
    "NeverStopWrapper": NeverStopWrapper,
    "ObsStateTmp": ObsStateTmp,
    "TimeLimitWrapper": TimeLimitWrapper,
    "MultiRobotCleanWrapper": MultiRobotCleanWrapper,
    "InfoLogWrapper": InfoLogWrapper,
    "ObsLaserStateTmp": ObsLaserStateTmp,
    "TestEpisodeWrapper": TestEpisodeWrapper,
}


def make_env(cfg):
    """envs/__init__.py:21-33"""
    if isinstance(cfg, str):
        cfg = read_yaml(cfg)
    if cfg["env_type"] != "robot_nav":
        raise ValueError("only env_type 'robot_nav' (ImageEnv) is implemented; %r is a different backend" % cfg["env_type"])
    env = ImageEnv(cfg)
    for name in cfg["wrapper"]:
        if name not in wrapper_dict:
            raise KeyError("wrapper %r is outside the step() path scope (evaluation / recording wrappers)" % name)
        env = wrapper_dict[name](env, cfg)
    cfg["node_id"] = cfg.get("node_id", 0) + 1
    return env
