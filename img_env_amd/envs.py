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


class EpisodeStats:
    """Running path statistics of every robot of an env at once, as tensors on the device the speeds live on: what the
    reference's evaluation helper (envs/wrapper/evaluation_wrapper/utils.py:60-129) computes from the stored (v, w) arrays of
    ONE robot -- variance of w, sign changes of w, mean |acceleration| and |jerk| of v and w, mean v and |w| -- accumulated step
    by step instead (sums, last two commands), so nothing grows with the episode and nothing loops over robots."""

    def __init__(self, n, dt, device):
        import torch
        self.dt = float(dt)
        z = lambda: torch.zeros(n, dtype=torch.float64, device=device)  # noqa: E731
        self.n = z()                        # commands so far
        self.sum_v, self.sum_w, self.sum_ww, self.sum_absw = z(), z(), z(), z()
        self.sum_abs_acc, self.sum_abs_jerk = [z(), z()], [z(), z()]  # v, w
        self.prev, self.prev2 = [z(), z()], [z(), z()]                # last and last-but-one command
        self.w_zero = z()

    def add(self, v, w):
        import torch
        v, w = v.to(torch.float64), w.to(torch.float64)
        has1, has2 = self.n >= 1, self.n >= 2
        for k, x in enumerate((v, w)):
            acc = (x - self.prev[k]) / self.dt
            acc_prev = (self.prev[k] - self.prev2[k]) / self.dt
            self.sum_abs_acc[k] += torch.where(has1, acc.abs(), torch.zeros_like(acc))
            self.sum_abs_jerk[k] += torch.where(has2, ((acc - acc_prev) / self.dt).abs(), torch.zeros_like(acc))
        last = self.prev[1]  # (0 before the first command, like the helper's `tmp`)
        self.w_zero += (((w == 0) & (last != 0)) | ((w > 0) & (last < 0)) | ((w < 0) & (last > 0))).to(torch.float64)
        self.prev2 = [self.prev[0], self.prev[1]]
        self.prev = [v, w]
        self.n += 1
        self.sum_v += v
        self.sum_w += w
        self.sum_ww += w * w
        self.sum_absw += w.abs()

    def finish(self):
        """the episode's figures per robot; the accumulators start over"""
        import torch
        n = self.n.clamp(min=1)
        mean_w = self.sum_w / n
        out = dict(w_variance=self.sum_ww / n - mean_w * mean_w, w_zero=self.w_zero.clone(),
                   v_acc=self.sum_abs_acc[0] / (n - 1).clamp(min=1), w_acc=self.sum_abs_acc[1] / (n - 1).clamp(min=1),
                   v_jerk=self.sum_abs_jerk[0] / (n - 2).clamp(min=1), w_jerk=self.sum_abs_jerk[1] / (n - 2).clamp(min=1),
                   v_avg=self.sum_v / n, w_avg=self.sum_absw / n)
        for t in (self.n, self.sum_v, self.sum_w, self.sum_ww, self.sum_absw, self.w_zero, *self.sum_abs_acc, *self.sum_abs_jerk,
                  *self.prev, *self.prev2):
            t.zero_()
        self.prev = [torch.zeros_like(self.n), torch.zeros_like(self.n)]
        self.prev2 = [torch.zeros_like(self.n), torch.zeros_like(self.n)]
        return out


class TestEpisodeWrapper(Wrapper):
    """Evaluation statistics over ``init_pose_bag_episodes`` episodes (the wrapper of that name in the reference's shipped
    test.yaml, envs/wrapper/evaluation_wrapper/TestEpisodeWrapper.py:8-119): how episodes end -- arrive / static, pedestrian,
    robot collision / time-out, by ``dones_info`` -- steps to arrive, mean speeds and the path figures of ``EpisodeStats``.
    Kept for EVERY robot of the env as device tensors (the reference tracks one robot, "suppose only one agent here"); the
    scalar attributes the reference exposes (``arrive_num``, ``v_sum``, ``w_variance_array`` ...) read robot 0, and the printed
    rates pool all robots, which for a one-robot env are the reference's numbers.  After ``max_episodes`` episodes the
    statistics are printed and, as in the reference, the process exits."""
    __test__ = False  # (not a pytest class)
    CODES = (5, 10, 1, 2, 3)  # arrive, time-out, static / pedestrian / other-robot collision

    def __init__(self, env, cfg):
        super().__init__(env)
        self.max_episodes = cfg["init_pose_bag_episodes"]
        self.dt = cfg["control_hz"]
        self.cur_episode = 0
        self.tmp_steps = 0
        self._stats = None      # EpisodeStats, made at the first step (robot count and device come with the speeds)
        self._ends = None       # [robot][code] episodes by how they ended
        self._arrive_steps = self._speed_steps = self._v_sum = self._w_sum = None
        self._episodes = []     # per counted episode: dict of [robot] tensors (rounded to 4 places like the reference's lists)

    def _init(self, speeds):
        import torch
        n, dev = speeds.shape[0], speeds.device
        self._stats = EpisodeStats(n, self.dt, dev)
        self._ends = torch.zeros(n, 11, dtype=torch.int64, device=dev)
        self._arrive_steps = torch.zeros(n, dtype=torch.int64, device=dev)
        self._speed_steps = torch.zeros(n, dtype=torch.int64, device=dev)
        self._v_sum = torch.zeros(n, dtype=torch.float64, device=dev)
        self._w_sum = torch.zeros(n, dtype=torch.float64, device=dev)

    def step(self, action):
        states, reward, done, info = self.env.step(action)
        speeds = info["speeds"]
        if self._stats is None:
            self._init(speeds)
        self.tmp_steps += 1
        self._v_sum += speeds[:, 0].double()
        self._w_sum += speeds[:, 1].double().abs()
        self._stats.add(speeds[:, 0], speeds[:, 1])
        return states, reward, done, info

    def reset(self, **kwargs):
        codes = kwargs.get("dones_info")
        if self.tmp_steps > 3:  # two or three steps: too short to count
            self.cur_episode += 1
            if codes is not None and self._stats is not None:
                self._count(codes)
        if self.cur_episode == self.max_episodes:
            self.screen_out()
        self.tmp_steps = 0
        return self.env.reset(**kwargs)

    def _count(self, codes):
        import torch
        codes = torch.as_tensor(codes, device=self._ends.device).long()
        known = torch.zeros_like(codes, dtype=torch.bool)
        for c in self.CODES:
            known |= codes == c
        if not bool(known.all()):
            raise ValueError("[TestEpisodeWrapper]: No dones info: %r" % codes[~known][0].item())
        self._ends.scatter_add_(1, codes.view(-1, 1), torch.ones_like(codes).view(-1, 1))
        self._speed_steps += self.tmp_steps
        self._arrive_steps += torch.where(codes == 5, self.tmp_steps, 0)
        ep = self._stats.finish()
        self._episodes.append({k: (torch.round(v * 1e4) / 1e4 if k != "w_zero" else v) for k, v in ep.items()})

    # ---- the reference's scalar attributes, for robot 0 ----
    def _end0(self, code):
        return 0 if self._ends is None else int(self._ends[0, code])

    arrive_num = property(lambda self: self._end0(5))
    stuck_num = property(lambda self: self._end0(10))
    static_coll_num = property(lambda self: self._end0(1))
    ped_coll_num = property(lambda self: self._end0(2))
    other_coll_num = property(lambda self: self._end0(3))
    steps = property(lambda self: 0 if self._arrive_steps is None else int(self._arrive_steps[0]))
    speed_step = property(lambda self: 0 if self._speed_steps is None else int(self._speed_steps[0]))
    v_sum = property(lambda self: 0.0 if self._v_sum is None else float(self._v_sum[0]))
    w_sum = property(lambda self: 0.0 if self._w_sum is None else float(self._w_sum[0]))
    w_variance_array = property(lambda self: [float(e["w_variance"][0]) for e in self._episodes])
    v_jerk_array = property(lambda self: [float(e["v_jerk"][0]) for e in self._episodes])
    w_jerk_array = property(lambda self: [float(e["w_jerk"][0]) for e in self._episodes])
    w_zero_array = property(lambda self: [float(e["w_zero"][0]) for e in self._episodes])

    def statistics(self):
        """rates and averages over all robots and counted episodes"""
        import torch
        if self._ends is None:
            return {}
        robots = self._ends.shape[0]
        n = self.max_episodes * robots
        ends = self._ends.sum(dim=0)
        speed_steps = max(1, int(self._speed_steps.sum()))
        per_ep = {k: float(torch.stack([e[k] for e in self._episodes]).sum()) / n for k in ("w_variance", "v_jerk", "w_jerk", "w_zero")} \
            if self._episodes else dict(w_variance=0.0, v_jerk=0.0, w_jerk=0.0, w_zero=0.0)
        return dict(arrive_rate=int(ends[5]) / n, static_coll_rate=int(ends[1]) / n, ped_coll_rate=int(ends[2]) / n,
                    other_coll_rate=int(ends[3]) / n, avg_arrive_steps=int(self._arrive_steps.sum()) / max(1, int(ends[5])),
                    stuck_rate=int(ends[10]) / n, avg_v=float(self._v_sum.sum()) / speed_steps, avg_w=float(self._w_sum.sum()) / speed_steps,
                    avg_w_variance=per_ep["w_variance"], avg_v_jerk=per_ep["v_jerk"], avg_w_jerk=per_ep["w_jerk"], avg_w_zero=per_ep["w_zero"])

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


class PedTrajectoryDatasetWrapper(Wrapper):
    """The feeder of the ``dataset`` pedestrian scene (envs/wrapper/evaluation_wrapper/PedTrajectoryDatasetWrapper.py:15-291): a file
    of recorded pedestrian tracks (ETH / UCY world coordinates: four rows frame, pedestrian, y, x) is cut into "worlds" -- ranges
    of pedestrians -- and every ``reset`` hands the env the current world's tracks as ``cur_ped_pos_v_datas``, one series of
    [x, y, theta, vx, vy] per pedestrian every ``control_hz`` seconds (img_env.cpp:361-386 replays them).  After
    ``repeated_time_per_env`` episodes the next world is up; every finished episode appends one line to ``output_file`` (how it
    ended + the path figures of its (v, w) commands); after the last world the process exits, as in the reference.

    numpy only (the reference goes through pandas and builds scipy interpolators it never evaluates).  The path figures follow
    envs/wrapper/evaluation_wrapper/utils.py:60-129 on the stored commands of robot 0 ("suppose only one agent here")."""

    def __init__(self, env, cfg):
        super().__init__(env)
        import os
        self.cfg = cfg
        self.dt = cfg.get("control_hz", 0.4)
        if cfg.get("ped_traj_dataset") is None:
            raise ValueError("PedTrajectoryDatasetWrapper needs cfg['ped_traj_dataset']")
        self._read_dataset(cfg["ped_traj_dataset"])
        self.repeated_time_per_env = cfg.get("repeated_time_per_env", 10)
        self.cur_repeated_time_per_env = 0
        self.ped_dataset_worlds = cfg.get("ped_dataset_worlds", [[0, 10]])
        self.max_worlds = len(self.ped_dataset_worlds)
        self.cur_world = 0
        lo, hi = self.ped_dataset_worlds[self.cur_world]
        cfg["ped_sim"]["total"] = self.cur_world_max_peds = hi - lo + 1
        self.node_id = cfg.get("node_id", 0)
        self.output_file = cfg.get("output_file", "../output/ped_dataset_{}/ppo_{}.txt".format(self.dataset_name, self.node_id))
        self.output_dir = "/".join(self.output_file.split("/")[:-1])
        if self.output_dir != "" and not os.path.exists(self.output_dir):
            os.mkdir(self.output_dir)
        self._v, self._w = [], []  # this episode's commands of robot 0

    # ---- the recorded tracks
    def _read_dataset(self, path):
        import os
        import sys
        if not os.path.exists(path):  # (run from the trainer's directory)
            path = sys.argv[0].split("runner")[0] + "env/drlnav_env/" + path
        rows = np.loadtxt(path, delimiter=",", ndmin=2)
        self._frame, self._ped = rows[0].astype(np.int64), rows[1].astype(np.int64)
        self._y, self._x = rows[2], rows[3]  # the file's third row is "y", its fourth "x"
        self.max_peds = int(self._ped.max())
        c = self.cfg
        self.dataset_name = c.get("ped_dataset_name", "eth")
        self.swapxy = c.get("swapxy", True)
        self.spawn_delay_s = c.get("spawn_delay_s", 0)
        self.offset = c.get("offset", [1.4, 14.4, 0])
        self.fps = c.get("fps", 15)
        if (self.dt * self.fps) % 1 != 0:
            raise ValueError("control_hz * fps must be a whole number of frames")
        self.skip_frame = int(self.dt * self.fps)
        self.start_t = c.get("start_t", 0)
        self.max_time = c.get("max_time", 20)
        self.scale_x, self.scale_y = c.get("scale_x", 1), c.get("scale_y", 1)

    def _track(self, ped_id, start_frame):
        """one pedestrian's series: times (with the spawn instant in front), poses with the heading of each move, finite-
        difference speeds along that heading; the first pose held until the pedestrian's first frame; every skip_frame-th kept"""
        sel = self._ped == ped_id
        frames, xs, ys = self._frame[sel], self._x[sel], self._y[sel]
        times = (frames - start_frame) * (1.0 / self.fps) + (self.spawn_delay_s + self.start_t)
        times = np.concatenate([[times[0] - self.start_t], times])
        a, b = (ys, xs) if self.swapxy else (xs, ys)
        sa, sb = (self.scale_y, self.scale_x) if self.swapxy else (self.scale_x, self.scale_y)
        a, b = sa * a, sb * b
        s, c = np.sin(self.offset[2]), np.cos(self.offset[2])
        px, py = a * c - b * s + self.offset[0], a * s + b * c + self.offset[1]
        th = np.arctan2(py[1:] - py[:-1], px[1:] - px[:-1])
        th = np.append(th, th[-1])
        pose = np.stack([px, py, th], 1)
        pose = np.insert(pose, [0], pose[0], axis=0)  # the spawn instant
        out = []
        for j in range(len(pose)):
            if j > 1:
                speed = np.sqrt((pose[j, 0] - pose[j - 1, 0]) ** 2 + (pose[j, 1] - pose[j - 1, 1]) ** 2) / (times[j] - times[j - 1])
                out.append([pose[j, 0], pose[j, 1], pose[j, 2], speed * np.cos(pose[j, 2]), speed * np.sin(pose[j, 2])])
            else:
                out.append([pose[j, 0], pose[j, 1], pose[j, 2], 0, 0])
        out = [out[0]] * int(frames[0] - start_frame) + out
        return out[::self.skip_frame], frames

    def _generate_humans(self, start_idx, max_agents):
        series, longest, start_frame = [], 0, None
        for i in range(max_agents):
            ped_id = i + start_idx + 1
            if not ped_id < self.max_peds + 1:
                raise AssertionError("PedTrajectoryDatasetWrapper: pedestrian %d is beyond the file's %d" % (ped_id, self.max_peds))
            first = int(self._frame[self._ped == ped_id][0])
            if i == 0:
                start_frame = first
            if (first - start_frame) / self.fps > self.max_time:
                break
            one, _ = self._track(ped_id, start_frame)
            longest = max(longest, len(one))
            series.append(one)
        for i in range(max_agents):  # (IndexError when the max_time cut dropped a pedestrian, as in the reference)
            series[i] = series[i] + (longest - len(series[i])) * [series[i][-1]]
        return series

    def change_world(self):
        return self._generate_humans(self.ped_dataset_worlds[self.cur_world][0], self.cur_world_max_peds)

    # ---- the env
    def step(self, action):
        states, reward, done, info = self.env.step(action)
        speeds = info.get("speeds")[0]  # robot 0
        self._v.append(float(speeds[0]))
        self._w.append(float(speeds[1]))
        return states, reward, done, info

    def reset(self, **kwargs):
        import sys
        self.out2logfile(kwargs.get("dones_info"))
        print("PedTrajectoryDatasetWrapper Reset ", flush=True)
        if self.cur_world == self.max_worlds:
            print("[PedTrajectoryDatasetWrapper]: Run Over.", flush=True)
            sys.exit()
        kwargs["cur_ped_pos_v_datas"] = self.change_world()
        return self.env.reset(**kwargs)

    def _metrics(self):
        v, w, dt = np.asarray(self._v, np.float64), np.asarray(self._w, np.float64), self.dt
        tmp, w_zero = 0, 0
        for x in w:  # sign changes of w (a zero after a turn counts)
            if x == 0:
                w_zero += 1 if tmp != 0 else 0
            elif (x > 0 and tmp < 0) or (x < 0 and tmp > 0):
                w_zero += 1
            tmp = x
        v_acc, w_acc = np.diff(v) / dt, np.diff(w) / dt
        return dict(v_avg=round(float(np.average(v)), 4), w_avg=round(float(np.average(np.abs(w))), 4),
                    v_acc=round(float(np.average(np.abs(v_acc))), 4), w_acc=round(float(np.average(np.abs(w_acc))), 4),
                    v_jerk=round(float(np.average(np.abs(np.diff(v_acc) / dt))), 4), w_jerk=round(float(np.average(np.abs(np.diff(w_acc) / dt))), 4),
                    w_zero=w_zero, path_time=round(len(v) * dt, 4), steps=len(v))

    def out2logfile(self, dones):
        if dones is None:
            return
        self.cur_repeated_time_per_env += 1
        code = int(dones[0])
        m = self._metrics()
        m.update(arrive=1 if code == 5 else 0, ped_collision=1 if code == 2 else 0, stuck=1 if code == 10 else 0, cur_world=self.cur_world)
        with open(self.output_file, "a") as f:
            f.write("{cur_world}, {arrive}, {ped_collision}, {stuck}, {v_avg}, {w_avg}, {v_acc}, {w_acc}, {v_jerk}, {w_jerk}, {w_zero}, "
                    "{path_time}, {steps}".format_map(m))
            f.write("\n")
        self._v, self._w = [], []
        if self.cur_repeated_time_per_env == self.repeated_time_per_env:
            self.cur_world += 1
            self.cur_repeated_time_per_env = 0


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
    "PedTrajectoryDatasetWrapper": PedTrajectoryDatasetWrapper,
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
