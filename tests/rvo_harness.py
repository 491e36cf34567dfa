"""ctypes drivers for the oracle's RVO restatement and the reference-built RVO2 (oracle/_ref)."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int)


def _fp(a):
    return a.ctypes.data_as(FP)


class RvoSimStruct(C.Structure):
    _fields_ = [("time_step", C.c_float), ("n_agents", C.c_int32), ("cap_agents", C.c_int32),
                ("px", FP), ("py", FP), ("vx", FP), ("vy", FP), ("prefx", FP), ("prefy", FP)]


class OracleRvo:
    def __init__(self, dt):
        self.lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L = self.lib
        L.rvo_create.restype = C.POINTER(RvoSimStruct)
        L.rvo_create.argtypes = [C.c_float]
        L.rvo_add_agent.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float,
                                    C.c_float, C.c_float]
        L.rvo_add_obstacle.argtypes = [C.c_void_p, FP, C.c_int]
        L.rvo_clear_obstacles.argtypes = [C.c_void_p]
        L.rvo_process_obstacles.argtypes = [C.c_void_p]
        L.rvo_do_step.argtypes = [C.c_void_p, C.c_int, FP, FP, C.c_int]
        L.rvo_destroy.argtypes = [C.c_void_p]
        L.rvo_set_bruteforce.argtypes = [C.c_int]
        self.h = L.rvo_create(dt)
        self.n = 0

    def add_agent(self, x, y, nd, mn, th, tho, r, ms):
        self.n += 1
        return self.lib.rvo_add_agent(self.h, x, y, nd, mn, th, tho, r, ms)

    def add_obstacle(self, xy):
        xy = np.ascontiguousarray(xy, np.float32)
        return self.lib.rvo_add_obstacle(self.h, _fp(xy), len(xy))

    def clear_obstacles(self):
        self.lib.rvo_clear_obstacles(self.h)

    def process_obstacles(self):
        self.lib.rvo_process_obstacles(self.h)

    def _arr(self, name):
        return np.ctypeslib.as_array(getattr(self.h.contents, name), shape=(self.n,))

    def set_position(self, i, x, y):
        self._arr("px")[i] = x
        self._arr("py")[i] = y

    def set_velocity(self, i, x, y):
        self._arr("vx")[i] = x
        self._arr("vy")[i] = y

    def set_pref(self, i, x, y):
        self._arr("prefx")[i] = x
        self._arr("prefy")[i] = y

    def do_step(self, ps=None, rs=None, brute=False):
        self.lib.rvo_set_bruteforce(1 if brute else 0)
        if ps is None:
            self.lib.rvo_do_step(self.h, self.n, None, None, -1)
        else:
            ps = np.ascontiguousarray(ps, np.float32)
            rs = np.ascontiguousarray(rs, np.float32)
            self.lib.rvo_do_step(self.h, self.n, _fp(ps), _fp(rs), len(rs))
        self.lib.rvo_set_bruteforce(0)

    def state(self):
        return np.stack([self._arr("px"), self._arr("py"), self._arr("vx"), self._arr("vy")], 1).copy()

    def close(self):
        self.lib.rvo_destroy(self.h)


class RefRvo:
    """The reference's own RVO2 sources compiled unmodified (oracle/_ref/librvo_ref.so)."""

    @staticmethod
    def path():
        return os.path.join(ROOT, "oracle", "_ref", "librvo_ref.so")

    def __init__(self, dt):
        self.lib = C.CDLL(self.path())
        L = self.lib
        L.rvoref_create.restype = C.c_void_p
        L.rvoref_create.argtypes = [C.c_float]
        L.rvoref_add_agent.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float,
                                       C.c_float, C.c_float]
        L.rvoref_add_obstacle.argtypes = [C.c_void_p, FP, C.c_int]
        for f in ("rvoref_clear_obstacles", "rvoref_process_obstacles", "rvoref_destroy"):
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("rvoref_set_position", "rvoref_set_velocity", "rvoref_set_pref_velocity"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.rvoref_do_step.argtypes = [C.c_void_p, FP, FP, C.c_int]
        L.rvoref_get_state.argtypes = [C.c_void_p, FP]
        self.h = L.rvoref_create(dt)
        self.n = 0

    def add_agent(self, x, y, nd, mn, th, tho, r, ms):
        self.n += 1
        return self.lib.rvoref_add_agent(self.h, x, y, nd, mn, th, tho, r, ms)

    def add_obstacle(self, xy):
        xy = np.ascontiguousarray(xy, np.float32)
        return self.lib.rvoref_add_obstacle(self.h, _fp(xy), len(xy))

    def clear_obstacles(self):
        self.lib.rvoref_clear_obstacles(self.h)

    def process_obstacles(self):
        self.lib.rvoref_process_obstacles(self.h)

    def set_position(self, i, x, y):
        self.lib.rvoref_set_position(self.h, i, x, y)

    def set_velocity(self, i, x, y):
        self.lib.rvoref_set_velocity(self.h, i, x, y)

    def set_pref(self, i, x, y):
        self.lib.rvoref_set_pref_velocity(self.h, i, x, y)

    def do_step(self, ps=None, rs=None, brute=False):
        if ps is None:
            self.lib.rvoref_do_step(self.h, None, None, -1)
        else:
            ps = np.ascontiguousarray(ps, np.float32)
            rs = np.ascontiguousarray(rs, np.float32)
            self.lib.rvoref_do_step(self.h, _fp(ps), _fp(rs), len(rs))

    def state(self):
        out = np.zeros((self.n, 4), np.float32)
        self.lib.rvoref_get_state(self.h, _fp(out))
        return out

    def close(self):
        self.lib.rvoref_destroy(self.h)


def run_scenario(sim, seed, n_agents=60, n_obs=4, steps=60, world=10.0, ervo=False, brute=False,
                 neighbor_dist=0.5, radius=0.5, dt=0.25):
    """Drive a simulator the way RVOScene does (rvoscene.h:36-66): pref velocity toward a goal,
    normalised when longer than 1; boxes given as the 4-vertex order rvoscene.h:19-26 uses."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(1.0, world - 1.0, (n_agents, 2)).astype(np.float32)
    goal = rng.uniform(1.0, world - 1.0, (n_agents, 2)).astype(np.float32)
    for i in range(n_agents):
        sim.add_agent(0.0, 0.0, neighbor_dist, 10, 5.0, 5.0, radius, float(np.float32(rng.uniform(0.3, 0.8))))
    for i in range(n_agents):
        sim.set_position(i, float(pos[i, 0]), float(pos[i, 1]))
    for k in range(n_obs):
        c = rng.uniform(2.0, world - 2.0, 2)
        a = c - rng.uniform(0.2, 0.8, 2) * rng.choice([-1, 1], 2)
        b = c + rng.uniform(0.2, 0.8, 2) * rng.choice([-1, 1], 2)
        sim.add_obstacle(np.array([[a[0], a[1]], [a[0], b[1]], [b[0], b[1]], [b[0], a[1]]], np.float32))
    sim.process_obstacles()
    trace = []
    for s in range(steps):
        st = sim.state()
        for i in range(n_agents):
            g = goal[i] - st[i, :2]
            g = g.astype(np.float32)
            sq = np.float32(g[0] * g[0] + g[1] * g[1])
            if sq > np.float32(1.0):
                inv = np.float32(1.0) / np.float32(np.sqrt(sq))
                g = np.array([g[0] * inv, g[1] * inv], np.float32)
            sim.set_pref(i, float(g[0]), float(g[1]))
        if ervo:
            ps = rng.uniform(0, world, (3, 2)).astype(np.float32)
            rs = rng.uniform(0.5, 2.0, 3).astype(np.float32)
            sim.do_step(ps, rs, brute=brute)
        else:
            sim.do_step(brute=brute)
        trace.append(sim.state())
        if s % 20 == 19:  # goals flip like go_back pedestrians
            goal = rng.uniform(1.0, world - 1.0, (n_agents, 2)).astype(np.float32)
    return np.stack(trace)
