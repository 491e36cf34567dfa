"""ctypes drivers for the oracle's SFM restatement and the reference-built libpedsim (oracle/_ref)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DP = C.POINTER(C.c_double)
FP = C.POINTER(C.c_float)


class OracleSfm:
    def __init__(self, n_peds, n_robots, max_speed):
        self.lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L = self.lib
        L.sfm_create.restype = C.c_void_p
        L.sfm_create.argtypes = [C.c_int, C.c_int, FP]
        L.sfm_add_obstacle.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.sfm_clear_obstacles.argtypes = [C.c_void_p]
        L.sfm_set_ped_pos.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.sfm_set_robot_pos.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.sfm_set_waypoints.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, DP, C.c_int]
        L.sfm_move_agents.argtypes = [C.c_void_p, C.c_double]
        L.sfm_get_agent.argtypes = [C.c_void_p, C.c_int, DP]
        L.sfm_get_vmax.argtypes = [C.c_void_p, C.c_int]
        L.sfm_get_vmax.restype = C.c_double
        L.sfm_destroy.argtypes = [C.c_void_p]
        ms = np.ascontiguousarray(max_speed, np.float32)
        self.h = L.sfm_create(n_peds, n_robots, ms.ctypes.data_as(FP))
        self.n_peds, self.n_robots = n_peds, n_robots

    def add_obstacle(self, a, b, c, d):
        self.lib.sfm_add_obstacle(self.h, a, b, c, d)

    def set_ped_pos(self, j, x, y):
        self.lib.sfm_set_ped_pos(self.h, j, x, y)

    def set_robot_pos(self, i, x, y):
        self.lib.sfm_set_robot_pos(self.h, i, x, y)

    def set_waypoints(self, j, gx, gy, traj):
        t = np.ascontiguousarray(traj, np.float64).reshape(-1, 3)
        self.lib.sfm_set_waypoints(self.h, j, gx, gy, t.ctypes.data_as(DP), len(t))

    def move(self, h):
        self.lib.sfm_move_agents(self.h, h)

    def state(self):
        out = np.zeros((self.n_peds + self.n_robots, 6))
        for i in range(len(out)):
            self.lib.sfm_get_agent(self.h, i, out[i].ctypes.data_as(DP))
        return out

    def vmax(self):
        return np.array([self.lib.sfm_get_vmax(self.h, i) for i in range(self.n_peds + self.n_robots)])


class RefSfm:
    """the reference's own libpedsim sources compiled unmodified (oracle/_ref/libpedsim_ref.so)"""

    @staticmethod
    def path():
        return os.path.join(ROOT, "oracle", "_ref", "libpedsim_ref.so")

    def __init__(self, n_peds, n_robots, max_speed):
        self.lib = C.CDLL(self.path())
        L = self.lib
        L.pedref_create.restype = C.c_void_p
        L.pedref_create.argtypes = [C.c_int, C.c_int, C.c_int, FP]
        L.pedref_add_obstacle.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.pedref_set_ped_pos.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.pedref_set_robot_pos.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.pedref_set_waypoints.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, DP, C.c_int]
        L.pedref_move_agents.argtypes = [C.c_void_p, C.c_double]
        L.pedref_get_ped.argtypes = [C.c_void_p, C.c_int, DP]
        L.pedref_get_robot.argtypes = [C.c_void_p, C.c_int, DP]
        L.pedref_get_vmax.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.pedref_get_vmax.restype = C.c_double
        ms = np.ascontiguousarray(max_speed, np.float32)
        self.h = L.pedref_create(n_peds, n_robots, 1, ms.ctypes.data_as(FP))
        self.n_peds, self.n_robots = n_peds, n_robots

    def add_obstacle(self, a, b, c, d):
        self.lib.pedref_add_obstacle(self.h, a, b, c, d)

    def set_ped_pos(self, j, x, y):
        self.lib.pedref_set_ped_pos(self.h, j, x, y)

    def set_robot_pos(self, i, x, y):
        self.lib.pedref_set_robot_pos(self.h, i, x, y)

    def set_waypoints(self, j, gx, gy, traj):
        t = np.ascontiguousarray(traj, np.float64).reshape(-1, 3)
        self.lib.pedref_set_waypoints(self.h, j, gx, gy, t.ctypes.data_as(DP), len(t))

    def move(self, h):
        self.lib.pedref_move_agents(self.h, h)

    def state(self):
        out = np.zeros((self.n_peds + self.n_robots, 6))
        for i in range(self.n_peds):
            self.lib.pedref_get_ped(self.h, i, out[i].ctypes.data_as(DP))
        for i in range(self.n_robots):
            self.lib.pedref_get_robot(self.h, i, out[self.n_peds + i].ctypes.data_as(DP))
        return out

    def vmax(self):
        return np.array([self.lib.pedref_get_vmax(self.h, 0, i) for i in range(self.n_peds)] +
                        [self.lib.pedref_get_vmax(self.h, 1, i) for i in range(self.n_robots)])


def run_scenario(cls, seed, n_peds=8, n_robots=3, n_obs=3, steps=80, world=10.0, h=0.25):
    """PedScene's call pattern (pedscene.h): reset positions / waypoints / obstacles, then per step
    moveAgents(h) followed by setRobotPos for every robot (img_env.cpp:343, 411-417)."""
    rng = np.random.default_rng(seed)
    sim = cls(n_peds, n_robots, np.full(n_peds, 0.5, np.float32))
    start = rng.uniform(1.0, world - 1.0, (n_peds, 2))
    goal = rng.uniform(1.0, world - 1.0, (n_peds, 2))
    for k in range(n_obs):
        a, b = rng.uniform(1.0, world - 1.0, 2), rng.uniform(1.0, world - 1.0, 2)
        sim.add_obstacle(a[0], a[1], b[0], b[1])
    for j in range(n_peds):
        sim.set_ped_pos(j, start[j, 0], start[j, 1])
        sim.set_waypoints(j, goal[j, 0], goal[j, 1], [[goal[j, 0], goal[j, 1], 0.0], [start[j, 0], start[j, 1], 0.0]])
    rob = rng.uniform(1.0, world - 1.0, (n_robots, 2))
    for i in range(n_robots):
        sim.set_robot_pos(i, rob[i, 0], rob[i, 1])
    trace = []
    for s in range(steps):
        sim.move(h)
        rob = rob + rng.uniform(-0.1, 0.1, rob.shape)
        for i in range(n_robots):
            sim.set_robot_pos(i, rob[i, 0], rob[i, 1])
        trace.append(sim.state())
    return np.stack(trace), sim.vmax()
