"""TEST-ONLY ctypes binding of the CPU oracle (oracle/liboracle.so).  Never imported by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

from img_env_amd import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


def load_oracle():
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build_oracle()
    lib = C.CDLL(path)
    lib.oracle_last_error.restype = C.c_char_p
    lib.oracle_create.argtypes = [C.POINTER(_cabi.Cfg), C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.oracle_destroy.argtypes = [C.c_void_p]
    lib.oracle_destroy.restype = None
    lib.oracle_reset.argtypes = [C.c_void_p, C.POINTER(_cabi.ResetBatch)]
    lib.oracle_step.argtypes = [C.c_void_p, C.c_void_p]
    lib.oracle_step_begin.argtypes = [C.c_void_p, C.c_void_p]
    lib.oracle_step_end.argtypes = [C.c_void_p]
    lib.oracle_records.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.oracle_outputs.argtypes = [C.c_void_p, C.POINTER(_cabi.Out)]
    lib.oracle_pedinfo.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.oracle_private_grid.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.oracle_grids.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.oracle_sfm_tree.argtypes = [C.c_void_p, C.c_void_p]
    lib.sfm_set_cr_atan2.argtypes = [C.c_int]
    lib.sfm_set_cr_atan2.restype = None
    return lib


def set_cr_atan2(on):
    """social-force oracle: every atan2 correctly rounded (libquadmath) instead of the host libm's -- process-wide switch"""
    load_oracle().sfm_set_cr_atan2(1 if on else 0)


class OracleWorld:
    """Same call surface as img_env_amd.world.World, on host numpy arrays."""

    def __init__(self, params, grid):
        self.lib = load_oracle()
        self.params = dict(params)
        self.grid = np.ascontiguousarray(grid, np.uint8)
        cfg, self._keep = _cabi.make_cfg(self.params)
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.lib.oracle_create(C.byref(cfg), self.grid.ctypes.data, self.grid.shape[0], self.grid.shape[1],
                                    C.byref(h))
        if rc != 0:
            raise ValueError("oracle_create: %s" % self.lib.oracle_last_error().decode())
        self.h = h
        self.n_robots, self.n_peds = cfg.n_robots, cfg.n_peds
        o = _cabi.Out()
        self.lib.oracle_outputs(self.h, C.byref(o))
        self.out = {}
        for name, (dt, shape) in _cabi.out_layout(o, self.n_peds, cfg.ped_image_size[0], cfg.ped_image_size[1]).items():
            ptr = getattr(o, name)
            if not ptr:  # a field only the library has (imgenv_out.step_*: the oracle knows no auto-reset)
                continue
            n = int(np.prod(shape))
            buf = (C.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(ptr)
            self.out[name] = np.frombuffer(buf, dtype=dt).reshape(shape)
        rec = C.c_void_p()
        bpr = C.c_int64()
        self.lib.oracle_records(self.h, C.byref(rec), C.byref(bpr))
        buf = (C.c_double * (self.n_robots * _cabi.RECORD_DOUBLES)).from_address(rec.value)
        self.records = np.frombuffer(buf, dtype=np.float64).reshape(self.n_robots, _cabi.RECORD_DOUBLES)
        self.n_local = o.n_local
        self.grid_shape = (o.grid_h, o.grid_w)  # after the load-time resize (grid_map.cpp:28-38)

    def _check(self, rc, what):
        if rc != 0:
            raise ValueError("%s: %s" % (what, self.lib.oracle_last_error().decode()))

    def reset(self, layout):
        b, keep = _cabi.make_reset_batch(layout if isinstance(layout, dict) else layout.as_batch(), self.n_robots,
                                         self.n_peds)
        self._check(self.lib.oracle_reset(self.h, C.byref(b)), "oracle_reset")
        return self.out

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float32).reshape(self.n_local, 3)
        self._check(self.lib.oracle_step(self.h, a.ctypes.data), "oracle_step")
        return self.out

    def step_begin(self, actions):
        a = np.ascontiguousarray(actions, np.float32).reshape(self.n_local, 3)
        self._check(self.lib.oracle_step_begin(self.h, a.ctypes.data), "oracle_step_begin")

    def step_end(self):
        self._check(self.lib.oracle_step_end(self.h), "oracle_step_end")
        return self.out

    def pedinfo(self):
        p = C.c_void_p()
        self._check(self.lib.oracle_pedinfo(self.h, C.byref(p)), "oracle_pedinfo")
        n = self.n_local * max(self.n_peds, 1) * 5
        if self.n_peds == 0:
            return np.zeros((self.n_local, 0, 5), np.float32)
        return np.frombuffer((C.c_float * n).from_address(p.value), np.float32).reshape(self.n_local, self.n_peds, 5).copy()

    def private_grid(self, robot):
        g = np.zeros(self.grid_shape, np.uint8)
        self._check(self.lib.oracle_private_grid(self.h, robot, g.ctypes.data), "oracle_private_grid")
        return g

    def sfm_tree(self):
        """(nodes, member entries, leaf hash, treehash hash) of the social-force crowd's quadtree (oracle_sfm.c: sfm_tree_digest)"""
        out = np.zeros(8, np.uint64)  # (the last four: one bit per agent that is in the tree)
        self._check(self.lib.oracle_sfm_tree(self.h, out.ctypes.data), "oracle_sfm_tree")
        return tuple(int(v) for v in out)

    def grids(self):
        a, b = C.c_void_p(), C.c_void_p()
        self.lib.oracle_grids(self.h, C.byref(a), C.byref(b))
        n = self.grid_shape[0] * self.grid_shape[1]
        obs = np.frombuffer((C.c_uint8 * n).from_address(a.value), np.uint8).reshape(self.grid_shape)
        ped = np.frombuffer((C.c_uint8 * n).from_address(b.value), np.uint8).reshape(self.grid_shape)
        return obs, ped

    def snapshot(self):
        return {k: v.copy() for k, v in self.out.items()}

    def close(self):
        if self.h:
            self.lib.oracle_destroy(self.h)
            self.h = None
