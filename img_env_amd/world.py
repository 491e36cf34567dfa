"""``World``: one img_env world living on one GPU, a thin Python object over the C ABI.

torch is plumbing only: it owns the output arena (so every output is a zero-copy ``torch.Tensor``
view of library-written HBM), provides the stream, and runs the RCCL all-gather of a robot-sharded
world.  All simulation work happens in ``csrc/libimgenv_hip.so``.
"""
import ctypes as C
import os

import numpy as np

from . import _cabi

_TORCH_DTYPES = None


def _torch_dtype(np_dtype):
    global _TORCH_DTYPES
    import torch
    if _TORCH_DTYPES is None:
        _TORCH_DTYPES = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                         np.dtype(np.float16): torch.float16, np.dtype(np.uint8): torch.uint8,
                         np.dtype(np.int8): torch.int8, np.dtype(np.int32): torch.int32}
    return _TORCH_DTYPES[np.dtype(np_dtype)]


class World:
    """create -> reset(layout) -> step(actions) ... ; ``out`` maps output names to device tensors."""

    def __init__(self, params, grid, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("img_env_amd.World needs a ROCm GPU (there is no CPU fallback)")
        self.lib = _cabi.load_library()
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index)
        self.params = dict(params)
        self.params["device"] = self.device.index
        # Output ownership.  The reference hands out fresh copies with every step (ROS responses); this library hands out its
        # working copies, read-only by contract (include/imgenv.h).  params["output_guard"] (or IMGENV_OUTPUT_GUARD in the
        # environment, which wins): "first" (THE DEFAULT) = IMGENV_FLAG_CHECK_OUTPUTS_FIRST, a write by the caller during the
        # handle's first 64 calls fails the next call, loudly, and the guard then switches itself off; "check" = the same for
        # good (IMGENV_FLAG_CHECK_OUTPUTS); "copy" = IMGENV_FLAG_FULL_REWRITE, `out` is then a second arena rewritten in full by
        # every call; "none" = the bare C-ABI default.
        guard = os.environ.get("IMGENV_OUTPUT_GUARD") or self.params.pop("output_guard", None) or "first"
        self.params.pop("output_guard", None)
        if guard not in ("none", "first", "check", "copy"):
            raise ValueError("output_guard: none | first | check | copy")
        if guard == "check":
            self.params["flags"] = int(self.params.get("flags", 0)) | _cabi.FLAG_CHECK_OUTPUTS
        elif guard == "copy":
            self.params["flags"] = int(self.params.get("flags", 0)) | _cabi.FLAG_FULL_REWRITE
        elif guard == "first" and not int(self.params.get("flags", 0)) & (_cabi.FLAG_CHECK_OUTPUTS | _cabi.FLAG_FULL_REWRITE):
            self.params["flags"] = int(self.params.get("flags", 0)) | _cabi.FLAG_CHECK_OUTPUTS_FIRST
        self.output_guard = guard
        self.grid = np.ascontiguousarray(grid, np.uint8)
        cfg, self._keep = _cabi.make_cfg(self.params)
        nbytes = self.lib.imgenv_arena_bytes(C.byref(cfg))
        if nbytes <= 0:
            raise ValueError("imgenv_arena_bytes rejected the configuration")
        with torch.cuda.device(self.device):
            self.arena = torch.zeros(int(nbytes), dtype=torch.uint8, device=self.device)
        cfg.out_arena = self.arena.data_ptr()
        cfg.out_arena_bytes = nbytes
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.lib.imgenv_create(C.byref(cfg), self.grid.ctypes.data, self.grid.shape[0], self.grid.shape[1],
                                    C.byref(h))
        if rc != 0:
            raise ValueError("imgenv_create: %s" % self.lib.imgenv_last_error().decode())
        self.h = h
        self.n_robots, self.n_peds = cfg.n_robots, cfg.n_peds
        self.n_worlds = max(1, cfg.n_worlds)
        self._trace = [0.0, 0] if os.environ.get("IMGENV_TRACE_RESET") else None  # seconds inside imgenv_reset_worlds, calls
        self._finished_buf = None
        o = _cabi.Out()
        self._check(self.lib.imgenv_outputs(self.h, C.byref(o)), "imgenv_outputs")
        self.n_local = o.n_local
        base = self.arena.data_ptr()
        self.out = {}
        for name, (dt, shape) in _cabi.out_layout(o, self.n_peds, cfg.ped_image_size[0], cfg.ped_image_size[1]).items():
            if not getattr(o, name):  # an optional output this handle does not produce
                continue
            off = getattr(o, name) - base
            n = int(np.prod(shape)) * np.dtype(dt).itemsize
            self.out[name] = self.arena[off:off + n].view(_torch_dtype(dt)).view(*shape)
        rec, bpr = C.c_void_p(), C.c_int64()
        self._check(self.lib.imgenv_records(self.h, C.byref(rec), C.byref(bpr)), "imgenv_records")
        off = rec.value - base
        n = self.n_robots * _cabi.RECORD_DOUBLES * 8
        # (output_guard "copy": the records stay in the library's private arena -- no tensor for them; such a handle is never a shard)
        self.records = (self.arena[off:off + n].view(torch.float64).view(self.n_robots, _cabi.RECORD_DOUBLES)
                        if 0 <= off and off + n <= int(nbytes) else None)
        self.robot_begin = cfg.robot_begin
        self.robot_end = cfg.robot_end if cfg.robot_end else cfg.n_robots

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.imgenv_last_error().decode()))

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, layout):
        """Reset everything.  A handle of several worlds (``n_worlds`` > 1) takes either one batch of all robots and
        pedestrians (world-major) with an obstacle list shared by every world, or a list of one layout per world."""
        if isinstance(layout, (list, tuple)):
            if len(layout) != self.n_worlds:
                raise ValueError("expected %d layouts, one per world" % self.n_worlds)
            return self.reset_worlds(list(range(self.n_worlds)), layout)
        b, keep = _cabi.make_reset_batch(layout if isinstance(layout, dict) else layout.as_batch(), self.n_robots,
                                         self.n_peds)
        self._check(self.lib.imgenv_reset(self.h, C.byref(b), self._stream()), "imgenv_reset")
        return self.out

    def prepare_reset(self, layout):
        """The C-ABI reset batch of one world's layout, built once and reusable: ``(batch, keepalive)`` for ``reset_worlds``."""
        return _cabi.make_reset_batch(layout if isinstance(layout, dict) else layout.as_batch(),
                                      self.n_robots // self.n_worlds, self.n_peds // self.n_worlds)

    def reset_worlds(self, worlds, layouts):
        """Reset several worlds of a multi-world handle in one call (one set of launches): ``layouts[q]`` -- a layout or
        what ``prepare_reset`` returned for it -- goes to world ``worlds[q]``."""
        n = len(worlds)
        if n == 0:
            return self.out
        prepared = [lay if isinstance(lay, tuple) else self.prepare_reset(lay) for lay in layouts]
        arr = (_cabi.ResetBatch * n)(*[b for b, _ in prepared])
        ids = (C.c_int32 * n)(*[int(k) for k in worlds])
        if self._trace is not None:
            import time
            t0 = time.perf_counter()
            rc = self.lib.imgenv_reset_worlds(self.h, n, ids, arr, self._stream())
            self._trace[0] += time.perf_counter() - t0
            self._trace[1] += 1
            self._check(rc, "imgenv_reset_worlds")
            return self.out
        self._check(self.lib.imgenv_reset_worlds(self.h, n, ids, arr, self._stream()), "imgenv_reset_worlds")
        return self.out

    def reset_worlds_spawn(self, worlds, spawn_cfg, seeds):
        """Reset the listed worlds from fresh random placements made inside the library (``imgenv_reset_worlds_spawn``):
        ``spawn_cfg`` = what ``spawn.make_spawn_cfg`` returned, ``seeds[q]`` seeds world ``worlds[q]``."""
        n = len(worlds)
        if n == 0:
            return self.out
        ids = (C.c_int32 * n)(*[int(k) for k in worlds])
        sd = (C.c_uint64 * n)(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in seeds])
        self._check(self.lib.imgenv_reset_worlds_spawn(self.h, n, ids, C.byref(spawn_cfg[0]), sd, self._stream()),
                    "imgenv_reset_worlds_spawn")
        return self.out

    def reset_world(self, world, layout):
        """Reset ONE world of a multi-world handle (ImageEnv.reset of one env process, yaml_env.py:296-317); the others
        keep their state and their time limits.  ``layout`` holds that world's robots, pedestrians and obstacles."""
        b, keep = _cabi.make_reset_batch(layout if isinstance(layout, dict) else layout.as_batch(),
                                         self.n_robots // self.n_worlds, self.n_peds // self.n_worlds)
        self._check(self.lib.imgenv_reset_world(self.h, int(world), C.byref(b), self._stream()), "imgenv_reset_world")
        return self.out

    def _actions(self, actions):
        import torch
        if not isinstance(actions, torch.Tensor):
            actions = torch.as_tensor(np.ascontiguousarray(actions, np.float32), device=self.device)
        if actions.dtype != torch.float32 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if actions.numel() != self.n_local * 3:
            raise ValueError("actions must be [%d, 3] (v, w, beep)" % self.n_local)
        return actions

    def step(self, actions, actions_ready=None):
        """One step, ordered on the current stream: behind whatever writes ``actions`` there, and behind whatever still reads the
        last step's outputs there.  ``actions_ready`` (``IMGENV_STEP_ACTIONS_READY``) is accepted for compatibility and has had no
        effect since round 6 (include/imgenv.h)."""
        import torch
        if actions_ready is None:
            actions_ready = not isinstance(actions, torch.Tensor)
        a = self._actions(actions)  # (host data: torch's copy from pageable memory returns when the data is on the device)
        self._check(self.lib.imgenv_step_flags(self.h, C.c_void_p(a.data_ptr()), _cabi.STEP_ACTIONS_READY if actions_ready else 0,
                                               self._stream()), "imgenv_step")
        return self.out

    def step_autoreset(self, actions, spawn_cfg, seed0):
        """``imgenv_step_autoreset``: one step, then every world whose robots are all done starts a new episode from a fresh
        placement drawn inside the library (the k-th such world, ascending, from ``seed0 + k``).  Returns the outputs and the
        list of worlds that were reset; ``out['step_*']`` keep what the step itself returned for them."""
        a = self._actions(actions)
        if self._finished_buf is None:
            self._finished_buf = (C.c_int32 * self.n_worlds)()
        n = C.c_int32(0)
        self._check(self.lib.imgenv_step_autoreset(self.h, C.c_void_p(a.data_ptr()), C.byref(spawn_cfg[0]),
                                                   C.c_uint64(int(seed0) & 0xFFFFFFFFFFFFFFFF), self._finished_buf,
                                                   self.n_worlds, C.byref(n), self._stream()), "imgenv_step_autoreset")
        return self.out, list(self._finished_buf[:n.value])

    def step_autoreset_device(self, actions, spawn_cfg, seed0):
        """``imgenv_step_autoreset_device``: the same without the host in the loop -- finished worlds are found, placed and
        reset by kernels alone, the call returns once everything is queued (no synchronisation, nothing read back).  The first
        call fixes ``seed0``: the k-th world reset from then on takes placement ``seed0 + k``."""
        a = self._actions(actions)
        self._check(self.lib.imgenv_step_autoreset_device(self.h, C.c_void_p(a.data_ptr()), C.byref(spawn_cfg[0]),
                                                          C.c_uint64(int(seed0) & 0xFFFFFFFFFFFFFFFF), self._stream()),
                    "imgenv_step_autoreset_device")
        return self.out

    def autoreset_last(self):
        """(worlds the last ``step_autoreset_device`` reset, placement number of the first of them); synchronises the stream"""
        if self._finished_buf is None:
            self._finished_buf = (C.c_int32 * self.n_worlds)()
        n, first = C.c_int32(0), C.c_uint64(0)
        self._check(self.lib.imgenv_autoreset_last(self.h, self._finished_buf, self.n_worlds, C.byref(n), C.byref(first), self._stream()),
                    "imgenv_autoreset_last")
        return list(self._finished_buf[:n.value]), int(first.value)

    def world_placement(self, world, n_obstacles):
        """the placement ``world`` currently runs, as its device-side reset received it: ``(ResetLayout, placement number)``"""
        from .worldgen import ResetLayout
        R, P, O = self.n_robots // self.n_worlds, self.n_peds // self.n_worlds, int(n_obstacles)
        out = dict(robot_pose=np.zeros((R, 4)), robot_goal=np.zeros((R, 2)), ped_pose=np.zeros((P, 4)), ped_goal=np.zeros((P, 2)),
                   ped_traj=np.zeros((P, 2, 3)), ped_traj_len=np.zeros(P, np.int32), obs_shape=np.zeros(O, np.int32),
                   obs_size=np.zeros((O, 4), np.float32), obs_pose=np.zeros((O, 4)))
        serial = C.c_uint64(0)
        self._check(self.lib.imgenv_world_placement(self.h, int(world), C.byref(serial), *[a.ctypes.data for a in out.values()]),
                    "imgenv_world_placement")
        return ResetLayout(**out), int(serial.value)

    def step_begin(self, actions):
        a = self._actions(actions)
        self._check(self.lib.imgenv_step_begin(self.h, C.c_void_p(a.data_ptr()), self._stream()), "imgenv_step_begin")

    def step_end(self):
        self._check(self.lib.imgenv_step_end(self.h, self._stream()), "imgenv_step_end")
        return self.out

    def init_comm(self, rank=None, n_ranks=None):
        """Give the library its own RCCL communicator so that ``step()`` runs the all-gather of a
        robot-sharded world itself.  The 128-byte id is created on rank 0 and broadcast through the
        already-initialised ``torch.distributed`` group (plumbing only)."""
        import torch
        import torch.distributed as dist
        rank = dist.get_rank() if rank is None else rank
        n_ranks = dist.get_world_size() if n_ranks is None else n_ranks
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            self._check(self.lib.imgenv_comm_unique_id(buf), "imgenv_comm_unique_id")
        t = torch.tensor(list(buf), dtype=torch.uint8, device=self.device)
        if n_ranks > 1:
            dist.broadcast(t, src=0)
        raw = bytes(t.cpu().tolist())
        buf2 = (C.c_ubyte * 128).from_buffer_copy(raw)
        self._check(self.lib.imgenv_comm_init(self.h, buf2, rank, n_ranks), "imgenv_comm_init")
        self.native_comm = True

    def comm_info(self):
        """(ranks, rank) as RCCL reports them for the library's own communicator"""
        n, r = C.c_int32(), C.c_int32()
        self._check(self.lib.imgenv_comm_info(self.h, C.byref(n), C.byref(r)), "imgenv_comm_info")
        return n.value, r.value

    def launches(self):
        return self.lib.imgenv_step_launches(self.h)

    def layer_mode(self):
        """what ``imgenv_create`` decided: {"layer": composed | stamped | counting, "shard_bitmaps", "early_observation", "crowd_ahead"}"""
        m = self.lib.imgenv_layer_mode(self.h)
        return dict(layer=("composed", "stamped", "counting")[m & 3], shard_bitmaps=bool(m & 4), early_observation=bool(m & 8),
                    crowd_ahead=bool(m & 16))

    def timing(self, mode, which=-1):
        """0 off, 1 every kernel, 2 only kernel id ``which`` (HIP events on the launch stream)"""
        self._check(self.lib.imgenv_timing(self.h, mode, which), "imgenv_timing")

    def timing_read(self):
        """{kernel name: (total ms, launches)} since the last timing() call"""
        ms = (C.c_double * _cabi.K_COUNT)()
        n = (C.c_int64 * _cabi.K_COUNT)()
        self._check(self.lib.imgenv_timing_read(self.h, ms, n), "imgenv_timing_read")
        return {self.lib.imgenv_kernel_name(q).decode(): (ms[q], n[q]) for q in range(_cabi.K_COUNT)}

    def snapshot(self):
        """host copies (numpy) of every output, after synchronising the stream"""
        import torch
        torch.cuda.synchronize(self.device)
        return {k: v.cpu().numpy().copy() for k, v in self.out.items()}

    def close(self):
        if getattr(self, "h", None):
            self.lib.imgenv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
