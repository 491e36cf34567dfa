"""ctypes mirror of ``include/imgenv.h`` -- the C ABI of the HIP step() library.

The product path loads ``img_env_amd/csrc/libimgenv_hip.so`` and nothing else: there is no CPU
fallback.  :func:`load_library` raises if the extension is missing or was built for another ABI.
"""
import ctypes as C
import os

import numpy as np

ABI_VERSION = 2
RECORD_DOUBLES = 8

SHAPE_CIRCLE, SHAPE_RECTANGLE, SHAPE_LEG = 0, 1, 2
SCENE_EMPTY, SCENE_RVO, SCENE_ERVO, SCENE_PEDSIM, SCENE_DATASET = 0, 1, 2, 3, 4
KTYPE_DIFF, KTYPE_OMNI = 0, 1
FLAG_PRIVATE_GRIDS = 1
FLAG_COMPOSE_DENSE = 2
FLAG_COMPOSE_SPARSE = 4
STEP_ACTIONS_READY = 1  # imgenv_step_flags: the actions are complete when the call is made (include/imgenv.h)
FLAG_LAYER_SUM = 512  # the counting class layer wherever it can run (include/imgenv.h)
FLAG_NO_VIEW_MAPS = 8  # imgenv_out.view_maps not wanted (include/imgenv.h)
FLAG_VIEW_TILED = 32  # views through the tiled kernels (csrc/view_big.h) / through k_view, where both can run
FLAG_VIEW_WAVE = 64
FLAG_AGENT_STATE_EXTRAS = 16  # AgentState.hits_x / hits_y / angular_map as well
FLAG_CHECK_OUTPUTS = 128  # debug: checksum every output array between calls, EINVAL "the caller wrote into imgenv_out.<field>"
FLAG_FULL_REWRITE = 256   # the outputs are copies rewritten in full by every call (the reference's value-copy ownership)
FLAG_CHECK_OUTPUTS_FIRST = 1024  # FLAG_CHECK_OUTPUTS for the handle's first 64 calls only (World's default)
ANGULAR_BINS = 72

SHAPES = {"circle": SHAPE_CIRCLE, "rectangle": SHAPE_RECTANGLE, "leg": SHAPE_LEG}
# Env.msg ped_scene_type strings (scenefactory.h:8-24): anything else is the EmptyScene
SCENES = {"rvoscene": SCENE_RVO, "ervoscene": SCENE_ERVO, "pedscene": SCENE_PEDSIM, "dataset": SCENE_DATASET}
KTYPES = {"diff": KTYPE_DIFF, "omni": KTYPE_OMNI}

_i32, _f32, _f64, _i64 = C.c_int32, C.c_float, C.c_double, C.c_int64
_pi32, _pf32, _pf64 = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)


class Limiter(C.Structure):
    _fields_ = [("has_velocity_limits", _i32), ("has_acceleration_limits", _i32), ("has_jerk_limits", _i32),
                ("min_velocity", _f32), ("max_velocity", _f32), ("min_acceleration", _f32),
                ("max_acceleration", _f32), ("min_jerk", _f32), ("max_jerk", _f32)]


class Cfg(C.Structure):
    _fields_ = [
        ("abi_version", _i32), ("struct_size", _i32),
        ("view_resolution", _f32), ("view_width", _f32), ("view_height", _f32), ("step_hz", _f32),
        ("state_dim", _i32), ("use_laser", _i32), ("range_total", _i32),
        ("view_angle_begin", _f32), ("view_angle_end", _f32), ("view_min_dist", _f32), ("view_max_dist", _f32),
        ("beep_r", _f32), ("ped_ca_p", _f32), ("relation_ped_robo", _i32),
        ("global_resolution", _f32), ("ped_scene_type", _i32), ("n_robots", _i32), ("n_peds", _i32),
        ("robot_ktype", _i32),
        ("robot_shape", _pi32), ("robot_size", _pf32), ("robot_sensor_cfg", _pf32),
        ("limiter_v", Limiter), ("limiter_w", Limiter),
        ("ped_shape", _pi32), ("ped_size", _pf32), ("ped_max_speed", _pf32),
        ("image_size", _i32 * 2), ("ped_image_size", _i32 * 2), ("max_ped", _i32), ("ped_vec_dim", _i32),
        ("ped_image_r", _f64), ("laser_max", _f64), ("laser_norm", _i32),
        ("robot_size_last", _pf64),
        ("ped_safety_space", _f64), ("time_max", _i32),
        ("robot_begin", _i32), ("robot_end", _i32),
        ("device", _i32), ("flags", _i32),
        ("out_arena", C.c_void_p), ("out_arena_bytes", _i64),
        ("n_worlds", _i32), ("reserved_", _i32),
    ]


class ResetBatch(C.Structure):
    _fields_ = [
        ("struct_size", _i32), ("n_obstacles", _i32),
        ("obs_shape", _pi32), ("obs_size", _pf32), ("obs_pose", _pf64),
        ("robot_pose", _pf64), ("robot_goal", _pf64),
        ("ped_pose", _pf64), ("ped_goal", _pf64), ("ped_traj_len", _pi32), ("ped_traj", _pf64),
        ("ped_traj_cap", _i32), ("ignore_obstacle", _i32),
        ("ped_traj_v", _pf64),
    ]


POSE_FIX, POSE_RAND_ANGLE, POSE_RANGE, POSE_RANGE_YAW, POSE_RANGE_VIEW = 0, 1, 2, 3, 4
POSE_RANGE_CIRCLE, POSE_RANGE_CIRCLE_FIX, POSE_CIRCLE_FIX, POSE_RANGE_MULTI = 5, 6, 7, 8


class SpawnAgent(C.Structure):
    _fields_ = [("begin_type", _i32), ("target_type", _i32), ("begin", C.c_double * 6), ("target", C.c_double * 6),
                ("module_size", C.c_double), ("begin_multi", C.POINTER(C.c_double)), ("target_multi", C.POINTER(C.c_double)),
                ("n_begin_multi", _i32), ("n_target_multi", _i32)]


class SpawnObstacle(C.Structure):
    _fields_ = [("shape", _i32), ("pose_type", _i32), ("size_range", C.c_double * 4), ("pose", C.c_double * 6)]


class SpawnCfg(C.Structure):
    _fields_ = [("struct_size", _i32), ("n_robots", _i32), ("n_peds", _i32), ("n_obstacles", _i32),
                ("agents", C.POINTER(SpawnAgent)), ("obstacles", C.POINTER(SpawnObstacle)),
                ("clearance", C.c_double), ("target_min_dist", C.c_double), ("circle_ranges", C.c_double * 2),
                ("go_back", _i32), ("ignore_obstacle", _i32)]


class Out(C.Structure):
    _fields_ = [
        ("struct_size", _i32), ("n_local", _i32), ("view_h", _i32), ("view_w", _i32), ("n_beams", _i32),
        ("state_dim", _i32), ("ped_vec_len", _i32), ("image_h", _i32), ("image_w", _i32), ("grid_h", _i32), ("grid_w", _i32),
        ("vector_states", C.c_void_p), ("view_maps", C.c_void_p), ("sensor_maps", C.c_void_p),
        ("lasers_raw", C.c_void_p), ("lasers", C.c_void_p), ("ped_vector_states", C.c_void_p),
        ("ped_maps", C.c_void_p), ("is_collisions", C.c_void_p), ("is_arrives", C.c_void_p),
        ("step_ds", C.c_void_p), ("ped_min_dists", C.c_void_p),
        ("base_rewards", C.c_void_p), ("base_dones", C.c_void_p),
        ("rewards", C.c_void_p), ("paper_rewards", C.c_void_p), ("dones", C.c_void_p), ("dones_info", C.c_void_p), ("is_clean", C.c_void_p),
        ("robot_pose", C.c_void_p), ("ped_state", C.c_void_p), ("counters", C.c_void_p),
        ("step_rewards", C.c_void_p), ("step_dones", C.c_void_p), ("step_dones_info", C.c_void_p), ("step_is_clean", C.c_void_p),
        ("step_is_arrives", C.c_void_p), ("step_is_collisions", C.c_void_p), ("step_all_down", C.c_void_p),
        ("hits_x", C.c_void_p), ("hits_y", C.c_void_p), ("angular_map", C.c_void_p),
    ]


#: name -> (numpy dtype, shape as a function of the Out header and the world sizes)
def out_layout(o, n_peds, hp, wp):
    R, B = o.n_local, max(o.n_beams, 1)
    return {
        "vector_states": (np.float32, (R, o.state_dim)),
        "view_maps": (np.uint8, (R, o.view_h, o.view_w)),
        "sensor_maps": (np.float16, (R, o.image_h, o.image_w)),
        "lasers_raw": (np.float32, (R, B)),
        "lasers": (np.float64, (R, B)),
        "ped_vector_states": (np.float32, (R, o.ped_vec_len)),
        "ped_maps": (np.float32, (R, 3, hp, wp)),
        "is_collisions": (np.int8, (R,)),
        "is_arrives": (np.uint8, (R,)),
        "step_ds": (np.float64, (R,)),
        "ped_min_dists": (np.float64, (R,)),
        "base_rewards": (np.int32, (R,)),
        "base_dones": (np.uint8, (R,)),
        "rewards": (np.float64, (R,)),
        "paper_rewards": (np.float64, (R,)),
        "dones": (np.uint8, (R,)),
        "dones_info": (np.int32, (R,)),
        "is_clean": (np.uint8, (R,)),
        "robot_pose": (np.float64, (R, 3)),
        "ped_state": (np.float64, (max(n_peds, 1), 4)),
        "counters": (np.int32, (4,)),
        "step_rewards": (np.float64, (R,)),
        "step_dones": (np.uint8, (R,)),
        "step_dones_info": (np.int32, (R,)),
        "step_is_clean": (np.uint8, (R,)),
        "step_is_arrives": (np.uint8, (R,)),
        "step_is_collisions": (np.int8, (R,)),
        "step_all_down": (np.uint8, (R,)),
        "hits_x": (np.float32, (R, B)),          # AgentState's remaining fields: null pointers unless FLAG_AGENT_STATE_EXTRAS
        "hits_y": (np.float32, (R, B)),
        "angular_map": (np.float32, (R, ANGULAR_BINS)),
    }


def _keep(arr, dtype):
    return np.ascontiguousarray(arr, dtype=dtype)


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


def make_cfg(p):
    """Build a :class:`Cfg` from a plain dict of python/numpy values (see ``World``).

    Returns ``(cfg, keepalive)``; ``keepalive`` owns the numpy buffers the struct points into.
    """
    R, P = int(p["n_robots"]), int(p["n_peds"])
    keep = {
        "robot_shape": _keep(p["robot_shape"], np.int32).reshape(R),
        "robot_size": _keep(p["robot_size"], np.float32).reshape(R, 4),
        "robot_sensor_cfg": _keep(p.get("robot_sensor_cfg", np.zeros((R, 2))), np.float32).reshape(R, 2),
        "ped_shape": _keep(p.get("ped_shape", np.zeros(P)), np.int32).reshape(P),
        "ped_size": _keep(p.get("ped_size", np.zeros((P, 6))), np.float32).reshape(P, 6),
        "ped_max_speed": _keep(p.get("ped_max_speed", np.zeros(P)), np.float32).reshape(P),
        "robot_size_last": _keep(p["robot_size_last"], np.float64).reshape(R),
    }
    c = Cfg()
    c.abi_version = ABI_VERSION
    c.struct_size = C.sizeof(Cfg)
    for k in ("view_resolution", "view_width", "view_height", "step_hz", "view_angle_begin", "view_angle_end",
              "view_min_dist", "view_max_dist", "global_resolution"):
        setattr(c, k, float(p[k]))
    c.beep_r = float(p.get("beep_r", 0.0))
    c.ped_ca_p = float(p.get("ped_ca_p", 0.0))
    for k in ("state_dim", "use_laser", "range_total", "relation_ped_robo", "ped_scene_type", "robot_ktype",
              "max_ped", "time_max"):
        setattr(c, k, int(p[k]))
    c.n_robots, c.n_peds = R, P
    c.ped_vec_dim = int(p.get("ped_vec_dim", 7))
    c.laser_norm = int(bool(p.get("laser_norm", True)))
    c.image_size[0], c.image_size[1] = int(p["image_size"][0]), int(p["image_size"][1])
    c.ped_image_size[0], c.ped_image_size[1] = int(p["ped_image_size"][0]), int(p["ped_image_size"][1])
    c.ped_image_r = float(p["ped_image_r"])
    c.laser_max = float(p["laser_max"])
    c.ped_safety_space = float(p["ped_safety_space"])
    for name in ("limiter_v", "limiter_w"):
        lim, src = getattr(c, name), p.get(name) or {}
        lim.has_velocity_limits = int(bool(src.get("has_velocity_limits", False)))
        lim.has_acceleration_limits = int(bool(src.get("has_acceleration_limits", False)))
        lim.has_jerk_limits = int(bool(src.get("has_jerk_limits", False)))
        for f in ("min_velocity", "max_velocity", "min_acceleration", "max_acceleration", "min_jerk", "max_jerk"):
            setattr(lim, f, float(src.get(f, 0.0)))
    c.robot_shape = _ptr(keep["robot_shape"], C.c_int32)
    c.robot_size = _ptr(keep["robot_size"], C.c_float)
    c.robot_sensor_cfg = _ptr(keep["robot_sensor_cfg"], C.c_float)
    c.ped_shape = _ptr(keep["ped_shape"], C.c_int32)
    c.ped_size = _ptr(keep["ped_size"], C.c_float)
    c.ped_max_speed = _ptr(keep["ped_max_speed"], C.c_float)
    c.robot_size_last = _ptr(keep["robot_size_last"], C.c_double)
    c.robot_begin = int(p.get("robot_begin", 0))
    c.robot_end = int(p.get("robot_end", R))
    c.device = int(p.get("device", 0))
    c.flags = int(p.get("flags", 0))
    c.n_worlds = int(p.get("n_worlds", 1))
    c.out_arena = None
    c.out_arena_bytes = 0
    return c, keep


def make_reset_batch(b, n_robots, n_peds):
    """``b``: dict with obstacles / robot / ped arrays (see ``worldgen.ResetLayout.as_batch``)."""
    nob = int(len(b.get("obs_shape", ())))
    P = n_peds
    cap = int(b.get("ped_traj_cap", 2))
    keep = {
        "obs_shape": _keep(b.get("obs_shape", np.zeros(0)), np.int32).reshape(nob),
        "obs_size": _keep(b.get("obs_size", np.zeros((0, 4))), np.float32).reshape(nob, 4),
        "obs_pose": _keep(b.get("obs_pose", np.zeros((0, 4))), np.float64).reshape(nob, 4),
        "robot_pose": _keep(b["robot_pose"], np.float64).reshape(n_robots, 4),
        "robot_goal": _keep(b["robot_goal"], np.float64).reshape(n_robots, 2),
        "ped_pose": _keep(b.get("ped_pose", np.zeros((P, 4))), np.float64).reshape(P, 4),
        "ped_goal": _keep(b.get("ped_goal", np.zeros((P, 2))), np.float64).reshape(P, 2),
        "ped_traj_len": _keep(b.get("ped_traj_len", np.zeros(P)), np.int32).reshape(P),
        "ped_traj": _keep(b.get("ped_traj", np.zeros((P, cap, 3))), np.float64).reshape(P, cap, 3),
    }
    r = ResetBatch()
    r.struct_size = C.sizeof(ResetBatch)
    r.n_obstacles = nob
    r.obs_shape = _ptr(keep["obs_shape"], C.c_int32)
    r.obs_size = _ptr(keep["obs_size"], C.c_float)
    r.obs_pose = _ptr(keep["obs_pose"], C.c_double)
    r.robot_pose = _ptr(keep["robot_pose"], C.c_double)
    r.robot_goal = _ptr(keep["robot_goal"], C.c_double)
    r.ped_pose = _ptr(keep["ped_pose"], C.c_double)
    r.ped_goal = _ptr(keep["ped_goal"], C.c_double)
    r.ped_traj_len = _ptr(keep["ped_traj_len"], C.c_int32)
    r.ped_traj = _ptr(keep["ped_traj"], C.c_double)
    r.ped_traj_cap = cap
    r.ignore_obstacle = int(bool(b.get("ignore_obstacle", False)))
    if b.get("ped_traj_v") is not None:  # dataset scene: recorded velocities beside the recorded positions
        keep["ped_traj_v"] = _keep(b["ped_traj_v"], np.float64).reshape(P, cap, 2)
        r.ped_traj_v = _ptr(keep["ped_traj_v"], C.c_double)
    return r, keep


#: every symbol include/imgenv.h declares
SYMBOLS = ("imgenv_backend", "imgenv_abi_version", "imgenv_last_error", "imgenv_create", "imgenv_arena_bytes",
           "imgenv_destroy", "imgenv_reset", "imgenv_step", "imgenv_step_begin", "imgenv_step_end",
           "imgenv_records", "imgenv_outputs", "imgenv_step_launches", "imgenv_timing", "imgenv_timing_read",
           "imgenv_kernel_name", "imgenv_comm_unique_id", "imgenv_comm_init", "imgenv_comm_info", "imgenv_reset_world", "imgenv_reset_worlds", "imgenv_spawn",
           "imgenv_reset_worlds_spawn", "imgenv_step_autoreset", "imgenv_step_autoreset_device", "imgenv_autoreset_last",
           "imgenv_world_placement", "imgenv_cv_resize_u8", "imgenv_build_id", "imgenv_step_flags", "imgenv_layer_mode")
K_COUNT = 14


def library_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libimgenv_hip.so")


_LIB = None


def bind(lib):
    """Attach argtypes / restypes of include/imgenv.h to a loaded CDLL."""
    lib.imgenv_backend.restype = C.c_char_p
    lib.imgenv_build_id.restype = C.c_char_p
    lib.imgenv_abi_version.restype = C.c_int32
    lib.imgenv_last_error.restype = C.c_char_p
    lib.imgenv_create.argtypes = [C.POINTER(Cfg), C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.imgenv_arena_bytes.argtypes = [C.POINTER(Cfg)]
    lib.imgenv_arena_bytes.restype = C.c_int64
    lib.imgenv_destroy.argtypes = [C.c_void_p]
    lib.imgenv_destroy.restype = None
    lib.imgenv_reset.argtypes = [C.c_void_p, C.POINTER(ResetBatch), C.c_void_p]
    lib.imgenv_reset_world.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ResetBatch), C.c_void_p]
    lib.imgenv_reset_worlds.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(ResetBatch), C.c_void_p]
    lib.imgenv_spawn.argtypes = [C.POINTER(SpawnCfg), C.c_uint64] + [C.c_void_p] * 9
    lib.imgenv_step_autoreset_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(SpawnCfg), C.c_uint64, C.c_void_p]
    lib.imgenv_autoreset_last.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.c_void_p]
    lib.imgenv_world_placement.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)] + [C.c_void_p] * 9
    lib.imgenv_reset_worlds_spawn.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(SpawnCfg),
                                              C.POINTER(C.c_uint64), C.c_void_p]
    lib.imgenv_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.imgenv_step_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.imgenv_step_autoreset.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(SpawnCfg), C.c_uint64, C.POINTER(C.c_int32), C.c_int32,
                                          C.POINTER(C.c_int32), C.c_void_p]
    lib.imgenv_step_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.imgenv_step_end.argtypes = [C.c_void_p, C.c_void_p]
    lib.imgenv_records.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.imgenv_outputs.argtypes = [C.c_void_p, C.POINTER(Out)]
    lib.imgenv_step_launches.argtypes = [C.c_void_p]
    lib.imgenv_layer_mode.argtypes = [C.c_void_p]
    lib.imgenv_timing.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.imgenv_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.imgenv_comm_unique_id.argtypes = [C.c_void_p]
    lib.imgenv_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.imgenv_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.imgenv_cv_resize_u8.argtypes = [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
    lib.imgenv_kernel_name.argtypes = [C.c_int]
    lib.imgenv_kernel_name.restype = C.c_char_p
    return lib


def load_library():
    """Load the HIP extension.  Raises (never falls back) when it is missing or mismatched."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "img_env_amd: HIP extension %s is missing -- run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
    lib = C.CDLL(path)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise RuntimeError("img_env_amd: %s does not export %s" % (path, s))
    bind(lib)
    if lib.imgenv_abi_version() != ABI_VERSION:
        raise RuntimeError("img_env_amd: ABI mismatch: library %d, python %d" % (lib.imgenv_abi_version(), ABI_VERSION))
    _LIB = lib
    return lib
