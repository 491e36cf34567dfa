"""YAML config (the reference's schema, envs/cfg/*.yaml) -> C-ABI parameter dict.

Follows what the reference copies from the YAML into ``InitEnv.srv`` and keeps on the Python side:
``ImageEnv._init_static_param`` / ``_init_req`` (envs/env/yaml_env.py:133-209) and
``EnvPos.init_robot`` / ``init_ped`` / ``_init_speed_limiter`` (envs/utils/reset_helper.py:348-412).
"""
import os

import numpy as np
import yaml

from . import _cabi


def read_yaml(file):
    """envs/__init__.py:9-18"""
    with open(file, "r", encoding="utf-8") as f:
        return yaml.load(f.read(), Loader=yaml.FullLoader)


def load_map(cfg):
    """The map image in its own pixels: ``imread(map_file, IMREAD_GRAYSCALE)`` of GridMap::read_image (grid_map.cpp:28-38),
    or an in-memory grid given as ``cfg['global_map']['map_array']`` (extension used by the synthetic worlds).  The resize to
    the view resolution (cv::resize INTER_LINEAR) happens inside ``imgenv_create``."""
    gm = cfg["global_map"]
    if gm.get("map_array") is not None:
        return np.ascontiguousarray(gm["map_array"], np.uint8)
    from PIL import Image
    path = gm["map_file"]
    if not os.path.isabs(path):
        for base in (cfg.get("map_dir"), os.path.join(os.path.dirname(__file__), "maps"), os.getcwd()):
            if base and os.path.exists(os.path.join(base, path)):
                path = os.path.join(base, path)
                break
    im = Image.open(path)
    if im.mode in ("L", "1"):
        return np.array(im.convert("L"), np.uint8)
    # cv::imread(IMREAD_GRAYSCALE) of a colour file: Y = 0.299 R + 0.587 G + 0.114 B in OpenCV's fixed point
    # ((R * 4899 + G * 9617 + B * 1868 + 8192) >> 14); equal to the channel value for the grey-in-RGB maps the reference ships
    rgb = np.array(im.convert("RGB"), np.int64)
    return ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def _limiter(cfg, key, defaults):
    src = cfg.get(key) or {}
    out = {"has_velocity_limits": False, "has_acceleration_limits": False, "has_jerk_limits": False,
           "min_velocity": 0.0, "max_velocity": 0.0, "min_acceleration": 0.0, "max_acceleration": 0.0,
           "min_jerk": 0.0, "max_jerk": 0.0}
    if src:  # reset_helper.py:351-370: only filled when the YAML block exists
        out.update(defaults)
        out.update({k: src[k] for k in src if k in out})
    return out


def params_from_cfg(cfg):
    """reference YAML dict -> dict accepted by ``_cabi.make_cfg``"""
    R = int(cfg["robot"]["total"])
    P = int(cfg["ped_sim"]["total"])
    rshape = np.array([_cabi.SHAPES[s] for s in cfg["robot"]["shape"][:R]], np.int32)
    rsize = np.zeros((R, 4), np.float32)
    rlast = np.zeros(R)
    for j in range(R):
        s = cfg["robot"]["size"][j]
        rsize[j, :len(s)] = s
        rlast[j] = s[-1]
    sens = np.zeros((R, 2), np.float32)
    if cfg["robot"].get("sensor_cfgs"):
        for j in range(R):
            sens[j] = cfg["robot"]["sensor_cfgs"][j][:2]
    pshape = np.zeros(P, np.int32)
    psize = np.zeros((P, 6), np.float32)
    for j in range(P):
        sh = cfg["ped_sim"]["shape"][j]
        s = list(cfg["ped_sim"]["size"][j])
        if sh == "leg":  # reset_helper.py:400-404
            s = s + [s[0], -s[1], s[2]]
        pshape[j] = _cabi.SHAPES[sh]
        psize[j, :len(s)] = s
    scene = _cabi.SCENES.get(cfg["ped_sim"].get("type", ""), _cabi.SCENE_EMPTY) if P > 0 else _cabi.SCENE_EMPTY
    return dict(
        view_resolution=cfg["view_map"]["resolution"], global_resolution=cfg["global_map"]["resolution"],
        view_width=cfg["view_map"]["width"], view_height=cfg["view_map"]["height"],
        step_hz=cfg["control_hz"], state_dim=cfg["state_dim"], use_laser=int(bool(cfg["use_laser"])),
        range_total=cfg["range_total"], view_angle_begin=cfg["view_angle_begin"],
        view_angle_end=cfg["view_angle_end"], view_min_dist=cfg["view_min_dist"],
        view_max_dist=cfg["view_max_dist"], relation_ped_robo=int(cfg["relation_ped_robo"]),
        # beep_r / ped_ca_p are never forwarded by yaml_env.py:183-200 => 0 at the node
        beep_r=0.0, ped_ca_p=0.0,
        ped_scene_type=scene, robot_ktype=_cabi.KTYPES[cfg["robot_type"]], n_robots=R, n_peds=P,
        robot_shape=rshape, robot_size=rsize, robot_sensor_cfg=sens, robot_size_last=rlast,
        limiter_v=_limiter(cfg, "speed_limiter_v", dict(min_velocity=0, max_velocity=0.6, min_acceleration=-2,
                                                         max_acceleration=2, min_jerk=-2, max_jerk=2)),
        limiter_w=_limiter(cfg, "speed_limiter_w", dict(min_velocity=-0.9, max_velocity=0.9, min_acceleration=-2,
                                                         max_acceleration=2, min_jerk=-2, max_jerk=2)),
        ped_shape=pshape, ped_size=psize,
        ped_max_speed=np.array(cfg["ped_sim"]["max_speed"][:P], np.float32) if P else np.zeros(0, np.float32),
        image_size=tuple(cfg["image_size"]), ped_image_size=tuple(cfg["ped_image_size"]),
        max_ped=cfg["max_ped"], ped_vec_dim=cfg["ped_vec_dim"], ped_image_r=cfg["ped_image_r"],
        laser_max=cfg["laser_max"], laser_norm=cfg.get("laser_norm", True),
        ped_safety_space=cfg["ped_safety_space"], time_max=cfg["time_max"],
        flags=int(cfg.get("flags", 0)),  # IMGENV_FLAG_* (not a reference key: library knobs such as the compose mode)
    )
