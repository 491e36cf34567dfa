"""Generate tests/golden/spawn_ref.npz: episodes placed by the REFERENCE's own ``EnvPos`` (envs/utils/reset_helper.py,
imported unmodified through ref_import) from the spawn sections of its shipped configs and a few synthetic ones, each
after ``random.seed(s)``.

Run in the build container only (needs /root/reference):  python tests/golden/gen_spawn_golden.py

The fixture holds, per case, the YAML subset ``EnvPos`` reads (data of the reference's config files), the seed and
what the reference produced: obstacle shapes / sizes / poses, start and target poses (x, y, yaw) of robots and
pedestrians and the pedestrians' trajectories.  tests/test_host_logic.py holds ``img_env_amd.spawn.EnvPos`` to them
bit for bit.
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
import yaml  # noqa: E402

KEYS = ("robot", "ped_sim", "object", "circle_ranges", "target_min_dist", "env_name", "robot_type")


def subset(cfg):
    out = {k: cfg[k] for k in KEYS if k in cfg}
    out["ped_sim"] = dict(out["ped_sim"])
    return json.loads(json.dumps(out))  # plain lists / dicts / numbers


def run_case(name, cfg, seed):
    envs = ref_import.import_reference_envs()  # noqa: F841  (installs the stand-ins)
    from envs.utils.reset_helper import EnvPos
    ep = EnvPos(cfg)
    random.seed(seed)
    obs, robots, peds = ep.reset()
    nr, npd = cfg["robot"]["total"], cfg["ped_sim"]["total"]
    rec = dict(
        init=np.array([p[:3] for p in ep.init_poses], float).reshape(nr + npd, 3),
        target=np.array([p[:3] for p in ep.target_poses], float).reshape(nr + npd, 3),
        obs_range=np.array(ep.obs_range, float).reshape(len(obs), 4) if obs else np.zeros((0, 4)),
        obs_shape=np.array([0 if o.shape == "circle" else 1 for o in obs], np.int32),
        obs_size=np.array([list(o.size) + [0.0] * (4 - len(o.size)) for o in obs], float).reshape(len(obs), 4),
        obs_quat=np.array([[o.init_pose.orientation.z, o.init_pose.orientation.w] for o in obs], float).reshape(len(obs), 2),
        robot_quat=np.array([[r.init_pose.orientation.z, r.init_pose.orientation.w] for r in robots], float).reshape(nr, 2),
        robot_goal=np.array([[r.goal.x, r.goal.y] for r in robots], float).reshape(nr, 2),
        ped_traj_len=np.array([len(p.trajectory) for p in peds], np.int32),
        ped_traj=np.array([[[t.x, t.y] for t in p.trajectory] + [[0.0, 0.0]] * (2 - len(p.trajectory)) for p in peds],
                          float).reshape(npd, 2, 2),
        circle_range=np.array(ep.circle_range),
    )
    out = {"%s/%s" % (name, k): v for k, v in rec.items()}
    out["%s/cfg" % name] = np.array(json.dumps(subset(cfg)))
    out["%s/seed" % name] = np.array(seed)
    print("%-28s seed %-4d robots %2d peds %2d obstacles %2d circle_range %.3f" % (name, seed, nr, npd, len(obs), ep.circle_range))
    return out


def shipped(name):
    path = os.path.join(ref_import.REFERENCE_ROOT, "envs", "cfg", name)
    with open(path, "r", encoding="utf-8") as f:
        return yaml.load(f.read(), Loader=yaml.FullLoader)


def synthetic():
    """pose types the shipped files do not use: fix, rand_angle, 6-number ranges, range_multi, range_circle_fix + circle_fix"""
    rng4, rng6 = [2.0, 9.0, 2.0, 9.0], [2.0, 9.0, 2.0, 9.0, -1.0, 1.0]
    multi = [[1.5, 3.5, 1.5, 3.5], [7.5, 9.5, 7.5, 9.5], [1.5, 3.5, 7.5, 9.5, 0.0, 0.5]]
    return dict(
        env_name="synthetic", robot_type="diff", circle_ranges=[2.2, 2.6], target_min_dist=1.5,
        robot=dict(total=6, shape=["circle", "rectangle", "circle", "circle", "circle", "circle"],
                   size=[[0, 0, 0.17], [-0.2, 0.2, -0.1, 0.1], [0, 0, 0.2], [0, 0, 0.17], [0, 0, 0.17], [0, 0, 0.17]],
                   begin_poses_type=["fix", "rand_angle", "range", "range_multi", "range_circle_fix", "range_circle_fix"],
                   begin_poses=[[1.0, 1.0, 0.5], [10.0, 1.0, -1.0, 1.0], rng6, multi, [5.5, 5.5], [5.5, 5.5]],
                   # (a fixed or rand_angle start needs a fixed or rand_angle target: with a random one the reference never
                   # leaves its `while reset_init` loop, reset_helper.py:218-300)
                   target_poses_type=["rand_angle", "fix", "range_view", "range_multi", "circle_fix", "range_circle_fix"],
                   target_poses=[[9.0, 9.5, 0.0, 3.0], [4.0, 10.0, 1.0], rng4, multi, [5.5, 5.5], [5.5, 5.5]]),
        ped_sim=dict(total=5, type="rvoscene", max_speed=[0.5] * 5, shape=["leg", "circle", "leg", "circle", "leg"],
                     size=[[0, 0.1, 0.1], [0, 0, 0.17], [0, 0.1, 0.1], [0, 0, 0.17], [0, 0.1, 0.1]],
                     begin_poses_type=["range", "range_circle", "range", "fix", "range"],
                     begin_poses=[rng4, [5.5, 5.5], rng6, [1.0, 10.0, 0.0], rng4],
                     target_poses_type=["range_view", "range_circle", "range", "rand_angle", "fix"],
                     target_poses=[rng4, [5.5, 5.5], rng4, [1.0, 7.0, -0.5, 0.5], [10.0, 10.0, 0.0]], go_back="random"),
        object=dict(total=4, shape=["circle", "rectangle", "circle", "rectangle"],
                    size_range=[[0.2, 0.5], [-0.3, 0.3, -0.2, 0.2], [0.1, 0.3], [-0.15, 0.15, -0.15, 0.15]],
                    poses_type=["range", "range", "fix", "fix"],
                    poses=[[3.0, 8.0, 3.0, 8.0], [3.0, 8.0, 3.0, 8.0, 0.0, 1.57], [6.0, 2.0], [2.0, 6.0, 0.7]]),
    )


if __name__ == "__main__":
    out = {}
    for fname in ("circle.yaml", "test.yaml", "10obs_5ped_baseline.yaml"):
        cfg = shipped(fname)
        for seed in (1, 2, 7):
            out.update(run_case("%s@%d" % (fname.replace(".yaml", ""), seed), cfg, seed))
    for seed in (3, 4, 5, 6):
        out.update(run_case("synthetic@%d" % seed, synthetic(), seed))
    crowded = shipped("circle.yaml")  # a circle too small for its cast: the "circle start failed 50 times" path (reset_helper.py:249-255)
    crowded = dict(crowded, circle_ranges=[0.9, 1.0])
    for seed in (1, 2):
        out.update(run_case("circle_crowded@%d" % seed, crowded, seed))
    path = os.path.join(HERE, "spawn_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KiB, %d cases)" % (path, os.path.getsize(path) / 1024, len([k for k in out if k.endswith("/seed")])))
