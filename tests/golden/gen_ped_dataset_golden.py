"""Generate tests/golden/ped_dataset_ref.npz: what the reference's PedTrajectoryDatasetWrapper makes of a recorded-trajectory file.

Run in the build container only (needs /root/reference):  python tests/golden/gen_ped_dataset_golden.py

The reference's own, unmodified class (envs/wrapper/evaluation_wrapper/PedTrajectoryDatasetWrapper.py:15-291) reads a synthetic
file in the layout of the ETH / UCY world-coordinate files it is written for (four rows: frame, pedestrian, y, x; frames every
6th of 15 per second) and builds ``cur_ped_pos_v_datas`` -- per pedestrian of a "world" a series of [x, y, theta, vx, vy] every
``control_hz`` seconds -- for three ranges of pedestrians.  The fixture holds the file's numbers (INPUT) and those series
(EXPECTED), plus the one-line episode records the wrapper appends to its output file for a scripted sequence of (v, w) commands
and episode endings."""
import os
import sys
import tempfile

import numpy as np
import pandas  # noqa: F401  (the real one, before tests/golden/ref_import.py starts answering for missing modules)
import scipy.interpolate  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402


def synthetic_dataset(seed=3, n_peds=9):
    """pedestrians that appear one after the other and walk roughly straight, sampled every 6 frames (2.5 per second at fps 15)"""
    rng = np.random.default_rng(seed)
    rows = []
    start = 0
    for p in range(1, n_peds + 1):
        start += int(rng.integers(0, 4)) * 6
        n = int(rng.integers(5, 12))
        pos = rng.uniform(-3.0, 3.0, 2)
        vel = rng.uniform(-0.6, 0.6, 2)
        for k in range(n):
            rows.append((start + 6 * k, p, pos[1], pos[0]))  # frame, ped, y, x
            pos = pos + vel * 0.4 + rng.normal(0, 0.02, 2)
    a = np.array(rows, dtype=np.float64)
    a = a[np.lexsort((a[:, 0], a[:, 1]))]  # grouped by pedestrian, by frame within
    return a.T  # the files hold one ROW per column name


class _Env:  # what the wrapper wraps here: it only ever calls step / reset on it
    def __init__(self):
        self.resets = []
        self.speeds = None

    def step(self, action):
        return None, None, None, {"speeds": [self.speeds]}

    def reset(self, **kw):
        self.resets.append(kw)
        return "state"


def main():
    envs = ref_import.import_reference_envs()  # noqa: F841  (the stub modules for gym / rospy / cv2 ...)
    from envs.wrapper.evaluation_wrapper.PedTrajectoryDatasetWrapper import PedTrajectoryDatasetWrapper
    data = synthetic_dataset()
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "world.csv")
    np.savetxt(path, data, delimiter=",", fmt="%.10g")
    data = np.loadtxt(path, delimiter=",")  # exactly what the file holds
    out_file = os.path.join(tmp, "log.txt")
    worlds = [[0, 3], [2, 7], [4, 8]]
    cfg = dict(control_hz=0.4, ped_traj_dataset=path, repeated_time_per_env=2, ped_dataset_worlds=worlds, ped_sim=dict(total=0), node_id=0,
               output_file=out_file, offset=[1.4, 14.4, 0.3], swapxy=True, fps=15, start_t=0, max_time=20, scale_x=1, scale_y=1,
               spawn_delay_s=0)
    env = _Env()
    w = PedTrajectoryDatasetWrapper(env, cfg)
    out = {"csv": data, "worlds": np.array(worlds), "offset": np.array(cfg["offset"]), "total0": np.array(cfg["ped_sim"]["total"])}
    # episodes: (commands, dones_info at the end); two per world
    rng = np.random.default_rng(8)
    script = []
    for e in range(6):
        cmds = np.stack([rng.uniform(0, 0.6, int(rng.integers(3, 9))), rng.uniform(-0.9, 0.9, 1)[0] * np.ones(1).repeat(1)], 0) if False else None
        n = int(rng.integers(3, 9))
        cmds = np.stack([rng.uniform(0, 0.6, n), np.round(rng.uniform(-0.9, 0.9, n), 1)], 1)
        script.append((cmds, int(rng.choice([5, 2, 10, 1]))))
    w.reset()  # the first reset: nothing to log, world 0
    for e, (cmds, code) in enumerate(script):
        series = env.resets[-1]["cur_ped_pos_v_datas"]
        out["series_%d" % e] = np.array(series, dtype=np.float64)  # [ped][t][5]
        out["world_%d" % e] = np.array(w.cur_world)
        for v, ww in cmds:
            env.speeds = (float(v), float(ww))
            w.step(None)
        out["cmds_%d" % e] = cmds
        out["code_%d" % e] = np.array(code)
        if e + 1 < len(script):
            w.reset(dones_info=[code])
        else:
            w.out2logfile([code])
    out["log"] = np.array(open(out_file).read())
    np.savez_compressed(os.path.join(HERE, "ped_dataset_ref.npz"), **out)
    print("wrote ped_dataset_ref.npz:", {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("series")})
    print(str(out["log"]))


if __name__ == "__main__":
    main()
