"""Field-by-field comparison of a HIP world and the oracle on the same seeded scenario.

Bars (BASELINE.json north_star): bit-exact for is_collisions / is_arrives and every other integer
or byte output; 1e-4 for lasers, maps and vectors.  The only arithmetic that is not bit-identical by
construction is sin/cos/atan2 (OCML vs glibc, <= 1 ulp), which enters the robot pose."""
import numpy as np

EXACT = ("is_collisions", "is_arrives", "view_maps", "sensor_maps", "base_rewards", "base_dones", "dones",
         "dones_info", "is_clean", "counters")
CLOSE = ("vector_states", "lasers_raw", "lasers", "ped_vector_states", "ped_maps", "step_ds", "ped_min_dists",
         "rewards", "paper_rewards", "robot_pose", "ped_state")
EXTRAS = ("hits_x", "hits_y", "angular_map")  # AgentState's remaining fields: only handles created with FLAG_AGENT_STATE_EXTRAS have them
TOL = 1e-4


def compare(g, c, fields=EXACT + CLOSE):
    """returns {field: description} for every field that misses its bar"""
    bad = {}
    for k in fields:
        if k in EXTRAS and k not in g:
            continue
        a, b = g[k], c[k]
        if k == "counters":  # [3] is a cumulative bench statistic of the HIP library only
            a, b = a[:3], b[:3]
        if k in EXACT:
            if not np.array_equal(a, b):
                idx = np.argwhere(np.asarray(a) != np.asarray(b))
                bad[k] = "%d mismatches, first at %s: hip %s oracle %s" % (len(idx), tuple(idx[0]), a[tuple(idx[0])],
                                                                          b[tuple(idx[0])])
        else:
            a64, b64 = a.astype(np.float64), b.astype(np.float64)
            fin = np.isfinite(a64) & np.isfinite(b64)
            if not np.array_equal(np.isfinite(a64), np.isfinite(b64)):
                bad[k] = "inf/nan pattern differs"
            elif fin.any():
                err = np.abs(a64[fin] - b64[fin]).max()
                if err > TOL:
                    bad[k] = "max abs err %.3g" % err
    return bad


def run_pair(gpu, cpu, layout, actions_seq, verbose=False, fields=EXACT + CLOSE):
    """reset + step both worlds; returns list of (step, {field: why})"""
    gpu.reset(layout)
    cpu.reset(layout)
    fails = []
    b = compare(gpu.snapshot(), cpu.snapshot(), fields)
    if b:
        fails.append((-1, b))
    for s, a in enumerate(actions_seq):
        gpu.step(a)
        cpu.step(a)
        b = compare(gpu.snapshot(), cpu.snapshot(), fields)
        if b:
            fails.append((s, b))
            if verbose:
                print("step", s, b)
    return fails
