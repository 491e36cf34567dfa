"""The table of DESIGN.md section 6 from a round's committed evidence:   python tools/round_summary.py [tag]   (no GPU needed)
Reads profiles/<tag>_bench.json (cfg-3) and profiles/<tag>_cfgN_bench.json, the kernel stats of the rocprofv3 traces
(profiles/<tag>_cfgN_kernel_stats.txt) and the counter files (profiles/<tag>_cfgN_pmc.json)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6_98"


def load(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError(path)


def rocprof_avg(cfg, kernel):
    """average duration (us) of the step's launch of `kernel` in the trace: the variant with the most calls"""
    best = None
    for line in open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.txt" % (tag, cfg))):
        m = re.match(r"(?:void )?(\w+)[<(].*?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m and m.group(1) == kernel and (best is None or int(m.group(2)) > best[0]):
            best = (int(m.group(2)), float(m.group(4)))
    return best[1] if best else float("nan")


print("| config | robot-steps/s (median of 5) | ms / step | dominant kernel | avg µs: rocprof trace / bench HIP events | algorithmic bytes per launch -> frac of 8 TB/s | "
      "counter bytes per launch -> frac | all kernels of a step, counters | `spec_policy` (time_max, frozen) | `full_rewrite` | CPU oracle, all cores / 1 thread |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for cfg in ("cfg2", "cfg3", "cfg4", "cfg5"):
    f = os.path.join(ROOT, "profiles", "%s_bench.json" % tag if cfg == "cfg3" else "%s_%s_bench.json" % (tag, cfg))
    d = load(f)
    r = d["roofline"]
    k = r["kernel"]
    alg = r["algorithmic_bytes_per_robot_step"].get(k, 0) * r["units_per_launch"]
    sp, fr, cb = d.get("spec_policy") or {}, d.get("full_rewrite") or {}, d.get("cpu_baseline") or {}
    trace_us = rocprof_avg(cfg, k)
    print("| %s | %.1f M | %.4f | `%s` | %.1f / %.1f | %.1f MB -> %.3f (%.3f by the trace's duration) | %s | %s | %s | %s | %s |" % (
        cfg, d["value"] / 1e6, d["ms_per_step"], k, trace_us, r["kernel_avg_us"], alg / 1e6, r["frac"], alg / (trace_us * 1e-6) / 8e12,
        "%.1f MB -> %.3f" % (r["traffic"] / 1e6, r["frac_by_counters"]) if r.get("traffic") else "--",
        "%.1f MB = %.2f of peak at the step rate" % (r["path_traffic_bytes_per_step"] / 1e6, r["path_traffic_frac"]) if r.get("path_traffic_bytes_per_step") else "--",
        "%.1f M (%d, %.3f)" % (sp["value"] / 1e6, sp["time_max"], sp["frozen_fraction"]) if sp.get("value") else "--",
        "%.1f M" % (fr["value"] / 1e6) if fr.get("value") else "--",
        "%.0f k / %.1f k" % (cb["value"] / 1e3, cb["single_thread_value"] / 1e3) if cb.get("value") else "--"))
for cfg in ("cfg4", "cfg3"):
    p = os.path.join(ROOT, "profiles", "%s_%s_shard_probe.json" % (tag, cfg))
    if os.path.exists(p):
        d = json.load(open(p))
        print("\n%s: shard %.1f us, unsharded %.1f us, ratio %.2f, kernels %s" % (d["probe"], d["us_per_step_shard"], d["us_per_step_unsharded_8192"], d["ratio"],
                                                                               d["per_rank_kernel_us"]))
