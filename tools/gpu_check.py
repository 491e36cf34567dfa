"""First-contact diagnostic for the GPU box: runs the parity cases and prints every mismatch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from img_env_amd.world import World  # noqa: E402
from oracle_binding import OracleWorld  # noqa: E402
from parity import run_pair  # noqa: E402
from scenarios import random_actions, small_world  # noqa: E402
from test_gpu_parity import CASES  # noqa: E402

print("device:", torch.cuda.get_device_name(0))
only = [a for a in sys.argv[1:] if not a.startswith("{")]
import json  # noqa: E402
for q, a in enumerate(x for x in sys.argv[1:] if x.startswith("{")):   # ad-hoc cases as JSON dicts
    CASES["adhoc%d" % q] = json.loads(a)
    only.append("adhoc%d" % q)
for name, kw in CASES.items():
    if only and name not in only:
        continue
    kw = dict(kw)
    grid, params, layout = small_world(**kw)
    t = time.time()
    gpu, cpu = World(params, grid), OracleWorld(params, grid)
    rng = np.random.default_rng(kw["seed"] + 50)
    acts = [random_actions(rng, kw["n_robots"]) for _ in range(40)]
    fails = run_pair(gpu, cpu, layout, acts)
    c = cpu.snapshot()
    print("%-24s %s  (%.1fs) collisions=%d arrives=%d frozen=%d" % (
        name, "OK" if not fails else "FAIL x%d" % len(fails), time.time() - t, int((c["is_collisions"] > 0).sum()),
        int(c["is_arrives"].sum()), int(c["counters"][2])))
    for s, b in fails[:4]:
        print("    step", s, b)
    gpu.close()
    cpu.close()
