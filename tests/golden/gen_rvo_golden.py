"""Generate tests/golden/rvo_ref_traces.npz from the REFERENCE's RVO2 (oracle/_ref/librvo_ref.so,
built by oracle/Makefile from /root/reference/src/3rdparty/ervo_ros/src/*.cpp).  Build container only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from rvo_harness import RefRvo, run_scenario  # noqa: E402
from test_oracle_rvo_ref import CASES  # noqa: E402

out = {}
for case, kw in CASES.items():
    for seed in (0, 1):
        out["%s_%d" % (case, seed)] = run_scenario(RefRvo(0.25), seed, **kw)
np.savez_compressed(os.path.join(HERE, "rvo_ref_traces.npz"), **out)
print({k: v.shape for k, v in out.items()})
