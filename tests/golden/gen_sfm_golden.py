"""Generate tests/golden/sfm_ref_traces.npz from the REFERENCE's libpedsim (oracle/_ref/libpedsim_ref.so, built
by oracle/Makefile from /root/reference/src/3rdparty/pedsimros/src/*.cpp).  Each scenario runs in a fresh
process because libpedsim's random vmax stream and libc rand() are process-global.  Build container only."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

if len(sys.argv) > 1:
    from sfm_harness import RefSfm, run_scenario
    from test_oracle_sfm_ref import CASES
    case = sys.argv[1]
    trace, vmax = run_scenario(RefSfm, 0, **CASES[case])
    np.savez(sys.argv[2], trace=trace, vmax=vmax)
else:
    from test_oracle_sfm_ref import CASES
    out = {}
    for case in CASES:
        tmp = os.path.join("/tmp", "sfm_%s.npz" % case)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), case, tmp])
        z = np.load(tmp)
        out[case], out[case + "_vmax"] = z["trace"], z["vmax"]
    np.savez_compressed(os.path.join(HERE, "sfm_ref_traces.npz"), **out)
    print({k: v.shape for k, v in out.items()})
