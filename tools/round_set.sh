#!/bin/bash
# The complete evidence set of a round, on ONE GPU box:   gpurun --timeout 4200 -- 'bash tools/round_set.sh [tag] > gpurun_out/<tag>_round_set.log 2>&1'
#   1. tools/profile_cfg.sh for every BASELINE config bench.py can time (kernel stats, timeline, FETCH / WRITE passes; SQ passes for
#      cfg3 and cfg5)
#   2. the counters of THIS build put where bench.py looks for them (profiles/pmc_latest.json, pmc_cfgN.json: on the box's copy of
#      the tree; copy gpurun_out/<tag>_cfgN_pmc.json into profiles/ yourself to commit them)
#   3. the bench line of every config (`roofline.traffic` from those counters: same build id), on the same box
#   4. rank 0 of the 8-GPU layouts of cfg-4 and cfg-3 (tools/shard_probe.py)
#   5. the whole GPU test-suite and smoke()
tag=${1:-r6_98}
set -x
cd /root/repo
for c in "cfg3 sq" "cfg2" "cfg4" "cfg5 sq"; do bash tools/profile_cfg.sh $tag $c; done
cp gpurun_out/${tag}_cfg3_pmc.json profiles/pmc_latest.json
for c in cfg2 cfg4 cfg5; do cp gpurun_out/${tag}_${c}_pmc.json profiles/pmc_$c.json; done
python bench.py 2> gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench.json
for c in cfg2 cfg4 cfg5; do python bench.py --config $c --no-multi-world 2> gpurun_out/${tag}_${c}_bench.err | tail -1 > gpurun_out/${tag}_${c}_bench.json; done
python tools/shard_probe.py cfg4 --time-max 1000 --out gpurun_out/${tag}_cfg4_shard_probe.json > /dev/null 2>&1
python tools/shard_probe.py cfg3 --time-max 1000 --out gpurun_out/${tag}_cfg3_shard_probe.json > /dev/null 2>&1
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -12 > gpurun_out/${tag}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
