#!/bin/bash
# Memory-path counters of one command, per dispatch: which level of the cache hierarchy a kernel's loads are served from.
# (a pass with the TA_* counters hung the profiler on this image: left out)
# tools/pmc_mem.sh <tag> [command...]; CSVs summarised by tools/pmc_big_launches.py
tag=${1:-mem}; shift
out=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
S=${*:-"python3 /root/repo/tools/shipped_probe.py --envs 2048 --steps 6"}
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $out/${tag}_mem_$i -o p --output-format csv -- $S > $out/${tag}_mem_$i.log 2>&1
done
cd /root/repo
python3 tools/pmc_big_launches.py $out/${tag}_mem_* > $out/${tag}_mem.txt
for d in $out/${tag}_mem_*; do [ -d $d ] && rm -rf $d; done
