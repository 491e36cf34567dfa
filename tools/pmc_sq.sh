#!/bin/bash
# SQ counter passes over a short bench run (on the GPU box): tools/pmc_sq.sh <tag>
cd /tmp && export TMPDIR=/tmp
tag=${1:-sq}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /root/repo/gpurun_out/pmc_${tag}_$i -o p --output-format csv -- python3 /root/repo/bench.py --steps 6 --warmup 3 --spinup 0 --no-cpu-baseline --no-episode --no-multi-world > /root/repo/gpurun_out/pmc_${tag}_$i.log 2>&1
done
