#!/bin/bash
# One profile set of ONE BASELINE config on the GPU box:   bash tools/profile_cfg.sh <tag> <cfg2|cfg3|cfg4|cfg5> [sq]
#   * rocprofv3 --kernel-trace --stats of `bench.py --config <cfg>` (100 steps)      -> <tag>_<cfg>_kernel_stats.txt, _timeline.txt
#   * --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, separate runs of a 6-step command -> <tag>_<cfg>_pmc_hbm.txt, <tag>_<cfg>_pmc.json
#   * with `sq`: four SQ counter passes (waves / busy / wait cycles, instruction mix)  -> <tag>_<cfg>_pmc_sq.txt (and into the json)
# Everything lands under gpurun_out/; copy what is judged into profiles/ (<tag>_<cfg>_pmc.json also as profiles/pmc_<cfg>.json, or
# pmc_latest.json for cfg3: what bench.py quotes for a library with the same build id).  The profiled commands never start child
# processes (--no-cpu-baseline): the profiler's preloaded library holds a GPU context before the program starts.
tag=${1:-set}
cfg=${2:-cfg3}
out=/root/repo/gpurun_out
p=$out/${tag}_${cfg}
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-episode --no-multi-world"
S="python3 /root/repo/bench.py --config $cfg --steps 6 --warmup 3 --spinup 0 --passes 1 --no-cpu-baseline --no-episode --no-multi-world"
rocprofv3 --kernel-trace --stats -d ${p}_trace -o t -- $B > ${p}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${p}_fetch -o p --output-format csv -- $S > ${p}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ${p}_write -o p --output-format csv -- $S > ${p}_write.log 2>&1
sq=""
if [ "$3" = "sq" ]; then
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH" \
             "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set -d ${p}_sq_$i -o p --output-format csv -- $S > ${p}_sq_$i.log 2>&1
    sq="$sq ${p}_sq_$i"
  done
fi
cd /root/repo
db=$(ls ${p}_trace/*/*.db ${p}_trace/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 tools/rocpd_stats.py $db > ${p}_kernel_stats.txt
[ -n "$db" ] && python3 tools/timeline.py $db > ${p}_timeline.txt 2>/dev/null
python3 tools/pmc_to_json.py ${p}_fetch ${p}_write ${p}_pmc.json $sq > /dev/null
python3 tools/pmc_summary.py ${p}_fetch ${p}_write > ${p}_pmc_hbm.txt
[ -n "$sq" ] && python3 tools/pmc_summary.py $sq > ${p}_pmc_sq.txt
grep -h '^{' ${p}_trace.log | tail -1 > ${p}_bench_under_trace.json
rm -rf ${p}_trace ${p}_fetch ${p}_write ${p}_sq_1 ${p}_sq_2 ${p}_sq_3 ${p}_sq_4
