#!/bin/bash
# One complete profile set of a round, on the GPU box: tools/profile_round.sh <tag>
#   * kernel trace + stats and the last-steps timeline of the headline bench command
#   * FETCH_SIZE / WRITE_SIZE passes (separate runs; corrected per the MI355X guide by tools/pmc_to_json.py)
#   * three SQ counter passes (instruction mix, busy / wait cycles) of the same command
#   * kernel traces of the configs that are not the headline: cfg-2, cfg-5, one GPU's share of cfg-4, and 1024 x 8 worlds
#   * the full bench line
# Everything lands under gpurun_out/<tag>_*; copy what is judged into profiles/.  The profiled commands never start child
# processes (--no-cpu-baseline): the profiler's preloaded library holds a GPU context before the program starts.
tag=${1:-set}
out=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-episode --no-multi-world"
S="python3 /root/repo/bench.py --steps 6 --warmup 3 --spinup 0 --no-cpu-baseline --no-episode --no-multi-world"
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o t -- $B > $out/${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_fetch -o p --output-format csv -- $S > $out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_write -o p --output-format csv -- $S > $out/${tag}_write.log 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH" \
           "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out/${tag}_sq_$i -o p --output-format csv -- $S > $out/${tag}_sq_$i.log 2>&1
done
for cfg in cfg2 cfg4 cfg5; do
  rocprofv3 --kernel-trace --stats -d $out/${tag}_$cfg -o t -- python3 /root/repo/tools/cfg_probe.py $cfg > $out/${tag}_${cfg}_probe.json 2> $out/${tag}_$cfg.log
done
rocprofv3 --kernel-trace --stats -d $out/${tag}_mw -o t -- python3 /root/repo/tools/multiworld_probe.py --worlds 1024 --robots 8 --peds 4 --steps 100 --warmup 50 > $out/${tag}_mw.log 2>&1
cd /root/repo
db=$(ls $out/${tag}_trace/*/*.db $out/${tag}_trace/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 tools/rocpd_stats.py $db > $out/${tag}_kernel_stats.txt
[ -n "$db" ] && python3 tools/timeline.py $db > $out/${tag}_timeline.txt 2>/dev/null
for cfg in cfg2 cfg4 cfg5 mw; do
  db=$(ls $out/${tag}_$cfg/*/*.db $out/${tag}_$cfg/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python3 tools/rocpd_stats.py $db > $out/${tag}_${cfg}_kernel_stats.txt
done
python3 tools/pmc_to_json.py $out/${tag}_fetch $out/${tag}_write $out/${tag}_pmc.json $out/${tag}_sq_1 $out/${tag}_sq_2 $out/${tag}_sq_3 > /dev/null
python3 tools/pmc_summary.py $out/${tag}_fetch $out/${tag}_write > $out/${tag}_pmc_hbm.txt
python3 tools/pmc_summary.py $out/${tag}_sq_1 $out/${tag}_sq_2 $out/${tag}_sq_3 $out/${tag}_sq_4 > $out/${tag}_pmc_sq.txt
python3 bench.py 2> $out/${tag}_bench.err | tail -1 > $out/${tag}_bench.json
# the reference's shipped geometry (test.yaml: 400 x 400 views shrunk to 48 x 48, 1000 beams): timeline + kernel stats at 256 envs,
# kernel stats + counters at 2048
bash tools/profile_shipped.sh ${tag}_shipped256 256
bash tools/profile_shipped.sh ${tag}_shipped2048 2048
PMC_CMD="python3 /root/repo/tools/shipped_probe.py --envs 2048 --steps 6" bash tools/profile_pmc.sh ${tag}_shipped2048
# BASELINE cfg-5 (96 x 96 views, 720 beams, 1000 ERVO pedestrians): FETCH / WRITE / SQ passes of a handful of steps
PMC_CMD="python3 /root/repo/tools/cfg5_probe.py 8192 0 short" bash tools/profile_pmc.sh ${tag}_cfg5
# (profile_pmc.sh averages a kernel over ALL its launches, the reset chain's few-robot ones included; the step's own launches alone:)
bash tools/pmc_mem.sh ${tag}_shipped2048
# keep the returned directory small: the raw traces stay on the box
rm -rf $out/${tag}_trace $out/${tag}_mw $out/${tag}_cfg2 $out/${tag}_cfg4 $out/${tag}_cfg5 $out/${tag}_sq_1 $out/${tag}_sq_2 $out/${tag}_sq_3 $out/${tag}_sq_4 $out/${tag}_fetch $out/${tag}_write
du -sh $out
# which library the set describes (imgenv_build_id = hash of the sources + flags; bench.py compares it with the library it runs)
python3 -c "import ctypes;l=ctypes.CDLL('/root/repo/img_env_amd/csrc/libimgenv_hip.so');l.imgenv_build_id.restype=ctypes.c_char_p;print(l.imgenv_build_id().decode())" > $out/${tag}_build_id.txt
