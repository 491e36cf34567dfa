#!/bin/bash
# The counter half of profile_round.sh alone: FETCH_SIZE / WRITE_SIZE passes and the three SQ passes of the headline bench
# command (separate runs).  tools/profile_pmc.sh <tag>; results under gpurun_out/<tag>_pmc*.
# PMC_CMD="python3 /root/repo/tools/shipped_probe.py --envs 2048 --steps 6" tools/profile_pmc.sh <tag> profiles another command.
tag=${1:-pmc}
out=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
S=${PMC_CMD:-"python3 /root/repo/bench.py --steps 6 --warmup 3 --spinup 0 --no-cpu-baseline --no-episode --no-multi-world"}
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
cd /root/repo
python3 tools/pmc_to_json.py $out/${tag}_fetch $out/${tag}_write $out/${tag}_pmc.json $out/${tag}_sq_1 $out/${tag}_sq_2 $out/${tag}_sq_3 > /dev/null
python3 tools/pmc_summary.py $out/${tag}_fetch $out/${tag}_write > $out/${tag}_pmc_hbm.txt
python3 tools/pmc_summary.py $out/${tag}_sq_1 $out/${tag}_sq_2 $out/${tag}_sq_3 $out/${tag}_sq_4 > $out/${tag}_pmc_sq.txt
rm -rf $out/${tag}_sq_1 $out/${tag}_sq_2 $out/${tag}_sq_3 $out/${tag}_sq_4 $out/${tag}_fetch $out/${tag}_write
# which library the set describes (imgenv_build_id = hash of the sources + flags; bench.py compares it with the library it runs)
python3 -c "import ctypes;l=ctypes.CDLL('/root/repo/img_env_amd/csrc/libimgenv_hip.so');l.imgenv_build_id.restype=ctypes.c_char_p;print(l.imgenv_build_id().decode())" > $out/${tag}_build_id.txt
