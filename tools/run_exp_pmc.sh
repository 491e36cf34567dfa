# instruction counts of k_view by phase: SQ counters over the product build and the three knock-out builds
cd /root/repo
export IMGENV_SERIAL=1
i=0
for f in "" -DIMGENV_EXP_SKIP_CROP -DIMGENV_EXP_SKIP_HITS -DIMGENV_EXP_SKIP_FINAL; do
  i=$((i+1))
  python3 tools/experiment.py $f --build-only --out=/tmp/exp_$i.so >/dev/null 2>&1
done
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d /root/repo/gpurun_out/pmc_exp_$i -o p --output-format csv -- python3 /root/repo/tools/exp_run.py /tmp/exp_$i.so 8 > /root/repo/gpurun_out/pmc_exp_$i.log 2>&1
  echo "== build $i" >> /root/repo/gpurun_out/pmc_exp.txt
  python3 /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmc_exp_$i | grep -A9 "k_view<true, true, false, 1>" >> /root/repo/gpurun_out/pmc_exp.txt
done
rm -rf /root/repo/gpurun_out/pmc_exp_[1234]
