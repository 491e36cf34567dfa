# instruction counts of k_view by phase: SQ counters over the product build, the builds that stop after the crop / after the
# first hits (earlier phases are unchanged by an early return, unlike knock-outs), the resolve knock-out and the final-pass
# knock-out; then the kernel's duration in each build (kernel trace, serial streams)
cd /root/repo
export IMGENV_SERIAL=1
rm -f /tmp/exp_*.so
i=0
for f in "" -DIMGENV_EXP_STOP_AFTER=1 -DIMGENV_EXP_STOP_AFTER=2 -DIMGENV_EXP_STOP_AFTER=3 -DIMGENV_EXP_STOP_AFTER=4; do
  i=$((i+1))
  python3 tools/experiment.py $f --build-only --out=/tmp/exp_$i.so >/dev/null 2>&1
done
cd /tmp && export TMPDIR=/tmp
rm -f /root/repo/gpurun_out/pmc_exp.txt
for i in ${EXP_BUILDS:-1 2 3 4 5}; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d /root/repo/gpurun_out/pmc_exp_$i -o p --output-format csv -- python3 /root/repo/tools/exp_run.py /tmp/exp_$i.so 8 > /root/repo/gpurun_out/pmc_exp_$i.log 2>&1
  echo "== build $i" >> /root/repo/gpurun_out/pmc_exp.txt
  python3 /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmc_exp_$i | grep -A9 "k_view<true, true, false, 1>" >> /root/repo/gpurun_out/pmc_exp.txt
  rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/pmc_expt_$i -o t -- python3 /root/repo/tools/exp_run.py /tmp/exp_$i.so 60 > /dev/null 2>&1
  db=$(ls /root/repo/gpurun_out/pmc_expt_$i/*/*.db /root/repo/gpurun_out/pmc_expt_$i/*.db 2>/dev/null | head -1)
  python3 /root/repo/tools/rocpd_stats.py $db | grep "k_view" >> /root/repo/gpurun_out/pmc_exp.txt
done
rm -rf /root/repo/gpurun_out/pmc_exp_[1234567] /root/repo/gpurun_out/pmc_expt_[1234567]
