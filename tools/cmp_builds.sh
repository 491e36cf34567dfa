cd /tmp && export TMPDIR=/tmp
export IMGENV_SERIAL=1
for v in old new; do
  so=/root/repo/img_env_amd/csrc/libimgenv_hip.so; [ $v = old ] && so=/root/repo/img_env_amd/csrc/libimgenv_hip_old.so
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d /root/repo/gpurun_out/pmc_cmp_$v -o p --output-format csv -- python3 /root/repo/tools/exp_run.py $so 8 > /root/repo/gpurun_out/pmc_cmp_$v.log 2>&1
  echo "== $v" >> /root/repo/gpurun_out/pmc_cmp.txt
  python3 /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmc_cmp_$v | grep -A9 "k_view<true, true, false, 1>" >> /root/repo/gpurun_out/pmc_cmp.txt
  rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/pmc_cmpt_$v -o t -- python3 /root/repo/tools/exp_run.py $so 60 > /dev/null 2>&1
  db=$(ls /root/repo/gpurun_out/pmc_cmpt_$v/*/*.db /root/repo/gpurun_out/pmc_cmpt_$v/*.db 2>/dev/null | head -1)
  python3 /root/repo/tools/rocpd_stats.py $db | grep "k_view" >> /root/repo/gpurun_out/pmc_cmp.txt
done
rm -rf /root/repo/gpurun_out/pmc_cmp_old /root/repo/gpurun_out/pmc_cmp_new /root/repo/gpurun_out/pmc_cmpt_old /root/repo/gpurun_out/pmc_cmpt_new
