# On the GPU box, after a change to k_view: (1) the whole -m gpu suite on the product build; (2) the parity suites again on a
# build whose resolve step has hardly any room (-DIMGENV_EXP_TINY_RESOLVE: 5 chunk descriptors, 2 result slots), so that the
# "no room" fallbacks of k_view's step (5) run on every robot; (3) optionally (PMC=1) the per-phase instruction counts;
# (4) one bench line.  Results under gpurun_out/.   usage: gpurun -- 'bash tools/check_k_view.sh'
python -m pytest tests -m gpu -x -q > gpurun_out/t_full.log 2>&1; grep -E "passed|failed|error" gpurun_out/t_full.log | tail -3 > gpurun_out/t.log
L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/lib.bak
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -DIMGENV_EXP_TINY_RESOLVE img_env_amd/csrc/imgenv_hip.hip -o $L
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multiworld.py -m gpu -x -q > gpurun_out/t_tiny_full.log 2>&1; grep -E "passed|failed|error" gpurun_out/t_tiny_full.log | tail -3 > gpurun_out/t_tiny.log
cp /tmp/lib.bak $L
if [ -n "$PMC" ]; then EXP_BUILDS="${EXP_BUILDS:-1 4 5 6}" bash tools/run_exp_pmc.sh; fi
python bench.py > gpurun_out/b.log 2>&1
cat gpurun_out/t.log gpurun_out/t_tiny.log
[ -n "$PMC" ] && grep -v "SQ_WAVES \|ACTIVE_INST\|WAIT_INST" gpurun_out/pmc_exp.txt
tail -1 gpurun_out/b.log | cut -c1-260
