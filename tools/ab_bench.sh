# A/B on ONE GPU box (boxes differ by a few per cent): the bench line of img_env_amd/csrc/libimgenv_hip_old.so (build it from the
# commit to compare with, it is git-ignored) against the current library, three times each, interleaved.
# usage: gpurun -- 'bash tools/ab_bench.sh'
L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/new.so
for r in 1 2 3; do
for v in old new; do
  if [ $v = old ]; then cp img_env_amd/csrc/libimgenv_hip_old.so $L; else cp /tmp/new.so $L; fi
  python bench.py --steps 300 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x,1) for k,x in d['kernel_us'].items()})"
done
done
cp /tmp/new.so $L
