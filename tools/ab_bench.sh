# A/B on ONE GPU box (boxes differ by a few per cent): the bench line of several builds of the library, three times each,
# interleaved.  Builds: img_env_amd/csrc/libimgenv_hip_*.so (git-ignored; build them from the commits / variants to compare)
# and the current library ("new").   usage: gpurun -- 'bash tools/ab_bench.sh'
L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/new.so
ARGS=${AB_ARGS:-"--steps 300 --no-cpu-baseline --no-episode --no-multi-world --passes 3"}
for r in 1 2 3; do
for f in img_env_amd/csrc/libimgenv_hip_*.so /tmp/new.so; do
  v=$(basename $f .so | sed 's/libimgenv_hip_//')
  cp $f $L
  python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x,1) for k,x in d['kernel_us'].items() if x})"
done
done
cp /tmp/new.so $L
