# two builds on one box at the shipped geometry, 2048 envs: img_env_amd/csrc/libimgenv_hip_prev.so against the current library
L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/new.so
for r in 1 2 3; do
for f in img_env_amd/csrc/libimgenv_hip_prev.so /tmp/new.so; do
  cp $f $L
  echo -n "$(basename $f) "; python tools/shipped_probe.py --envs 2048 --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['us_per_step'],1), d['kernel_us'])"
done
done
cp /tmp/new.so $L
