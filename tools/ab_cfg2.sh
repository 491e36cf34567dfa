# two builds on cfg-2 (1024 robots, no pedestrians: two launches per step), one box: libimgenv_hip_old.so against the current library
L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/new.so
for r in 1 2 3; do
for f in img_env_amd/csrc/libimgenv_hip_old.so /tmp/new.so; do
  cp $f $L
  echo -n "$(basename $f) "; python tools/cfg_probe.py cfg2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['us_per_step'],2), d['kernel_us'])"
done
done
cp /tmp/new.so $L
