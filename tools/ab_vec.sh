# two builds on one box: img_env_amd/csrc/libimgenv_hip_prev.so against the current library (vec_env with device-side resets, shipped 256 envs)
L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/new.so
for r in 1 2 3; do
for f in img_env_amd/csrc/libimgenv_hip_prev.so /tmp/new.so; do
  cp $f $L
  echo "== $f"
  python tools/vec_env_probe.py --device-only --steps 400 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['device_reset']; print('vec_env', round(d['robot_steps_per_s']/1e6,2), round(d['us_per_step'],1))"
  python tools/shipped_probe.py --envs 256 --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shipped256', round(d['value']))"
done
done
cp /tmp/new.so $L
