L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/new.so
for r in 1 2; do
for f in img_env_amd/csrc/libimgenv_hip_prev.so /tmp/new.so; do
  cp $f $L
  echo "== $f"
  python tools/vec_env_probe.py --steps 300 2>&1 | tail -1 | cut -c1-400
  python tools/shipped_probe.py --envs 256 --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']))"
done
done
cp /tmp/new.so $L
