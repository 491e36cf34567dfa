L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/new.so
for r in 1 2; do
for f in img_env_amd/csrc/libimgenv_hip_vA.so /tmp/new.so; do
  cp $f $L
  python tools/multiworld_probe.py --worlds 512 --robots 8 --peds 6 --scene pedscene --steps 200 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $f) 512x8+6', round(d['value']/1e6,2), round(d['us_per_step'],1))"
  python tools/multiworld_probe.py --worlds 64 --robots 64 --peds 32 --scene pedscene --steps 200 --warmup 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $f) 64x64+32', round(d['value']/1e6,2), round(d['us_per_step'],1))"
done
done
cp /tmp/new.so $L
