# shipped geometry, builds img_env_amd/csrc/libimgenv_hip_*.so against the current library on ONE box, interleaved
L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/new.so
for r in 1 2 3; do
for f in img_env_amd/csrc/libimgenv_hip_*.so /tmp/new.so; do
  cp $f $L
  for e in 256 2048; do
    python tools/shipped_probe.py --envs $e --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $f)', $e, round(d['value']), d['kernel_us']['k_crop_big'], d['kernel_us']['k_raster'])"
  done
done
done
cp /tmp/new.so $L
