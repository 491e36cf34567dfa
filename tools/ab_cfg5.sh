L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/new.so
for r in 1 2 3; do
for f in img_env_amd/csrc/libimgenv_hip_old.so /tmp/new.so; do
  cp $f $L
  echo -n "$(basename $f) "; python tools/cfg5_probe.py 8192 2>&1 | tail -2 | tr '\n' ' '; echo
done
done
cp /tmp/new.so $L
