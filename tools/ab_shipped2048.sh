cp img_env_amd/csrc/libimgenv_hip.so /tmp/new.so
for r in 1 2; do for f in img_env_amd/csrc/libimgenv_hip_old.so /tmp/new.so; do cp $f img_env_amd/csrc/libimgenv_hip.so; python tools/shipped_probe.py --envs 2048 --steps 100 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['value']/1e6,3), round(d['us_per_step'],1), d['kernel_us'])"; done; done
cp /tmp/new.so img_env_amd/csrc/libimgenv_hip.so
