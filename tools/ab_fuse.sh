for r in 1 2; do
for f in 0 1; do
echo "== FUSE=$f"
IMGENV_FUSE_MOVE=$f python tools/cfg_probe.py cfg2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', round(d['value']/1e6,2), round(d['us_per_step'],1), d['kernel_us'])"
IMGENV_FUSE_MOVE=$f python tools/shipped_probe.py --envs 256 --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shipped256', round(d['value']))"
IMGENV_FUSE_MOVE=$f python tools/cfg2_flags.py 4 2>&1 | tail -1 | cut -c1-300
done
done
