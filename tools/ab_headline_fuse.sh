for r in 1 2 3; do
for f in 0 1; do
echo -n "FUSE=$f  "
IMGENV_FUSE_MOVE=$f python tools/cfg_probe.py cfg3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['us_per_step'],1), d['kernel_us'])"
done
done
