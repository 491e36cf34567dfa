# the step's move inside the raster launch (k_move_raster) forced off / on, one box: pedestrian-free handles beyond 4096 robots
one() { env "$@" python tools/multiworld_probe.py --worlds $W --robots $R --peds $P --steps 200 --warmup 150 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d.get('us_per_step', 0),1))"; }
for shape in "8192 1 0" "64 128 0" "2048 2 0"; do
set -- $shape; W=$1; R=$2; P=$3
echo "== worlds $W x ($R + $P)"
for r in 1 2; do
echo -n "off  "; one IMGENV_FUSE_MOVE=0
echo -n "on   "; one IMGENV_FUSE_MOVE=1
done
done
