# launches of 1025-4096 robots: two wavefronts per view (IMGENV_VIEW_NW=1: one) and robots / pedestrians in raster blocks of their own
# (IMGENV_RASTER_SPLIT=-1: one block draws both), one box
one() { env "$@" python tools/multiworld_probe.py --worlds $W --robots $R --peds $P --steps 200 --warmup 150 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d.get('us_per_step', 0),1))"; }
for shape in "1024 4 3" "512 4 3" "2048 2 2" "32 128 16" "16 128 16" "64 32 8"; do
set -- $shape; W=$1; R=$2; P=$3
echo "== worlds $W x ($R + $P)"
for r in 1 2; do
echo -n "old  "; one IMGENV_VIEW_NW=1 IMGENV_RASTER_SPLIT=-1
echo -n "new  "; one A=1
done
done
for R in 2048 4096; do
echo "== cfg-2 world, $R robots"
for r in 1 2; do
echo -n "old  "; IMGENV_VIEW_NW=1 python tools/cfg2_flags.py 0 $R 2>&1 | tail -1 | cut -c1-200
echo -n "new  "; python tools/cfg2_flags.py 0 $R 2>&1 | tail -1 | cut -c1-200
done
done
