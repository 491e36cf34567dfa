one() { env "$@" python tools/multiworld_probe.py --worlds $W --robots $R --peds $P --steps 200 --warmup 150 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d.get('us_per_step', 0),1))"; }
for shape in "16 128 16" "512 4 3" "64 32 8" "1024 2 2"; do
set -- $shape; W=$1; R=$2; P=$3
echo "== worlds $W x ($R + $P)"
for r in 1 2; do
echo -n "NW=2  "; one IMGENV_VIEW_NW=2
echo -n "NW=4  "; one IMGENV_VIEW_NW=4
done
done
