# cfg-5 (96 x 96 views: the LDS bounds the occupancy) with one / two / four wavefronts per view, one box
for r in 1 2; do
for nw in 1 2 4; do
echo "== cfg5 IMGENV_VIEW_NW=$nw"
IMGENV_VIEW_NW=$nw python tools/cfg5_probe.py 8192 2>&1 | tail -2 | cut -c1-400
done
done
