# vec_env (1024 envs x (4 + 3), device-side resets) under the library's measurement switches, one box
run() { echo "== $*"; env "$@" python tools/vec_env_probe.py --device-only --steps 400 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['device_reset']; print(round(d['robot_steps_per_s']/1e6,2), round(d['us_per_step'],1))"; }
for r in 1 2; do
run IMGENV_FUSE_MOVE=0
run IMGENV_FUSE_MOVE=1
run IMGENV_FUSE_MOVE=1 IMGENV_RASTER_SPLIT=1 IMGENV_VIEW_NW=2
run IMGENV_FUSE_MOVE=0 IMGENV_RASTER_SPLIT=1 IMGENV_VIEW_NW=2
done
