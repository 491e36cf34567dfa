# The headline bench line several times in fresh processes (the step time has two modes from process to process on the pool's
# boxes), then with other numbers of hardware queues.  usage: gpurun -- 'bash tools/bimodal_probe.sh'
run() { python bench.py --steps 200 --no-cpu-baseline --no-episode --no-multi-world --passes 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x,1) for k,x in d['kernel_us'].items() if x})"; }
for r in 1 2 3 4 5 6; do run default; done
for q in 1 2 8; do for r in 1 2 3; do GPU_MAX_HW_QUEUES=$q run queues$q; done; done
