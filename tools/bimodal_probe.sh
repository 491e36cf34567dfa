# The headline bench line several times in fresh processes (the step time is bimodal across processes: see profiles/README.md),
# then with different numbers of hardware queues.  usage: gpurun -- 'bash tools/bimodal_probe.sh'
run() { python bench.py --steps 300 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e6,2), round(d['ms_per_step'],4), {k: round(x,1) for k,x in d['kernel_us'].items()})"; }
for r in 1 2 3 4; do run full ""; done
for q in 3 5 6; do for r in 1 2; do GPU_MAX_HW_QUEUES=$q run queues$q "--no-cpu-baseline --no-episode --no-multi-world"; done; done
for r in 1 2 3; do run noepisode "--no-cpu-baseline"; done
