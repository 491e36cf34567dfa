#!/bin/bash
# One profile set of a round, on the GPU box: tools/profile_round.sh <tag>
#   kernel trace + stats of the headline bench command, FETCH_SIZE / WRITE_SIZE passes (separate runs), a kernel trace of
#   the many-worlds probe, and the full bench line.  Everything lands under gpurun_out/<tag>_*; copy what is judged into profiles/.
tag=${1:-set}
out=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-episode --no-multi-world"
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o t -- $B > $out/${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_fetch -o p --output-format csv -- python3 /root/repo/bench.py --steps 6 --warmup 3 --spinup 0 --no-cpu-baseline --no-episode --no-multi-world > $out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_write -o p --output-format csv -- python3 /root/repo/bench.py --steps 6 --warmup 3 --spinup 0 --no-cpu-baseline --no-episode --no-multi-world > $out/${tag}_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/${tag}_mw -o t -- python3 /root/repo/tools/multiworld_probe.py --worlds 1024 --robots 8 --peds 4 --steps 100 --warmup 50 > $out/${tag}_mw.log 2>&1
cd /root/repo
db=$(ls $out/${tag}_trace/*/*.db $out/${tag}_trace/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 tools/rocpd_stats.py $db > $out/${tag}_kernel_stats.txt
[ -n "$db" ] && python3 tools/timeline.py $db > $out/${tag}_timeline.txt 2>/dev/null
db=$(ls $out/${tag}_mw/*/*.db $out/${tag}_mw/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 tools/rocpd_stats.py $db > $out/${tag}_mw_kernel_stats.txt
python3 tools/pmc_to_json.py $out/${tag}_fetch $out/${tag}_write $out/${tag}_pmc.json > /dev/null
python3 tools/pmc_summary.py $out/${tag}_fetch $out/${tag}_write > $out/${tag}_pmc_hbm.txt
python3 bench.py 2> $out/${tag}_bench.err | tail -1 > $out/${tag}_bench.json
# keep the returned directory small: the raw traces stay on the box
rm -rf $out/${tag}_trace $out/${tag}_mw
find $out/${tag}_fetch $out/${tag}_write -name "*.db" -delete 2>/dev/null
du -sh $out
