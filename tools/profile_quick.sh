#!/bin/bash
# The short version of profile_round.sh when GPU minutes are scarce: kernel trace + stats + last-steps timeline of the headline
# bench command, and the full bench line.  tools/profile_quick.sh <tag>; results under gpurun_out/<tag>_*.
tag=${1:-quick}
out=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o t -- python3 /root/repo/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-episode --no-multi-world > $out/${tag}_trace.log 2>&1
cd /root/repo
db=$(ls $out/${tag}_trace/*/*.db $out/${tag}_trace/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 tools/rocpd_stats.py $db > $out/${tag}_kernel_stats.txt
[ -n "$db" ] && python3 tools/timeline.py $db > $out/${tag}_timeline.txt 2>/dev/null
python3 bench.py 2> $out/${tag}_bench.err | tail -1 > $out/${tag}_bench.json
rm -rf $out/${tag}_trace
# which library the set describes (imgenv_build_id = hash of the sources + flags; bench.py compares it with the library it runs)
python3 -c "import ctypes;l=ctypes.CDLL('/root/repo/img_env_amd/csrc/libimgenv_hip.so');l.imgenv_build_id.restype=ctypes.c_char_p;print(l.imgenv_build_id().decode())" > $out/${tag}_build_id.txt
