#!/bin/bash
# Kernel trace + stats + timeline of VecImageEnv with device-side auto-reset (tools/host_issue_probe.py device): tools/profile_vec_env.sh <tag>
tag=${1:-vecenv}
out=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o t -- python3 /root/repo/tools/host_issue_probe.py device > $out/${tag}_trace.log 2>&1
cd /root/repo
db=$(ls $out/${tag}_trace/*/*.db $out/${tag}_trace/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python3 tools/rocpd_stats.py $db > $out/${tag}_kernel_stats.txt
[ -n "$db" ] && python3 tools/timeline.py $db 70 > $out/${tag}_timeline.txt 2>/dev/null
rm -rf $out/${tag}_trace
