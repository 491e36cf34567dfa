#!/bin/bash
# ONE build under several settings of the library's measurement switches (tools/README.md), interleaved on one box.
#   usage: gpurun -- 'bash tools/ab_switches.sh IMGENV_EARLY_OBS=0 IMGENV_EARLY_OBS=1 -- python tools/headline_probe.py'
#          a setting may hold several assignments: "IMGENV_EARLY_OBS=0 IMGENV_SERIAL=1"
settings=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do settings+=("$1"); shift; done
shift
for r in 1 2 3; do
  for s in "${settings[@]}"; do
    echo -n "[$s]  "; env $s "$@" 2>/dev/null | tail -1 | cut -c1-400
  done
done
