#!/bin/bash
# Two (or more) BUILDS of the library on ONE GPU box, interleaved, under any probe: boxes of the pool differ by a few per cent and
# some alternate between two step times from process to process, so numbers are only ever compared on one box, run after run.
# Builds: img_env_amd/csrc/libimgenv_hip_*.so (git-ignored: build them from the commits to compare, e.g. with
# `python -c "import __graft_entry__ as g, subprocess; subprocess.check_call(g.hip_command('img_env_amd/csrc/libimgenv_hip_old.so'))"`
# in a checkout of the old commit) against the current library.
#   usage: gpurun -- 'bash tools/ab_probe.sh python tools/cfg_probe.py cfg2'      (prints the probe's last line per build and round)
L=img_env_amd/csrc/libimgenv_hip.so
cp $L /tmp/new.so
for r in 1 2 3; do
  for f in img_env_amd/csrc/libimgenv_hip_*.so /tmp/new.so; do
    [ -e "$f" ] || continue
    cp $f $L
    echo -n "$(basename $f)  "; "$@" 2>/dev/null | tail -1 | cut -c1-400
  done
done
cp /tmp/new.so $L
