#!/bin/bash
# A/B/... of several engine builds on the stage-kernel probes, alternating, on ONE box (call-to-call spread of a probe is ~3 %).
# usage: ab_probe.sh <tag> <reps> "<lib1> <lib2> ..." <probe args...>   e.g.  ab_probe.sh r4f 3 "build_tmp/a.so build_tmp/b.so" "cp16 30000 40"
tag=$1; reps=$2; libs=$3; shift 3
O=gpurun_out/$tag; mkdir -p $O
for args in "$@"; do
  for r in $(seq $reps); do
    for L in $libs; do
      echo -n "$(basename $L .so | sed s/librp_engine_//) "; RP_ENGINE_LIB=$PWD/$L python scripts/probe_stage32.py $args 2>&1 | grep -v amdgpu
    done
  done
done > $O/ab.log 2>&1
cat $O/ab.log
