#!/bin/bash
# round-3 (second session) experiment: slot groups per GPU (co-scheduled tree / evaluator phases).  usage: r4_groups.sh <tag>
tag=${1:-r4a}; O=gpurun_out/$tag; mkdir -p $O
for g in 1 2 3; do
  python bench.py --steps 1 --warmup 0 --no-cpu-baseline --groups $g > $O/bench_g$g.json 2> $O/bench_g$g.err || exit 1
  tail -c 400 $O/bench_g$g.json; echo
done
