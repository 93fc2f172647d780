#!/bin/bash
# A/B of engine builds on the default bench (one pool each, alternating) on ONE box.  usage: ab_bench.sh <tag> <reps> "<lib1> <lib2> ..." [bench args]
tag=$1; reps=$2; libs=$3; shift 3
O=gpurun_out/$tag; mkdir -p $O
for r in $(seq $reps); do
  for L in $libs; do
    n=$(basename $L .so | sed s/librp_engine_//)
    RP_ENGINE_LIB=$PWD/$L python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $O/bench_${n}_$r.json 2> $O/bench_${n}_$r.err || { tail -5 $O/bench_${n}_$r.err; exit 1; }
    python - $O/bench_${n}_$r.json $n <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "%.1f eps/s" % d["value"], "pool %.2f s" % d["pool_seconds"][-1], {k: round(v, 4) for k, v in d["phase_ms_per_launch"].items()}, {k: round(v, 4) for k, v in d["kernel_ms_per_launch"].items()}, "frac_eval %.3f" % d["roofline_evaluator"]["frac"])
PY
  done
done | tee $O/ab.log
