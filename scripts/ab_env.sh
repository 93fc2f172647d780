#!/bin/bash
# A/B of an environment knob on the default bench (one pool each, alternating) on ONE box.  usage: ab_env.sh <tag> <reps> <VAR> "<values>"
tag=$1; reps=$2; var=$3; vals=$4
O=gpurun_out/$tag; mkdir -p $O
for r in $(seq $reps); do
  for v in $vals; do
    env $var=$v python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_${v}_$r.json 2> $O/bench_${v}_$r.err || { tail -5 $O/bench_${v}_$r.err; exit 1; }
    python - $O/bench_${v}_$r.json "$var=$v" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "%.1f eps/s" % d["value"], "pool %.2f s" % d["pool_seconds"][-1], {k: round(v, 4) for k, v in d["phase_ms_per_launch"].items()}, {k: round(v, 4) for k, v in d["kernel_ms_per_launch"].items()}, "frac_eval %.3f" % d["roofline_evaluator"]["frac"], "roofline %.3f" % d["roofline"]["frac"])
PY
  done
done | tee $O/ab.log
