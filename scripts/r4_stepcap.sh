#!/bin/bash
# step_cap sweep at the current engine state (one pool each, one box).  usage: r4_stepcap.sh <tag> "<caps>"
tag=${1:-r4v}; O=gpurun_out/$tag; mkdir -p $O
for c in $2; do
  python bench.py --steps 1 --warmup 0 --no-cpu-baseline --step-cap $c > $O/cap$c.json 2> $O/cap$c.err || { tail -3 $O/cap$c.err; exit 1; }
  python - $O/cap$c.json $c <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("step_cap", sys.argv[2], "%.1f eps/s" % d["value"], "pool %.2f s" % d["pool_seconds"][-1], "waves", d["waves"], {k: round(v, 4) for k, v in d["phase_ms_per_launch"].items()}, "leaves/launch %.0f" % d["roofline"]["leaves_per_launch"])
PY
done | tee $O/sweep.log
