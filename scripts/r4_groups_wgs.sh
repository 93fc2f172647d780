#!/bin/bash
# experiment: slot groups co-scheduled with the persistent stage kernels at ONE workgroup per CU (room for the other group's tree kernels)
tag=${1:-r4k}; O=gpurun_out/$tag; mkdir -p $O
run() { n=$1; shift; "$@" python bench.py --steps 1 --warmup 0 --no-cpu-baseline $BARGS > $O/$n.json 2> $O/$n.err || { tail -3 $O/$n.err; return; }
  python - $O/$n.json $n <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "%.1f eps/s" % d["value"], "pool %.2f s" % d["pool_seconds"][-1])
PY
}
BARGS="--groups 1" run g1_wgs2 env
BARGS="--groups 1" run g1_wgs1 env RP_STAGE16_WGS=1 RP_CONVPOOL_WGS=1 RP_STAGE32_WGS=1
BARGS="--groups 2" run g2_wgs1 env RP_STAGE16_WGS=1 RP_CONVPOOL_WGS=1 RP_STAGE32_WGS=1
BARGS="--groups 3" run g3_wgs1 env RP_STAGE16_WGS=1 RP_CONVPOOL_WGS=1 RP_STAGE32_WGS=1
BARGS="--groups 4" run g4_wgs1 env RP_STAGE16_WGS=1 RP_CONVPOOL_WGS=1 RP_STAGE32_WGS=1
