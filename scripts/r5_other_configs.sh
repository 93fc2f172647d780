#!/bin/bash
# c5 window + kernel trace + whole pool, c4 Coach iteration (one rank, one rank over RCCL), c2 bench at the round's last engine state.  usage: r5_other_configs.sh <tag>
tag=${1:-r5y}; R=$PWD; O=$R/gpurun_out/$tag; mkdir -p $O
bash scripts/r3_c5.sh $tag > $O/c5.log 2>&1
python bench.py --config c5 --steps 1 --warmup 0 --budget 600 --no-cpu-baseline > $O/c5_pool.json 2> $O/c5_pool.err
python bench.py --coach-iter --config c4 > $O/coach_c4_1rank.json 2> $O/coach_c4_1rank.err
RP_DIST_FORCE=1 RP_DIST_BACKEND=nccl RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29731 python bench.py --coach-iter --config c4 --games 8192 > $O/coach_c4_1rank_rccl.json 2> $O/coach_c4_1rank_rccl.err
python bench.py --config c2 --steps 1 --warmup 0 --no-cpu-baseline > $O/c2.json 2> $O/c2.err
ls -la $O; tail -c 400 $O/c5_pool.json; echo; tail -c 600 $O/coach_c4_1rank.json; echo; tail -c 300 $O/c2.json
