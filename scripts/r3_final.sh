#!/bin/bash
# Round-3 final measurement pass (one gpurun call): committed profiles of the default bench, c5 window / trace / whole pool, Coach iteration.
tag=${1:-r3r}; R=$PWD; O=$R/gpurun_out/$tag; mkdir -p $O
bash scripts/r3_prof.sh $tag > $O/prof.log 2>&1
bash scripts/r3_c5.sh $tag > $O/c5.log 2>&1
python bench.py --config c5 --steps 1 --warmup 0 --budget 600 --no-cpu-baseline > $O/c5_pool.json 2> $O/c5_pool.err
python bench.py --coach-iter --config c4 > $O/coach_c4_1rank.json 2> $O/coach_c4_1rank.err
RP_DIST_BACKEND=gloo RP_SINGLE_DEVICE=1 python bench.py --coach-iter --config c4 --gpus 2 --games 8192 > $O/coach_c4_2rank_gloo.json 2> $O/coach_c4_2rank_gloo.err
RP_DIST_BACKEND=gloo RP_SINGLE_DEVICE=1 python bench.py --gpus 2 --games 8192 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_2rank_coach_block.json 2> $O/bench_2rank_coach_block.err
RP_DIST_FORCE=1 RP_DIST_BACKEND=nccl RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29731 python bench.py --coach-iter --config c4 --games 8192 > $O/coach_c4_1rank_rccl.json 2> $O/coach_c4_1rank_rccl.err
ls -la $O; tail -c 300 $O/c5_pool.json; tail -c 700 $O/coach_c4_1rank_rccl.json
