#!/bin/bash
# round-3 measurement pass on the GPU box: default bench (1 timed pool), c4 Coach iteration on one rank, and a 2-rank gloo rehearsal of the
# Coach iteration on this box's single GPU.  usage: r3_bench.sh <tag>
tag=${1:-r3d}; O=gpurun_out/$tag; mkdir -p $O
python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --coach-iter --config c4 > $O/coach_c4_1rank.json 2> $O/coach_c4_1rank.err
RP_DIST_BACKEND=gloo RP_SINGLE_DEVICE=1 python bench.py --coach-iter --config c4 --gpus 2 --games 8192 > $O/coach_c4_2rank_gloo.json 2> $O/coach_c4_2rank_gloo.err
tail -c 600 $O/bench_c3.json; echo; tail -c 1500 $O/coach_c4_1rank.json; echo; tail -c 1500 $O/coach_c4_2rank_gloo.json; tail -3 $O/*.err
