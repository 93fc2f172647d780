#!/bin/bash
# Round-3 (second session) closing pass, one gpurun call: GPU tests, committed profiles of the default bench, the driver's command,
# c5 window + whole pool, Coach iteration (one rank, two ranks over gloo on one GPU, one rank over RCCL).  usage: r4_final.sh <tag>
tag=${1:-r4z}; R=$PWD; O=$R/gpurun_out/$tag; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests.log
bash scripts/r3_prof.sh $tag > $O/prof.log 2>&1; echo "prof rc=$?"
( time timeout -k 10 560 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err ) 2> $O/driver_time.txt; echo "bench rc=$?"; cat $O/driver_time.txt | tail -3
tail -c 400 $O/bench_driver_command.json; echo
