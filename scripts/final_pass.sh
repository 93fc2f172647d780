#!/bin/bash
# The round's closing pass on the GPU box (repository root, through gpurun): the GPU test suite, the committed profile set
# (scripts/profile_round.sh: kernel trace, PMC traffic, SQ counters) and the driver's exact bench command with its wall time.
#   usage: bash scripts/final_pass.sh <tag>      -> gpurun_out/<tag>/
set -o pipefail
tag=${1:-final}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/$tag/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/$tag/gpu_tests.log
timeout -k 10 500 bash scripts/profile_round.sh $tag > /dev/null 2>&1; echo "profile rc=$?"
cd $GRAFT_REPO_ROOT
( time timeout -k 10 560 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/driver_bench.jsonl 2> gpurun_out/$tag/driver_bench.err ) 2> gpurun_out/$tag/driver_time.txt; echo "bench rc=$?"
cat gpurun_out/$tag/driver_time.txt; wc -l gpurun_out/$tag/driver_bench.jsonl
