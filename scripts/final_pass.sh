set -o pipefail
mkdir -p gpurun_out/r2j
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r2j/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2j/gpu_tests.log
timeout -k 10 500 bash scripts/profile_round.sh r2j > /dev/null 2>&1; echo "profile rc=$?"
cd $GRAFT_REPO_ROOT
( time timeout -k 10 560 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2j/driver_bench.jsonl 2> gpurun_out/r2j/driver_bench.err ) 2> gpurun_out/r2j/driver_time.txt; echo "bench rc=$?"
cat gpurun_out/r2j/driver_time.txt; wc -l gpurun_out/r2j/driver_bench.jsonl
