# bench lines of the other BASELINE configurations on one GPU (c2, c4's per-GPU workload, a bounded window of c5), for the record
set -o pipefail
O=gpurun_out/${1:-extra}; mkdir -p $O
timeout -k 10 200 python3 bench.py --config c2 --steps 2 --warmup 1 --no-cpu-baseline > $O/c2.jsonl 2> $O/c2.err; echo "c2 rc=$?"
timeout -k 10 300 python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/c4.jsonl 2> $O/c4.err; echo "c4 rc=$?"
timeout -k 10 300 python3 bench.py --config c5 --waves 400 --no-cpu-baseline > $O/c5.jsonl 2> $O/c5.err; echo "c5 rc=$?"
for c in c2 c4 c5; do tail -1 $O/$c.jsonl | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:60], d['value'], d['unit'], d.get('expansions_per_s'), d.get('phase_ms_per_launch'))"; done
