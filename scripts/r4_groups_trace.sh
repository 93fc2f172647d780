#!/bin/bash
# kernel trace of a bounded run with several slot groups (are the groups' kernels really in flight together?).  usage: r4_groups_trace.sh <tag> <groups>
tag=${1:-r4b}; G=${2:-2}
R=$PWD; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/bench.py --profile-waves 600 --no-cpu-baseline --groups $G > $O/trace_bench.jsonl 2> $O/trace.err
python3 $R/scripts/summarize_trace.py $O/trace $O/kernel_trace_g$G.md > /dev/null
python3 $R/scripts/timeline_trace.py $O/trace $O/timeline_g$G.md > /dev/null 2>&1
# keep a slice of the raw trace for a closer look
f=$(find $O/trace -name '*_kernel_trace.csv' | head -1)
python3 - "$f" "$O/slice_g$G.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows); lo = int(n * 0.7)
t0 = int(rows[lo]["Start_Timestamp"])
with open(sys.argv[2], "w") as fh:
    for r in rows[lo:lo + 400]:
        fh.write("%s,%d,%d,%s\n" % (r.get("Queue_Id"), int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, (r.get("Kernel_Name") or "")[:40]))
PY
rm -rf $O/trace
ls -la $O
