#!/bin/bash
# usage (repository root, GPU box): scripts/pmc_tlb.sh <tag>   address-translation counters of the tree kernels, first 150 waves of the default bench
tag=${1:-tlb}
R=$PWD; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE -d $O/t1 --output-format csv -- python3 $R/bench.py --profile-waves 150 --no-cpu-baseline > /dev/null 2> $O/t1.err
python3 - $O/t1 > $O/tlb.md <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | launches | UTCL1 requests | UTCL1 misses | miss share | UTCL2 busy / GUI active |")
print("|---|---|---|---|---|---|")
for k, m in sorted(acc.items()):
    mean = {c: sum(v) / len(v) for c, v in m.items()}
    n = len(next(iter(m.values())))
    rq, ms = mean.get("TCP_UTCL1_REQUEST_sum", 0), mean.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0)
    print("| `%s` | %d | %.0f | %.0f | %.3f | %.3f |" % (k, n, rq, ms, ms / max(rq, 1), mean.get("GRBM_UTCL2_BUSY", 0) / max(mean.get("GRBM_GUI_ACTIVE", 1), 1)))
PY
rm -rf $O/t1
cat $O/tlb.md
