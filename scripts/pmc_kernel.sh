#!/bin/bash
# usage: pmc_kernel.sh <out dir> <kernel substring> -- <program...>   two rocprofv3 --pmc passes (8 SQ counters each), per-kernel means printed
out=$1; pat=$2; shift 3
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INSTS_SALU -d $out/p1 --output-format csv -- "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM -d $out/p2 --output-format csv -- "$@" > $out/p2.log 2>&1
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m): print("%-28s %16.0f  (%d dispatches)" % (k, m[k], len(acc[k])))
if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print("kernel cycles %.0f ; matrix pipe busy %.3f of 1024 SIMDs" % (cyc, m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)))
if "SQ_WAVES" in m:
    w = m["SQ_WAVES"]
    for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in m: print("%-20s per wave: %.0f cycles" % (k, 4 * m[k] / w))
PY
grep "us per launch" $out/p1.log
