#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters (counter_collection CSV) -> small markdown table.
usage: summarize_pmc.py <dir> <out.md> [kernel-name substring ...]"""
import csv
import glob
import os
import sys


def main(src, out, only=()):
    files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if only and not any(k in r.get("Kernel_Name", "") for k in only):
                    continue
                key = (r.get("Kernel_Name", "?"), r.get("Counter_Name", "?"))
                v = float(r.get("Counter_Value", 0) or 0)
                a = acc.setdefault(key, [0, 0.0])
                a[0] += 1; a[1] += v
    with open(out, "w") as o:
        o.write("| kernel | counter | dispatches | mean per dispatch | total |\n|---|---|---|---|---|\n")
        for (k, c), (n, tot) in sorted(acc.items(), key=lambda kv: (kv[0][0], kv[0][1]))[:200]:
            o.write("| `%s` | %s | %d | %.1f | %.1f |\n" % (k[:100], c, n, tot / n, tot))
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], tuple(sys.argv[3:]))
