#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace CSV into a per-kernel table (calls, total, mean, p50, p90, max, share) so that
the summary fits in the repository; the raw trace can then be deleted.  usage: summarize_trace.py <dir with *_kernel_trace.csv> <out.md>"""
import csv
import glob
import os
import sys

import numpy as np


def main(src, out):
    files = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)
    dur = {}
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r.get("Kernel_Name") or r.get("Name")
                d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                dur.setdefault(name, []).append(d)
    total = sum(sum(v) for v in dur.values())
    rows = sorted(dur.items(), key=lambda kv: -sum(kv[1]))
    with open(out, "w") as o:
        o.write("| kernel | calls | total ms | share | mean us | p50 us | p90 us | max us |\n|---|---|---|---|---|---|---|---|\n")
        for name, v in rows[:60]:
            a = np.array(v) / 1e3
            o.write("| `%s` | %d | %.2f | %.1f%% | %.1f | %.1f | %.1f | %.1f |\n" % (name[:110], len(a), a.sum() / 1e3, 100 * a.sum() * 1e3 / total,
                                                                                  a.mean(), np.percentile(a, 50), np.percentile(a, 90), a.max()))
        o.write("\ntotal kernel time %.2f ms over %d dispatches, %d distinct kernels\n" % (total / 1e6, sum(len(v) for v in dur.values()), len(dur)))
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
