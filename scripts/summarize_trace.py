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
    dur, rows_all = {}, []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r.get("Kernel_Name") or r.get("Name")
                d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                dur.setdefault(name, []).append(d)
                rows_all.append((float(r["Start_Timestamp"]), d, name))
    total = sum(sum(v) for v in dur.values())
    rows = sorted(dur.items(), key=lambda kv: -sum(kv[1]))
    with open(out, "w") as o:
        o.write("| kernel | calls | total ms | share | mean us | p50 us | p90 us | max us |\n|---|---|---|---|---|---|---|---|\n")
        for name, v in rows[:60]:
            a = np.array(v) / 1e3
            o.write("| `%s` | %d | %.2f | %.1f%% | %.1f | %.1f | %.1f | %.1f |\n" % (name[:110], len(a), a.sum() / 1e3, 100 * a.sum() * 1e3 / total,
                                                                                  a.mean(), np.percentile(a, 50), np.percentile(a, 90), a.max()))
        o.write("\ntotal kernel time %.2f ms over %d dispatches, %d distinct kernels\n" % (total / 1e6, sum(len(v) for v in dur.values()), len(dur)))
        # the longest launches of the tree-walk kernel: which launch of the run they are and what ran right before them
        rows_all.sort()
        t0 = rows_all[0][0] if rows_all else 0.0
        ks = [(d, k) for k, (st, d, nm) in enumerate(rows_all) if nm.startswith("void k_search")]
        order = {k: n for n, (_, k) in enumerate(ks)}
        if ks:
            o.write("\nlongest `k_search` launches (of %d):\n\n| us | launch # | ms after the first kernel | previous kernel |\n|---|---|---|---|\n" % len(ks))
            for d, k in sorted(ks, reverse=True)[:5]:
                o.write("| %.1f | %d | %.1f | `%s` |\n" % (d / 1e3, order[k], (rows_all[k][0] - t0) / 1e6, rows_all[k - 1][2][:60] if k else "-"))
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
