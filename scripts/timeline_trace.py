#!/usr/bin/env python3
"""Concurrency view of a rocprofv3 --kernel-trace CSV: over the steady-state tail of the trace, how much wall time has
0, 1, 2, ... kernels in flight, how much has at least one matrix-core kernel in flight, and the kernel sequence of one
wave (one graph launch) on one queue with its gaps.  usage: timeline_trace.py <dir with *_kernel_trace.csv> <out.md> [tail fraction]"""
import csv
import glob
import os
import sys

import numpy as np

MFMA = ("k_resblock", "k_resstage", "k_convpool", "Cijk_", "igemm_", "grouped_conv")


def main(src, out, tail=0.5):
    rows = []
    for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name"),
                             r.get("Queue_Id", "0")))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - (t1 - t0) * tail
    rows = [r for r in rows if r[0] >= cut]
    span = max(r[1] for r in rows) - rows[0][0]
    ev = []
    for s, e, n, q in rows:
        m = any(k in n for k in MFMA)
        ev.append((s, 1, m))
        ev.append((e, -1, m))
    ev.sort()
    hist, mf_busy, cur, curm, last = {}, 0, 0, 0, ev[0][0]
    for t, d, m in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        if curm > 0:
            mf_busy += t - last
        last = t
        cur += d
        if m:
            curm += d
    with open(out, "w") as o:
        o.write("steady-state tail: %.2f ms wall, %d dispatches, %.2f ms summed kernel time\n\n" %
                (span / 1e6, len(rows), sum(r[1] - r[0] for r in rows) / 1e6))
        o.write("| kernels in flight | share of wall |\n|---|---|\n")
        for k in sorted(hist):
            o.write("| %d | %.1f%% |\n" % (k, 100.0 * hist[k] / span))
        o.write("\nwall time with >= 1 matrix-core kernel in flight: %.1f%%\n\n" % (100.0 * mf_busy / span))
        # one wave on the busiest queue: from one k_moves to the next
        qs = {}
        for r in rows:
            qs.setdefault(r[3], []).append(r)
        q = max(qs, key=lambda k: len(qs[k]))
        seq = qs[q]
        starts = [i for i, r in enumerate(seq) if "k_moves" in r[2]]
        if len(starts) >= 3:
            a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
            o.write("one wave on queue %s (%.3f ms from k_moves to the next k_moves):\n\n| kernel | start us | dur us | gap before us |\n|---|---|---|---|\n" %
                    (q, (seq[b][0] - seq[a][0]) / 1e6))
            prev_end = seq[a][0]
            for r in seq[a:b]:
                o.write("| `%s` | %.1f | %.1f | %.1f |\n" % (r[2][:70], (r[0] - seq[a][0]) / 1e3, (r[1] - r[0]) / 1e3, (r[0] - prev_end) / 1e3))
                prev_end = r[1]
            o.write("| (next k_moves) | %.1f | | %.1f |\n" % ((seq[b][0] - seq[a][0]) / 1e3, (seq[b][0] - prev_end) / 1e3))
    print(open(out).read()[:8000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.5)
