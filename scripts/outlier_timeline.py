#!/usr/bin/env python3
"""Places the long `__amd_rocclr_fillBufferAligned` dispatches and the launch-time outliers of the engine's kernels on ONE timeline of a
rocprofv3 --kernel-trace CSV: every dispatch longer than `min_ms`, with start / end relative to the first kernel, the kernel dispatched
before and after it on the same queue, and whether any other listed dispatch overlaps it or lies within `near_ms` of it.
usage: outlier_timeline.py <dir with *_kernel_trace.csv> <out.md> [min_ms=5] [near_ms=50]"""
import csv
import glob
import os
import sys


def main(src, out, min_ms=5.0, near_ms=50.0):
    rows = []
    for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Kernel_Name") or r.get("Name"))[:70], r.get("Queue_Id", "0")))
    rows.sort()
    t0 = rows[0][0]
    long_ = [(i, r) for i, r in enumerate(rows) if (r[1] - r[0]) / 1e6 >= min_ms]
    lines = ["| # | kernel | start ms | end ms | duration ms | previous dispatch | next dispatch | other long dispatch within %.0f ms |" % near_ms, "|---|---|---|---|---|---|---|---|"]
    for i, (s, e, n, q) in long_:
        prev = next((rows[j][2] for j in range(i - 1, -1, -1) if rows[j][3] == q), "-")
        nxt = next((rows[j][2] for j in range(i + 1, len(rows)) if rows[j][3] == q), "-")
        near = []
        for j, (s2, e2, n2, _) in long_:
            if j == i:
                continue
            gap = max(s2 - e, s - e2, 0) / 1e6
            if gap <= near_ms:
                near.append("%s (%s)" % (n2.split("(")[0][:28], "overlaps" if gap == 0 else "%.1f ms away" % gap))
        lines.append("| %d | `%s` | %.1f | %.1f | %.2f | `%s` | `%s` | %s |" % (i, n, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, prev[:40], nxt[:40], "; ".join(near) or "none"))
    span = (max(r[1] for r in rows) - t0) / 1e6
    fills = [r for _, r in long_ if "fillBuffer" in r[2]]
    others = [r for _, r in long_ if "fillBuffer" not in r[2]]
    head = ["# Long dispatches on one timeline", "",
            "%d dispatches over %.0f ms; %d take >= %.1f ms: %d `fillBufferAligned`, %d kernels of the engine / evaluator." % (len(rows), span, len(long_), min_ms, len(fills), len(others)), ""]
    open(out, "w").write("\n".join(head + lines) + "\n")
    print("\n".join(head + lines))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2], float(a[3]) if len(a) > 3 else 5.0, float(a[4]) if len(a) > 4 else 50.0)
