#!/usr/bin/env python3
"""HBM bytes per launch of the engine's kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes).  Units: rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
for wide coalesced reads, so it is doubled (same guide, section HBM).  The output carries the SHA-256 of the rp_engine.hip it was measured on: bench.py reports `traffic` only while that still matches.
usage: pmc_traffic.py <fetch dir> <write dir> <leaves per launch> <out.json>"""
import csv
import glob
import hashlib
import json
import os
import sys

ENGINE_SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "resource_packing_self_play_amd", "csrc", "rp_engine.hip")

KERNELS = ("k_resstage16", "k_resstage32", "k_convpool32", "k_search", "k_commit", "k_leaf_stem", "k_moves")


def means(src, counter):
    acc = {}
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                name = r.get("Kernel_Name", "")
                for k in KERNELS:
                    if k in name:
                        if k in ("k_resstage32", "k_convpool32"):  # two instantiations per wave: keep them apart by template arguments
                            k = k + name[name.index("<"):name.index(">") + 1] if "<" in name else k
                        a = acc.setdefault(k, [0, 0.0])
                        a[0] += 1; a[1] += float(r.get("Counter_Value", 0) or 0)
    return {k: (n, tot / n) for k, (n, tot) in acc.items()}


def main(fetch_dir, write_dir, leaves, out):
    fe, wr = means(fetch_dir, "FETCH_SIZE"), means(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        f_kib, w_kib = fe.get(k, (0, 0.0))[1], wr.get(k, (0, 0.0))[1]
        res[k] = {"launches": fe.get(k, (0, 0))[0], "fetch_size_kib_raw": f_kib, "write_size_kib": w_kib,
                  "hbm_bytes_per_launch": 2 * f_kib * 1024 + w_kib * 1024, "leaves_per_launch": int(leaves)}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench  # engine_hash(): comment- and white-space-insensitive SHA-256 of the engine source (the same function checks it)
    doc = {"engine_sha256": bench.engine_hash(ENGINE_SRC), "kernels": res}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
