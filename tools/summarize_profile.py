#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel-trace --stats + separate --pmc FETCH_SIZE / WRITE_SIZE passes) into a
small text summary for profiles/ and a per-launch HBM traffic figure.

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B
(hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024) and on gfx950 FETCH_SIZE reports exactly half of a wide
coalesced read stream, so the read side is doubled.  usage: summarize_profile.py <dir> <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))


def main():
    d, tag = sys.argv[1], sys.argv[2]
    print("# rocprofv3 summary %s" % tag)
    stats = find(os.path.join(d, "trace"), "*kernel_stats.csv")
    for path in stats:
        print("\n## kernel stats (%s)" % os.path.relpath(path, d))
        rows = list(csv.DictReader(open(path)))
        if rows:
            cols = list(rows[0].keys())
            print(" | ".join(cols))
            for r in rows[:25]:
                print(" | ".join(str(r[c])[:90] for c in cols))
    per_kernel = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        sub = "pmc_fetch" if cname == "FETCH_SIZE" else "pmc_write"
        files = find(os.path.join(d, sub), "*counter_collection.csv")
        acc = defaultdict(lambda: [0.0, 0])
        for path in files:
            for r in csv.DictReader(open(path)):
                if r.get("Counter_Name") != cname:
                    continue
                k = r.get("Kernel_Name", "?")
                acc[k][0] += float(r.get("Counter_Value", 0) or 0)
                acc[k][1] += 1
        # one row per dispatch per counter instance; rocprofv3 may split a counter over dimensions (XCC/instances):
        # sum values per dispatch id first
        acc2 = defaultdict(lambda: defaultdict(float))
        for path in files:
            for r in csv.DictReader(open(path)):
                if r.get("Counter_Name") == cname:
                    acc2[r.get("Kernel_Name", "?")][r.get("Dispatch_Id", "?")] += float(r.get("Counter_Value", 0) or 0)
        for k, disp in acc2.items():
            vals = list(disp.values())
            per_kernel.setdefault(k, {})[cname] = (sum(vals) / len(vals), len(vals))
    print("\n## HBM traffic per launch (PMC, separate passes; FETCH doubled per gfx950 correction)")
    out = {}
    for k, v in sorted(per_kernel.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[0] + kv[1].get("WRITE_SIZE", (0, 0))[0])):
        f, nf = v.get("FETCH_SIZE", (0.0, 0))
        w, nw = v.get("WRITE_SIZE", (0.0, 0))
        hbm = (2.0 * f + w) * 1024.0
        out[k] = {"fetch_size_kib_raw": f, "write_size_kib": w, "hbm_bytes_per_launch": hbm, "launches": max(nf, nw)}
        print("%-70s FETCH_SIZE=%.1f (x2 corr) WRITE_SIZE=%.1f -> %.3f MB/launch (%d launches)" % (k[:70], f, w, hbm / 1e6, max(nf, nw)))
    json.dump(out, open(os.path.join(d, "traffic_by_kernel.json"), "w"), indent=1)
    for log in ("bench_trace.log", "bench_pmc_fetch.log"):
        p = os.path.join(d, log)
        if os.path.exists(p):
            lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if lines:
                print("\n## bench line under %s\n%s" % (log, lines[-1][:3000]))


if __name__ == "__main__":
    main()
