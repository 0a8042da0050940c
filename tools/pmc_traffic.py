#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in
separate runs as MI355X_MICROARCH.md prescribes: the two counters do not fit one TCC pass).

usage: tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]

Units / corrections (guide, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so the
read side is doubled; WRITE_SIZE is exact for 16 B/lane stores and float atomics.  Narrower access
widths are uncalibrated in the guide -- the table therefore also lists the raw values, and the
torch elementwise copy/add kernels of the same run (known byte counts) serve as calibration rows.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    n = n.replace("at::native::", "").replace("(anonymous namespace)::", "")
    return n[:90]


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    return acc


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w)):
        fk, fn = f.get(k, [0.0, 0])
        wk, wn = w.get(k, [0.0, 0])
        n = max(fn, wn, 1)
        fetch_raw = fk / max(fn, 1) * 1024.0
        write_raw = wk / max(wn, 1) * 1024.0
        rows.append({"kernel": k, "launches": n, "fetch_raw_bytes": fetch_raw, "write_bytes": write_raw,
                     "traffic_bytes": 2.0 * fetch_raw + write_raw})
    rows.sort(key=lambda r: -r["traffic_bytes"] * r["launches"])
    print("| kernel | launches | FETCH_SIZE raw MB | x2 (gfx950) MB | WRITE_SIZE MB | traffic MB / launch |")
    print("|---|---:|---:|---:|---:|---:|")
    for r in rows:
        if r["traffic_bytes"] < 1e6:
            continue
        print(f"| `{r['kernel']}` | {r['launches']} | {r['fetch_raw_bytes']/1e6:.1f} | {2*r['fetch_raw_bytes']/1e6:.1f} | "
              f"{r['write_bytes']/1e6:.1f} | {r['traffic_bytes']/1e6:.1f} |")
    if len(sys.argv) > 3:
        json.dump({r["kernel"]: r for r in rows}, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
