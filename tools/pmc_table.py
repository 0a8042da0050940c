#!/usr/bin/env python3
"""Condense a rocprofv3 --pmc counter_collection CSV: per kernel dispatch (in order) one row of counters."""
import csv
import sys
from collections import OrderedDict

rows = OrderedDict()
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:70])
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for v in rows.values() for c in v})
print("dispatch | kernel | " + " | ".join(names))
for (d, k), v in rows.items():
    if "lago::" not in k:
        continue
    print(f"{d} | {k} | " + " | ".join(f"{v.get(c, float('nan')):.4g}" for c in names))
