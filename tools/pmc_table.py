#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 counter_collection.csv files.
usage: tools/pmc_table.py <csv> [<csv> ...]  (kernels filtered to lago::*)"""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
        if not k.startswith("lago::"):
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k)
    print("   " + "  ".join(f"{c}={v[0]/v[1]:.4g}" for c, v in sorted(cs.items())))
