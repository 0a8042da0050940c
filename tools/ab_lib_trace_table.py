#!/usr/bin/env python3
"""Condense the output of tools/ab_lib_trace.sh (stdin or a file): mean per-kernel duration (us) and ms per shoot per library build."""
import collections
import re
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
res = collections.defaultdict(lambda: collections.defaultdict(list))
cur = None
for line in txt.split("\n"):
    m = re.match(r"== (\w+)\s+ms_per_step ([\d.]+)", line)
    if m:
        cur = m.group(1)
        res[cur]["step"].append(float(m.group(2)))
        continue
    m = re.match(r"\| `lago::(\w+)<.*?\| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
    if m and cur:
        res[cur][m.group(1)].append(float(m.group(4)))
ks = ["step", "ad_star3_tile_kernel", "compose3_window_kernel", "zy_forward_kernel", "fluid_xpass2_persist_kernel", "zy_inverse_kernel"]
print("%8s" % "", *["%10s" % k[:10] for k in ks], "  sum5   runs")
for v, d in res.items():
    row = [sum(d[k]) / len(d[k]) if d[k] else 0 for k in ks]
    print("%8s" % v, *["%10.1f" % x for x in row], "%8.1f" % sum(row[1:]), [round(x, 2) for x in d["step"]])
