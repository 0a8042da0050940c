#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` (from --kernel-trace --stats) into a short table:
kernel names are cut at the argument list and template noise is trimmed.
usage: tools/rocprof_summary.py <kernel_stats.csv> [out.md]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)          # drop argument list
    name = re.sub(r"at::native::", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if len(name) > 110:
        name = name[:107] + "..."
    return name


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / tot < 0.001:
            continue
        lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {t/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {100*t/tot:.2f} |")
    lines.append(f"| **total GPU kernel time** | | {tot/1e6:.3f} | | 100 |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(out)
    print(out)


if __name__ == "__main__":
    main()
