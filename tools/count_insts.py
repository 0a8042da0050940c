#!/usr/bin/env python3
"""Instruction mix of kernels in a gfx950 assembly file (hipcc -S --cuda-device-only ...): per kernel whose mangled name
contains one of the given substrings, the number of instructions by class (static count, loops counted once).
usage: python tools/count_insts.py file.s substring [substring ...]"""
import re
import sys
from collections import Counter

text = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not any(k in name for k in sys.argv[2:]):
        continue
    ins = []
    for line in body.split("\n"):
        t = line.strip()
        if not t or t.startswith((".", ";")) or t.split()[0].endswith(":"):
            continue
        ins.append(t.split()[0])
    cls = Counter()
    for i in ins:
        p = i.split("_")[0]
        cls["valu" if p == "v" else "salu" if p == "s" else "lds" if p == "ds" else "vmem" if p in ("buffer", "global", "flat", "scratch") else p] += 1
    print(f"{name[:110]}\n   {len(ins)} instructions: {dict(cls)}")
    print("   " + ", ".join(f"{k} {v}" for k, v in Counter(ins).most_common(18)))
