#!/usr/bin/env python3
"""bench.py's per-operator table (micro_ops, batch 8 x 3 x 128^3) for ONE library build (LAGO_HIP_LIBRARY picks it), one line.
usage: LAGO_HIP_LIBRARY=... python tools/ab_lib_ops.py <tag>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import lagomorph_amd as lm

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
r = bench.micro_ops(lm, torch.device("cuda"), 128, 8)
print(f"{tag:>8s}: " + "  ".join(f"{k.split('(')[0][:14]}{'(' + k.split('(')[1][:6] if '(' in k else ''} {v['ms'] * 1e3:.1f}" for k, v in r["ops"].items()), flush=True)
