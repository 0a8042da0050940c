#!/bin/bash
# usage (on the GPU box): tools/ab_run.sh  -- alternates lagomorph_amd/_lib/ab_old.so / ab_new.so through
# LAGO_HIP_LIBRARY (the product library is never overwritten; the loader refuses a library of another ABI version)
cd "$(dirname "$0")/.."
for v in old new old new; do LAGO_HIP_LIBRARY=$PWD/lagomorph_amd/_lib/ab_$v.so python tools/ab_splat.py $v 2>/dev/null; done
