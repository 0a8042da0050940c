#!/bin/bash
# usage (on the GPU box): tools/ab_run.sh  -- alternates lagomorph_amd/_lib/ab_old.so / ab_new.so under the product name
cd "$(dirname "$0")/.."
for v in old new old new; do cp lagomorph_amd/_lib/ab_$v.so lagomorph_amd/_lib/liblagomorph_hip.so; python tools/ab_splat.py $v 2>/dev/null; done
