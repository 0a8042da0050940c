#!/bin/bash
# usage (on the GPU box): tools/ab_fft_libs.sh tagA tagB ...  -- runs tools/ab_fft_libs.py with lagomorph_amd/_lib/ab_<tag>.so, twice round
cd "$(dirname "$0")/.."
for r in 1 2; do for v in "$@"; do LAGO_HIP_LIBRARY=$PWD/lagomorph_amd/_lib/ab_$v.so python tools/ab_fft_libs.py $v 2>/dev/null; done; done
