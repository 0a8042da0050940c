#!/bin/bash
# per-kernel durations of the generic FFT passes on the fallback shapes (case indices of tools/run_fft_fallback.py)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for c in ${@:-1 2 4}; do
out=gpurun_out/r04/fftg_prof_$c
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/run_fft_fallback.py $c > $out.out 2> $out.err < /dev/null
f=$(find $out -name "*kernel_stats.csv" | head -1)
echo "== case $c"
if [ -n "$f" ]; then head -12 "$f" | cut -c1-220; else tail -5 $out.err; fi
done
