#!/bin/bash
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
cd "$repo" || exit 1
mkdir -p gpurun_out
{
S=128 B=8 SEQ=1,2,3,4,1,2,3,4 timeout 300 python tools/probe_stream_spikes.py 2>/dev/null
S=128 B=8 SEQ=1,2,1,2,1,2 timeout 300 python tools/probe_stream_spikes.py 2>/dev/null
S=128 B=16 SEQ=2,2,2,1,2 timeout 300 python tools/probe_stream_spikes.py 2>/dev/null
} > gpurun_out/r05_stream_spikes.txt
cat gpurun_out/r05_stream_spikes.txt
