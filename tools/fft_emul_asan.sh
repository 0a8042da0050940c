#!/bin/bash
# The FFT passes' phase functions (lagomorph_amd/csrc/fft_lds.hpp: every phase between two workgroup barriers is a
# function of (phase, thread id)) run for ALL thread ids on the host under AddressSanitizer + UBSan: the index
# arithmetic of the kernels against exactly-sized global buffers and the LDS image.  GPU sanitizers are not available
# on the pool; this is the CPU-side substitute for the one kernel family whose code is host-compilable.
# ~1 min to build, ~3 min to run (tests/test_fft_emulation.py is the uninstrumented form of the same program).
set -e
repo=$(cd "$(dirname "$0")/.." && pwd)
out=${TMPDIR:-/tmp}/lago_fft_emul_asan
hipcc -O1 -g -std=c++17 --offload-host-only -fsanitize=address,undefined -fno-omit-frame-pointer -o "$out" "$repo/tests/native/fft_emul.hip"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 "$out" | tail -3
