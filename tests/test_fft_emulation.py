"""Host emulation of the three LDS-tiled FFT passes of the fluid metric (no GPU needed).

lagomorph_amd/csrc/fft_lds.hpp writes every phase between two workgroup barriers as a function of
(phase, thread id); tests/native/fft_emul.hip runs those phases for all thread ids in turn and
compares with a double-precision DFT + the per-frequency operator (cuda/metric.cu:103-160)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_fft_passes_host_emulation(tmp_path):
    exe = str(tmp_path / "fft_emul")
    src = os.path.join(HERE, "native", "fft_emul.hip")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-o", exe, src], check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all ok" in r.stdout
