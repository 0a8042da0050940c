"""No hot kernel of the built library may spill registers (CPU test: reads the code-object metadata of the `.so`).

Scratch is HBM-backed: a spilled value costs every lane a store (and a load) of real memory traffic.  Round 3 found two
such kernels only through their PMC write counters -- the first LDS-window `compose` (134 MB of scratch writes per
launch) and the geometry-once splat (335 MB, the hoisted coordinates of a rare path) -- so the metadata is checked here.
The cold instantiations that are allowed to spill are listed with their bounds."""
import os
import re
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import check_spills  # noqa: E402

ALLOWED = [  # (pattern on the demangled name, spilled VGPRs allowed)
    (r"lago::splat_tiled_kernel<float, [03], (true|false), true, 1024, 4, true>", 2),   # multi-channel general splat: affine / regrid callers only
    (r"lago::splat_tiled_kernel<double, [03], (true|false), true, 1024, 4, false>", 4),  # float64 general splat
    # persistent zy passes of mixed planes (not BASELINE shapes; 160 x 160 itself must be clean): re-deriving the per-plane
    # indices instead (no spill) measured 2-4 % SLOWER than these few spilled registers (tools/ab_fft_libs.py, round 3)
    (r"lago::zy_(forward|inverse)_persist_kernel<(128, 160|160, 192|192, 160)>", 7),
    # ... and of the 176 / 208 planes (round 6: nine or ten float4 of the next plane in flight beside a radix-11 / radix-13 level
    # under the 128 registers of a 1024-thread workgroup).  Measured WITH these spills against the one-shot kernels, which
    # have none (tools/time_oasis_fft.py, profiles/r06_radix11_13.md): 208 x 176 planes 870 against 987 us per sharp at
    # batch 8, 176 x 176 373 against 450, 176 x 208 460 against 527.
    (r"lago::zy_(forward|inverse)_persist_kernel<(208, 176|176, 176|176, 208)>", 12),
    # ... and of the 224 x 160, 160 x 224 and 144 x 176 planes (odd factors 7 and 9; inverse only, 3 - 11 registers): sharp at 4 x 192
    # x 224 x 160 514 against 580 us one-shot, 4 x 224 x 160 x 224 630 against 719, 8 x 144 x 144 x 176 506 against 610
    (r"lago::zy_inverse_persist_kernel<(224, 160|160, 224|144, 176)>", 11),
]


@pytest.mark.skipif(not os.path.exists(os.path.join(check_spills.LLVM, "llvm-readelf")) or not shutil.which("c++filt"),
                    reason="needs ROCm's llvm-readelf / llvm-objdump and c++filt")
def test_hot_kernels_do_not_spill():
    import lagomorph_amd

    res = check_spills.kernel_resources(lagomorph_amd.lagomorph_ext.LIB_PATH)
    assert len(res) > 400, len(res)
    spilled = {k: v for k, v in res.items() if v[2] or v[3]}
    pretty = dict(zip(spilled, check_spills.demangle(list(spilled))))
    offenders = []
    for k, v in spilled.items():
        bound = next((b for pat, b in ALLOWED if re.search(pat, pretty[k])), 0)
        if v[2] > bound:
            offenders.append(f"{pretty[k][:140]}: {v[2]} spilled VGPRs, {v[3]} B scratch (allowed {bound})")
    assert not offenders, "\n".join(offenders)
    # and the kernels the benchmarks live in are there, at the occupancy DESIGN.md states
    by_name = {n: res[k] for k, n in zip(res, check_spills.demangle(list(res)))}
    for frag, max_vgprs in (("lago::compose3_window_kernel<512, 8, false>", 128), ("lago::splat_shear_mc_kernel<1024, false, false, 2>", 64),
                            ("lago::ad_star3_tile_kernel<float, 512, 2, 5, 2, 128>", 64), ("lago::ad_star3_tile_kernel<float, 512, 2, 5, 3, 160>", 64),
                            ("lago::ad_star3_tile_kernel<float, 512, 2, 5, 2, 0>", 64), ("lago::zy_forward_persist_kernel<160, 160>", 128),
                            ("lago::zy_inverse_persist_kernel<160, 160>", 128), ("lago::fluid_xpass2_persist_kernel<128, true, 256>", 256),
                            ("lago::fluid_xpass2_persist_kernel<160, true, 256>", 256), ("lago::zy_forward_kernel<128, 128>", 128), ("lago::splat_shear_kernel<1024, true, true, false, 0>", 64)):
        hits = [v for n, v in by_name.items() if frag in n]
        assert hits, frag
        assert all(v[0] <= max_vgprs and v[2] == 0 for v in hits), (frag, hits)
