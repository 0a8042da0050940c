"""GPU: a short randomised parity sweep (tools/fuzz_parity.py): random 2D / 3D shapes incl. thin and ragged ones, both
dtypes, batch / channel counts, broadcast images, displacements from sub-voxel to far out of range (smooth, rough,
integer-valued), unit and non-unit steps -- interp forward / d_u, the Jacobian products and their backward forms, compose,
Ad_star, affine forward and regrid forward BIT FOR BIT against the oracle, the fluid metric on random extents and the
scatter-adds (d_I, d_A, d_T, regrid backward) at north_star's bound or, where thousands of float32 terms pile onto one
border cell or cancel, by the float32 summation bound against the float64 oracle; every third case under a random
combination of the library's sibling implementations.  The long form (`python tools/fuzz_parity.py 150 <seed>`: 6 000 - 9 000 cases per
run; `LAGO_FUZZ_BIG=1` for volumes of up to 2 M voxels) found no mismatch in over 109 000 cases (profiles/r05_fuzz.md)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_cases_against_the_oracle(seed):
    import fuzz_parity

    try:
        n, worst, yard = fuzz_parity.run(budget=12.0, seed=seed)
    except SystemExit as e:   # the tool reports a mismatch this way
        pytest.fail(str(e))
    assert n >= 20, n
    assert all(v <= 1.0 for v in worst.values()), worst


def test_random_matching_steps_against_the_oracle_backend():
    """tools/fuzz_step.py, short form: random `lddmm_step` problems (2D / 3D, multiscale momenta, 1-4 integration steps,
    preconditioning, 1-6 subjects, the stream-split option, both dtypes) through HIP against the oracle backend.  The
    long form ran 19 226 steps without a mismatch (profiles/r05_fuzz.md, incl. the fourteen cell-face events, each proved on its own case by tools/debug_step_event.py)."""
    import fuzz_step

    try:
        n, worst, yard = fuzz_step.run(budget=15.0, seed=7)
    except SystemExit as e:
        pytest.fail(str(e))
    assert n >= 20, n
    assert all(v <= 1.0 for v in worst.values()), worst


def test_a_constructed_cell_face_event_is_found_and_proved():
    """tools/debug_step_event.py on a case BUILT to contain one event: the momenta are adjusted (the operator is linear)
    until the float64 displacement of voxel (12, 12, 40) is -1.2e-6 along z -- 0.31 float32 ulps below the grid point, so
    `40 + h` rounds to 40.0 in float32 (cell 40) and lies below it in float64 (cell 39).  The analysis must find that
    voxel as an outlier of the position gradient ON a cell face, nothing off a face, and the float32 step with the on-face
    values taken from the float64 run must be within north_star's bound of the float64 step."""
    import numpy as np
    import torch

    import debug_step_event
    import lagomorph_amd as lm
    from test_gpu_lddmm_step import smooth_np

    rng = np.random.default_rng(41)
    sp, x0 = (24, 24, 48), (12, 12, 40)
    base = torch.from_numpy(smooth_np(rng, (1, 1) + sp, 1.5))
    base = base / base.std()
    imgs = (base + 0.5 * torch.from_numpy(smooth_np(rng, (1, 1) + sp, 1.0))).contiguous()
    met = lm.FluidMetric([0.1, 0.0, 0.01])
    m = torch.from_numpy(smooth_np(rng, (1, 3) + sp, 1.5)).cuda()
    m = m * (0.5 / met.sharp(m).abs().max())
    e = torch.zeros_like(m)
    e[(0, 2) + x0] = 1.0
    s_e = float(met.sharp(e)[(0, 2) + x0])
    m32 = m.float()
    target = -1.2e-6
    for _ in range(6):   # h = -sharp(m) (one integration step, dt = 1): drive h_z(x0) of the float32-representable momenta to the target
        h = -float(met.sharp(m32.double())[(0, 2) + x0])
        m32 = (m32.double() + ((h - target) / s_e) * e).float()
    h = -float(met.sharp(m32.double())[(0, 2) + x0])
    assert abs(h - target) < 2e-7, h
    assert float(np.float32(40.0) + np.float32(h)) == 40.0 and 40.0 + h < 40.0   # the two floors differ
    kw = dict(integration_steps=1, reg_weight=1e-2, learning_rate_pose=1e-3, momentum_preconditioning=False)
    lines = []
    ev = debug_step_event.analyse(base, imgs, m32.cpu(), 1, kw, say=lines.append)
    print("\n".join(lines))
    assert ev["on_face"] >= 1 and ev["off_face"] == 0, ev
    assert any(f"voxel {x0}" in ln for ln in lines), lines
    assert max(ev["after"].values()) <= 1e-5, ev
    assert ev["after"]["m"] <= ev["before"]["m"], ev
