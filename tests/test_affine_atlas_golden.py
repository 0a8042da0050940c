"""`affine_atlas` and `StandardizedDataset` against fixtures from the REFERENCE's own loop:
tests/golden/ref_affine_atlas.npz holds four runs of /root/reference/lagomorph/affine.py:288-415 (imported by
tools/gen_golden_atlas_from_reference.py with the CPU oracle standing in for the CUDA-only extension): 2D and 3D,
`image_update_freq` 0 / 1 / 2 (incl. the reference's carry-over of a partial gradient accumulation into the next
epoch), `affine_steps` 1 / 2 / 3, both regularisers, ragged last minibatches, a given initial image, float64 and
float32 -- atlas, A, T, every per-iteration and per-epoch loss, and the standardized (inverse-mapped) subjects.

CPU: the host mirror on the oracle backend reproduces them to 1e-12 (float64) / 1e-6 (float32: the oracle *is* the
arithmetic that produced the fixtures; what differs is torch's summation order in the loss reductions).  GPU: the same
through the HIP kernels (float64 1e-9; float32 at north_star's 1e-5)."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "ref_affine_atlas.npz"))

# must equal AFFINE_CASES of tools/gen_golden_atlas_from_reference.py
CASES = {
    "a2d": dict(num_epochs=3, batch_size=2, learning_rate_A=2e-3, learning_rate_T=5e-2, learning_rate_I=1.0),
    "b3d_freq_steps": dict(num_epochs=3, batch_size=2, image_update_freq=2, affine_steps=2, reg_weightA=0.3,
                           reg_weightT=0.05, learning_rate_A=3e-3, learning_rate_T=4e-2, learning_rate_I=0.7),
    "c2d_givenI_freq1": dict(num_epochs=2, batch_size=3, image_update_freq=1, affine_steps=3, reg_weightT=0.1,
                             learning_rate_A=1e-3, learning_rate_T=2e-2, learning_rate_I=0.5),
    "d3d_f32": dict(num_epochs=2, batch_size=2, learning_rate_A=2e-3, learning_rate_T=5e-2, learning_rate_I=1.0),
}


def run_case(lm, name, device):
    data = torch.from_numpy(G[name + "_data"]).to(device)
    n, d = data.shape[0], data.dim() - 2
    As = torch.zeros((n, d, d), dtype=data.dtype, device=device)
    Ts = torch.zeros((n, d), dtype=data.dtype, device=device)
    I0 = torch.from_numpy(G[name + "_I0"]).to(device) if name + "_I0" in G.files else None
    I, A, T, ep, it = lm.affine_atlas(data, As, Ts, I=I0, **CASES[name])
    sd = lm.StandardizedDataset([data[i] for i in range(n)], A, T, device=device)
    std = torch.stack([sd[i] for i in range(len(sd))])
    return I, A, T, ep, it, std


def check(name, got, tol):
    I, A, T, ep, it, std = got
    for key, val in (("I", I), ("A", A), ("T", T), ("standardized", std)):
        want = G[f"{name}_{key}"].astype(np.float64)
        err = np.abs(val.detach().cpu().numpy().astype(np.float64) - want).max()
        assert err <= tol * max(np.abs(want).max(), 1e-30), (name, key, err, np.abs(want).max())
    assert len(it) == len(G[name + "_iter_losses"]) and len(ep) == len(G[name + "_epoch_losses"])
    assert it == pytest.approx(list(G[name + "_iter_losses"]), rel=max(tol, 1e-12))
    assert ep == pytest.approx(list(G[name + "_epoch_losses"]), rel=max(tol, 1e-12))


@pytest.mark.parametrize("name", list(CASES))
def test_affine_atlas_matches_the_reference_loop_oracle_backend(oracle_ext, name):
    import lagomorph_amd as lm

    check(name, run_case(lm, name, "cpu"), 1e-6 if name.endswith("f32") else 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_affine_atlas_matches_the_reference_loop_hip(name):
    import lagomorph_amd as lm

    lm.set_debug_mode(True)
    try:
        check(name, run_case(lm, name, "cuda"), 1e-5 if name.endswith("f32") else 1e-9)
    finally:
        lm.set_debug_mode(False)


@pytest.mark.gpu
def test_standardized_dataset_hip_equals_oracle():
    """StandardizedDataset (affine.py:418-438) on the GPU: affine_inverse + affine_interp through the HIP kernel,
    against the oracle's affine_interp_forward on the inverse map computed on the host (bit-exact forward kernel)."""
    import lagomorph_amd as lm
    from oracle import lago_oracle as orc

    rng = np.random.default_rng(4)
    for sp in ((13, 11), (7, 9, 8)):
        d = len(sp)
        imgs = rng.standard_normal((3, 2) + sp)
        A = 0.1 * rng.standard_normal((3, d, d))
        T = rng.standard_normal((3, d))
        At, Tt = torch.from_numpy(A).cuda(), torch.from_numpy(T).cuda()
        sd = lm.StandardizedDataset([torch.from_numpy(imgs[i]) for i in range(3)], At, Tt, device="cuda")
        assert len(sd) == 3
        for i in range(3):
            Ainv, Tinv = lm.affine_inverse(At[[i]] + torch.eye(d, dtype=torch.float64, device="cuda")[None], Tt[[i]])
            want = orc.affine_interp_forward(imgs[i][None], Ainv.cpu().numpy(), Tinv.cpu().numpy())[0]
            got = sd[i]
            assert got.shape == (2,) + sp and got.is_cuda
            assert np.array_equal(got.cpu().numpy(), want)
