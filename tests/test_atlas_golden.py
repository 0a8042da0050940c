"""LDDMMAtlasBuilder against runs of the REFERENCE's own LDDMMAtlasBuilder (lddmm.py:108-375) recorded in
tests/golden/ref_atlas.npz by tools/gen_golden_atlas_from_reference.py: `lddmm_steps`, `image_update_freq`,
`image_shape` regrid of a given I0, multiscale momenta (`momentum_shape` != image shape), momentum
preconditioning, ragged minibatches, the four loss histories.  On the oracle backend the two Python layers
sit on the same extension, so every number must agree to rounding of the host arithmetic; through the HIP
kernels (float64) the scatter-add gradients differ in summation order only."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_atlas.npz"))

CASES = {
    "a3d": dict(batch_size=2, lddmm_integration_steps=2, reg_weight=1e-1, learning_rate_pose=2e-6,
                learning_rate_image=5e-2),
    "b2d_multiscale": dict(batch_size=2, lddmm_steps=2, lddmm_integration_steps=3, image_update_freq=2, reg_weight=5e-2,
                           learning_rate_pose=5e-4, learning_rate_image=1e-1, momentum_shape=(5, 5),
                           momentum_preconditioning=True),
    "c3d_multiscale_I0": dict(batch_size=2, lddmm_integration_steps=2, reg_weight=1e-1, learning_rate_pose=2e-4,
                              learning_rate_image=5e-2, momentum_shape=(4, 4, 5), image_update_freq=1),
}
EPOCHS = {"a3d": 3, "b2d_multiscale": 3, "c3d_multiscale_I0": 2}
BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


def run_case(lm, name, dev, **extra):
    data = torch.from_numpy(G[name + "_data"]).to(dev)
    I0 = torch.from_numpy(G[name + "_I0"]).to(dev) if name + "_I0" in G.files else None
    b = lm.LDDMMAtlasBuilder(data, I0=I0, **CASES[name], **extra)
    b.run(num_epochs=EPOCHS[name])
    return b


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("backend", BACKENDS)
def test_builder_reproduces_the_reference_run(request, backend, name):
    import lagomorph_amd as lm

    if backend == "oracle":
        request.getfixturevalue("oracle_ext")
        dev, rtol = "cpu", 1e-12
    else:
        lm.set_debug_mode(True)
        dev, rtol = "cuda", 1e-9
    b = run_case(lm, name, dev)
    lm.set_debug_mode(False) if backend == "hip" else None

    def close(got, key):
        want = G[name + "_" + key]
        got = np.asarray(got, dtype=np.float64).reshape(want.shape)
        scale = max(np.abs(want).max(), 1e-300)
        err = np.abs(got - want).max() / scale
        assert err <= rtol, f"{name}.{key}: max rel-to-max error {err:.3e} > {rtol:.0e}"

    close(b.I.detach().cpu().numpy(), "I")
    close(torch.cat([m.detach() for m in b.ms]).cpu().numpy(), "ms")
    for k in ("epoch_losses", "epoch_reg_terms", "iter_losses", "iter_reg_terms"):
        assert len(getattr(b, k)) == len(G[name + "_" + k]), k
        close(getattr(b, k), k)
    assert b.regrid_momenta == ("multiscale" in name)


def test_lddmm_steps_take_the_image_gradient_from_the_last_step_only(oracle_ext):
    """lddmm.py:331-332: I.requires_grad is switched on for the last inner step only."""
    import lagomorph_amd as lm

    data = torch.from_numpy(G["a3d_data"])
    seen = []
    orig = lm.lddmm.lddmm_step

    def spy(I, *a, **k):
        seen.append(I.requires_grad)
        return orig(I, *a, **k)

    lm.lddmm.lddmm_step = spy
    try:
        b = lm.LDDMMAtlasBuilder(data, batch_size=4, lddmm_steps=3, lddmm_integration_steps=1, learning_rate_pose=1e-6)
        b.epoch()
    finally:
        lm.lddmm.lddmm_step = orig
    assert seen == [False, False, True]


def test_hdf5_layout_with_a_stand_in_h5py(oracle_ext, tmp_path, monkeypatch):
    """save() writes the reference's dataset names (lddmm.py:251-262: atlas, momenta + batch_sizes attribute,
    epoch_losses, epoch_reg_terms, iter_losses, iter_reg_terms) through the h5py API when h5py is importable.
    h5py is absent from this image, so a minimal in-memory stand-in records the calls."""
    import sys
    import types

    import lagomorph_amd as lm

    store = {}

    class DS:
        def __init__(self, data=None, shape=None, dtype=None):
            self.a = np.array(data) if data is not None else np.zeros(shape, dtype=dtype)
            self.attrs = {}

        def __setitem__(self, k, v):
            self.a[k] = v

        def __getitem__(self, k):
            return self.a[k]

        def __array__(self, dtype=None, copy=None):
            return self.a if dtype is None else self.a.astype(dtype)

    class File:
        def __init__(self, path, mode):
            self.path, self.mode = path, mode
            if mode == "w":
                store[path] = {}
                open(path, "wb").write(b"\x89HDF\r\n\x1a\n")  # the HDF5 signature load() looks for

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def create_dataset(self, name, data=None, shape=None, dtype=None):
            store[self.path][name] = DS(data, shape, dtype)
            return store[self.path][name]

        def __getitem__(self, name):
            return store[self.path][name]

    fake = types.ModuleType("h5py")
    fake.File = File
    monkeypatch.setitem(sys.modules, "h5py", fake)

    data = torch.from_numpy(G["a3d_data"])
    kw = dict(batch_size=3, lddmm_integration_steps=1, learning_rate_pose=1e-6, learning_rate_image=1e-2)
    a = lm.LDDMMAtlasBuilder(data, **kw)
    a.run(num_epochs=2)
    path = a.save(str(tmp_path / "atlas.h5"))
    f = store[path]
    assert sorted(f) == ["atlas", "epoch_losses", "epoch_reg_terms", "iter_losses", "iter_reg_terms", "momenta"]
    assert f["atlas"].a.shape == (1, 1, 6, 6, 6) and f["momenta"].a.shape == (4, 3, 6, 6, 6)
    assert list(f["momenta"].attrs["batch_sizes"]) == [3, 1]
    assert f["iter_losses"].a.shape == (4,) and f["epoch_losses"].a.shape == (2,)
    b = lm.LDDMMAtlasBuilder(data, **kw)
    b.load(path)
    assert torch.equal(a.I, b.I) and all(torch.equal(x, y) for x, y in zip(a.ms, b.ms))
    assert a.iter_losses == b.iter_losses and a.epoch_reg_terms == b.epoch_reg_terms
