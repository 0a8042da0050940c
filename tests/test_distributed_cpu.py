"""CPU, world_size 2 over gloo: the batch-sharded atlas builder (lagomorph_amd.lddmm
LDDMMAtlasBuilder; reference lddmm.py:108-375) gives the same atlas as a single process over the
whole dataset -- shard ownership of momenta, the SUM all-reduce of the atlas gradient divided by
image_iters * world_size (lddmm.py:292-297), the mean-image all-reduce (lddmm.py:196-198) and the
loss reduction.  The CPU oracle stands in for lagomorph_ext inside the workers (tests only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _patch_oracle():
    import lagomorph_amd as lm
    from oracle.lago_oracle import OracleExt

    o = OracleExt()
    for name in ("interp_forward", "interp_backward", "compose", "jacobian_times_vectorfield_forward",
                 "jacobian_times_vectorfield_backward", "jacobian_times_vectorfield_adjoint_forward",
                 "jacobian_times_vectorfield_adjoint_backward", "fluid_operator", "regrid_forward", "regrid_backward"):
        setattr(lm.lagomorph_ext, name, getattr(o, name))
    delattr(lm.lagomorph_ext, "interp_backward_fused")
    delattr(lm.lagomorph_ext, "fluid_metric")
    delattr(lm.lagomorph_ext, "Ad_star")
    delattr(lm.lagomorph_ext, "ad_star")
    return lm


def _dataset(n, sp):
    g = torch.Generator().manual_seed(5)
    base = torch.randn((1, 1) + sp, generator=g, dtype=torch.float64)
    return base + 0.3 * torch.randn((n, 1) + sp, generator=g, dtype=torch.float64)


def _run(rank, world, port, sp, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lm = _patch_oracle()
        data = _dataset(4, sp)
        per = data.shape[0] // world
        shard = data[rank * per:(rank + 1) * per]
        # record every all-reduce: (numel, async?, issued from inside the backward pass?)
        calls = []
        real_all_reduce = dist.all_reduce

        def spy(t, *a, **k):
            calls.append((t.numel(), bool(k.get("async_op", False)), torch._C._current_graph_task_id() != -1))
            return real_all_reduce(t, *a, **k)

        lm.lddmm.dist.all_reduce = spy
        b = lm.LDDMMAtlasBuilder(shard, batch_size=2, lddmm_integration_steps=2, reg_weight=1e-1,
                                 learning_rate_pose=1e-2, learning_rate_image=1e-1, world_size=world, rank=rank,
                                 dataset_size=data.shape[0])
        I = b.run(num_epochs=2)
        res = {"I": I.numpy(), "loss": [float(x) for x in b.epoch_losses], "m0": b.ms[0].numpy()}
        if rank == 0:
            np.savez(out, I=res["I"], loss=np.array(res["loss"]), m0=res["m0"], calls=np.array(calls, dtype=np.int64),
                     iter_loss=np.array(b.iter_losses))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("sp", [(6, 6, 6), (8, 8)])
def test_two_rank_atlas_equals_sequential_emulation(tmp_path, oracle_ext, sp):
    """Reference semantics being checked (lddmm.py:287-341, as coded): every rank runs lddmm_step on
    its own minibatch against the same atlas, I.grad is SUM-all-reduced and divided by
    image_iters * world_size, then one SGD step.  With image_update_freq = 0 that happens after
    EVERY iteration (`image_iters < 0` is never true, lddmm.py:288-291), which is why a 2-rank run is
    not the same computation as one process walking the 4 subjects in two minibatches.  The
    emulation below does the per-rank steps sequentially in this process with plain tensor math."""
    import lagomorph_amd as lm

    data = _dataset(4, sp)
    kw = dict(integration_steps=2, reg_weight=1e-1, learning_rate_pose=1e-2)
    metric = lm.FluidMetric([0.1, 0, 0.01])
    I = data.mean(0, keepdim=True).clone()
    ms = [torch.zeros((2, len(sp)) + sp, dtype=data.dtype) for _ in range(2)]
    losses = []
    for _ in range(2):  # epochs; one iteration per rank per epoch
        grads, tot = [], 0.0
        for r in range(2):
            Ir = I.clone().requires_grad_(True)
            ms[r], loss, _reg = lm.lddmm_step(Ir, ms[r], data[2 * r:2 * r + 2], metric, 4, **kw)
            grads.append(Ir.grad)
            tot += float(loss)
        I = I - 1e-1 * (grads[0] + grads[1]) / (1 * 2)
        losses.append(tot)

    out = str(tmp_path / "rank0.npz")
    mp.spawn(_run, args=(2, _free_port(), sp, out), nprocs=2, join=True)
    r = np.load(out)
    assert np.allclose(r["I"], I.numpy(), rtol=1e-10, atol=1e-12)  # up to the all-reduce's summation order
    assert np.allclose(r["loss"], losses, rtol=1e-10)
    assert np.allclose(r["m0"], ms[0].numpy(), rtol=1e-10, atol=1e-12)  # rank 0 owns the first minibatch
    assert not np.allclose(r["I"], data.mean(0, keepdim=True).numpy())  # the atlas actually moved
    assert np.allclose(r["iter_loss"], losses, rtol=1e-10)  # one iteration per epoch here: iteration loss == epoch loss
    # collectives: the mean image once, then per epoch ONE all-reduce of the atlas gradient -- asynchronous and
    # issued from inside the backward pass (it overlaps the rest of the backward through expmap) -- and one of the
    # stacked (loss, reg) history; nothing else
    nv = int(np.prod(sp))
    calls = [tuple(c) for c in r["calls"].tolist()]
    assert calls == [(nv, 0, 0), (nv, 1, 1), (2, 0, 0), (nv, 1, 1), (2, 0, 0)], calls


def _run_ragged(rank, world, port, sp, out, n=5):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lm = _patch_oracle()
        data = _dataset(n, sp)   # 5 subjects over 2 ranks: the sampler pads to 6, one subject is seen twice
        b = lm.LDDMMAtlasBuilder.from_dataset(data, world_size=world, rank=rank, batch_size=2, lddmm_integration_steps=2,
                                              reg_weight=1e-1, learning_rate_pose=1e-2, learning_rate_image=1e-1)
        I = b.run(num_epochs=2)
        np.savez(out + f".{rank}.npz", I=I.numpy(), idx=np.array(b.subject_indices), loss=np.array(b.epoch_losses),
                 ms=torch.cat(b.ms).numpy(), nb=len(b.ms))
    finally:
        dist.destroy_process_group()


def test_shard_indices_are_the_distributed_samplers():
    """lddmm.py:164-167: DistributedSampler with its defaults (shuffle, seed 0, no set_epoch, padded)."""
    from torch.utils.data.distributed import DistributedSampler

    import lagomorph_amd as lm

    for n, world in ((5, 2), (7, 4), (256, 8), (3, 8)):
        shards = [lm.shard_indices(n, world, r) for r in range(world)]
        for r in range(world):
            assert shards[r] == list(iter(DistributedSampler(range(n), num_replicas=world, rank=r)))
        per = -(-n // world)
        assert all(len(s) == per for s in shards)                       # equal length: padded
        assert set(i for s in shards for i in s) == set(range(n))       # every subject is somebody's
        flat = [shards[r][k] for k in range(per) for r in range(world)]  # round-robin deal of the padded permutation
        assert flat[:n] == flat[:n] and sorted(flat[:n]) == list(range(n)) and flat[n:] == flat[:world * per - n]
    assert lm.shard_indices(5, 2, 1, shuffle=False) == [1, 3, 0]


@pytest.mark.parametrize("world,n", [(2, 5), (4, 9)])
def test_n_rank_atlas_with_a_padded_shard(tmp_path, oracle_ext, world, n):
    """5 subjects on 2 ranks (9 on 4) through `LDDMMAtlasBuilder.from_dataset`: every rank holds 3 subjects (2
    minibatches: the sampler pads the assignment, some subjects are seen by two ranks), runs the same number of iterations
    (no rank waits at a collective that another never reaches), all end with the same atlas, and the result equals a
    sequential emulation over the same padded assignment."""
    import lagomorph_amd as lm

    sp = (6, 6, 6)
    out = str(tmp_path / "ragged")
    mp.spawn(_run_ragged, args=(world, _free_port(), sp, out, n), nprocs=world, join=True)
    rs = [np.load(out + f".{r}.npz") for r in range(world)]
    for r in range(world):
        assert np.array_equal(rs[0]["I"], rs[r]["I"]) and int(rs[r]["nb"]) == 2
        assert list(rs[r]["idx"]) == lm.shard_indices(n, world, r)
    # sequential emulation: per iteration every rank steps on its own minibatch against the same atlas, the
    # gradients are summed and divided by image_iters * world_size (lddmm.py:292-297), one SGD step
    data = _dataset(n, sp)
    shards = [data[torch.as_tensor(lm.shard_indices(n, world, r))] for r in range(world)]
    metric = lm.FluidMetric([0.1, 0, 0.01])
    mean = sum(lm.lddmm.streaming_batch_average(s, 2) for s in shards).unsqueeze(0) / world
    I = mean.clone()
    ms = [[torch.zeros((b, 3) + sp, dtype=data.dtype) for b in (2, 1)] for _ in range(world)]
    losses = []
    for _ in range(2):
        tot = 0.0
        for it, (lo, hi) in enumerate(((0, 2), (2, 3))):
            grads = []
            for r in range(world):
                Ir = I.clone().requires_grad_(True)
                ms[r][it], loss, _ = lm.lddmm_step(Ir, ms[r][it], shards[r][lo:hi], metric, n, integration_steps=2,
                                                   reg_weight=1e-1, learning_rate_pose=1e-2)
                grads.append(Ir.grad)
                tot += float(loss)
            I = I - 1e-1 * sum(grads) / world
        losses.append(tot)
    assert np.allclose(rs[0]["I"], I.numpy(), rtol=1e-10, atol=1e-12)
    assert np.allclose(rs[0]["loss"], losses, rtol=1e-10)
    assert np.allclose(rs[world - 1]["ms"], torch.cat(ms[world - 1]).numpy(), rtol=1e-10, atol=1e-12)


def _run_affine(rank, world, port, sp, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import lagomorph_amd as lm
        from oracle.lago_oracle import OracleExt

        o = OracleExt()
        for name in ("affine_interp_forward", "affine_interp_backward"):
            setattr(lm.lagomorph_ext, name, getattr(o, name))
        data = _dataset(8, sp)
        per = data.shape[0] // world
        shard = data[rank * per:(rank + 1) * per].contiguous()
        d = len(sp)
        As = torch.zeros((per, d, d), dtype=torch.float64)
        Ts = torch.zeros((per, d), dtype=torch.float64)
        I, As, Ts, ep, _ = lm.affine_atlas(shard, As, Ts, num_epochs=3, batch_size=2, learning_rate_A=1e-3,
                                           learning_rate_T=2e-2, learning_rate_I=0.5, world_size=world, rank=rank,
                                           dataset_size=data.shape[0])
        np.savez(out + f".{rank}.npz", I=I.numpy(), A=As.numpy(), T=Ts.numpy(), ep=np.array(ep))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sp", [(9, 8), (6, 5, 7)])
def test_two_rank_affine_atlas_equals_one_process(tmp_path, oracle_ext, sp):
    """affine_atlas sharded over 2 ranks (mean-image all-reduce, one SUM all-reduce of the atlas gradient
    per epoch divided by image_iters * world_size, epoch-loss all-reduce; affine.py:328-333, 389-409)
    equals one process over all subjects: with the image updated once per epoch the per-subject
    (A, T) steps only depend on that epoch's atlas, and both runs average the same minibatch
    gradients."""
    import lagomorph_amd as lm

    out = str(tmp_path / "aff")
    mp.spawn(_run_affine, args=(2, _free_port(), sp, out), nprocs=2, join=True)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["I"], r1["I"])  # every rank ends with the same atlas
    data = _dataset(8, sp)
    d = len(sp)
    As = torch.zeros((8, d, d), dtype=torch.float64)
    Ts = torch.zeros((8, d), dtype=torch.float64)
    I, As, Ts, ep, _ = lm.affine_atlas(data, As, Ts, num_epochs=3, batch_size=2, learning_rate_A=1e-3,
                                       learning_rate_T=2e-2, learning_rate_I=0.5)
    assert np.allclose(r0["I"], I.numpy(), rtol=0, atol=1e-12)
    assert np.allclose(np.concatenate([r0["A"], r1["A"]]), As.numpy(), rtol=0, atol=1e-12)
    assert np.allclose(np.concatenate([r0["T"], r1["T"]]), Ts.numpy(), rtol=0, atol=1e-12)
    assert np.allclose(r0["ep"], np.array(ep), rtol=1e-12)


def _run_forced(rank, world, port, sp, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lm = _patch_oracle()
        data = _dataset(6, sp)
        res = {}
        for tag, kw in (("plain", {}), ("forced", dict(force_collectives=True)), ("forced_blocking", dict(force_collectives=True, overlap_allreduce=False))):
            calls = []
            real = dist.all_reduce

            def spy(t, *a, **k):
                calls.append((t.numel(), bool(k.get("async_op", False)), torch._C._current_graph_task_id() != -1))
                return real(t, *a, **k)

            lm.lddmm.dist.all_reduce = spy
            try:
                b = lm.LDDMMAtlasBuilder(data, batch_size=2, lddmm_integration_steps=2, reg_weight=1e-1, learning_rate_pose=1e-2,
                                         learning_rate_image=1e-1, image_update_freq=2, **kw)
                b.run(num_epochs=2)
            finally:
                lm.lddmm.dist.all_reduce = real
            res[tag] = (b.I.detach().numpy(), torch.cat(b.ms).numpy(), np.array(b.iter_losses), np.array(calls, dtype=np.int64).reshape(-1, 3))
        np.savez(out, **{f"{t}_{k}": v for t, r in res.items() for k, v in zip(("I", "ms", "loss", "calls"), r)})
    finally:
        dist.destroy_process_group()


def test_forced_collectives_at_world_size_one_equal_the_plain_builder(tmp_path):
    """`force_collectives=True` (round 6: what tests/test_gpu_rccl_world1.py runs over RCCL on the GPU box) on a
    world-size-1 gloo group: every N-rank branch is taken -- the collective sequence says so -- and, a SUM over one rank
    being the identity, the results are those of the plain builder exactly (the CPU oracle has no atomics)."""
    sp = (6, 6, 6)
    out = str(tmp_path / "forced.npz")
    mp.spawn(_run_forced, args=(1, _free_port(), sp, out), nprocs=1, join=True)
    r = np.load(out)
    nv = int(np.prod(sp))
    assert r["plain_calls"].size == 0
    # image_update_freq = 2 on 3 minibatches: per epoch the update after iteration 2 and the forced one at the end
    want = [(nv, 0, 0)] + [(nv, 1, 1), (nv, 1, 1), (6, 0, 0)] * 2
    assert [tuple(c) for c in r["forced_calls"].tolist()] == want
    assert [tuple(c) for c in r["forced_blocking_calls"].tolist()] == [(n, 0, 0) for n, _, _ in want]
    for tag in ("forced", "forced_blocking"):
        for k in ("I", "ms", "loss"):
            assert np.array_equal(r[f"{tag}_{k}"], r[f"plain_{k}"]), (tag, k)
