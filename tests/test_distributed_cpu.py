"""CPU tier: the caption-block sharding (aladin_amd/distributed.py) under gloo, world_size 2.
The HIP scorer cannot run here, so the collectives are exercised with the torch restatement from
oracle/ injected as scores_fn / hinge_fn; the result must equal the single-process loss and
gradients on the concatenated batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, 'oracle')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import faithful_torch as FT
    from aladin_amd import synth
    from aladin_amd.distributed import sharded_alignment_loss
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    B, R, T, D = 4, 12, 15, 32
    im, s, il, sl = synth.alignment_batch(B * world, R, T, D, seed=321, ragged=True)
    a = torch.from_numpy(im[rank * B:(rank + 1) * B]).requires_grad_(True)
    b = torch.from_numpy(s[rank * B:(rank + 1) * B]).requires_grad_(True)
    loss, S = sharded_alignment_loss(
        a, b, il[rank * B:(rank + 1) * B], sl[rank * B:(rank + 1) * B], 0.2, True,
        scores_fn=lambda x, y, xl, yl: FT.alignment_scores_faithful(x, y, [int(v) for v in xl], [int(v) for v in yl]),
        hinge_fn=FT.hinge_faithful)
    loss.backward()
    ret[rank] = (loss.item(), S.detach().numpy(), a.grad.numpy(), b.grad.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_loss_equals_single_process():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import faithful_torch as FT
    from aladin_amd import synth
    world, B = 2, 4
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    im, s, il, sl = synth.alignment_batch(B * world, 12, 15, 32, seed=321, ragged=True)
    loss, S, dim, ds = FT.alignment_triplet_step(torch.from_numpy(im), torch.from_numpy(s), il, sl, 0.2, True)
    for r in range(world):
        l_r, S_r, ga, gb = ret[r]
        np.testing.assert_allclose(l_r, loss.item(), rtol=1e-6)
        np.testing.assert_allclose(S_r, S.numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ga, dim.numpy()[r * B:(r + 1) * B], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(gb, ds.numpy()[r * B:(r + 1) * B], rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------------
# SparseImageExchange (the max_violation backward's pair-driven all-to-all), gloo, world_size 3
# ------------------------------------------------------------------------------------------------
def _sparse_pattern(world, B, seed):
    """Replicated dS with a hinge-like support: the diagonal, one entry per row and one per column,
    plus an all-zero caption block column range on the last rank when seed is odd."""
    rng = np.random.RandomState(seed)
    n = world * B
    dS = np.zeros((n, n), np.float32)
    dS[np.arange(n), np.arange(n)] = -2.0
    dS[np.arange(n), rng.randint(0, n, n)] += 1.0
    dS[rng.randint(0, n, n), np.arange(n)] += 1.0
    if seed % 2:
        dS[:, (world - 1) * B:] = 0.0                   # a rank that needs nothing
    return dS


def _contribution(i_global, b, R, D):
    """What caption block b would send back for image i (any deterministic function)."""
    base = (i_global[:, None, None] * 7 + b * 1000).astype(np.float32)
    return base + np.arange(R, dtype=np.float32)[None, :, None] + 0.001 * np.arange(D, dtype=np.float32)[None, None, :]


def _exchange_worker(rank, world, port, ret):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from aladin_amd.distributed import SparseImageExchange
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    B, R, D = 5, 3, 4
    out = {}
    for seed in (0, 1, 2):
        dS = _sparse_pattern(world, B, seed)
        im_all = np.arange(world * B * R * D, dtype=np.float32).reshape(world * B, R, D)
        ex = SparseImageExchange(torch.from_numpy(dS), B)
        got = ex.fetch(torch.from_numpy(im_all[rank * B:(rank + 1) * B]))
        need = ex.need_idx.numpy()
        d_need = _contribution(need, rank, R, D)
        d_local = ex.give_back(torch.from_numpy(d_need), (B, R, D))
        out[seed] = (need, got.numpy(), d_local.numpy())
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sparse_image_exchange_world3():
    world, B, R, D = 3, 5, 3, 4
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_exchange_worker, args=(world, port, ret), nprocs=world, join=True)
    im_all = np.arange(world * B * R * D, dtype=np.float32).reshape(world * B, R, D)
    for seed in (0, 1, 2):
        dS = _sparse_pattern(world, B, seed)
        for r in range(world):
            need, got, d_local = ret[r][seed]
            want_need = np.nonzero((dS[:, r * B:(r + 1) * B] != 0).any(axis=1))[0]
            np.testing.assert_array_equal(need, want_need)                       # ascending global ids
            np.testing.assert_array_equal(got, im_all[want_need])                # exactly those sets, fp32 bits
            want = np.zeros((B, R, D), np.float32)
            for b in range(world):                                               # every block that used my images
                used = np.nonzero((dS[r * B:(r + 1) * B, b * B:(b + 1) * B] != 0).any(axis=1))[0]
                want[used] += _contribution(used + r * B, b, R, D)
            np.testing.assert_allclose(d_local, want, rtol=1e-6)
