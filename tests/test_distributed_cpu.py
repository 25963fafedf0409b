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
