"""Worker of tests/test_gpu_parity.py::test_fast_sharded_node_across_processes_on_one_gpu: one RANK of the fast sharded step with the REAL
HIP kernels.  No node with more than one GPU has been available to this build, and RCCL refuses two ranks on one device -- so the ranks
are separate processes that share cuda:0 and exchange through gloo (which stages device tensors through the host): every line of
aladin_amd.distributed._ShardedTriplet runs at W > 1 on real operands -- segment views into ONE gathered buffer, remote blocks scored from
their segments in place, the score-block permute, the dense exchange (reduce-scatter = all-reduce + slice under gloo) and the pair-driven
one (all-to-all staged through the host under gloo) -- only the transport differs from the 8-GPU run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, shape, exchanges, bwd_mode, ret):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from aladin_amd import distributed as AD, ops, synth
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ops.set_backward_precision(bwd_mode)
    B, R, T, D, ragged = shape
    im, s, il, sl = synth.structured_alignment_batch(B * world, R, T, D, seed=4321 + world, noise=3.0, ragged=ragged)
    blk = slice(rank * B, (rank + 1) * B)
    out = {}
    for exchange in exchanges:
        a = torch.from_numpy(im[blk].copy()).to(dev).requires_grad_(True)
        b = torch.from_numpy(s[blk].copy()).to(dev).requires_grad_(True)
        loss, S = AD.sharded_alignment_loss_fast(a, b, il[blk], sl[blk], 0.2, True, exchange=exchange)
        took_sparse = loss.grad_fn.exchange is not None
        (loss * 0.5).backward()
        torch.cuda.synchronize()
        out[exchange] = (loss.detach().cpu(), S.cpu() if rank == 0 else None, float(S.double().sum()), a.grad.cpu(), b.grad.cpu(), took_sparse)
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def model_worker(rank, world, port, B, loss_type, weights, epochs, bwd_mode, ret):
    """One rank of ALADModel(config, shard_group=None).forward_loss_total -- every loss head of the shipped YAMLs on the GLOBAL batch --
    with the real kernels, ranks sharing cuda:0 over gloo."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    from aladin_amd import ops, synth
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.evaluation import LogCollector
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ops.set_backward_precision(bwd_mode)
    config = {'training': {'loss-type': loss_type, 'loss-weights': weights, 'margin': 0.2, 'measure': 'dot',
                           'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}}
    im, s, il, sl = synth.structured_alignment_batch(B * world, 34, 50, 768, seed=977, noise=3.0, ragged=True)
    ie, ce = synth.global_embeddings(B * world, 768, seed=978, noise=3.0)
    blk = slice(rank * B, (rank + 1) * B)
    out = []
    for epoch in epochs:
        m = ALADModel(config, shard_group=None)
        m.logger = LogCollector()
        leaves = [torch.from_numpy(x[blk].copy()).to(dev).requires_grad_(True) for x in (ie, ce, im, s)]
        loss, d = m.forward_loss_total(leaves[0], leaves[1], leaves[2].permute(1, 0, 2), leaves[3].permute(1, 0, 2), il[blk], sl[blk], 0,
                                       epoch=epoch, distill_epoch=2)
        loss.backward()
        torch.cuda.synchronize()
        out.append((float(loss), {k: float(v) for k, v in d.items()}, {k: (mm.val, mm.count) for k, mm in m.logger.meters.items()},
                    [None if t.grad is None else t.grad.cpu() for t in leaves]))
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()
