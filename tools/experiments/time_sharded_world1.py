#!/usr/bin/env python3
"""Single-GPU timing of the sharded fast path's two backward exchanges under RCCL, world_size 1:
what the planner + compact backward of SparseImageExchange cost next to the dense form when no
data actually has to travel (the multi-GPU runs are the driver's)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29655')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda:0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    from aladin_amd import synth
    from aladin_amd.distributed import sharded_alignment_loss_fast
    B = int(os.environ.get('B', 256))
    im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=1234)
    a = torch.from_numpy(im).to(dev).requires_grad_(True)
    b = torch.from_numpy(s).to(dev).requires_grad_(True)
    for mode in ('dense', 'sparse', 'dense', 'sparse'):
        def step():
            a.grad = None
            b.grad = None
            loss, _ = sharded_alignment_loss_fast(a, b, il, sl, 0.2, True, exchange=mode)
            loss.backward()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        print('%-6s %.3f ms/step' % (mode, (time.perf_counter() - t0) * 10))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
