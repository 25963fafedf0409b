#!/usr/bin/env python3
"""How fast does the REFERENCE's formulation run on this MI355X through PyTorch-ROCm?

The reference is plain PyTorch, so on an MI355X it would run as ATen/rocBLAS kernels.  This tool
times, on the same synthetic batch as bench.py (config 2: B=256, R=34, T=50, D=768, full lengths,
margin 0.2, max_violation, forward + backward):

  faithful    the dataflow of alad/loss.py:80-125 + :42-67 as the reference issues it: expand both
              sets to (B, B, ., D), one batched matmul over B*B tiny problems, boolean masks built
              per batch row, masked_fill, max over regions, sum over words, VSE++ hinge  (fp32)
  one-gemm    the same maths as one (B*R') x D x (B*T') GEMM + reshape/max/sum (fp32) with the
              masks built without Python loops: the best a maintainer gets inside eager PyTorch
  one-gemm16  one-gemm with fp16 operands (torch.autocast), the precision class of the HIP path
  hip         this repo's fused node (aladin_amd.ops.alignment_triplet_loss), eager launches

and prints one JSON line per variant.  It restates the formulations itself (tools/ may not import
oracle/); tests/test_gpu_parity.py is where parity is checked, not here.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F


def masks(im_len, s_len, Rp, Tp, dev):
    # the reference fills these row by row in Python (alad/loss.py:103-115)
    im_mask = torch.zeros(len(im_len), Rp, dtype=torch.bool, device=dev)
    for row, l in zip(im_mask, im_len):
        row[l - 1:] = True
    s_mask = torch.zeros(len(s_len), Tp, dtype=torch.bool, device=dev)
    for row, l in zip(s_mask, s_len):
        row[l - 3:] = True
    return im_mask, s_mask


def masks_vectorised(im_len, s_len, Rp, Tp, dev):
    il = torch.as_tensor(im_len, device=dev) - 1
    sl = torch.as_tensor(s_len, device=dev) - 3
    return (torch.arange(Rp, device=dev)[None, :] >= il[:, None]), (torch.arange(Tp, device=dev)[None, :] >= sl[:, None])


def hinge(scores, margin):
    diag = scores.diag().view(-1, 1)
    cost_s = (margin + scores - diag).clamp(min=0)
    cost_im = (margin + scores - diag.t()).clamp(min=0)
    eye = torch.eye(scores.size(0), device=scores.device) > .5
    cost_s = cost_s.masked_fill(eye, 0)
    cost_im = cost_im.masked_fill(eye, 0)
    return cost_s.max(1)[0].sum() + cost_im.max(0)[0].sum()


def faithful(im, s, im_len, s_len, margin):
    a = F.normalize(im, p=2, dim=2)[:, 1:, :]
    b = F.normalize(s, p=2, dim=2)[:, 1:-2, :]
    Bi, Rp, Bc, Tp = a.size(0), a.size(1), b.size(0), b.size(1)
    a4 = a.unsqueeze(1).expand(-1, Bc, -1, -1)
    b4 = b.unsqueeze(0).expand(Bi, -1, -1, -1)
    al = torch.matmul(a4, b4.permute(0, 1, 3, 2))
    im_mask, s_mask = masks(im_len, s_len, Rp, Tp, im.device)
    dead = im_mask[:, None, :, None] | s_mask[None, :, None, :]
    al = al.masked_fill(dead, 0)
    return hinge(al.max(2)[0].sum(2), margin)


def one_gemm(im, s, im_len, s_len, margin):
    a = F.normalize(im, p=2, dim=2)[:, 1:, :]
    b = F.normalize(s, p=2, dim=2)[:, 1:-2, :]
    Bi, Rp, Bc, Tp = a.size(0), a.size(1), b.size(0), b.size(1)
    al = (a.reshape(Bi * Rp, -1) @ b.reshape(Bc * Tp, -1).t()).view(Bi, Rp, Bc, Tp).float()
    im_mask, s_mask = masks_vectorised(im_len, s_len, Rp, Tp, im.device)
    al = al.masked_fill(im_mask[:, :, None, None] | s_mask[None, None, :, :], 0)
    return hinge(al.max(1)[0].sum(2), margin)


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    args = ap.parse_args()
    from aladin_amd import ops, synth
    B, R, T, D = args.batch, 34, 50, 768
    dev = torch.device('cuda:0')
    im_np, s_np, im_len, s_len = synth.alignment_batch(B, R, T, D, seed=1234)
    im = torch.from_numpy(im_np).to(dev).requires_grad_(True)
    s = torch.from_numpy(s_np).to(dev).requires_grad_(True)

    def step_of(loss_fn, autocast=False):
        def step():
            im.grad = None
            s.grad = None
            if autocast:
                with torch.autocast('cuda', dtype=torch.float16):
                    loss = loss_fn(im, s, im_len, s_len, 0.2)
            else:
                loss = loss_fn(im, s, im_len, s_len, 0.2)
            loss.backward()
            return loss
        return step

    def hip_step():
        im.grad = None
        s.grad = None
        loss, _ = ops.alignment_triplet_loss(im, s, im_len, s_len, 0.2, True)
        loss.backward()
        return loss

    variants = [('hip', hip_step, args.steps * 20), ('one-gemm', step_of(one_gemm), args.steps),
                ('one-gemm16', step_of(one_gemm, True), args.steps), ('faithful', step_of(faithful), args.steps)]
    ref_loss = None
    for name, fn, steps in variants:
        try:
            loss = float(fn().detach())
            sec = timed(fn, steps, args.warmup)
        except Exception as exc:                                   # e.g. out of memory on the 16 GB expand
            print(json.dumps({'variant': name, 'error': str(exc)[:200]}))
            continue
        ref_loss = loss if ref_loss is None else ref_loss
        print(json.dumps({'variant': name, 'ms_per_step': round(sec * 1e3, 4), 'pairs_per_s': round(B * B / sec, 1),
                          'loss': round(loss, 5), 'steps': steps, 'batch': B,
                          'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
        torch.cuda.reset_peak_memory_stats()
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
