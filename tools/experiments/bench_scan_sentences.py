#!/usr/bin/env python3
"""Time aggregation='scan-sentences' (fwd + bwd of the triplet loss) on one MI355X: this library vs
the reference's own dataflow in eager PyTorch-ROCm (restated here; tools/ may not import oracle/)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F


def reference_style(im, s, im_len, s_len):
    a = F.normalize(im, p=2, dim=2)[:, 1:, :]
    b = F.normalize(s, p=2, dim=2)[:, 1:-2, :]
    Bi, Rp, Bc, Tp = a.size(0), a.size(1), b.size(0), b.size(1)
    a4 = a.unsqueeze(1).expand(-1, Bc, -1, -1)
    b4 = b.unsqueeze(0).expand(Bi, -1, -1, -1)
    al = torch.matmul(a4, b4.permute(0, 1, 3, 2))
    dev = im.device
    rdead = torch.arange(Rp, device=dev)[None, :] >= (torch.as_tensor(im_len, device=dev) - 1)[:, None]
    wdead = torch.arange(Tp, device=dev)[None, :] >= (torch.as_tensor(s_len, device=dev) - 3)[:, None]
    dead = rdead[:, None, :, None] | wdead[None, :, None, :]
    al = al.masked_fill(dead, 0)
    nrm = F.normalize(F.relu(al), p=2, dim=2)
    w = torch.softmax(nrm.masked_fill(dead, float('-inf')), dim=3).unsqueeze(3)
    att = torch.matmul(w, b4.unsqueeze(2).expand(-1, -1, Rp, -1, -1)).squeeze(3)      # (Bi,Bc,R',D) via (Bi,Bc,R',T',D)
    cos = F.cosine_similarity(a4, att, dim=3)
    return cos.masked_fill(rdead[:, None, :].expand_as(cos), 0).sum(2)


def timed(fn, steps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    from aladin_amd import ops, synth
    dev = torch.device('cuda:0')
    for B in (32, 128, 256):
        im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=1234)
        a = torch.from_numpy(im).to(dev).requires_grad_(True)
        b = torch.from_numpy(s).to(dev).requires_grad_(True)

        def ours():
            a.grad = None
            b.grad = None
            ops.hinge_loss(ops.alignment_scan_scores(a, b, il, sl), 0.2, True).backward()

        def ref():
            a.grad = None
            b.grad = None
            S = reference_style(a, b, il, sl)
            d = S.diag().view(-1, 1)
            eye = torch.eye(B, device=dev) > .5
            ((0.2 + S - d).clamp(min=0).masked_fill(eye, 0).max(1)[0].sum()
             + (0.2 + S - d.t()).clamp(min=0).masked_fill(eye, 0).max(0)[0].sum()).backward()

        row = {'B': B, 'hip_ms': round(timed(ours, 5), 3)}
        if B <= 32:
            with torch.no_grad():
                err = float((ops.alignment_scan_scores(a, b, il, sl) - reference_style(a, b, il, sl)).abs().max())
            row.update(torch_rocm_eager_ms=round(timed(ref, 3), 3), max_abs_score_diff=err,
                       eager_peak_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1))
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
