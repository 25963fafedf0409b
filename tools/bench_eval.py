#!/usr/bin/env python3
"""Timing of BASELINE configs[2] (evaluation): 5000 x 25000 x 768 matching-head similarity + ranks,
and the alignment-head grid 1000 x 5000 at padded length 71.  Prints one JSON line per workload.
Not the driver's bench (that is bench.py); used for DESIGN.md / profiles."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from aladin_amd import evaluation as E, ops, synth


def timed(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    dev = torch.device('cuda:0')
    n_img = 5000
    img, cap = synth.retrieval_embeddings(n_img, 768, seed=303, sigma=12.0)
    a = torch.from_numpy(img[0::5]).to(dev)
    b = torch.from_numpy(cap).to(dev)
    sim = ops.sim_matrix(a, b)
    ms_sim = timed(lambda: ops.sim_matrix(a, b))
    ms_rank = timed(lambda: ops.recall_ranks(sim))
    ms_torch = timed(lambda: torch.mm(a, b.t()))
    ms_fused = timed(lambda: ops.retrieval_ranks(a, b))
    for x, y in zip(ops.retrieval_ranks(a, b), ops.recall_ranks(sim)):
        assert torch.equal(x, y)
    flops = 2.0 * n_img * 5 * n_img * 768
    print(json.dumps({'workload': 'configs[2] matching head 5000x25000x768', 'sim_ms': round(ms_sim, 3),
                      'rank_ms': round(ms_rank, 3), 'torch_mm_fp32_ms': round(ms_torch, 3),
                      'fused_sim_plus_rank_ms': round(ms_fused, 3),
                      'sim_tflops_algorithmic': round(flops / ms_sim / 1e9, 1)}))
    n = 1000
    images, captions, il, cl = synth.eval_sets(n, 768, seed=9)
    ia = torch.from_numpy(images[0::5]).to(dev)
    ca = torch.from_numpy(captions).to(dev)
    images_d = torch.from_numpy(images).to(dev)
    ilen = il[0::5]
    pairs = n * 5 * n
    from aladin_amd.store import PackedSetStore
    for prec in ('split', 'fp16'):                       # 'split' = the rank-exact evaluation default, 'fp16' = training operands
        ops.set_eval_precision(prec)
        ms_align = timed(lambda: E.compute_sim_matrix(ia, ca, ilen, cl, mode='alignment'), iters=5)
        print(json.dumps({'workload': 'alignment-head grid 1000x5000 (sets padded to L=71; trimmed to the real lengths + one masked region), precision ' + prec,
                          'ms': round(ms_align, 3), 'pairs_per_s': round(pairs / ms_align * 1e3, 1)}))
        # i2t + t2i drivers on device-resident (N, 71, D) tensors: one grid (memoised), ranks, top-50 lists
        def drivers():
            E.clear_eval_cache()
            E.i2t(images_d, ca, il, cl, sim_function='alignment')
            return E.t2i(images_d, ca, il, cl, sim_function='alignment', return_ranks=True)
        ms_drv = timed(drivers, iters=3, warm=1)
        print(json.dumps({'workload': 'i2t + t2i (alignment head, COCO-1k protocol, ranks + top-50, metrics on the host), precision ' + prec,
                          'ms': round(ms_drv, 3)}))
        # the same grid from the packed stores (aladin_amd/store.py): operands are row copies
        si, sc = PackedSetStore(768, 0, dev, precision=prec), PackedSetStore(768, 2, dev, precision=prec)
        for k0 in range(0, images.shape[0], 500):
            si.append(images_d[k0:k0 + 500], il[k0:k0 + 500])
            sc.append(ca[k0:k0 + 500], cl[k0:k0 + 500])
        view = si.view(slice(0, None, 5))
        assert torch.equal(E.compute_sim_matrix(view, sc, mode='alignment'), E.compute_sim_matrix(ia, ca, ilen, cl, mode='alignment'))
        ms_store = timed(lambda: E.compute_sim_matrix(view, sc, mode='alignment'), iters=5)
        print(json.dumps({'workload': 'alignment-head grid 1000x5000 from PackedSetStore (packed by true length), bit-identical scores, precision ' + prec,
                          'ms': round(ms_store, 3), 'pairs_per_s': round(pairs / ms_store * 1e3, 1),
                          'store_MB': round((si.nbytes() + sc.nbytes()) / 2 ** 20, 1),
                          'dense_fp32_MB': round((images.nbytes + captions.nbytes) / 2 ** 20, 1)}))
    ops.set_eval_precision('fp16')
    ms_align_padded = timed(lambda: ops._align_forward(ia, ca, ops.lengths_tensor(ilen, dev), ops.lengths_tensor(cl, dev))[0], iters=3)
    print(json.dumps({'workload': 'the same grid scored on the padded 70 x 68 blocks (no trimming), fp16', 'ms': round(ms_align_padded, 3)}))

    # COCO-5k sized alignment-head retrieval (5000 images x 25000 captions, lengths as in COCO: 12..34 regions,
    # 7..30 tokens) from packed stores filled batch by batch on the device; device-generated features.
    g = torch.Generator(device=dev).manual_seed(5)
    il5 = torch.randint(12, 35, (5000,), generator=g, device=dev).tolist()
    cl5 = torch.randint(7, 31, (25000,), generator=g, device=dev).tolist()
    for prec in ('split', 'fp16'):
        si5 = PackedSetStore(768, 0, dev, capacity_rows=5000 * 34, precision=prec)
        sc5 = PackedSetStore(768, 2, dev, capacity_rows=25000 * 28, precision=prec)
        g.manual_seed(5)
        for k0 in range(0, 5000, 500):
            si5.append(torch.randn((500, 34, 768), generator=g, device=dev), il5[k0:k0 + 500])
        for k0 in range(0, 25000, 500):
            sc5.append(torch.randn((500, 30, 768), generator=g, device=dev), cl5[k0:k0 + 500])
        torch.cuda.synchronize()

        def full():
            S5 = E.compute_sim_matrix(si5, sc5, mode='alignment')
            return ops.recall_ranks(S5)

        ms_full = timed(full, iters=3, warm=1)
        print(json.dumps({'workload': 'COCO-5k sized alignment-head retrieval: 5000 x 25000 grid from PackedSetStores + all four rank outputs, precision ' + prec,
                          'ms': round(ms_full, 2), 'pairs_per_s': round(5000 * 25000 / ms_full * 1e3, 1),
                          'store_MB': round((si5.nbytes() + sc5.nbytes()) / 2 ** 20, 1),
                          'reference_buffers_MB': round((5000 * 5 + 25000) * 71 * 768 * 4 / 2 ** 20, 1)}))
        del si5, sc5


if __name__ == '__main__':
    main()
