#!/usr/bin/env python3
"""Per-rank COMPUTE of the sharded step at world size W, emulated on one GPU (no communication):
what one rank runs between the collectives -- pack, (W*B x B) score block, hinge on the (W*B)^2
matrix, and the backward of its caption block in both exchange forms.  usage: emulate_rank_compute.py [W ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def timed(fn, n=30, preroll_s=0.5):
    """HIP-event time per call after a clock-settling pre-roll (round 2 timed 10 eager calls on a cold clock)."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < preroll_s:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    from aladin_amd import distributed as DD, ops, synth
    dev = torch.device('cuda:0')
    B, R, T, D = 256, 34, 50, 768
    for W in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
        ims, caps = [], []
        for r in range(W):
            im, s, il, sl = synth.alignment_batch(B, R, T, D, seed=1234 + 17 * r)
            ims.append(torch.from_numpy(im).to(dev))
            caps.append(torch.from_numpy(s).to(dev))
        ilt = ops.lengths_tensor(il, dev)
        slt = ops.lengths_tensor(sl, dev)
        g_loc, g_glob, ok = DD._local_and_global_geometry(B, W, R, T, D)
        assert ok
        packs = [ops.pack_images(x, ilt, g_loc) for x in ims]
        xm_all, xe_all = torch.cat([p[0] for p in packs]), torch.cat([p[1] for p in packs])
        il_all = torch.cat([ilt] * W)
        im_all = torch.cat(ims)
        t_pack = timed(lambda: ops.pack_images(ims[0], ilt, g_loc))
        t_scores = timed(lambda: DD.rank_scores_block(xm_all, xe_all, caps[0], slt, g_glob))
        # its parts: caption pack, side GEMM + score kernel, score kernel alone (side result reused)
        y0 = ops.pack_captions(caps[0], slt, g_glob)
        S0 = torch.empty((W * B, B), dtype=torch.float32, device=dev)
        e0 = torch.empty(g_glob.e_bytes, dtype=torch.uint8, device=dev)
        t_packc = timed(lambda: ops.pack_captions(caps[0], slt, g_glob))
        t_side_score = timed(lambda: ops.scores_from_packed(xm_all, xe_all, y0, g_glob, S0, e0))
        t_score_only = timed(lambda: ops.scores_from_packed(xm_all, xe_all, y0, g_glob, S0, e0, reuse_side=True))
        blocks = [DD.rank_scores_block(xm_all, xe_all, caps[r], slt, g_glob) for r in range(W)]
        S_full = torch.cat([b[0] for b in blocks], dim=1).contiguous()
        t_hinge = timed(lambda: ops._hinge_raw(S_full, 0.2, True, True))
        loss, dS_full, _ = ops._hinge_raw(S_full, 0.2, True, True)
        t_dense = timed(lambda: DD.rank_backward_block(im_all, il_all, caps[0], slt, dS_full, 0, g_glob, xm_all, xe_all, blocks[0][1]))
        need = torch.nonzero((dS_full[:, :B] != 0).any(dim=1)).flatten()
        im_need, il_need = im_all.index_select(0, need), il_all.index_select(0, need)
        dS_need = dS_full.index_select(0, need)[:, :B].contiguous()
        t_sparse = timed(lambda: ops._align_backward(im_need, caps[0], il_need, slt, dS_need))
        # round 6: the pair-driven exchange packs its compact problem (one launch) so that the pair kernel runs on the fp16 MFMA and the
        # row step can take its partner rows from the packed operands (ops.set_backward_precision)
        g_need = ops.align_geometry(im_need.shape[0], B, R, T, D)

        def compact_packed():
            packed = ops.pack_sets(im_need, caps[0], il_need, slt, g_need)
            return ops._align_backward(im_need, caps[0], il_need, slt, dS_need, packed=packed)
        t_sparse_packed = {}
        for mode in ('exact', 'fp16'):
            old = ops.set_backward_precision(mode)
            t_sparse_packed[mode] = round(timed(compact_packed), 3)
            ops.set_backward_precision(old)
        print(json.dumps({'W': W, 'pack_ms': round(t_pack, 3), 'scores_block_ms': round(t_scores, 3), 'pack_captions_ms': round(t_packc, 4),
                          'side_plus_score_ms': round(t_side_score, 4), 'score_kernel_ms': round(t_score_only, 4),
                          'score_kernel_us_per_256x256_block': round(t_score_only * 1e3 / W, 2), 'hinge_ms': round(t_hinge, 3),
                          'bwd_dense_ms': round(t_dense, 3), 'bwd_compact_ms': round(t_sparse, 3), 'bwd_compact_packed_ms': t_sparse_packed, 'images_needed': int(need.numel()),
                          'of': W * B, 'dense_exchange_MB': round(2 * W * B * R * D * 4 / 2 ** 20, 1),
                          'sparse_exchange_MB': round(2 * int(need.numel()) * R * D * 4 / 2 ** 20 * (W - 1) / W, 1)}), flush=True)


if __name__ == '__main__':
    main()
