#!/usr/bin/env python3
"""VERDICT r3 item 8, the measurement it asks for BEFORE a re-score kernel is written: on the COCO-1k alignment-head grid
(tests/golden/eval_coco1k_d768.npz's generator: 1000 images x 5000 captions, D = 768), how many (image, caption) pairs does a
RIGOROUS per-pair error band around the fp16-operand score leave undecided against the ground truth's score?

  score      S[i,c] = sum_w max_r <x^_ir, y^_cw>                                  (alad/loss.py:97-125)
  operands   x^ = hi + lo, hi = fp16(x^):  <x^,y^> - <hi_x,hi_y> = <lo_x,hi_y> + <hi_x,lo_y> + <lo_x,lo_y>
  band       |max_r a_r - max_r b_r| <= max_r |a_r - b_r|, Cauchy-Schwarz per (r, w), summed over the caption's words:
             band(i,c) = P_i Q_c + R_i T_c + P_i T_c,  P_i = max_r |lo_ir|, R_i = max_r |hi_ir|, Q_c = sum_w |hi_cw|, T_c = sum_w |lo_cw|
             (+ the fp32 accumulation of 768-term dot products, not counted here: the band below is a LOWER bound of a rigorous one)
  undecided  i2t: |S16[i,c] - S16[i,gt(i)]| <= band(i,c) + band(i,gt(i)) for the image's best ground-truth caption;
             t2i: |S16[i,c] - S16[gt(c),c]| <= band(i,c) + band(gt(c),c)
Prints the fractions, the measured |S16 - S_split| next to the band, and what a re-score pass would cost.
  python tools/alignment_band_probe.py [fixture] > gpurun_out/alignment_band_probe.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from aladin_amd import evaluation as E, ops, synth


def main():
    fixture = sys.argv[1] if len(sys.argv) > 1 else 'eval_coco1k_d768'
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', fixture + '.npz'))
    n_img, D = int(g['n_img']), int(g['D'])
    images, captions, il, cl = synth.eval_sets(n_img, D, int(g['seed']), base_weight=float(g['gen_base_weight']),
                                               img_len_range=tuple(int(v) for v in g['gen_img_len_range']),
                                               cap_len_range=tuple(int(v) for v in g['gen_cap_len_range']), n_full=int(g['gen_n_full']))
    dev = torch.device('cuda:0')
    ia, ca = torch.from_numpy(images[0::5]).to(dev), torch.from_numpy(captions).to(dev)
    ilen = il[0::5]
    S = {}
    for prec in ('split', 'fp16'):
        ops.set_eval_precision(prec)
        E.clear_eval_cache()
        S[prec] = E.compute_sim_matrix(ia, ca, ilen, cl, mode='alignment').double()
    ops.set_eval_precision('split')

    def norms(sets, lens, tail):
        """per sample: (max |lo|, max |hi|, sum |hi|, sum |lo|) over the scored positions 1 .. len-1-tail"""
        x = sets.double()
        xh = x / x.norm(dim=2, keepdim=True).clamp_min(1e-300)
        hi = xh.float().half().double()
        lo = xh - hi
        nh, nl = hi.norm(dim=2), lo.norm(dim=2)
        pos = torch.arange(sets.shape[1], device=sets.device)[None, :]
        L = torch.tensor(lens, device=sets.device)[:, None]
        m = (pos >= 1) & (pos < L - tail)
        z = torch.zeros_like(nh)
        return (torch.where(m, nl, z).amax(1), torch.where(m, nh, z).amax(1), torch.where(m, nh, z).sum(1), torch.where(m, nl, z).sum(1))
    P, R, _, _ = norms(ia, ilen, 0)
    _, _, Q, Tn = norms(ca, cl, 2)
    band = P[:, None] * Q[None, :] + R[:, None] * Tn[None, :] + P[:, None] * Tn[None, :]
    err = (S['fp16'] - S['split']).abs()
    print('grid %d x %d, D = %d (fixture %s)' % (n_img, 5 * n_img, D, fixture))
    print('|S_fp16 - S_split|: max %.3e  mean %.3e;   rigorous band: min %.3e  mean %.3e  max %.3e;   worst err / band %.3f'
          % (err.max(), err.mean(), band.min(), band.mean(), band.max(), (err / band).max()))
    print('score spread: std over captions of one image %.3f (mean over images); ground-truth margin to the row median %.3f'
          % (S['split'].std(dim=1).mean(), (S['split'][torch.arange(n_img, device=dev).repeat_interleave(5), torch.arange(5 * n_img, device=dev)]
                                           - S['split'].median(dim=1).values.repeat_interleave(5)).mean()))
    S16 = S['fp16']
    gt_img = torch.arange(5 * n_img, device=dev) // 5
    # i2t: the best of the image's five ground truths decides the rank (alad/evaluation.py:196-223)
    gts = S16.view(n_img, n_img, 5)[torch.arange(n_img), torch.arange(n_img)]          # (n_img, 5) scores of the own captions
    best = gts.argmax(dim=1) + 5 * torch.arange(n_img, device=dev)
    s_gt, b_gt = S16[torch.arange(n_img), best], band[torch.arange(n_img), best]
    und_i = ((S16 - s_gt[:, None]).abs() <= band + b_gt[:, None])
    und_i[torch.arange(n_img), best] = False
    # t2i: the caption's own image
    s_gc, b_gc = S16[gt_img, torch.arange(5 * n_img)], band[gt_img, torch.arange(5 * n_img)]
    und_t = ((S16 - s_gc[None, :]).abs() <= band + b_gc[None, :])
    und_t[gt_img, torch.arange(5 * n_img)] = False
    either = und_i | und_t
    n = S16.numel()
    print('undecided pairs  i2t %.4f %%  t2i %.4f %%  either %.4f %% of %d  (queries with any: i2t %d / %d, t2i %d / %d)'
          % (100.0 * und_i.sum() / n, 100.0 * und_t.sum() / n, 100.0 * either.sum() / n, n, int(und_i.any(1).sum()), n_img, int(und_t.any(0).sum()), 5 * n_img))
    # what the measured error would need (NOT rigorous): a band of 4 x the worst observed error
    emp = 4.0 * float(err.max())
    und_e = (((S16 - s_gt[:, None]).abs() <= 2 * emp) | ((S16 - s_gc[None, :]).abs() <= 2 * emp))
    print('for comparison, an EMPIRICAL band of 4 x the worst measured error (%.2e, not a bound): %.4f %% undecided' % (emp, 100.0 * und_e.sum() / n))
    # ranks the reference protocol also returns top lists (t2i: top 50 images per caption), which need the ORDER among the first
    # 50 scores of every column, not only the comparison with the ground truth
    top = S16.topk(51, dim=0).values                       # (51, 5 n_img)
    gaps = top[:-1] - top[1:]
    bmax = band.amax(dim=0)
    print('t2i top-50 lists: %.1f %% of the captions have two of their first 51 scores closer than twice the band (order undecided)'
          % (100.0 * (gaps <= 2 * bmax[None, :]).any(dim=0).double().mean()))


if __name__ == '__main__':
    main()
