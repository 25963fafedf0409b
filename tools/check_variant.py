#!/usr/bin/env python3
"""Scores of the bench batch (B = 256, full lengths) and of a ragged batch from whatever library / variant the environment
selects (ALADIN_LIB, ALADIN_SCORE_VARIANT ... of the diag build), saved for a bit-for-bit comparison between variants:

    python tools/check_variant.py save gpurun_out/S_base.pt
    ALADIN_LIB=aladin_amd/lib/libaladin_hip_diag.so ALADIN_SCORE_VARIANT=1 python tools/check_variant.py save gpurun_out/S_v1.pt
    python tools/check_variant.py cmp gpurun_out/S_base.pt gpurun_out/S_v1.pt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    if sys.argv[1] == 'cmp':
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        ok = True
        for k in a:
            same = torch.equal(a[k], b[k])
            ok &= same
            print('%-12s %s  max |diff| %.3g' % (k, 'bit-identical' if same else 'DIFFERENT', float((a[k] - b[k]).abs().max())))
        sys.exit(0 if ok else 1)
    from aladin_amd import ops, synth
    dev = torch.device('cuda:0')
    out = {}
    for tag, ragged, seed in (('full', False, 1234), ('ragged', True, 99)):
        im, s, il, sl = synth.alignment_batch(256, 34, 50, 768, seed=seed, ragged=ragged)
        with torch.no_grad():
            out[tag] = ops.alignment_scores(torch.from_numpy(im).to(dev), torch.from_numpy(s).to(dev), il, sl, precision='fp16').cpu()
    torch.save(out, sys.argv[2])
    print('saved', sys.argv[2], {k: float(v.double().sum()) for k, v in out.items()})


if __name__ == '__main__':
    main()
