#!/usr/bin/env python3
"""Phase stamps of the backward's pair kernel (diagnostic build): where a workgroup's time goes.

    python tools/pair_probe.py [B ...]        # default 32 256

Prints per batch size the median over workgroups of each phase in microseconds (s_memrealtime runs at 100 MHz):
header loads, MFMA phase (K loop), word scan, exact fp32 candidates, table write; and the median candidate count."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("ALADIN_LIB", os.path.join(ROOT, "aladin_amd", "lib", "libaladin_hip_diag.so"))
import numpy as np
import torch
from aladin_amd import _lib, synth
from aladin_amd.loss import AlignmentContrastiveLoss


def main():
    lib = _lib.load()
    lib.aladin_debug_read_pair_probe.restype = C.c_int
    batches = [int(v) for v in sys.argv[1:]] or [32, 256]
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=True, aggregation='MrSw')
    for B in batches:
        im_np, s_np, il, sl = synth.alignment_batch(B, seed=1234, ragged=False)
        im = torch.from_numpy(im_np).cuda().requires_grad_(True)
        s = torch.from_numpy(s_np).cuda().requires_grad_(True)
        for _ in range(3):
            im.grad = s.grad = None
            crit(im, s, il, sl).backward()
        torch.cuda.synchronize()
        nb = min(1024, 3 * B)
        buf = (C.c_ulonglong * (8 * nb))()
        assert lib.aladin_debug_read_pair_probe(buf, nb) == 0
        a = np.array(buf, dtype=np.int64).reshape(nb, 8)
        a = a[a[:, 5] > a[:, 0]]
        d = np.diff(a[:, :6], axis=1) / 100.0
        names = ["header", "mfma", "scan", "candidates", "write"]
        print(json.dumps({"batch": B, "workgroups": int(len(a)), **{n + "_us": round(float(np.median(d[:, k])), 2) for k, n in enumerate(names)},
                          "total_us": round(float(np.median(a[:, 5] - a[:, 0])) / 100.0, 2), "candidates": int(np.median(a[:, 6])),
                          "span_us": round(float(a[:, 5].max() - a[:, 0].min()) / 100.0, 2)}))


if __name__ == "__main__":
    main()
