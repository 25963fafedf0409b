#!/usr/bin/env python3
"""The forward chain (pack + side GEMM + score kernel) with the side GEMM on a second stream beside the main rows' packing
(ops.SIDE_OVERLAP) against the single-stream chain: eager launches and a replayed HIP graph, interleaved on one box; the score
matrices must be bit-identical.   python tools/forward_overlap_probe.py [B R T D]..."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aladin_amd import ops, synth


def ev_ms(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    shapes = [(256, 34, 50, 768), (256, 51, 38, 768), (256, 36, 50, 768), (128, 34, 50, 768), (64, 34, 50, 768)]
    if len(sys.argv) >= 5:
        shapes = [tuple(int(v) for v in sys.argv[1:5])]
    dev = torch.device('cuda:0')
    for B, R, T, D in shapes:
        im, s, il, sl = synth.alignment_batch(B, R, T, D, seed=B + R, ragged=False)
        a, b = torch.from_numpy(im).to(dev), torch.from_numpy(s).to(dev)
        ilt, slt = ops.lengths_tensor(il, dev), ops.lengths_tensor(sl, dev)
        ops.SIDE_OVERLAP_MIN_PAIRS = 1
        out = {}
        for flag in (False, True):
            ops.SIDE_OVERLAP = flag
            out[flag] = ops._align_forward(a, b, ilt, slt)[0].clone()
        assert torch.equal(out[False], out[True]), 'scores differ'
        res = {('eager', False): [], ('eager', True): [], ('graph', False): [], ('graph', True): []}
        graphs = {}
        for flag in (False, True):
            ops.SIDE_OVERLAP = flag
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                for _ in range(3):
                    ops._align_forward(a, b, ilt, slt)
            torch.cuda.current_stream().wait_stream(st)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                keep = ops._align_forward(a, b, ilt, slt)
            graphs[flag] = (g, keep)
        for rep in range(5):
            for flag in (False, True):
                ops.SIDE_OVERLAP = flag
                fn = lambda: ops._align_forward(a, b, ilt, slt)
                for _ in range(20):
                    fn()
                res[('eager', flag)].append(ev_ms(fn, 200))
                g = graphs[flag][0]
                for _ in range(20):
                    g.replay()
                res[('graph', flag)].append(ev_ms(g.replay, 200))
        assert torch.equal(graphs[True][1][0], out[False]), 'graph scores differ'
        print('B %d R %d T %d D %d   forward chain, median of 5 x 200 (us):  eager single-stream %.1f  overlapped %.1f   graph single-stream %.1f  overlapped %.1f'
              % (B, R, T, D, 1e3 * statistics.median(res[('eager', False)]), 1e3 * statistics.median(res[('eager', True)]),
                 1e3 * statistics.median(res[('graph', False)]), 1e3 * statistics.median(res[('graph', True)])), flush=True)


if __name__ == '__main__':
    main()
