"""phase stamps of sim_screen_kernel (ALADIN_LIB=aladin_amd/lib/libaladin_hip_diag.so: make -C aladin_amd/csrc diag): per-tile wall-clock stamps in the tile's list segment"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aladin_amd import ops, synth, _lib
dev = torch.device('cuda:0')
lib = _lib.load()
for sigma, exact in ((3.0, False), (6.0, False), (8.0, False), (12.0, False), (12.0, True)):
    i, c = synth.retrieval_embeddings(5000, 768, seed=303, sigma=sigma)
    a, b = torch.from_numpy(i[0::5]).to(dev), torch.from_numpy(c).to(dev)
    n_img, n_cap, D = 5000, 25000, 768
    nbytes = lib.aladin_retrieval_workspace_bytes(n_img, n_cap, D)
    for _ in range(3): ops.retrieval_ranks(a, b, exact=exact)
    # call through ops but keep the workspace: re-implement the call
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    outs = [torch.empty(n, dtype=torch.int32, device=dev) for n in (n_img, n_img, n_cap, n_cap)]
    fn = lib.aladin_retrieval_ranks_exact if exact else lib.aladin_retrieval_ranks
    rc = fn(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), n_img, n_cap, D, 5, *[o.data_ptr() for o in outs], ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    so = lib.aladin_retrieval_stats_offset(n_img, n_cap, D)
    n_tiles = 20 * 66
    CAP = 512                                        # SIM_LIST_CAP (csrc/recall.hip)
    r256 = lambda b: (b + 255) // 256 * 256
    # workspace behind the statistics (retr_layout): lob_i2t (Mp words), lob_t2i (Np words), list_cnt (n_tiles words), lists
    lo = so + 256 + r256(20 * 256 * 4) + r256(66 * 384 * 4) + r256(n_tiles * 4)
    seg = ws[lo:lo + n_tiles * CAP * 16].view(torch.int64).view(n_tiles, 2 * CAP)[:, 2 * CAP - 12:].cpu().numpy().astype(np.float64)
    t0 = seg[:, 0].min()
    d = np.diff(seg[:, :8], axis=1) / 100.0          # 100 MHz -> us
    ep = np.round(np.median(np.diff(np.concatenate([seg[:, 6:7], seg[:, 8:12], seg[:, 7:8]], axis=1), axis=1) / 100.0, axis=0), 2).tolist()
    print('   exact epilogue: init+barrier, rows, cols, barrier, global atomics:', ep)
    print('   diag counters [exact tiles, listed pairs, cheap group tests, full group evaluations, waves in phase 2, rescored, analysed, overflowed, skipped]:', ws[so:so + 36].view(torch.int32).cpu().tolist())
    if not exact:
        listed = seg[:, 8] > 0
        if listed.any():
            print('   listed tiles: phase 2a (incl. barrier) / 2b+flush us, median:', round(float(np.median(seg[listed, 8] - seg[listed, 3]) / 100.0), 2), round(float(np.median(seg[listed, 4] - seg[listed, 8]) / 100.0), 2))
    print('sigma', sigma, 'exact' if exact else 'screen', 'per-phase us (median over tiles):', np.round(np.median(d, axis=0), 2).tolist(),
          'tile total median', round(float(np.median((seg[:, 7] if (exact or sigma > 10) else seg[:, 4]) - seg[:, 0]) / 100.0), 2),
          'kernel span', round(float((seg.max() - t0) / 100.0), 1))
    if '--detail' in sys.argv:
        end = np.where(seg[:, 7] > 0, seg[:, 7], seg[:, 4])
        dur = (end - seg[:, 0]) / 100.0
        order = np.argsort(seg[:, 0])
        print('   tile duration us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f' % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max()))
        ev = sorted([(t, 1) for t in seg[:, 0]] + [(t, -1) for t in end])
        c = m = 0
        for _, d_ in ev:
            c += d_; m = max(m, c)
        print('   max concurrent tiles', m, ' start times (us) of every 128th tile:', np.round((np.sort(seg[:, 0]) - t0)[::128] / 100.0, 1).tolist())
        slow = np.argsort(-dur)[:5]
        for k in slow:
            print('   slow tile', int(k), 'phases', np.round(np.diff(seg[k, :8]) / 100.0, 1).tolist(), 'start', round(float((seg[k, 0] - t0) / 100.0), 1))
