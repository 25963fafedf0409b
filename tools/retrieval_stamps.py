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
    # a tile that listed more than CAP - 6 pairs wrote list entries over its stamps: leave it out (counted below)
    ref = np.median(seg[:, 0])
    ok = (np.abs(seg[:, 0] - ref) < 1e8) & np.all((seg[:, 1:12] == 0) | ((seg[:, 1:12] >= seg[:, :1]) & (seg[:, 1:12] < seg[:, :1] + 1e6)), axis=1)
    n_lost = int((~ok).sum())
    seg = seg[ok]
    t0 = seg[:, 0].min()
    us = lambda a, b: (seg[:, a] - seg[:, b]) / 100.0     # 100 MHz -> us; slots: 0 start, 1 prefix GEMM done, 2 thresholds, 3 phase 1 / 1b, 8 phase 2a,
    med = lambda v: round(float(np.median(v)), 2) if len(v) else None      # 4 phase 2b + flush | 5 continuation starts, 6 done, 7 epilogue done (epilogue: 8 init, 9 rows, 10 columns, 11 barrier)
    done_exact = seg[:, 7] > 0                        # the tile continued its chains in place (or the launch was all-exact)
    analysed = seg[:, 2] > 0
    screened = analysed & ~done_exact                 # decided by the prefix: ends at slot 4
    counters = ws[so:so + 36].view(torch.int32).cpu().tolist()
    print('sigma', sigma, 'all-exact launch' if exact else 'screened launch', '| tiles: screened %d, analysed then exact %d, exact at once %d' %
          (int(screened.sum()), int((analysed & done_exact).sum()), int((~analysed & done_exact).sum())),
          '| counters [exact tiles, listed pairs, (diag) cheap group tests, full group evaluations, waves in phase 2, rescored, analysed, overflowed, skipped]:', counters)
    if screened.any():
        m = screened
        p2a = m & (seg[:, 8] > 0)
        print('   screened tiles, median us: prefix GEMM %s | loads + thresholds %s | phase 1 + 1b %s | phase 2a (incl. barrier) %s | phase 2b + flush %s | tile total %s' %
              (med(us(1, 0)[m]), med(us(2, 1)[m]), med(us(3, 2)[m]), med(us(8, 3)[p2a]), med(us(4, 8)[p2a]), med(us(4, 0)[m])))
    if done_exact.any():
        m = done_exact
        ma = m & analysed
        print('   exact tiles, median us: prefix GEMM %s | analysis before giving up %s | continuation GEMM %s | epilogue %s (init + barrier %s, rows %s, columns %s, barrier %s, global atomics %s) | tile total %s' %
              (med(us(1, 0)[m]), med(us(5, 1)[ma]), med(us(6, 5)[m]), med(us(7, 6)[m]), med(us(8, 6)[m]), med(us(9, 8)[m]), med(us(10, 9)[m]), med(us(11, 10)[m]), med(us(7, 11)[m]), med(us(7, 0)[m])))
    print('   kernel span %.1f us' % float((seg[:, :12].max() - t0) / 100.0) + ('   (%d tiles left out: their lists overwrote the stamps)' % n_lost if n_lost else ''))
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
