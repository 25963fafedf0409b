"""Device-resident, length-packed fp16 embedding store for evaluation.

The reference's encode_data (alad/evaluation.py:80-155) copies every batch to the host into
(N, 71, D) fp32 buffers and i2t / t2i copy slices back per query.  A PackedSetStore keeps what the
two retrieval heads read and nothing else:

    rows   fp16 (total_rows, Dp)  the positions [1, len - tail) of every set, L2-normalised exactly
                                  as the alignment pack kernels do, contiguous by TRUE length
    glob   fp32 (N, D)            the slot-0 global embedding (matching head, alad/evaluation.py:127-128)
    lengths                       the raw length list the reference returns

COCO-5k captions (25 000 x ~12 scored words x 768) take 0.46 GB instead of 5.4 GB; alignment scores
computed from a store are bit-identical to those computed from the fp32 sets (same fp16 operands).
"""
import ctypes as C

import torch

from . import _lib, ops


class PackedSetStore:
    def __init__(self, feat_dim, tail, device, capacity_rows=4096, precision='split', padded_len=71):
        """tail = trailing positions the alignment head drops: 0 for image sets, 2 for captions
        (reference alad/loss.py:87-90).  precision: 'split' keeps every unit vector as an fp16 hi/lo pair
        (rows twice as wide) so that alignment scores are rank-exact (ops.set_eval_precision); 'fp16' keeps
        the single-rounding training operand.  padded_len: the length the reference's encode_data pads every
        set to (max_len = 71, alad/evaluation.py:98-99): a sample shorter than that competes with the zero
        fill in the max over regions (alad/loss.py:116,124), one that fills it does not."""
        lib = _lib.load()
        self.D = int(feat_dim)
        self.precision = precision
        self.padded_len = int(padded_len)
        self.Dp = int(lib.aladin_store_row_width(self.D, ops._precision_code(precision)))
        if self.Dp < self.D:
            raise ValueError('aladin_amd: bad feature size %r' % (feat_dim,))
        self.tail = int(tail)
        self.device = torch.device(device)
        self.rows = torch.empty((max(int(capacity_rows), 1), self.Dp), dtype=torch.float16, device=self.device)
        self.n_rows = 0
        self.lengths = []                     # raw lengths, as the reference's encode_data returns them
        self._counts = []                     # usable positions per sample (host copy)
        self._glob, self._offsets_t, self._counts_t = [], None, None

    # ------------------------------------------------------------------------------------------ filling
    def _reserve(self, extra):
        need = self.n_rows + extra
        if need > self.rows.shape[0]:
            grown = torch.empty((max(need, 2 * self.rows.shape[0]), self.Dp), dtype=torch.float16, device=self.device)
            grown[:self.n_rows] = self.rows[:self.n_rows]
            self.rows = grown

    def append(self, sets, lengths, glob=None):
        """sets: (B, L, D) fp32 on the device (any strides with a unit inner stride); lengths: B ints;
        glob: (B, D) global embeddings for the matching head (defaults to slot 0 of the sets)."""
        ops._require_gpu(sets)
        B, L, D = sets.shape
        if D != self.D or len(lengths) != B:
            raise ValueError('aladin_amd: store.append got a (%d,%d,%d) batch with %d lengths (store D=%d)'
                             % (B, L, D, len(lengths), self.D))
        lengths = [int(v) for v in lengths]
        if L > self.padded_len:
            raise ValueError('aladin_amd: store.append got sets of %d positions, more than padded_len=%d' % (L, self.padded_len))
        counts = [min(max(v - 1 - self.tail, 0), L - 1) for v in lengths]
        offs, run = [], self.n_rows
        for c in counts:
            offs.append(run)
            run += c
        self._reserve(run - self.n_rows)
        sets = ops._rows_inner_contig(sets)
        lens_t = torch.tensor(lengths, dtype=torch.int32, device=self.device)
        offs_t = torch.tensor(offs, dtype=torch.int64, device=self.device)
        if L >= 2:
            _lib.check(_lib.load().aladin_store_append(ops._ptr(sets), sets.stride(0), sets.stride(1), ops._ptr(lens_t), B, L,
                                                            D, self.tail, ops._ptr(offs_t), ops._ptr(self.rows),
                                                            ops._precision_code(self.precision), ops._stream()),
                       'store_append')
        self._glob.append((sets[:, 0, :] if glob is None else glob).to(torch.float32).clone())
        self.n_rows = run
        self.lengths.extend(lengths)
        self._counts.extend(counts)
        self._offsets_t = self._counts_t = None

    # ------------------------------------------------------------------------------------------ reading
    def __len__(self):
        return len(self.lengths)

    @property
    def glob(self):
        if len(self._glob) != 1:
            self._glob = [torch.cat(self._glob)] if self._glob else [torch.empty((0, self.D), device=self.device)]
        return self._glob[0]

    def nbytes(self):
        return self.n_rows * self.Dp * 2 + len(self) * (self.D * 4 + 12)      # Dp already counts hi and lo for split stores

    def _tables(self):
        if self._offsets_t is None:
            offs, run = [], 0
            for c in self._counts:
                offs.append(run)
                run += c
            self._offsets_t = torch.tensor(offs, dtype=torch.int64, device=self.device)
            self._counts_t = torch.tensor(self._counts, dtype=torch.int32, device=self.device)
        return self._offsets_t, self._counts_t

    def view(self, index):
        """A selection (slice or index list) sharing this store's rows, e.g. store.view(slice(0, None, 5))
        for the de-duplicated images of alad/evaluation.py:171."""
        ids = list(range(len(self)))[index] if isinstance(index, slice) else [int(v) for v in index]
        return StoreView(self, ids)

    def max_count(self, ids=None):
        cs = self._counts if ids is None else [self._counts[k] for k in ids]
        return max(cs) if cs else 0


class StoreView:
    def __init__(self, store, ids):
        self.store, self.ids = store, ids
        self._ids_t = None

    def __len__(self):
        return len(self.ids)

    @property
    def lengths(self):
        return [self.store.lengths[k] for k in self.ids]

    @property
    def glob(self):
        return self.store.glob.index_select(0, self.ids_t.to(torch.int64))

    @property
    def ids_t(self):
        if self._ids_t is None:
            self._ids_t = torch.tensor(self.ids, dtype=torch.int32, device=self.store.device)
        return self._ids_t


def _unwrap(x):
    return (x.store, x.ids, x.ids_t) if isinstance(x, StoreView) else (x, None, None)


def alignment_scores_from_stores(img, cap):
    """(N_img, N_cap) 'MrSw' scores (reference alad/loss.py:80-125) between two stores / views: operands are row
    copies of the stores (no fp32 read, no normalisation).  Large ragged grids are scored in length classes
    (ops.bucket_plan: a pair pays for the tile class of its own image and caption); each class block is
    _store_scores_block."""
    si, ids_i, _ = _unwrap(img)
    sc, ids_c, _ = _unwrap(cap)
    if len(img) < 1 or len(cap) < 1:
        raise ValueError('aladin_amd: empty store')
    if len(img) * len(cap) < ops.BUCKET_MIN_PAIRS:
        return _store_scores_block(img, cap)
    ids_i = list(range(len(si))) if ids_i is None else list(ids_i)
    ids_c = list(range(len(sc))) if ids_c is None else list(ids_c)
    cap_x = max(si.padded_len - 1 - si.tail, 1)

    def need_x():
        return [min(max(si._counts[k], 1) + 1, cap_x) if si._counts[k] < cap_x else cap_x for k in ids_i]

    def need_y():
        return [max(sc._counts[k], 1) for k in ids_c]
    # keyed on the stores' identity and fill state: an append invalidates the plan
    plan = ops.grid_plan(('store', id(si), si.n_rows, len(si), id(sc), sc.n_rows, len(sc), tuple(ids_i), tuple(ids_c)),
                         need_x, need_y, si.device)
    if plan is None:
        return _store_scores_block(img, cap)
    # class views: python id lists for the host-side geometry, device id tensors by an index_select of the callers' ids
    # (no blocking copy between the blocks); nothing that references the stores is kept in the cached plan
    base_i = img.ids_t if isinstance(img, StoreView) else torch.arange(len(si), dtype=torch.int32, device=si.device)
    base_c = cap.ids_t if isinstance(cap, StoreView) else torch.arange(len(sc), dtype=torch.int32, device=sc.device)
    vx, vy = [], []
    for g, ig in zip(plan.gx, plan.ix):
        v = StoreView(si, [ids_i[k] for k in g])
        v._ids_t = base_i.index_select(0, ig)
        vx.append(v)
    for g, ig in zip(plan.gy, plan.iy):
        v = StoreView(sc, [ids_c[k] for k in g])
        v._ids_t = base_c.index_select(0, ig)
        vy.append(v)
    blocks = {(a, b): _store_scores_block(va, vb) for a, va in enumerate(vx) for b, vb in enumerate(vy)}
    return plan.assemble(blocks)


def _store_scores_block(img, cap):
    """One geometry for the whole block.  One score launch, or one per caption chunk when the side-row scratch
    (N_img x 16*tp16*N_cap floats when R' = 33) would pass E_SCRATCH_LIMIT -- 16 GB for a 5000 x 25000 grid otherwise;
    a score does not depend on the chunking."""
    si, ids_i, idt_i = _unwrap(img)
    sc, ids_c, idt_c = _unwrap(cap)
    if si.D != sc.D:
        raise ValueError('aladin_amd: feature sizes differ (%d vs %d)' % (si.D, sc.D))
    Bi, Bc = len(img), len(cap)
    if Bi < 1 or Bc < 1:
        raise ValueError('aladin_amd: empty store')
    if si.precision != sc.precision:
        raise ValueError('aladin_amd: the two stores hold different precisions (%s vs %s)' % (si.precision, sc.precision))
    # Max side: every sample shorter than the padded set (encode_data's 71 positions) takes the zero fill into
    # its max over regions (alad/loss.py:116,124), so the geometry keeps ONE position past the longest count
    # (a zero row in the operand) unless a sample fills the padded set; the sum side needs none.
    Rq = min(max(si.max_count(ids_i), 1) + 1, si.padded_len - 1 - si.tail)
    Tq = max(sc.max_count(ids_c), 1)
    precision = si.precision
    lib = _lib.load()
    dev = si.device
    oi, ci = si._tables()
    oc, cc = sc._tables()
    geom = ops.align_geometry(Bi, Bc, Rq + 1 + si.tail, Tq + 1 + sc.tail, si.D, si.tail, sc.tail, precision)
    chunk = Bc
    if geom.e_bytes > ops.E_SCRATCH_LIMIT:
        chunk = max(geom.cap_unit, int(Bc * ops.E_SCRATCH_LIMIT // geom.e_bytes) // geom.cap_unit * geom.cap_unit)
    xm = torch.empty(geom.xm_bytes // 2, dtype=torch.float16, device=dev)
    xe = torch.empty(max(geom.xe_bytes // 2, 8), dtype=torch.float16, device=dev)
    _lib.check(lib.aladin_align_pack_store_x(ops._ptr(si.rows), ops._ptr(oi), ops._ptr(ci), ops._ptr(idt_i), C.byref(geom),
                                             ops._ptr(xm), ops._ptr(xe), ops._stream()), 'align_pack_store_x')
    if chunk >= Bc:
        y = torch.empty(geom.y_bytes // 2, dtype=torch.float16, device=dev)
        _lib.check(lib.aladin_align_pack_store_y(ops._ptr(sc.rows), ops._ptr(oc), ops._ptr(cc), ops._ptr(idt_c), C.byref(geom),
                                                 ops._ptr(y), ops._stream()), 'align_pack_store_y')
        return ops.scores_from_packed(xm, xe, y, geom)
    S = torch.empty((Bi, Bc), dtype=torch.float32, device=dev)
    all_ids = idt_c if idt_c is not None else torch.arange(Bc, dtype=torch.int32, device=dev)
    for j0 in range(0, Bc, chunk):
        j1 = min(Bc, j0 + chunk)
        g = ops.align_geometry(Bi, j1 - j0, Rq + 1 + si.tail, Tq + 1 + sc.tail, si.D, si.tail, sc.tail, precision)   # same x layout
        y = torch.empty(g.y_bytes // 2, dtype=torch.float16, device=dev)
        ids = all_ids[j0:j1].contiguous()
        _lib.check(lib.aladin_align_pack_store_y(ops._ptr(sc.rows), ops._ptr(oc), ops._ptr(cc), ops._ptr(ids), C.byref(g),
                                                 ops._ptr(y), ops._stream()), 'align_pack_store_y')
        ops.scores_from_packed(xm, xe, y, g, out=S[:, j0:j1])
    return S
