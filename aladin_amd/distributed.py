"""Caption-block sharding of the alignment triplet loss over the GPUs of one node (one process per
GPU, torch.distributed backend "nccl" == RCCL over xGMI).

The reference is single-device; parity target = the single-device result
on the concatenated global batch.  S[i][j] depends only on image i and caption j, so:

    rank r owns captions [r*B, (r+1)*B) and images [r*B, (r+1)*B)
    1. all-gather the image sets (+ lengths)                       -- the one exchange of the path
    2. S[:, block r] = HIP alignment scores (all images x own captions)
    3. all-gather the (W*B x B) score blocks -> full S on every rank (2 MB per rank at B=256)
    4. hinge on the full S (replicated, identical bits on every rank)
    5. backward: each rank differentiates its own column block; d(captions) is local.  d(image sets):
       dense dS (sum of violations) -> reduce-scatter of the (W*B, R, D) contributions; sparse dS
       (max_violation: <= 3 non-zeros per row/column) -> SparseImageExchange: only the image sets a
       caption block actually pairs with travel (fp32, all-to-all), and only their gradients return.

`scores_fn` / `hinge_fn` default to the HIP ops; CPU tests (gloo, world_size 2) inject the torch
restatement from oracle/ to exercise the collectives without a GPU.
"""
import torch
import torch.distributed as dist


class _HostStamp:
    def __init__(self):
        import time
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class PhaseRecorder:
    """Device-side timeline of one sharded step, for bench.py --gpus N (config.phases_ms): mark(name) drops an
    event on the current stream at a phase boundary; summary() gives the mean milliseconds between consecutive
    marks per phase name.  Collective phases measure what the compute stream WAITED for the exchange (their
    overlap with scoring is the point of the design), not the wire time."""

    def __init__(self):
        self.steps = []
        self._cur = None

    def begin(self):
        self._cur = [('start', self._event())]

    @staticmethod
    def _event():
        if not torch.cuda.is_available():          # the CPU tier (gloo, tests/helpers/cpu_standins.py): host time stamps
            return _HostStamp()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def mark(self, name):
        if self._cur is not None:
            self._cur.append((name, self._event()))

    def end(self):
        if self._cur is not None:
            self.steps.append(self._cur)
            self._cur = None

    def summary(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        tot, order = {}, []
        for st in self.steps:
            for (_, e0), (name, e1) in zip(st[:-1], st[1:]):
                if name not in tot:
                    tot[name] = 0.0
                    order.append(name)
                tot[name] += e0.elapsed_time(e1)
        n = max(len(self.steps), 1)
        return {k: round(tot[k] / n, 4) for k in order}


_RECORDER = [None]


def set_phase_recorder(rec):
    """Install (or with None remove) a PhaseRecorder that the fast sharded step marks its phases on."""
    _RECORDER[0] = rec


def _mark(name):
    if _RECORDER[0] is not None:
        _RECORDER[0].mark(name)


def _world(group):
    return dist.get_world_size(group), dist.get_rank(group)


def _reduce_scatter_rows(g, n, group):
    """Sum over ranks of the (W*n, ...) tensor g, this rank's n rows of it.  RCCL: one reduce-scatter; gloo (the CPU tier) has none:
    all-reduce + slice, the same sums."""
    W, r = _world(group)
    if dist.get_backend(group) == 'gloo':
        dist.all_reduce(g, group=group)
        return g[r * n:(r + 1) * n].clone()
    out = torch.empty((n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
    dist.reduce_scatter_tensor(out, g, group=group)
    return out


class _AllGatherRows(torch.autograd.Function):
    """cat over ranks along dim 0; backward = reduce-scatter(sum) of the incoming gradient."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        W, _ = _world(group)
        x = x.contiguous()
        out = torch.empty((W * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        W, r = _world(ctx.group)
        g = g.contiguous()
        n = g.shape[0] // W
        return _reduce_scatter_rows(g, n, ctx.group), None


class _GatherColumnBlocks(torch.autograd.Function):
    """(N x b) block per rank -> (N x W*b) on every rank.  The loss built on the result is
    REPLICATED (same value on every rank), so backward hands each rank its own block of the
    gradient -- no sum over ranks."""

    @staticmethod
    def forward(ctx, blk, group):
        ctx.group = group
        W, _ = _world(group)
        parts = torch.empty((W * blk.shape[0], blk.shape[1]), dtype=blk.dtype, device=blk.device)
        dist.all_gather_into_tensor(parts, blk.contiguous(), group=group)
        return parts.view(W, blk.shape[0], blk.shape[1]).permute(1, 0, 2).reshape(blk.shape[0], W * blk.shape[1])

    @staticmethod
    def backward(ctx, g):
        W, r = _world(ctx.group)
        b = g.shape[1] // W
        return g[:, r * b:(r + 1) * b].contiguous(), None


def gather_lengths(lens, device, group=None):
    W, _ = _world(group)
    t = torch.tensor([int(v) for v in lens], dtype=torch.int32, device=device)
    out = torch.empty(W * t.numel(), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def sharded_alignment_loss(im_set, s_seq, im_len, s_len, margin=0.2, max_violation=True, group=None,
                           scores_fn=None, hinge_fn=None):
    """(loss, S_full) of the alignment triplet loss on the GLOBAL batch (all ranks' samples), every
    rank passing its local (B,R,D) / (B,T,D) sets; all ranks must use the same B, R, T.
    Equals AlignmentContrastiveLoss(...)(cat(im), cat(s), ...) on one device."""
    if scores_fn is None or hinge_fn is None:
        from . import ops
        scores_fn = scores_fn or ops.alignment_scores
        hinge_fn = hinge_fn or ops.hinge_loss
    im_all = _AllGatherRows.apply(im_set, group)
    il_all = gather_lengths(im_len, im_set.device, group)
    S_blk = scores_fn(im_all, s_seq, il_all, s_len)                   # (W*B, B)
    S_full = _GatherColumnBlocks.apply(S_blk, group)
    return hinge_fn(S_full, margin, max_violation), S_full


# ------------------------------------------------------------------------------------------------
# Fast path (GPU, RCCL): packed-fp16 all-gather for the forward, raw fp32 all-gather overlapped with
# the score kernel for the backward, fused hinge on the replicated global matrix.
# The per-rank pieces are pure functions of already-gathered tensors so that one GPU can validate
# the rank-offset logic by emulating every rank (tests/test_gpu_parity.py).
# ------------------------------------------------------------------------------------------------
def _local_and_global_geometry(B, W, R, T, D):
    from . import ops
    g_loc = ops.align_geometry(B, B, R, T, D)
    g_glob = ops.align_geometry(W * B, B, R, T, D)          # all images x this rank's captions
    ok = (g_glob.xm_rows == W * g_loc.xm_rows and g_glob.xe_rows == W * g_loc.xe_rows and g_loc.Bi_pad == B)
    return g_loc, g_glob, ok


def rank_scores_block(xm_all, xe_all, s_local, s_len_t, g_glob):
    """(W*B x B) column block of the global score matrix from the gathered packed image operands."""
    from . import ops
    y = ops.pack_captions(s_local, s_len_t, g_glob)
    return ops.scores_from_packed(xm_all, xe_all, y, g_glob), y


def rank_scores_rows(xm_all, xe_all, y, S_blk, first, count, B, R, T, D):
    """Fill rows [first*B, (first+count)*B) of this rank's (W*B x B) score block: the images of `count`
    consecutive ranks against the local captions.  The packed operands of a rank range are contiguous
    slices of the gathered buffers (the fast path requires xe_rows == B per rank), and a score depends
    only on its own image / caption rows, so any split into ranges gives the bits of one launch."""
    from . import ops
    if count <= 0:
        return
    g = ops.align_geometry(count * B, B, R, T, D)
    per_m = (g.xm_bytes // 2) // count
    per_e = (g.xe_bytes // 2) // count
    xe = xe_all[first * per_e:(first + count) * per_e] if per_e else xe_all
    ops.scores_from_packed(xm_all[first * per_m:(first + count) * per_m], xe, y, g, out=S_blk[first * B:(first + count) * B])


def rank_backward_block(im_all, il_all_t, s_local, s_len_t, dS_full, rank, g_glob, xm_all, xe_all, y, gscale=None, rnorm=None):
    """This rank's contribution: d(all image sets) restricted to its caption block, and d(its captions).
    rnorm: the inverse norms of [xm_all rows | xe_all rows | y rows] (g_glob.rnorm_bytes) -- with them the row step follows
    ops.set_backward_precision (fp16 partner rows by default), without them it reads the raw fp32 sets."""
    from . import ops
    B = s_local.shape[0]
    dS_blk = dS_full[:, rank * B:(rank + 1) * B].contiguous()
    packed = (g_glob, xm_all, xe_all, y) if rnorm is None else (g_glob, xm_all, xe_all, y, rnorm)
    return ops._align_backward(im_all, s_local, il_all_t, s_len_t, dS_blk, gscale=gscale, packed=packed)


class _PinnedPool:
    """Pinned host buffers for asynchronous device->host copies, OWNED by whoever took them until handed back:
    `get` hands out a free buffer (allocating one when none is free -- a step in steady state allocates nothing),
    `put` returns it.  A buffer is reused only after its owner returned it AND the copy last issued into it has
    completed (event), so any number of plans may be outstanding at once -- e.g. several micro-batch forwards before
    one backward -- without one reading another's counts.  (Round 3 rotated 8 slots guarded by the copy event alone:
    the ninth forward before the first backward overwrote the first plan's counts.)"""

    def __init__(self, alloc=None):
        self.free = {}
        self.allocated = 0
        self._alloc = alloc or (lambda n: (torch.empty(n, dtype=torch.int64).pin_memory(), torch.cuda.Event()))

    def get(self, numel):
        lst = self.free.setdefault(int(numel), [])
        if lst:
            buf, ev = lst.pop()
            ev.synchronize()                     # an owner that died unresolved may have left its copy in flight
            return buf, ev
        self.allocated += 1
        return self._alloc(int(numel))

    def put(self, buf, ev):
        self.free.setdefault(int(buf.numel()), []).append((buf, ev))


_PINNED = _PinnedPool()


class SparseImageExchange:
    """Pair-driven exchange of image sets for the backward of a SPARSE dS (max_violation hinge).

    dS_full is replicated (the hinge runs on the gathered score matrix on every rank), so every rank
    derives the whole send/receive plan from it without talking: caption block b needs image i iff
    dS_full[i, block b] has a non-zero.  The split sizes of the variable-length all-to-all have to be known on
    the HOST: the (W x W) count matrix leaves the device by an asynchronous copy issued when the plan is built (in
    the forward) and is waited for only when the exchange is first used (fetch(), in the backward) -- the forward
    itself never blocks the host.  (A fixed-capacity, sync-free all-to-all cannot be both exact and smaller than
    the dense form: the only bound on what a caption block needs from one peer is that peer's whole batch.)

        fetch(im_local)        -> (n_need, R, D) fp32 sets of the images this rank's captions pair with,
                                  in ascending global index order (self.need_idx)
        give_back(d_im_need)   -> (B, R, D) gradient of the local image sets, summed over the ranks
                                  that used them in a fixed rank order (deterministic)

    At B=256, W=8 a rank needs ~600 of the 2048 image sets (diagonal + one row- and one column-maximum
    per sample), so ~45 MB travel each way instead of the 214 MB all-gather + 214 MB reduce-scatter of
    the dense form.  Pure torch + torch.distributed: runs under gloo on CPU for the tests."""

    def __init__(self, dS_full, B, group=None, pool=None):
        self.group = group
        W, r = _world(group)
        self.W, self.r, self.B = W, r, B
        nz = (dS_full.view(W * B, W, B) != 0).any(dim=2).t().contiguous()          # nz[b, i]
        counts = nz.view(W, W, B).sum(dim=2).reshape(-1)                           # counts[b * W + a]
        self._event = None
        self._pool = pool if pool is not None else (_PINNED if counts.is_cuda else None)       # `pool`: tests drive the ownership logic on CPU
        if self._pool is not None:
            self._counts, self._event = self._pool.get(W * W)          # ours until _resolve() (or __del__) hands it back
            self._counts.copy_(counts, non_blocking=True)
            self._event.record()
        else:
            self._counts = counts
        # ascending indices of the True entries without a sync: stable sort of the negated mask, cut at resolve time
        self._need_order = torch.argsort((~nz[r]).to(torch.uint8), stable=True)
        mine = nz[:, r * B:(r + 1) * B].reshape(-1)                                # dest-major, local image minor
        self._send_order = torch.argsort((~mine).to(torch.uint8), stable=True)
        self._resolved = False

    def _resolve(self):
        """The one host wait of the exchange (first use)."""
        if self._resolved:
            return
        if self._event is not None:
            self._event.synchronize()
        W, r, B = self.W, self.r, self.B
        counts = self._counts.view(W, W).tolist()                                  # counts[b][a]
        self._release()
        self.recv_splits = [int(counts[r][a]) for a in range(W)]
        self.send_splits = [int(counts[b][r]) for b in range(W)]
        self._need_idx = self._need_order[:sum(self.recv_splits)]
        self.send_rows = self._send_order[:sum(self.send_splits)] % B
        self._resolved = True

    def _release(self):
        if self._pool is not None and self._counts is not None:
            self._pool.put(self._counts, self._event)
        self._counts, self._pool = None, None

    def __del__(self):                               # a plan whose backward never ran: the buffer goes back (get() waits for its copy)
        try:
            self._release()
        except Exception:
            pass

    @property
    def need_idx(self):
        self._resolve()
        return self._need_idx

    def _all_to_all(self, send, out_splits, in_splits):
        if send.is_cuda and dist.get_backend(self.group) == 'gloo':
            # gloo has no device all-to-all: stage through the host.  (Only the one-GPU test of the real node across processes gets
            # here -- tests/helpers/gpu_shard_worker.py; the 8-GPU run is RCCL.)
            out = torch.empty((sum(out_splits),) + tuple(send.shape[1:]), dtype=send.dtype)
            dist.all_to_all_single(out, send.contiguous().cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
            return out.to(send.device)
        out = torch.empty((sum(out_splits),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(out, send.contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                               group=self.group)
        return out

    def fetch(self, im_local):
        self._resolve()
        return self._all_to_all(im_local.index_select(0, self.send_rows), self.recv_splits, self.send_splits)

    def give_back(self, d_im_need, shape):
        self._resolve()
        back = self._all_to_all(d_im_need, self.send_splits, self.recv_splits)
        d_im = torch.zeros(shape, dtype=d_im_need.dtype, device=d_im_need.device)
        off = 0
        for n in self.send_splits:                     # one source rank at a time: indices unique per call
            if n:
                d_im.index_add_(0, self.send_rows[off:off + n], back[off:off + n])
            off += n
        return d_im


class FlatSegments:
    """The one forward exchange of the fast path: every rank contributes ONE segment [xm | xe | image lengths] (256-byte aligned
    parts) to a single all-gather.  Pure tensor bookkeeping (device-agnostic: the gloo tests run it on CPU tensors)."""

    def __init__(self, xm_bytes, xe_bytes, B, rn_rows=0):
        """rn_rows > 0: the segment also carries the inverse norms of the rank's packed image rows ([xm rows | xe rows] fp32, what
        aladin_align_pack writes into `rnorm`), so that the backward's fp16 row step has them for every rank's images."""
        up = lambda v: (int(v) + 255) // 256 * 256
        self.xm_b, self.xe_b, self.B = int(xm_bytes), max(int(xe_bytes), 16), int(B)
        self.o_xe = up(self.xm_b)
        self.o_il = self.o_xe + up(self.xe_b)
        self.o_rn = self.o_il + up(4 * self.B)
        self.rn_b = 4 * int(rn_rows)
        self.seg = self.o_rn + up(self.rn_b)

    def alloc(self, device):
        """-> (flat, xm, xe, il): this rank's segment and typed views of its parts (fp16, fp16, int32)."""
        flat = torch.empty(self.seg, dtype=torch.uint8, device=device)
        return flat, flat[:self.xm_b].view(torch.float16), flat[self.o_xe:self.o_xe + self.xe_b].view(torch.float16), \
            flat[self.o_il:self.o_il + 4 * self.B].view(torch.int32)

    def rnorm_view(self, flat):
        """this rank's inverse-norm part (fp32) of its own segment"""
        return flat[self.o_rn:self.o_rn + self.rn_b].view(torch.float32)

    def rnorms(self, flat_all):
        """(W, rn_rows) fp32: every rank's inverse norms, in place"""
        W = flat_all.numel() // self.seg
        return flat_all.view(W, self.seg)[:, self.o_rn:self.o_rn + self.rn_b].view(torch.float32)

    def gather(self, flat, group, async_op=True):
        W, _ = _world(group)
        flat_all = torch.empty(W * self.seg, dtype=torch.uint8, device=flat.device)
        return flat_all, dist.all_gather_into_tensor(flat_all, flat, group=group, async_op=async_op)

    def rank_views(self, flat_all, w):
        """(xm, xe) of rank w, in place."""
        base = w * self.seg
        return flat_all[base:base + self.xm_b].view(torch.float16), flat_all[base + self.o_xe:base + self.o_xe + self.xe_b].view(torch.float16)

    def lengths(self, flat_all):
        W = flat_all.numel() // self.seg
        return flat_all.view(W, self.seg)[:, self.o_il:self.o_il + 4 * self.B].contiguous().view(torch.int32).reshape(-1)

    def contiguous_operands(self, flat_all):
        """(xm_all, xe_all): the ranks' parts laid out as ONE operand each (device copies; the dense backward's pair kernel wants that)."""
        W = flat_all.numel() // self.seg
        segs = flat_all.view(W, self.seg)
        return segs[:, :self.xm_b].contiguous().view(-1).view(torch.float16), \
            segs[:, self.o_xe:self.o_xe + self.xe_b].contiguous().view(-1).view(torch.float16)


def place_image_rnorms(rn_glob, rn_ranks, n_m, n_e):
    """The gathered ranks' inverse norms -> the global layout the backward reads.  rn_ranks: (W, n_m + n_e) fp32, rank w's packed image
    rows [its xm rows | its xe rows] (what aladin_align_pack wrote for the LOCAL geometry); rn_glob: the GLOBAL geometry's buffer
    [all ranks' xm rows | all ranks' xe rows | y rows] whose y part the caption packer fills.  A pure function of its arguments so that
    one GPU can check it against the single-device layout (tests/test_gpu_parity.py: the emulated ranks)."""
    W = rn_ranks.shape[0]
    rn_glob[:W * n_m].view(W, n_m).copy_(rn_ranks[:, :n_m])
    if n_e:
        rn_glob[W * n_m:W * (n_m + n_e)].view(W, n_e).copy_(rn_ranks[:, n_m:n_m + n_e])
    return rn_glob


class _ShardedTriplet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, s, im_len_t, s_len_t, margin, max_violation, group, exchange='auto'):
        from . import ops
        W, r = _world(group)
        B, R, D = im.shape
        T = s.shape[1]
        g_loc, g_glob, ok = _local_and_global_geometry(B, W, R, T, D)
        if not ok:
            # the gathered packed operands must be the plain concatenation of the per-rank ones: the per-rank batch
            # has to fill whole image tiles (B % img_unit == 0) and whole 64-row side-operand groups
            raise ValueError('aladin_amd.distributed: sharded_alignment_loss_fast needs a per-rank batch whose packed '
                             'image operand concatenates across ranks (B a multiple of 64 covers every shape; got '
                             'B=%d, tile unit %d); use sharded_alignment_loss(), the composable path, instead'
                             % (B, g_loc.img_unit))
        im_c = im.contiguous()
        # ONE exchange for everything the forward needs from the other ranks (round 5; rounds 2-4 issued three gathers): each rank's
        # segment is [xm | xe | image lengths], packed straight into it; the remote images are then scored rank by rank from their
        # segments (a score depends only on its own image / caption rows, so any split gives the bits of one launch).
        need = any(ctx.needs_input_grad[:2])
        # <= 3 non-zeros of dS per row/column under max_violation: pair-driven exchange
        # (its fixed cost -- one host sync for the split sizes -- pays off once the dense form would move
        # >= 4 ranks' worth of fp32 sets; bench.py times both)
        sparse = (bool(max_violation) and W >= 4) if exchange == 'auto' else (exchange == 'sparse')
        # dense exchange: the inverse norms of the packed rows travel with them when the backward's row step will take its partner rows
        # from the packed fp16 operands (ops.set_backward_precision: the library default) -- 4 bytes per 1.5 KB row.  (The pair-driven
        # exchange packs its compact problem again in the backward and gets them there.)
        fp16_rows = need and not sparse and ops._BWD_PARTNERS[0] == 'fp16'
        n_m, n_e = int(g_loc.xm_rows), int(g_loc.xe_rows)
        fs = FlatSegments(g_loc.xm_bytes, g_loc.xe_bytes, B, rn_rows=(n_m + n_e) if fp16_rows and W > 1 else 0)
        flat, xm, xe, il_seg = fs.alloc(im.device)
        il_seg.copy_(im_len_t)
        rn_glob = None
        if fp16_rows:                          # [all ranks' xm rows | all ranks' xe rows | y rows]: the captions' part now, the images' after the gather
            rn_glob = torch.empty(int(g_glob.rnorm_bytes) // 4, dtype=torch.float32, device=im.device)
        # one rank: the local rows ARE the global ones -- their inverse norms go straight into place (no copies after the gather)
        rn_img = None if not fp16_rows else (rn_glob if W == 1 else fs.rnorm_view(flat))
        ops.pack_images(im_c, im_len_t, g_loc, rnorm=rn_img, out=(xm, xe))
        flat_all, gather = fs.gather(flat, group)
        im_all, work = None, None
        if need and not sparse:                    # raw fp32 sets: only the exact backward reads them (the second and last gather)
            im_all = torch.empty((W * B, R, D), dtype=im.dtype, device=im.device)
            work = dist.all_gather_into_tensor(im_all, im_c, group=group, async_op=True)
        # The local images' block does not need the exchange: score it while the gather is in flight, then the other ranks' images.
        y = ops.pack_captions(s, s_len_t, g_glob, rnorm=rn_glob)
        _mark('pack+issue_gathers')
        S_blk = torch.empty((W * B, B), dtype=torch.float32, device=im.device)
        e_scr = torch.empty(max(int(g_loc.e_bytes), 16), dtype=torch.uint8, device=im.device)
        ops.scores_from_packed(xm, xe, y, g_loc, out=S_blk[r * B:(r + 1) * B], e_scratch=e_scr)
        _mark('local_block')
        gather.wait()
        _mark('gather_wait')
        for k in range(1, W):
            w_ = (r + k) % W
            xm_w, xe_w = fs.rank_views(flat_all, w_)
            ops.scores_from_packed(xm_w, xe_w, y, g_loc, out=S_blk[w_ * B:(w_ + 1) * B], e_scratch=e_scr)
        il_all = fs.lengths(flat_all)
        _mark('remote_rows')
        parts = torch.empty((W * S_blk.shape[0], B), dtype=S_blk.dtype, device=im.device)
        dist.all_gather_into_tensor(parts, S_blk, group=group)
        if work is not None:
            # The raw-set gather was issued BEFORE the score-block gather on the same communicator, so it has
            # completed by now: joining it here costs nothing and leaves no collective writing into im_all behind
            # this call if backward() is never run.
            work.wait()
        _mark('S_allgather')
        S_full = parts.view(W, W * B, B).permute(1, 0, 2).reshape(W * B, W * B)
        loss, dS_full, _ = ops._hinge_raw(S_full, margin, max_violation, need)
        _mark('hinge')
        ctx.exchange = None
        if need and sparse:
            # the plan only: its split sizes travel to the host asynchronously and are first needed by the fetch, which
            # runs in the backward -- the forward never blocks the host
            ex = SparseImageExchange(dS_full, B, group)
            _mark('sparse_plan')
            ctx.save_for_backward(im_c, il_all, s, s_len_t, dS_full)
            ctx.exchange, ctx.im_shape, ctx.rank = ex, tuple(im.shape), r
        elif need:
            # the dense backward's pair kernel reads all ranks' packed images as ONE operand: lay the segments' parts out contiguously
            # (device copies, no collective)
            xm_all, xe_all = fs.contiguous_operands(flat_all) if W > 1 else (xm, xe)
            if rn_glob is not None:
                if W > 1:
                    place_image_rnorms(rn_glob, fs.rnorms(flat_all), n_m, n_e)
                ctx.save_for_backward(im_all, il_all, s, s_len_t, dS_full, xm_all, xe_all, y, rn_glob)
            else:
                ctx.save_for_backward(im_all, il_all, s, s_len_t, dS_full, xm_all, xe_all, y)
            ctx.g_glob, ctx.group = g_glob, group
        ctx.mark_non_differentiable(S_full)
        ctx.set_materialize_grads(False)
        return loss, S_full

    @staticmethod
    def backward(ctx, g_loss, _g_scores):
        if g_loss is None:
            return (None,) * 8
        if ctx.exchange is not None:
            from . import ops
            im_c, il_all, s, s_len_t, dS_full = ctx.saved_tensors
            gscale = g_loss.to(torch.float32).contiguous()
            _mark('bwd_start')
            ex, r, B = ctx.exchange, ctx.rank, s.shape[0]
            im_need = ex.fetch(im_c)                              # waits for the plan's split sizes (the one host sync)
            il_need = il_all.index_select(0, ex.need_idx)
            dS_need = dS_full.index_select(0, ex.need_idx)[:, r * B:(r + 1) * B].contiguous()
            _mark('bwd_sparse_fetch')
            if im_need.shape[0]:
                # the compact problem (the images this caption block pairs with x its captions) packed again -- one launch over ~n_need x R
                # + B x T rows -- so that its arg-maxima come from the fp16 MFMA pair kernel (re-decided exactly in fp32, like the single-GPU
                # step's) and its row step can take the partner rows from the packed operands; without `packed` both run in fp32 throughout
                packed = None
                g_need = ops.align_geometry(im_need.shape[0], s.shape[0], im_need.shape[1], s.shape[1], s.shape[2])
                if ops._pair_kernel_covers(g_need):
                    packed = ops.pack_sets(im_need, s, il_need, s_len_t, g_need)
                d_im_need, d_s = ops._align_backward(im_need, s, il_need, s_len_t, dS_need, gscale=gscale, packed=packed)
            else:                                  # no violation anywhere in this caption block
                d_im_need, d_s = torch.zeros_like(im_need), torch.zeros_like(s)
            _mark('bwd_compute_compact')
            d_im = ctx.exchange.give_back(d_im_need, ctx.im_shape)
            _mark('bwd_give_back')
            return d_im, d_s, None, None, None, None, None, None
        im_all, il_all, s, s_len_t, dS_full, xm_all, xe_all, y = ctx.saved_tensors[:8]
        rn_glob = ctx.saved_tensors[8] if len(ctx.saved_tensors) > 8 else None
        W, r = _world(ctx.group)
        _mark('bwd_start')
        d_im_all, d_s = rank_backward_block(im_all, il_all, s, s_len_t, dS_full, r, ctx.g_glob, xm_all, xe_all, y,
                                            gscale=g_loss.to(torch.float32).contiguous(), rnorm=rn_glob)
        _mark('bwd_compute_dense')
        d_im = _reduce_scatter_rows(d_im_all.contiguous(), s.shape[0], ctx.group)
        _mark('bwd_reduce_scatter')
        return d_im, d_s, None, None, None, None, None, None


def sharded_alignment_loss_fast(im_set, s_seq, im_len, s_len, margin=0.2, max_violation=True, group=None,
                                exchange='auto'):
    """Same result as sharded_alignment_loss on GPUs over RCCL, with the forward exchange in packed
    fp16 (13 MB per rank at B=256 instead of 27 MB), one fused autograd node, and for the backward
    either the pair-driven SparseImageExchange (max_violation) or the raw all-gather overlapped with
    scoring + reduce-scatter (dense dS); ``exchange`` = 'auto' | 'sparse' | 'dense' overrides the choice.
    Returns (loss, S_full.detach()).

    Gradient convention: the loss is the GLOBAL-batch loss, replicated on every rank, and each rank receives
    d(loss)/d(its own inputs) in full -- the sum over ranks of the per-rank parameter gradients is the
    single-device gradient.  A data-parallel wrapper that AVERAGES gradients (DistributedDataParallel's default)
    therefore ends up with 1/W of it: scale the loss by the world size, or reduce with a sum."""
    if exchange not in ('auto', 'sparse', 'dense'):
        raise ValueError("aladin_amd.distributed: exchange must be 'auto', 'sparse' or 'dense'")
    from . import ops
    im_len_t, s_len_t = ops._check_sets(im_set, s_seq, im_len, s_len)
    return _ShardedTriplet.apply(im_set, s_seq, im_len_t, s_len_t, margin, max_violation, group, exchange)


# ------------------------------------------------------------------------------------------------
# All loss heads of a training step on the GLOBAL batch (reference alad/alad_model.py:371-454 on the
# concatenation of every rank's samples): the shipped distillation configs across the GPUs of a node.
# ------------------------------------------------------------------------------------------------
def sharded_loss_heads(img_emb, cap_emb, im_set, s_seq, im_len, s_len, margin, max_violation, heads, weights,
                       group=None, exchange='auto', temperature=6.0, eps=1e-10,
                       align_fn=None, dot_fn=None, hinge_fn=None, listnet_fn=None):
    """Matching hinge, alignment hinge and ListNet distillation of the global batch, every rank passing its local
    (B, D) global embeddings and (B, R, D) / (B, T, D) sets -> (total, terms, S_full, M_full):

        total   sum_k w_k L_k over `heads` with a non-zero weight (alad_model.py:450-453), replicated on every rank
        terms   dict head -> loss value (detached; a head in `heads` with weight 0 is computed for logging only,
                like the distillation term before distill_epoch, :442-444)
        S_full  (W*B, W*B) alignment scores (the detached teacher, alad/loss.py:370) or None
        M_full  (W*B, W*B) matching scores or None

    Sharding = caption blocks, as for the alignment triplet (module docstring): rank r owns column block r of BOTH
    score matrices.  The image-side operands are all-gathered (packed fp16 sets for S, the (B, D) fp32 global
    embeddings for M), each rank computes its (W*B x B) blocks, the blocks are all-gathered (2 x 2 MB per rank at
    B=256, W=8) and the heads run on the replicated matrices -- the row-wise softmaxes / maxima of alad/loss.py:
    431-445 and :60-67 need every caption block.  Backward: a rank keeps its own column block of dL/dM and dL/dS;
    d(cap_emb), d(s_seq) are local, d(img_emb) is a reduce-scatter of (W*B, D), d(im_set) the alignment exchange.
    Same gradient convention as sharded_alignment_loss_fast (the loss is the global loss on every rank).
    The *_fn hooks default to the HIP ops; the CPU tests inject the oracle."""
    heads = list(heads)
    unknown = set(heads) - {'matching', 'alignment', 'distillation'}
    if unknown or not heads:
        raise ValueError('aladin_amd.distributed: heads must be a non-empty subset of matching / alignment / distillation')
    from . import ops
    dot_fn = dot_fn or ops.dot_scores
    hinge_fn = hinge_fn or ops.hinge_loss
    listnet_fn = listnet_fn or (lambda t, m: ops.listnet_loss(t, m, temperature, eps))
    w = {k: float(weights.get(k, 0.0)) for k in heads}
    terms, S_full, M_full = {}, None, None
    total = None

    def add(k, value):
        nonlocal total
        terms[k] = value.detach()
        if w[k] != 0.0:
            total = value * w[k] if total is None else total + value * w[k]

    if 'alignment' in heads or 'distillation' in heads:
        if align_fn is None:
            def align_fn(a, b, al, bl):
                return sharded_alignment_loss_fast(a, b, al, bl, margin, max_violation, group=group, exchange=exchange)
        live = 'alignment' in heads and w['alignment'] != 0.0
        if live:
            a_loss, S_full = align_fn(im_set, s_seq, im_len, s_len)
        else:                                        # teacher only (or logged only): no backward through the sets
            with torch.no_grad():
                a_loss, S_full = align_fn(im_set.detach(), s_seq.detach(), im_len, s_len)
        S_full = S_full.detach()
    if 'matching' in heads or 'distillation' in heads:
        img_all = _AllGatherRows.apply(img_emb, group)                      # (W*B, D)
        M_full = _GatherColumnBlocks.apply(dot_fn(img_all, cap_emb), group)  # (W*B, W*B), differentiable
    # the reference's key order: matching, alignment, distillation (alad_model.py:380-408)
    if 'matching' in heads:
        add('matching', hinge_fn(M_full, margin, max_violation))
    if 'alignment' in heads:
        add('alignment', a_loss)
    if 'distillation' in heads:
        add('distillation', listnet_fn(S_full, M_full))
    if total is None:
        raise ValueError('aladin_amd.distributed: every selected head has weight 0')
    return total, terms, S_full, (M_full.detach() if M_full is not None else None)
