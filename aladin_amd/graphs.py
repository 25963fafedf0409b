"""HIP-graph replay of the loss-head step for the shipped batch size (every YAML: bs 32).

At bs = 32 the loss heads -- matching scores + hinge, alignment scores + hinge, listnet, and their backward --
are ~20 short launches whose eager issue time (Python + ctypes, ~15 us each) is several times the GPU time
(0.33 ms per step against ~0.1 ms of kernels, profiles/).  The C ABI neither allocates nor synchronises, so the
whole step -- ALADModel.forward_loss, the weighted sum of alad_model.py:442-453 and its backward down to the
four encoder outputs -- is captured ONCE per batch shape into a HIP graph and replayed:

    step = GraphedLossStep(model)
    loss, loss_dict = step(img_emb, cap_emb, img_set, cap_seq, img_len, cap_len, epoch=epoch)
    loss.backward()            # continues into the encoder: the graph already holds d loss / d (its inputs)

`loss` is connected to the four embedding tensors through a custom autograd node whose backward hands out the
gradients the replay produced (scaled by the incoming gradient).  Shapes are part of the capture: the encoder
slices its sets to the batch maxima (alad_model.py:174-175) and a set padded beyond the batch maximum would
change the max over regions (the zero fill competes, alad/loss.py:116,124), so graphs are cached per
(B, R, T, D, distillation active) -- a few dozen small graphs over a COCO epoch (LRU, `cache_size`).
Host cost per step is what is left around the replay, so it is kept minimal: the four inputs are staged by ONE
multi-tensor copy, the lengths by ONE host->device copy from a rotating pinned buffer (skipped when they equal the
previous step's), the logged terms leave by ONE asynchronous device->host copy that `model.logger` receives when the
next step is issued (log='deferred'; log='sync' keeps the reference's immediate `.item()` protocol) -- the host never
waits for the device, so issuing step n + 1 overlaps the replay of step n.
"""
from collections import OrderedDict, deque

import numpy as np
import torch


class _Entry:
    __slots__ = ('graph', 'inputs', 'lens', 'lens_both', 'lens_key', 'lens_slots', 'lens_k', 'loss', 'terms', 'grads', 'logged',
                 'log_buf', 'log_names', 'generation')


class _PinnedSlot:
    """One pinned staging buffer + the event of the last asynchronous copy that used it."""
    __slots__ = ('host', 'view', 'event', 'used')

    def __init__(self, numel, dtype):
        self.host = torch.empty(numel, dtype=dtype).pin_memory()
        self.view = self.host.numpy()                       # same memory: filled without a tensor construction
        self.event = torch.cuda.Event()
        self.used = False

    def acquire(self):
        if self.used:
            self.event.synchronize()                        # the copy last issued from / into this buffer has EXECUTED
        self.used = True
        return self


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, entry, *embs):
        # also lays permuted (S,B,D)<->(B,S,D) views out as captured; one multi-tensor launch when the layouts agree
        if all(d.stride() == s.stride() and d.dtype == s.dtype for d, s in zip(entry.inputs, embs)):
            torch._foreach_copy_(entry.inputs, [s.detach() for s in embs])
        else:
            for dst, src in zip(entry.inputs, embs):
                dst.copy_(src)
        entry.graph.replay()
        entry.generation += 1                                # the static gradient buffers now belong to THIS replay
        ctx.entry, ctx.generation = entry, entry.generation
        return entry.loss.clone()

    @staticmethod
    def backward(ctx, g):
        # the replay already differentiated the captured loss w.r.t. its static inputs; they live in the graph's static
        # buffers, which the next replay of the same batch shape overwrites
        if ctx.generation != ctx.entry.generation:
            raise RuntimeError('aladin_amd.graphs: backward() of a GraphedLossStep result after a later step of the same batch '
                               'shape was run -- the captured graph keeps ONE set of gradient buffers per shape; call '
                               'backward() before the next step (or use ALADModel.forward for overlapping steps)')
        live = [t for t in ctx.entry.grads if t is not None]               # an input no active term depends on has none
        scaled = iter(torch._foreach_mul(live, g.to(torch.float32)) if live else [])
        return (None,) + tuple(next(scaled) if t is not None else None for t in ctx.entry.grads)


class GraphedLossStep:
    """log: 'deferred' (default) -- the logged loss terms leave the device by ONE asynchronous copy per step and reach
    `model.logger` when the next step is issued (or at flush()): the host never waits for the GPU, so issuing step n + 1
    overlaps the replay of step n; 'sync' -- the reference's protocol (`.item()` right after the forward,
    alad_model.py:383,390,408): the logger is current when the call returns, at the price of one host wait per step."""

    def __init__(self, model, cache_size=32, log='deferred'):
        if log not in ('deferred', 'sync'):
            raise ValueError("aladin_amd.graphs: log must be 'deferred' or 'sync'")
        self.model = model
        self.cache_size = int(cache_size)
        self.log = log
        self._cache = OrderedDict()
        self._pending = deque()                              # (slot, [(key, index, n)], logger)
        self._log_slots = [None] * 4
        self._log_k = 0

    def _run(self, e, epoch_before_distill):
        m = self.model
        for t in e.inputs:
            t.grad = None
        m._graph_bypass = True                               # ALADModel(graphed=True) routes its training steps HERE: not recursively
        try:
            loss, losses = m.forward_loss_total(e.inputs[0], e.inputs[1], e.inputs[2], e.inputs[3], e.lens[0], e.lens[1], 0,
                                                0 if epoch_before_distill else 1, 1, log=False)
        finally:
            m._graph_bypass = False
        e.logged = m.pending_log
        loss.backward()
        return loss, losses

    def _capture(self, key, embs):
        B, R, T, D, before, dev = key
        e = _Entry()
        e.generation = 0
        e.lens_key = None
        e.lens_slots = [_PinnedSlot(2 * B, torch.int32) for _ in range(3)]
        e.lens_k = 0
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            e.inputs = [torch.zeros_like(x, memory_format=torch.preserve_format).requires_grad_(True) for x in embs]
            for dst, src in zip(e.inputs, embs):
                dst.data.copy_(src)
            e.lens_both = torch.cat([torch.full((B,), R, dtype=torch.int32, device=dev), torch.full((B,), T, dtype=torch.int32, device=dev)])
            e.lens = (e.lens_both[:B], e.lens_both[B:])      # static addresses: refreshed by ONE host->device copy per new batch
            for _ in range(2):                               # warm-up outside capture: lazy initialisation, LDS reservations
                self._run(e, before)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        for t in e.inputs:
            t.grad = None
        e.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(e.graph):
            loss, losses = self._run(e, before)
            e.loss = loss.detach()
            e.terms = OrderedDict((k, v.detach()) for k, v in losses.items())
            # the logged values as ONE device buffer, filled inside the graph: a single copy brings them to the host
            e.log_buf = torch.stack([torch.as_tensor(t).detach().reshape(()).to(torch.float32) for _, t, _ in e.logged]) \
                if e.logged else None
        e.grads = [t.grad for t in e.inputs]
        e.log_names = [(key, n) for key, _, n in e.logged] if e.logged else []
        return e

    # ------------------------------------------------------------------------------------------ logging
    def _drain(self, wait):
        while self._pending:
            slot, names, logger = self._pending[0]
            if not wait and not slot.event.query():
                break
            slot.event.synchronize()
            vals = slot.view.tolist()
            for (key, n), v in zip(names, vals):
                logger.update(key, v, n)
            self._pending.popleft()

    def flush(self):
        """Hand every outstanding logged value to its logger (waits for the device)."""
        self._drain(True)

    def _log(self, e):
        # (the model's own slot, not its `logger` property: reading THAT delivers the outstanding values, i.e. waits for the device)
        logger = self.model.__dict__['_logger'] if '_logger' in self.model.__dict__ else self.model.logger
        if logger is None or e.log_buf is None:
            return
        k = self._log_k % len(self._log_slots)
        self._log_k += 1
        if self._log_slots[k] is None or self._log_slots[k].host.numel() != e.log_buf.numel():
            self._log_slots[k] = _PinnedSlot(e.log_buf.numel(), torch.float32)
        if len(self._pending) >= len(self._log_slots) - 1:
            self._drain(True)                                # never more copies in flight than staging buffers
        slot = self._log_slots[k].acquire()
        slot.host.copy_(e.log_buf, non_blocking=True)
        slot.event.record()
        self._pending.append((slot, e.log_names, logger))
        if self.log == 'sync':
            self._drain(True)

    def __call__(self, img_emb, cap_emb, img_emb_set, cap_emb_seq, img_lengths, cap_lengths, reg_loss=0, epoch=0,
                 distill_epoch=2):
        if 'regularizehidden' in self.model.losses_types:
            raise NotImplementedError('aladin_amd.graphs: the regularisation term comes from the encoder; add it outside')
        self._drain(False)                                   # values of earlier steps that have arrived meanwhile
        embs = (img_emb, cap_emb, img_emb_set, cap_emb_seq)
        B, D = img_emb.shape
        key = (B, img_emb_set.shape[0], cap_emb_seq.shape[0], D, bool(epoch < distill_epoch), img_emb.device)
        e = self._cache.get(key)
        if e is None:
            e = self._capture(key, embs)
            self._cache[key] = e
            while len(self._cache) > self.cache_size:
                self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(key)
        # plain ints, whatever the caller holds them in (lists, numpy, CPU or device tensors: one .tolist() each, never a
        # per-element tensor comparison)
        lk = (tuple(int(v) for v in (img_lengths.tolist() if torch.is_tensor(img_lengths) else img_lengths)),
              tuple(int(v) for v in (cap_lengths.tolist() if torch.is_tensor(cap_lengths) else cap_lengths)))
        if lk != e.lens_key:
            # one pinned buffer per in-flight copy (rotation + event): the host may run ahead of the device by several
            # steps, and a staging buffer must not be rewritten before its host->device copy has executed
            slot = e.lens_slots[e.lens_k % len(e.lens_slots)].acquire()
            e.lens_k += 1
            slot.view[:B] = lk[0]
            slot.view[B:] = lk[1]
            e.lens_both.copy_(slot.host, non_blocking=True)
            slot.event.record()
            e.lens_key = lk
        loss = _Replay.apply(e, *embs)
        self._log(e)
        return loss, OrderedDict(e.terms)

