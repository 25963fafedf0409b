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
The lengths travel as device int32 tensors refreshed before each replay; `model.logger` is fed after the replay
from one device->host copy (or not at all when it is None).
"""
from collections import OrderedDict

import torch


class _Entry:
    __slots__ = ('graph', 'inputs', 'lens', 'lens_host', 'lens_event', 'loss', 'terms', 'grads', 'logged', 'generation')


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, entry, *embs):
        for dst, src in zip(entry.inputs, embs):
            dst.copy_(src)                                   # also lays permuted (S,B,D)<->(B,S,D) views out as captured
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
    def __init__(self, model, cache_size=32):
        self.model = model
        self.cache_size = int(cache_size)
        self._cache = OrderedDict()

    def _run(self, e, epoch_before_distill):
        m = self.model
        for t in e.inputs:
            t.grad = None
        loss, losses = m.forward_loss_total(e.inputs[0], e.inputs[1], e.inputs[2], e.inputs[3], e.lens[0], e.lens[1], 0,
                                            0 if epoch_before_distill else 1, 1, log=False)
        e.logged = m.pending_log
        loss.backward()
        return loss, losses

    def _capture(self, key, embs):
        B, R, T, D, before, dev = key
        e = _Entry()
        e.generation = 0
        e.lens_event = None
        e.lens_host = (torch.empty(B, dtype=torch.int32).pin_memory(), torch.empty(B, dtype=torch.int32).pin_memory())
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            e.inputs = [torch.zeros_like(x, memory_format=torch.contiguous_format).requires_grad_(True) for x in embs]
            for dst, src in zip(e.inputs, embs):
                dst.data.copy_(src)
            e.lens = (torch.full((B,), R, dtype=torch.int32, device=dev), torch.full((B,), T, dtype=torch.int32, device=dev))
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
        e.grads = [t.grad for t in e.inputs]
        return e

    def __call__(self, img_emb, cap_emb, img_emb_set, cap_emb_seq, img_lengths, cap_lengths, reg_loss=0, epoch=0,
                 distill_epoch=2):
        if 'regularizehidden' in self.model.losses_types:
            raise NotImplementedError('aladin_amd.graphs: the regularisation term comes from the encoder; add it outside')
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
        if e.lens_event is not None:
            # the pinned staging buffers are reused: the previous step's asynchronous host->device copies must have been
            # EXECUTED before they are rewritten (with logger None nothing else makes the host wait for the device)
            e.lens_event.synchronize()
        for host, dst, src in zip(e.lens_host, e.lens, (img_lengths, cap_lengths)):
            host.copy_(torch.as_tensor([int(v) for v in src], dtype=torch.int32))
            dst.copy_(host, non_blocking=True)
        if e.lens_event is None:
            e.lens_event = torch.cuda.Event()
        e.lens_event.record()
        loss = _Replay.apply(e, *embs)
        m = self.model
        m.pending_log = e.logged
        m.flush_log()
        return loss, OrderedDict(e.terms)
