"""Secondary measurements of bench.py's JSON line (the claims DESIGN.md makes next to the headline are
timed by the driver's own run, not only by builder-run tools).  Each function takes a few seconds on one MI355X and
returns a small dict; bench.py wraps every call in try / except and records {'error': ...} instead of failing the line.

    loss_heads_bs32            the shipped distillation YAML's loss-head step at its batch size 32: eager modules,
                               GraphedLossStep (end to end incl. staging + backward()), graph replay alone
    e2e_config4                configs[4]'s training step over the VinVL-base sized backbone with random weights: fp32 as the
                               reference trains, bf16 autocast with both BERT passes batched; the loss heads' share
    alignment_retrieval_coco1k the alignment-head 1000 x 5000 grid (sets padded to 71): rank-exact split precision and fp16
    forward_chain              pack + side GEMM + score kernel of the headline batch, event-timed
    cpu_baseline_b256          the reference's dataflow at B = 256 on the host cores, live (1 warm-up + 2 timed steps)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ev_ms(fn, iters, warm=3):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _wall_ms(fn, iters, warm=3, warm_s=0.0):
    """Wall clock per call.  warm_s: keep calling for that long first -- a HOST-bound step (bs 32: Python + launches) is only comparable
    once the host core holds its working clock: measured on this pool, the same eager step takes 0.32 ms in the first ~100 ms of
    a process and 0.20 ms a few seconds later (tools/experiments/host_slowdown_probe.py), like the GPU's clock-settling pre-roll."""
    import torch
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(20):
            fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def loss_heads_bs32(dev, R=51, Tn=38, B=32):
    """alad-alignment-and-matching-distill.yaml (loss-type 'alignment-distillation', listnet, margin 0.2, max_violation) at the
    YAML's batch size and the shipped data shape (50 regions + 35 tokens): matching scores + hinge, alignment scores + hinge,
    listnet, backward.  Wall-clock per step (the host is part of what bounds it at this size)."""
    import torch
    from aladin_amd import synth
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.graphs import GraphedLossStep
    from aladin_amd.loss import AlignmentContrastiveLoss, ContrastiveLoss, DistillationLoss
    im, s, il, sl = synth.alignment_batch(B, R, Tn, 768, seed=7, ragged=True)
    gi, gc = synth.global_embeddings(B, 768, seed=8)
    a = torch.from_numpy(im).to(dev).requires_grad_(True)
    b = torch.from_numpy(s).to(dev).requires_grad_(True)
    x = torch.from_numpy(gi).to(dev).requires_grad_(True)
    y = torch.from_numpy(gc).to(dev).requires_grad_(True)
    mc, ac, dc = ContrastiveLoss(0.2, 'dot', True), AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw'), DistillationLoss('listnet')

    def eager():
        for t in (a, b, x, y):
            t.grad = None
        _, M = mc(x, y, return_similarity_mat=True)
        la, S = ac(a, b, il, sl, return_similarity_mat=True)
        (la + dc(S, M)).backward()

    model = ALADModel({'training': {'loss-type': 'alignment-distillation', 'loss-weights': [1, 1], 'margin': 0.2, 'measure': 'dot',
                                    'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}})
    gstep = GraphedLossStep(model)
    a_s = a.detach().permute(1, 0, 2).contiguous().requires_grad_(True)
    b_s = b.detach().permute(1, 0, 2).contiguous().requires_grad_(True)
    seed = torch.ones((), device=dev)

    def graphed():
        for t in (a_s, b_s, x, y):
            t.grad = None
        loss, _ = gstep(x, y, a_s, b_s, il, sl, epoch=5)
        loss.backward(gradient=seed)

    def replay_only():
        next(iter(gstep._cache.values())).graph.replay()

    # the import swap with NO change to the train loop (round 6): ALADModel(config, graphed=True) called as alad/train.py:416 calls it;
    # `model_flag_logged_ms` additionally reads model.logger every step, as the reference's loop does for TensorBoard (:447)
    from aladin_amd.evaluation import LogCollector
    model_g = ALADModel(model.config, graphed=True)
    model_g.logger = LogCollector()
    model_g.forward_emb = lambda p, q: (x, y, a_s, b_s, il, sl, 0)
    model_e = ALADModel(model.config, graphed=False)
    model_e.logger = LogCollector()
    model_e.forward_emb = model_g.forward_emb

    def via_model(m, read_logger):
        def run():
            for t in (a_s, b_s, x, y):
                t.grad = None
            loss, _ = m(None, None, epoch=5, distill_epoch=2)
            loss.backward(gradient=seed)
            if read_logger:
                str(m.logger)
        return run

    graphed()
    out = {'workload': 'loss heads of alad-alignment-and-matching-distill.yaml at bs %d, R=%d, T=%d (50 regions + 35 tokens), D=768, '
                       'ragged lengths: matching + alignment hinge + listnet, forward + backward; wall clock per step' % (B, R, Tn),
           'host_preroll_s': 0.6,
           'eager_ms': round(_wall_ms(eager, 300, warm_s=0.6), 4), 'graphed_step_ms': round(_wall_ms(graphed, 500, warm_s=0.6), 4),
           'graph_replay_only_ms': round(_wall_ms(replay_only, 500), 4),
           'model_eager_ms': round(_wall_ms(via_model(model_e, False), 300, warm_s=0.6), 4),
           'model_flag_ms': round(_wall_ms(via_model(model_g, False), 500, warm_s=0.6), 4),
           'model_flag_logged_ms': round(_wall_ms(via_model(model_g, True), 500, warm_s=0.6), 4)}
    gstep.flush()
    return out


def e2e_config4(dev, bs=32):
    """configs[4] as far as it goes offline: the step of alad-alignment-and-matching-distill.yaml over the VinVL-base sized
    BertImgModel with RANDOM weights (no checkpoint / COCO features here), 35 tokens + 50 regions."""
    import numpy as np
    import torch
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.backbone import BertConfig, ImageBertForSequenceClassification
    config = {'model': {'embed-size': 768, 'text-aggregation': 'first', 'image-aggregation': 'first', 'freeze-teran': False,
                        'teran-layers': 0, 'tern-layers': 2, 'post-layers': 0, 'shared-transformer': True,
                        'depth-aggregation-alignment': False, 'depth-aggregation-matching': False, 'dropout': 0.1},
              'training': {'max-violation': True, 'loss-type': 'alignment-distillation', 'loss-weights': [1, 1], 'alignment-mode': 'MrSw',
                           'distillation-mode': 'listnet', 'measure': 'dot', 'margin': 0.2, 'bs': bs}}
    n_tok, n_reg = 35, 50
    rng = np.random.default_rng(1)
    cap_len = [int(v) for v in rng.integers(8, 22, bs)]
    feat_len = [int(v) for v in rng.integers(15, n_reg + 1, bs)]
    cap_len[0], feat_len[1] = n_tok, n_reg
    ids = torch.from_numpy(rng.integers(1, 30000, (bs, n_tok))).to(dev)
    feats = torch.from_numpy(rng.standard_normal((bs, n_reg, 2054)).astype(np.float32)).to(dev)
    tmask = (torch.arange(n_tok)[None, :] < torch.tensor(cap_len)[:, None]).long().to(dev)
    rmask = (torch.arange(n_reg)[None, :] < torch.tensor(feat_len)[:, None]).long().to(dev)
    types = torch.zeros_like(ids)
    ex_txt = (ids * tmask, tmask, types, None, cap_len)
    ex_img = (ids * tmask, torch.cat([tmask, rmask], 1), types, feats * rmask[:, :, None], None, feat_len)
    out = {'workload': 'configs[4] shape-level step: alad-alignment-and-matching-distill.yaml, bs %d, %d tokens + %d regions, VinVL-base sized '
                       'BertImgModel (random weights), forward + backward; GPU time per step (events)' % (bs, n_tok, n_reg)}
    for tag, ac in (('fp32', None), ('bf16_autocast_batched_passes', torch.bfloat16)):
        torch.manual_seed(0)
        model = ALADModel(config, backbone=ImageBertForSequenceClassification(BertConfig()), backbone_autocast=ac).to(dev).train()
        params = [p for p in model.parameters() if p.requires_grad]

        def full_step():
            for p in params:
                p.grad = None
            loss, _ = model(ex_img, ex_txt, epoch=5, distill_epoch=2)
            loss.backward()

        out[tag + '_step_ms'] = round(_ev_ms(full_step, 10, warm=3), 3)
        if ac is None:
            out['parameters_M'] = round(sum(p.numel() for p in params) / 1e6, 1)
            with torch.no_grad():
                sets = [t.detach() if isinstance(t, torch.Tensor) else t for t in model.forward_emb(ex_img, ex_txt)]
            leaves = [sets[k].clone().requires_grad_(True) for k in range(4)]

            def heads_only():
                for t in leaves:
                    t.grad = None
                loss, _ = model.forward_loss_total(leaves[0], leaves[1], leaves[2], leaves[3], sets[4], sets[5], 0, 5, 2)
                loss.backward()

            out['loss_heads_ms'] = round(_ev_ms(heads_only, 50), 4)
            out['loss_heads_share_of_fp32_step'] = round(out['loss_heads_ms'] / out['fp32_step_ms'], 4)
        del model, params
        torch.cuda.empty_cache()
    return out


def alignment_retrieval_coco1k(dev):
    """The alignment-head retrieval grid of COCO-1k (1000 images x 5000 captions, sets padded to 71 positions as encode_data lays
    them out, synthetic COCO-like lengths): one launch per length class, rank-exact split precision and fp16 operands."""
    import torch
    from aladin_amd import evaluation as E, ops, synth
    n = 1000
    images, captions, il, cl = synth.eval_sets(n, 768, seed=9)
    ia = torch.from_numpy(images[0::5]).to(dev)
    ca = torch.from_numpy(captions).to(dev)
    ilen = il[0::5]
    out = {'workload': 'alignment-head grid 1000 x 5000 (COCO-1k protocol, sets padded to 71, lengths 12-34 regions / 7-30 tokens), D=768'}
    old = ops.set_eval_precision('split')
    try:
        for prec in ('split', 'fp16'):
            ops.set_eval_precision(prec)
            E.clear_eval_cache()
            out[prec + '_ms'] = round(_ev_ms(lambda: E.compute_sim_matrix(ia, ca, ilen, cl, mode='alignment'), 5, warm=2), 3)
    finally:
        ops.set_eval_precision(old)
        E.clear_eval_cache()
    out['pairs_per_s_split'] = round(n * 5 * n / (out['split_ms'] * 1e-3), 1)
    return out


def forward_chain(im, s, il, sl, groups=5, iters=100):
    """pack + side GEMM + score kernel of the bench batch (what `alignment scoring` costs end to end on the device), event-timed:
    median of `groups` groups of `iters` chains."""
    import statistics
    import torch
    from aladin_amd import ops
    dev = im.device
    ilt, slt = ops.lengths_tensor(il, dev), ops.lengths_tensor(sl, dev)

    def chain():
        ops._align_forward(im, s, ilt, slt)               # pack_both + side GEMM + score kernel (operand buffers from torch's caching allocator)
    for _ in range(50):
        chain()
    ms = statistics.median(_ev_ms(chain, iters, warm=0) for _ in range(groups))
    return ms


def cpu_baseline_b256(threads, budget_s=60.0):
    """SURVEY 8(d) CPU timing plan, live: the reference's dataflow (oracle/faithful_torch.py = alad/loss.py:79-159 op for op) at
    B = 256, R = 34, T = 50, D = 768, forward + backward, at the thread count the B = 16 sweep found best: 1 warm-up + 2 timed
    steps (~20 s, ~40 GB of host memory).  Raises MemoryError / RuntimeError when the host cannot hold it (bench.py then attaches
    the committed one-off measurement and says so)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import faithful_torch as FT
    from aladin_amd import synth
    avail = None
    try:
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable:'):
                avail = int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    # a container's memory cap does not show in /proc/meminfo: an OOM kill there takes the whole bench line with it
    for lim in ('/sys/fs/cgroup/memory.max', '/sys/fs/cgroup/memory/memory.limit_in_bytes'):
        try:
            v = open(lim).read().strip()
            if v.isdigit():
                cap = int(v) / 2 ** 30
                used = 0.0
                for cur in ('/sys/fs/cgroup/memory.current', '/sys/fs/cgroup/memory/memory.usage_in_bytes'):
                    try:
                        used = int(open(cur).read().strip()) / 2 ** 30
                        break
                    except (OSError, ValueError):
                        pass
                avail = cap - used if avail is None else min(avail, cap - used)
            break
        except OSError:
            continue
    if avail is not None and avail < 60.0:
        raise MemoryError('only %.0f GiB of host memory available (cgroup limit included); the faithful B = 256 step needs ~40 GiB' % avail)
    im, s, il, sl = synth.alignment_batch(256, 34, 50, 768, seed=1234, ragged=False)
    a, b = torch.from_numpy(im), torch.from_numpy(s)
    torch.set_num_threads(int(threads))
    t0 = time.perf_counter()
    FT.alignment_triplet_step(a, b, il, sl, 0.2, True, backward=True)                  # warm-up
    first = time.perf_counter() - t0
    times = []
    for _ in range(2):
        if times and time.perf_counter() - t0 > budget_s:
            break
        t1 = time.perf_counter()
        FT.alignment_triplet_step(a, b, il, sl, 0.2, True, backward=True)
        times.append(time.perf_counter() - t1)
    best = min(times) if times else first
    return {'workload': 'reference dataflow (oracle/faithful_torch.py) B=256 R=34 T=50 D=768 fp32, fwd+bwd', 'source': 'live',
            'threads': int(threads), 's_per_step': round(best, 3), 'steps_timed': len(times), 'warmup_s': round(first, 3),
            'pairs_per_s_fwd_bwd': round(256 * 256 / best, 1), 'host_mem_available_GiB': None if avail is None else round(avail, 1)}


if __name__ == '__main__':
    import torch
    dev = torch.device('cuda:0')
    for fn in (loss_heads_bs32, e2e_config4, alignment_retrieval_coco1k):
        print(json.dumps(fn(dev)), flush=True)
