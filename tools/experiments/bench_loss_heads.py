#!/usr/bin/env python3
"""The whole loss-head step of the shipped distillation config (configs/alad-alignment-and-matching-
distill.yaml: loss-type 'alignment-distillation', listnet, margin 0.2, max_violation) -- matching
scores + hinge, alignment scores + hinge, listnet distillation, backward -- on one MI355X:
this library's modules vs the reference's formulas in eager PyTorch-ROCm (restated here, masks built
by Python loops as alad/loss.py:103-115 does).  bs = 32 is the YAML's batch size."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F


def ref_hinge(scores, margin=0.2):
    d = scores.diag().view(-1, 1)
    eye = torch.eye(scores.size(0), device=scores.device) > .5
    cs = (margin + scores - d).clamp(min=0).masked_fill_(eye, 0)
    ci = (margin + scores - d.t()).clamp(min=0).masked_fill_(eye, 0)
    return cs.max(1)[0].sum() + ci.max(0)[0].sum()


def ref_alignment(im_set, s_seq, im_len, s_len):
    im_set = F.normalize(im_set, p=2, dim=2)[:, 1:, :]
    s_seq = F.normalize(s_seq, p=2, dim=2)[:, 1:-2, :]
    im_len = [l - 1 for l in im_len]
    s_len = [l - 3 for l in s_len]
    Bi, Ri, Bc, Tc = im_set.size(0), im_set.size(1), s_seq.size(0), s_seq.size(1)
    a = im_set.unsqueeze(1).expand(-1, Bc, -1, -1)
    b = s_seq.unsqueeze(0).expand(Bi, -1, -1, -1)
    al = torch.matmul(a, b.permute(0, 1, 3, 2))
    im_mask = torch.zeros(Bi, Ri, dtype=torch.bool, device=al.device)
    for row, l in zip(im_mask, im_len):
        row[l:] = True
    s_mask = torch.zeros(Bc, Tc, dtype=torch.bool, device=al.device)
    for row, l in zip(s_mask, s_len):
        row[l:] = True
    al = al.masked_fill(im_mask[:, None, :, None] | s_mask[None, :, None, :], 0)
    return al.max(2)[0].sum(2)


def ref_listnet(teacher, student):
    teacher = teacher.detach()
    loss = 0
    for dim in (1, 0):
        p = F.softmax(teacher, dim=dim)
        q = F.softmax(6.0 * student, dim=dim) + 1e-10
        loss = loss + torch.mean(-torch.sum(p * torch.log(q), dim=dim))
    return loss


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    from aladin_amd import synth
    from aladin_amd.loss import AlignmentContrastiveLoss, ContrastiveLoss, DistillationLoss
    dev = torch.device('cuda:0')
    profile = '--profile' in sys.argv
    batches = [int(v) for v in sys.argv[1:] if not v.startswith('--')] or [32, 256]
    # --shape=R,T: the set lengths (default BASELINE's 34 x 50; the shipped data config is 50 regions + 35 tokens: --shape=51,38)
    R, Tn = next(([int(x) for x in v.split('=')[1].split(',')] for v in sys.argv[1:] if v.startswith('--shape=')), [34, 50])
    for B in batches:
        im, s, il, sl = synth.alignment_batch(B, R, Tn, 768, seed=7, ragged=True)
        gi, gc = synth.global_embeddings(B, 768, seed=8)
        a = torch.from_numpy(im).to(dev).requires_grad_(True)
        b = torch.from_numpy(s).to(dev).requires_grad_(True)
        x = torch.from_numpy(gi).to(dev).requires_grad_(True)
        y = torch.from_numpy(gc).to(dev).requires_grad_(True)
        mc, ac, dc = ContrastiveLoss(0.2, 'dot', True), AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw'), DistillationLoss('listnet')

        def zero():
            for t in (a, b, x, y):
                t.grad = None

        def ours():
            zero()
            _, M = mc(x, y, return_similarity_mat=True)
            la, S = ac(a, b, il, sl, return_similarity_mat=True)
            (la + dc(S, M)).backward()

        def ref():
            zero()
            M = x.mm(y.t())
            ref_hinge(M)                                               # computed and dropped, alad_model.py:380-381
            S = ref_alignment(a, b, il, sl)
            (ref_hinge(S) + ref_listnet(S, M)).backward()

        # the same step captured once into a HIP graph and replayed (aladin_amd.graphs.GraphedLossStep over ALADModel.forward_loss)
        from aladin_amd.alad_model import ALADModel
        from aladin_amd.graphs import GraphedLossStep
        model = ALADModel({'training': {'loss-type': 'alignment-distillation', 'loss-weights': [1, 1], 'margin': 0.2, 'measure': 'dot',
                                        'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}})
        gstep = GraphedLossStep(model)
        a_s, b_s = a.detach().permute(1, 0, 2).contiguous().requires_grad_(True), b.detach().permute(1, 0, 2).contiguous().requires_grad_(True)

        seed = torch.ones((), device=dev)                   # d loss / d loss, allocated once (as bench.py does)

        def graphed():
            for t in (a_s, b_s, x, y):
                t.grad = None
            loss, _ = gstep(x, y, a_s, b_s, il, sl, epoch=5)
            loss.backward(gradient=seed)

        def graph_only():                                       # the captured kernels alone (no input copies, no autograd glue)
            next(iter(gstep._cache.values())).graph.replay()

        graphed()
        # the same with a logger attached: deferred logging (one asynchronous D2H per step, the host never waits) and the
        # reference's protocol (a host wait per step)
        from aladin_amd.evaluation import LogCollector
        model.logger = LogCollector()
        t_logged = timed(graphed, 200)
        gstep.flush()
        gsync = GraphedLossStep(model, log='sync')

        def graphed_sync():
            for t in (a_s, b_s, x, y):
                t.grad = None
            loss, _ = gsync(x, y, a_s, b_s, il, sl, epoch=5)
            loss.backward(gradient=seed)
        t_sync = timed(graphed_sync, 200)
        model.logger = None
        if profile:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            timed(graphed, 50)
            pr.enable()
            for _ in range(500):
                graphed()
            pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
        with torch.no_grad():
            l1 = float(ac(a, b, il, sl)) + float(dc(ac(a, b, il, sl, return_loss=False, return_similarity_mat=True), x.mm(y.t())))
            Sr = ref_alignment(a, b, il, sl)
            l2 = float(ref_hinge(Sr)) + float(ref_listnet(Sr, x.mm(y.t())))
        print(json.dumps({'batch': B, 'R': R, 'T': Tn, 'hip_ms': round(timed(ours, 50), 4), 'hip_graphed_step_ms': round(timed(graphed, 200), 4),
                          'hip_graphed_step_deferred_log_ms': round(t_logged, 4), 'hip_graphed_step_sync_log_ms': round(t_sync, 4),
                          'hip_graph_replay_only_ms': round(timed(graph_only, 500), 4), 'torch_rocm_eager_ms': round(timed(ref, 10), 3),
                          'loss_hip': round(l1, 5), 'loss_eager': round(l2, 5)}), flush=True)


if __name__ == '__main__':
    main()
