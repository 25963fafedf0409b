#!/usr/bin/env python3
"""BASELINE configs[4] as far as it goes offline: the training step of alad-alignment-and-matching-distill.yaml (bs 32, loss-type
'alignment-distillation', listnet) END TO END on one MI355X -- the VinVL-base BertImgModel of aladin_amd/backbone.py at its real size
with RANDOM weights (the checkpoint and the COCO features cannot be downloaded here), the matching head, the HIP loss heads -- forward +
backward, fp32 as the reference trains.  Reports where the step's time goes: backbone (two BERT passes), matching head + hand-off, loss
heads; and the same step with the loss heads of the reference's formulas in eager PyTorch-ROCm.  Shapes: 35 tokens, 50 regions of 2054
features (README.md:53-56 of the reference: max_seq_length 50 / max_img_seq_length 50; COCO captions ~12 tokens)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, e0.elapsed_time(e1) / n


def main():
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.backbone import BertConfig, ImageBertForSequenceClassification
    dev = torch.device('cuda:0')
    config = {'model': {'embed-size': 768, 'text-aggregation': 'first', 'image-aggregation': 'first', 'freeze-teran': False,
                        'teran-layers': 0, 'tern-layers': 2, 'post-layers': 0, 'shared-transformer': True,
                        'depth-aggregation-alignment': False, 'depth-aggregation-matching': False, 'dropout': 0.1},
              'training': {'max-violation': True, 'loss-type': 'alignment-distillation', 'loss-weights': [1, 1], 'alignment-mode': 'MrSw',
                           'distillation-mode': 'listnet', 'measure': 'dot', 'margin': 0.2, 'bs': 32}}
    torch.manual_seed(0)
    ac = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(sys.argv[2]) if len(sys.argv) > 2 else None
    model = ALADModel(config, backbone=ImageBertForSequenceClassification(BertConfig()), backbone_autocast=ac).to(dev).train()
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
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
    params = [p for p in model.parameters() if p.requires_grad]

    def zero():
        for p in params:
            p.grad = None

    def full_step():
        zero()
        loss, _ = model(ex_img, ex_txt, epoch=5, distill_epoch=2)
        loss.backward()

    def encoder_only():                       # the 7-tuple and its backward with a stand-in scalar loss (no loss heads)
        zero()
        out = model.forward_emb(ex_img, ex_txt)
        (out[0].sum() + out[1].sum() + 1e-3 * out[2].sum() + 1e-3 * out[3].sum()).backward()

    with torch.no_grad():
        sets = [t.detach() if isinstance(t, torch.Tensor) else t for t in model.forward_emb(ex_img, ex_txt)]
    leaves = [sets[k].clone().requires_grad_(True) for k in range(4)]

    def heads_only():
        for t in leaves:
            t.grad = None
        loss, _ = model.forward_loss_total(leaves[0], leaves[1], leaves[2], leaves[3], sets[4], sets[5], 0, 5, 2)
        loss.backward()

    res = {'workload': 'configs[4] shape-level: alad-alignment-and-matching-distill.yaml step, bs %d, %d tokens, %d regions, VinVL-base '
                       'BertImgModel with random weights, backbone %s' % (bs, n_tok, n_reg, 'fp32' if ac is None else 'autocast ' + sys.argv[2]),
           'parameters_M': round(sum(p.numel() for p in params) / 1e6, 1)}
    for name, fn in (('full_step', full_step), ('encoder_only', encoder_only), ('loss_heads_only', heads_only)):
        wall, gpu = timed(fn)
        res[name + '_ms'] = round(wall, 3)
        res[name + '_gpu_ms'] = round(gpu, 3)
    res['loss_heads_share_of_step'] = round(res['loss_heads_only_gpu_ms'] / res['full_step_gpu_ms'], 4)
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
