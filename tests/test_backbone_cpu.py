"""CPU tier: aladin_amd/backbone.py (VinVL / Oscar BertImgModel, plain PyTorch) against tests/golden/backbone_bertimg.npz --
outputs and gradients of the REFERENCE's own BertImgModel.forward (oscar/modeling/modeling_bert.py:150-279; attention
arithmetic :28-70, layer loop :88-147) recorded in the build container over a restatement of the un-vendored
`transformers.pytorch_transformers` layers (tests/golden/make_golden.py: gen_backbone, which also checks those layers
against the installed transformers' BertModel).  Also: reference parameter names (strict loading of a VinVL-style
checkpoint directory and of a reference ALADIN checkpoint's `img_txt_enc.*` keys)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden


def backbone_case(device='cpu', **flags):
    """-> (golden dict, model with the golden's weights, inputs) -- shared with the GPU tier."""
    import importlib.util
    from conftest import GOLDEN
    from aladin_amd import synth
    from aladin_amd.backbone import BertConfig, BertImgModel
    g = load_golden('backbone_bertimg')
    cfg = BertConfig(**json.loads(str(g['cfg_json'])), **flags)
    model = BertImgModel(cfg).eval()
    names = [str(n) for n in g['param_names']]
    own = dict(model.named_parameters())
    assert sorted(own) == sorted(names), 'parameter names differ from the reference BertImgModel'
    vals = synth.module_parameters([(n, tuple(own[n].shape)) for n in names], int(g['seed']), scale=0.08)
    with torch.no_grad():
        for n in names:
            own[n].copy_(torch.from_numpy(vals[n]))
    spec = importlib.util.spec_from_file_location('make_golden_inputs', os.path.join(GOLDEN, 'backbone_inputs.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ids, tmask, fmask, types_, feats = mod.backbone_inputs(int(g['seed']) + 50)
    assert abs(synth.checksum(ids) - float(g['ids_checksum'])) < 1e-6 and abs(synth.checksum(feats) - float(g['feats_checksum'])) < 1e-4
    to = lambda x: torch.from_numpy(x).to(device)             # noqa: E731
    return g, model.to(device), (to(ids), to(tmask), to(fmask), to(types_), to(feats))


@pytest.mark.parametrize('explicit', [False, True])
def test_bertimg_matches_the_reference_forward_and_backward(explicit):
    """explicit=False: fused SDPA attention (what the encoder runs); True: the matmul / softmax path with every hidden state
    and attention map returned, as the reference configures it (alad_model.py:41-42)."""
    from aladin_amd import synth
    g, model, (ids, tmask, fmask, types_, feats) = backbone_case(output_attentions=explicit, output_hidden_states=explicit)
    o_txt = model(ids, token_type_ids=types_, attention_mask=tmask, img_feats=None)
    f = feats.clone().requires_grad_(True)
    o_img = model(ids, token_type_ids=types_, attention_mask=fmask, img_feats=f)
    assert len(o_txt) == len(o_img) == (4 if explicit else 2)
    tol = dict(rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(o_txt[0].detach().numpy(), g['txt_seq'], **tol)
    np.testing.assert_allclose(o_txt[1].detach().numpy(), g['txt_pooled'], **tol)
    np.testing.assert_allclose(o_img[0].detach().numpy(), g['img_seq'], **tol)
    np.testing.assert_allclose(o_img[1].detach().numpy(), g['img_pooled'], **tol)
    if explicit:
        assert len(o_img[2]) == int(g['n_hidden']) and len(o_img[3]) == int(g['n_att'])
        np.testing.assert_allclose(o_img[2][1].detach().numpy(), g['img_hidden_1'], **tol)
        np.testing.assert_allclose(o_img[3][-1].detach().numpy(), g['img_att_last'], **tol)
    w = torch.from_numpy(synth.normal(tuple(o_img[0].shape), int(g['seed']) + 60))
    model.zero_grad()
    (o_img[0] * w).sum().add(o_img[1].sum()).backward()
    def close(got, ref):                                    # fp32 sums of differently ordered terms: judged against the largest entry
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-6 * float(np.abs(ref).max()))
    close(f.grad.numpy(), g['d_feats'])
    close(model.img_embedding.weight.grad.numpy(), g['d_img_embedding_weight'])
    close(model.encoder.layer[0].attention.self.query.weight.grad.numpy(), g['d_q0'])
    np.testing.assert_allclose(float(model.embeddings.word_embeddings.weight.grad.abs().sum()), float(g['d_word_embeddings_abs']), rtol=1e-4)


def test_padded_positions_do_not_reach_attended_ones():
    """The -10000 additive mask (modeling_bert.py:226-227): what sits in padded token / region slots must not change the
    states of attended positions."""
    g, model, (ids, tmask, fmask, types_, feats) = backbone_case()
    with torch.no_grad():
        a = model(ids, token_type_ids=types_, attention_mask=fmask, img_feats=feats)[0]
        ids2, feats2 = ids.clone(), feats.clone()
        n_tok = ids.shape[1]
        ids2[tmask == 0] = 7
        feats2[fmask[:, n_tok:] == 0] = 3.0
        b = model(ids2, token_type_ids=types_, attention_mask=fmask, img_feats=feats2)[0]
    keep = fmask.bool()
    np.testing.assert_allclose(b[keep].numpy(), a[keep].numpy(), rtol=1e-5, atol=1e-6)


def test_forward_pair_equals_the_two_passes():
    """BertImgModel.forward_pair (the caption pass and the tags + regions pass of alad_model.py:124-140 as one pass of 2B
    sequences) against the two separate passes -- themselves pinned to the reference above: states of the real positions and
    gradients."""
    from aladin_amd import synth
    g, model, (ids, tmask, fmask, types_, feats) = backbone_case()
    f1 = feats.clone().requires_grad_(True)
    a_txt = model(ids, token_type_ids=types_, attention_mask=tmask, img_feats=None)[0]
    a_img = model(ids, token_type_ids=types_, attention_mask=fmask, img_feats=f1)[0]
    w_t = torch.from_numpy(synth.normal(tuple(a_txt.shape), 5)) * tmask[:, :, None]
    w_i = torch.from_numpy(synth.normal(tuple(a_img.shape), 6)) * fmask[:, :, None]
    model.zero_grad()
    ((a_txt * w_t).sum() + (a_img * w_i).sum()).backward()
    g_ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    f2 = feats.clone().requires_grad_(True)
    b_txt, b_img = model.forward_pair(ids, types_, tmask, ids, types_, fmask, f2)
    model.zero_grad()
    ((b_txt * w_t).sum() + (b_img * w_i).sum()).backward()
    keep_t, keep_i = tmask.bool(), fmask.bool()
    np.testing.assert_allclose(b_txt[keep_t].detach().numpy(), a_txt[keep_t].detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(b_img[keep_i].detach().numpy(), a_img[keep_i].detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(f2.grad.numpy(), f1.grad.numpy(), rtol=1e-4, atol=1e-6 * float(f1.grad.abs().max()))
    top = max(float(v.abs().max()) for v in g_ref.values())
    for n, p in model.named_parameters():
        if n in g_ref:
            if n.endswith('key.bias'):          # zero in exact arithmetic (softmax is shift-invariant): rounding noise in both
                assert float(p.grad.abs().max()) <= 1e-4 * top and float(g_ref[n].abs().max()) <= 1e-4 * top
                continue
            scale = max(float(g_ref[n].abs().max()), 1e-3 * top)
            np.testing.assert_allclose(p.grad.numpy(), g_ref[n].numpy(), rtol=1e-4, atol=2e-6 * scale, err_msg=n)


def test_checkpoint_directory_loads_strictly_and_feeds_the_encoder(tmp_path):
    """A VinVL-style checkpoint directory (config.json + pytorch_model.bin with the reference's key names) -> encoder built
    from it as alad_model.py:39-43 does -> reference ALADIN checkpoint keys load with strict=True."""
    from aladin_amd.backbone import BertConfig, ImageBertForSequenceClassification
    from aladin_amd.encoder import JointTextImageTransformerEncoder
    cfgd = dict(vocab_size=50, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=64,
                max_position_embeddings=32, img_feature_dim=10, use_img_layernorm=1, img_layer_norm_eps=1e-12)
    torch.manual_seed(3)
    src = ImageBertForSequenceClassification(BertConfig(**cfgd))
    keys = set(src.state_dict())
    for k in ('bert.embeddings.word_embeddings.weight', 'bert.embeddings.LayerNorm.bias', 'bert.img_embedding.weight', 'bert.LayerNorm.weight',
              'bert.encoder.layer.0.attention.self.query.weight', 'bert.encoder.layer.0.attention.output.LayerNorm.weight',
              'bert.encoder.layer.0.intermediate.dense.bias', 'bert.encoder.layer.0.output.dense.weight', 'bert.pooler.dense.weight',
              'classifier.weight'):
        assert k in keys, k
    with open(tmp_path / 'config.json', 'w') as f:
        json.dump(cfgd, f)
    torch.save(src.state_dict(), tmp_path / 'pytorch_model.bin')
    config = {'model': {'embed-size': 768, 'teran-layers': 0, 'tern-layers': 2, 'post-layers': 0, 'dropout': 0.1,
                        'shared-transformer': True, 'text-aggregation': 'first', 'image-aggregation': 'first'},
              'training': {'loss-type': 'alignment-distillation', 'measure': 'dot'}}
    enc = JointTextImageTransformerEncoder(config, oscar_checkpoint=str(tmp_path)).eval()
    for k, v in src.state_dict().items():
        assert torch.equal(enc.oscar_model.state_dict()[k], v)
    # the key set of the reference encoder for this configuration (alad_model.py:43,55-56,104-108)
    top = {k.split('.')[0] for k in enc.state_dict()}
    assert top == {'oscar_model', 'img_proj', 'cap_proj', 'final_projection_net'}
    # like the loader the reference goes through (pytorch_transformers' from_pretrained), keys outside the backbone are
    # tolerated with a warning: a pre-training checkpoint has cls.* and no classifier.*, newer exports carry position_ids
    other = dict(src.state_dict())
    other.pop('classifier.bias')
    other['cls.predictions.bias'] = torch.zeros(50)
    other['bert.embeddings.position_ids'] = torch.arange(32)[None]
    torch.save(other, tmp_path / 'pytorch_model.bin')
    with pytest.warns(UserWarning) as rec:
        m2 = ImageBertForSequenceClassification.from_pretrained(str(tmp_path))
    msgs = ' '.join(str(w.message) for w in rec)
    assert 'classifier.bias' in msgs and 'cls.predictions.bias' in msgs and 'position_ids' in msgs
    assert torch.equal(m2.bert.pooler.dense.weight, src.bert.pooler.dense.weight)
    with pytest.raises(RuntimeError, match='lacks backbone parameters'):        # a MISSING bert.* parameter is an error
        bad = dict(src.state_dict())
        bad.pop('bert.encoder.layer.0.output.dense.weight')
        torch.save(bad, tmp_path / 'pytorch_model.bin')
        ImageBertForSequenceClassification.from_pretrained(str(tmp_path))
    torch.save(src.state_dict(), tmp_path / 'pytorch_model.bin')
    # drive the 7-tuple once on CPU up to the HIP l2norm (which needs the GPU): shapes of the hand-off
    B, n_tok, n_reg = 2, 6, 4
    ids = torch.randint(1, 50, (B, n_tok))
    with torch.no_grad():
        seq = enc.oscar_model.bert(input_ids=ids, attention_mask=torch.ones(B, n_tok + n_reg, dtype=torch.long),
                                   token_type_ids=torch.zeros_like(ids), img_feats=torch.randn(B, n_reg, 10))[0]
    assert tuple(seq.shape) == (B, n_tok + n_reg, 768)
