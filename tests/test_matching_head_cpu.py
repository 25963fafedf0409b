"""CPU tier: the matching head against tests/golden/matching_head.npz, which was made by
running the REFERENCE's JointTextImageTransformerEncoder.forward (alad/alad_model.py:119-247) on a fake backbone.
The head is host PyTorch code, so its arithmetic can be checked here; the full hand-off (with the HIP l2norm) and
the configs[4] shape-level smoke run in the GPU tier."""
import numpy as np
import torch

from conftest import load_golden, matching_head_case


def _l2norm_torch(x):                          # alad/utils.py:134-139, torch ops (the product path uses the HIP kernel)
    return x / x.pow(2).sum(dim=1, keepdim=True).sqrt()


def test_slot0_transformer_matches_reference_forward_and_backward():
    from aladin_amd.encoder import _pad_mask, slot0_transformer
    g = load_golden('matching_head')
    enc, a, b, cap_len, feat_len, n_tok, w = matching_head_case(g, torch.device('cpu'))
    c_emb = a[:, :max(cap_len)].permute(1, 0, 2)
    i_emb = b[:, n_tok:n_tok + max(feat_len)].permute(1, 0, 2)
    cap_glob = _l2norm_torch(slot0_transformer(enc.final_projection_net, c_emb, _pad_mask(cap_len, max(cap_len), a.device)))
    img_glob = _l2norm_torch(slot0_transformer(enc.final_projection_net, i_emb, _pad_mask(feat_len, max(feat_len), a.device)))
    np.testing.assert_allclose(cap_glob.detach().numpy(), g['cap_glob'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(img_glob.detach().numpy(), g['img_glob'], rtol=2e-5, atol=2e-6)
    img_set = torch.nn.functional.normalize(i_emb, p=2, dim=2)
    cap_seq = torch.nn.functional.normalize(c_emb, p=2, dim=2)
    assert list(img_set.shape) == list(g['img_set_shape']) and list(cap_seq.shape) == list(g['cap_seq_shape'])
    np.testing.assert_allclose(img_set.detach().numpy()[:, :, ::16], g['img_set_s'], rtol=1e-6, atol=1e-7)
    ((img_glob * w[0]).sum() + (cap_glob * w[1]).sum() + 0.05 * (img_set * w[2]).sum() + 0.05 * (cap_seq * w[3]).sum()).backward()
    for got, key in ((a.grad, 'd_txt_seq'), (b.grad, 'd_img_seq')):
        ref = g[key + '_s']
        scale = float(np.abs(ref).max())
        np.testing.assert_allclose(got.numpy()[:, :, ::16], ref, rtol=1e-4, atol=2e-6 * scale)
        np.testing.assert_allclose(float(got.abs().sum()), float(g[key + '_abs']), rtol=1e-5)
    for n, p in enc.final_projection_net.named_parameters():
        key = n.replace('.', '__')
        gr = p.grad.numpy()
        ref = g['dp_s__' + key]
        scale = max(1e-12, float(np.abs(ref).max()))
        np.testing.assert_allclose(gr.reshape(-1)[::max(1, gr.size // 512)][:512], ref, rtol=2e-4, atol=5e-6 * scale)
        np.testing.assert_allclose(float(np.abs(gr).sum()), float(g['dp_abs__' + key]), rtol=2e-5)


def test_row0_equals_full_transformer_encoder():
    """slot0_transformer == nn.TransformerEncoder(...)[0] (torch's own module, training-path arithmetic), ragged masks."""
    from aladin_amd.encoder import _pad_mask, slot0_transformer
    torch.manual_seed(0)
    layer = torch.nn.TransformerEncoderLayer(d_model=64, nhead=4, dim_feedforward=64, dropout=0.1)
    net = torch.nn.TransformerEncoder(layer, num_layers=2).eval()
    x = torch.randn(9, 6, 64, requires_grad=True)
    mask = _pad_mask([9, 4, 7, 1, 9, 5], 9, x.device)
    ref = net(x, src_key_padding_mask=mask)[0]
    got = slot0_transformer(net, x, mask)
    np.testing.assert_allclose(got.detach().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    gr, = torch.autograd.grad(ref.sin().sum(), x)
    gg, = torch.autograd.grad(got.sin().sum(), x)
    np.testing.assert_allclose(gg.numpy(), gr.numpy(), rtol=1e-4, atol=1e-6)


def test_state_dict_names_match_the_reference_head():
    """`img_txt_enc.final_projection_net.*` keys of a reference checkpoint load unchanged (the golden stores the
    reference head's parameter names)."""
    from aladin_amd.encoder import JointTextImageTransformerEncoder
    from standins import StandInBackbone
    g = load_golden('matching_head')
    cfg = {'model': {'embed-size': 768, 'teran-layers': 0, 'tern-layers': 2, 'post-layers': 0, 'dropout': 0.1,
                     'shared-transformer': True}, 'training': {'loss-type': 'alignment-distillation', 'measure': 'dot'}}
    enc = JointTextImageTransformerEncoder(cfg, StandInBackbone(hidden=768, feat_dim=16, vocab=50))
    names = [n for n, _ in enc.final_projection_net.named_parameters()]
    assert names == [str(n) for n in g['param_names']]
