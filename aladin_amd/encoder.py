"""The step immediately before the hot path: the matching head
`final_projection_net` and the hand-off from the backbone to the 7-tuple the loss heads consume.

    JointTextImageTransformerEncoder   <- alad/alad_model.py:29-247 for the model section every shipped YAML
                                          uses (teran-layers 0, no depth aggregation, post-layers 0): the
                                          backbone's last hidden states are sliced to the batch maxima
                                          (:174-175), run through the 2-layer nn.TransformerEncoder matching
                                          head with key-padding masks (:104-108, :231-233), slot 0 is taken,
                                          sets are F.normalize'd and globals l2norm'd (:237-241)
    slot0_transformer                  the head evaluated for what the model consumes -- output row 0 only

The VinVL / Oscar BERT (`oscar/modeling/modeling_bert.py:150-279`) is aladin_amd/backbone.py; it -- or any module with the
`.bert(input_ids=, attention_mask=, token_type_ids=, img_feats=) -> (sequence_output, ...)` call the reference
makes (alad_model.py:129,139) -- is handed in as `backbone`, or built from a VinVL checkpoint directory
(`oscar_checkpoint`, as the reference does at :40-43).

This module is host code on PyTorch-ROCm (north_star: "host code stays Python on PyTorch-ROCm for the
backbone").  Parameter names equal the reference's and the two projections the reference constructs but never calls
for these configurations (`img_proj`, `cap_proj`, :55-56) are kept, so the `img_txt_enc.*` entries of a reference
ALADIN checkpoint (train.py:329-337) load with strict=True: oscar_model.bert.*, oscar_model.classifier.*, img_proj.*,
cap_proj.*, final_projection_net.* (the 'first' aggregators of the shipped YAMLs hold no parameters).
Only `l2norm` runs in this package's HIP kernels.
"""
import torch
from torch import nn
from torch.nn import functional as F


def _layer_full(layer, x, pad_mask):
    """One post-norm nn.TransformerEncoderLayer (the reference's: relu, norm_first=False, batch_first=False) on
    the whole (S, B, D) sequence, written out so that it never takes torch's inference fast path (whose
    nested-tensor form zeroes padded rows): x = LN1(x + drop(SA(x))); x = LN2(x + drop(W2 drop(relu(W1 x))))."""
    a = layer.self_attn(x, x, x, key_padding_mask=pad_mask, need_weights=False)[0]
    x = layer.norm1(x + layer.dropout1(a))
    f = layer.linear2(layer.dropout(F.relu(layer.linear1(x))))
    return layer.norm2(x + layer.dropout2(f))


def _layer_row0(layer, x, pad_mask):
    """The same layer evaluated for OUTPUT ROW 0 only -> (B, D).  Keys and values still come from every
    position, but the query projection, the attention rows, the output projection, both LayerNorms and the
    feed-forward block are computed for slot 0 alone: 1/S of the layer's work past the K/V projection, which
    is all the model reads of the last layer (alad_model.py:231-233 take `[0]`)."""
    mha = layer.self_attn
    S, B, D = x.shape
    H, hd = mha.num_heads, D // mha.num_heads
    w, b = mha.in_proj_weight, mha.in_proj_bias
    q = F.linear(x[0], w[:D], b[:D])                                   # (B, D)
    kv = F.linear(x, w[D:], b[D:])                                     # (S, B, 2D)
    k, v = kv[..., :D], kv[..., D:]
    q = q.view(B, H, hd) * (hd ** -0.5)
    k = k.reshape(S, B, H, hd)
    v = v.reshape(S, B, H, hd)
    att = torch.einsum('bhd,sbhd->bhs', q, k)
    if pad_mask is not None:
        att = att.masked_fill(pad_mask[:, None, :], float('-inf'))
    att = F.dropout(torch.softmax(att, dim=-1), p=mha.dropout, training=layer.training)
    ctx = torch.einsum('bhs,sbhd->bhd', att, v).reshape(B, D)
    a = mha.out_proj(ctx)
    y = layer.norm1(x[0] + layer.dropout1(a))
    f = layer.linear2(layer.dropout(F.relu(layer.linear1(y))))
    return layer.norm2(y + layer.dropout2(f))


def slot0_transformer(encoder, x, pad_mask):
    """`encoder(x, src_key_padding_mask=pad_mask)[0]` for an nn.TransformerEncoder built as the reference builds
    its heads (alad_model.py:104-108): all layers but the last on the full sequence, the last one for row 0."""
    layers = list(encoder.layers)
    for layer in layers[:-1]:
        x = _layer_full(layer, x, pad_mask)
    out = _layer_row0(layers[-1], x, pad_mask)
    return encoder.norm(out) if encoder.norm is not None else out


def _pad_mask(lengths, max_len, device):
    """True at padded positions (alad_model.py:152-160), built on the device without the per-sample Python loop."""
    lens = torch.as_tensor([int(v) for v in lengths], device=device)
    return torch.arange(max_len, device=device)[None, :] >= lens[:, None]


class JointTextImageTransformerEncoder(nn.Module):
    """reference alad/alad_model.py:29-247, for the configurations the shipped YAMLs select.

    forward(examples_imgs, examples_txts) takes the reference's collated tuples
        examples_txts = (input_ids, attention_mask, token_type_ids, <unused>, cap_len)
        examples_imgs = (input_ids, attention_mask, token_type_ids, img_feats, <unused>, feat_len)
    and returns (img_glob (B,D), cap_glob (B,D), img_set (R,B,D), cap_seq (T,B,D), feat_len, cap_len, reg_loss)."""

    def __init__(self, config, backbone=None, oscar_checkpoint=None, backbone_autocast=None):
        """backbone_autocast: None (default: the backbone runs in fp32, as the reference trains) or a torch dtype
        (torch.bfloat16 / torch.float16): the two BERT passes run under torch.autocast -- 16-bit MFMA GEMMs instead of fp32
        ones; the hand-off, the matching head and the loss heads stay fp32.  An MI355X-side option, not reference behaviour:
        the end-to-end step of the shipped YAML is 99 % backbone (tools/experiments/bench_e2e_config4.py)."""
        super().__init__()
        self.backbone_autocast = backbone_autocast
        # The backbone's forward_pair (both BERT passes as one pass of 2B sequences) halves the launches but pads the caption pass
        # to the image pass's length: measured at bs 32 (tools/experiments/bench_e2e_config4.py) it pays when the step is launch-bound
        # (16-bit autocast: 32.1 -> 19.4 ms) and costs 5 % when the fp32 GEMMs dominate (33.5 -> 35.3 ms)
        self.batch_passes = backbone_autocast is not None
        m = config['model']
        if backbone is None:
            if oscar_checkpoint is None:
                raise ValueError('aladin_amd.encoder: pass a backbone module or the VinVL checkpoint directory')
            from .backbone import BertConfig, ImageBertForSequenceClassification
            bert_config = BertConfig.from_pretrained(oscar_checkpoint)                  # alad_model.py:40-43
            # the reference switches output_attentions / output_hidden_states on here (:41-42) and reads the extra
            # outputs only under depth aggregation, which the supported configurations do not use: left off, so that
            # attention runs as one fused kernel per layer
            backbone = ImageBertForSequenceClassification.from_pretrained(oscar_checkpoint, config=bert_config)
        if m.get('teran-layers', 0) != 0 or m.get('post-layers', 0) != 0 or m.get('depth-aggregation-alignment') \
                or m.get('depth-aggregation-matching') or m.get('depth-aggregation'):
            raise NotImplementedError('aladin_amd: only the model section of the shipped configs is provided '
                                      '(teran-layers 0, post-layers 0, no depth aggregation; alad/configs/*.yaml:10-16)')
        self.oscar_model = backbone                                   # alad_model.py:43 (injected instead of from_pretrained)
        self.freeze_teran = m.get('freeze-teran', False)
        embed_size = m['embed-size']
        self.embed_size = embed_size
        hidden_size = 768                                             # :54 (hard-wired in the reference)
        self.img_proj = nn.Linear(hidden_size, embed_size)           # :55-56: constructed, never called (kept for its keys)
        self.cap_proj = nn.Linear(hidden_size, embed_size)
        layer = nn.TransformerEncoderLayer(d_model=embed_size, nhead=4, dim_feedforward=embed_size, dropout=m['dropout'])
        self.final_projection_net = nn.TransformerEncoder(layer, num_layers=m['tern-layers'], enable_nested_tensor=False)      # :104-108
        self.l1_regularization = 'regularizehidden' in config['training']['loss-type']
        if self.l1_regularization:
            raise NotImplementedError("aladin_amd: 'regularizehidden' needs the backbone's hidden states (alad_model.py:222-227)")

    def forward(self, examples_imgs, examples_txts):
        from .loss import l2norm
        with torch.set_grad_enabled(torch.is_grad_enabled() and not self.freeze_teran), \
                torch.autocast('cuda', dtype=self.backbone_autocast or torch.bfloat16, enabled=self.backbone_autocast is not None):   # :121-123
            pair = getattr(self.oscar_model.bert, 'forward_pair', None) if self.batch_passes else None
            if pair is not None and examples_imgs[1] is not None:
                # both passes (:124-140) as one pass of 2B sequences: half the launches of a launch-bound step (backbone.py)
                t_seq, i_seq = pair(examples_txts[0], examples_txts[2], examples_txts[1], examples_imgs[0], examples_imgs[2],
                                    examples_imgs[1], examples_imgs[3])
                txt_out, img_out = (t_seq,), (i_seq,)
            else:
                txt_out = self.oscar_model.bert(input_ids=examples_txts[0], attention_mask=examples_txts[1],
                                                token_type_ids=examples_txts[2], img_feats=None)
                img_out = self.oscar_model.bert(input_ids=examples_imgs[0], attention_mask=examples_imgs[1],
                                                token_type_ids=examples_imgs[2], img_feats=examples_imgs[3])
        if self.backbone_autocast is not None:                         # everything downstream is fp32
            txt_out, img_out = (txt_out[0].float(),), (img_out[0].float(),)
        cap_len, feat_len = examples_txts[4], examples_imgs[5]
        n_tok = examples_txts[0].shape[1]                              # max_language_token_len (:147)
        max_cap, max_img = max(cap_len), max(feat_len)
        dev = txt_out[0].device
        txt_mask, img_mask = _pad_mask(cap_len, max_cap, dev), _pad_mask(feat_len, max_img, dev)
        c_emb = txt_out[0][:, :max_cap].permute(1, 0, 2)               # (T, B, D)   :174
        i_emb = img_out[0][:, n_tok:n_tok + max_img].permute(1, 0, 2)  # (R, B, D)   :175
        cap_glob = slot0_transformer(self.final_projection_net, c_emb, txt_mask)     # :231-233
        img_glob = slot0_transformer(self.final_projection_net, i_emb, img_mask)
        img_set = F.normalize(i_emb, p=2, dim=2)                       # :237-238 (teran-layers 0: the sets are the backbone's)
        cap_seq = F.normalize(c_emb, p=2, dim=2)
        return l2norm(img_glob), l2norm(cap_glob), img_set, cap_seq, feat_len, cap_len, 0      # :240-247
