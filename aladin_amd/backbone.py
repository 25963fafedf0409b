"""VinVL / Oscar backbone on PyTorch-ROCm (the last piece): `BertImgModel`
(reference oscar/modeling/modeling_bert.py:150-279) and the `ImageBertForSequenceClassification` shell the
ALADIN encoder instantiates (alad/alad_model.py:39-43, oscar/modeling/modeling_bert.py:290-320).

north_star keeps this part as host PyTorch code: it is the producer of the (B, T, 768) / (B, R, 768) states the
HIP alignment / matching path consumes, not part of that path.  Written for ROCm: attention goes through
`F.scaled_dot_product_attention` with the reference's additive -10000 mask (one fused kernel per layer instead of
matmul / add / softmax / dropout / matmul) unless the attention maps themselves are requested.

What comes from where:
  * oscar/modeling/modeling_bert.py (in /root/reference): BertImgModel.__init__/forward -- image embedding
    Linear(img_feature_dim -> hidden) [+ LayerNorm(img_layer_norm_eps)] + dropout, concatenation after the text
    embeddings, the extended attention mask, the layer loop (CaptionBertEncoder :88-127), the attention arithmetic
    (CaptionBertSelfAttention :28-70), layer wiring (CaptionBertLayer / CaptionBertAttention :72-147), the outputs tuple.
  * huggingface/transformers @ 067923d3267325f525f4e46f357360c191ba562e (`pytorch_transformers`, an EMPTY submodule in
    the reference, .gitmodules:1-4): BertEmbeddings, BertSelfOutput, BertIntermediate (erf GELU), BertOutput, BertPooler,
    BertLayerNorm(eps) and transpose_for_scores -- restated here from the published BERT algorithm.  PARITY UNPINNED for
    these (no reference test or vector exists at that boundary); tests/golden/backbone_bertimg.npz pins this module
    against the reference's OWN BertImgModel.forward running over a restatement of those layers, and against the
    installed transformers' BertModel on the text-only path (tests/golden/make_golden.py: gen_backbone).

Parameter names are the reference's, so `pytorch_model.bin` of a VinVL checkpoint loads with strict=True into
`ImageBertForSequenceClassification` (keys bert.embeddings.*, bert.encoder.layer.N.*, bert.pooler.*,
bert.img_embedding.*, bert.LayerNorm.*, classifier.*).
"""
import json
import math
import os

import torch
from torch import nn
from torch.nn import functional as F


class BertConfig:
    """The fields of the checkpoint's config.json that BertImgModel reads (defaults = VinVL base)."""

    DEFAULTS = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                    intermediate_size=3072, hidden_act='gelu', hidden_dropout_prob=0.1,
                    attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                    initializer_range=0.02, layer_norm_eps=1e-12, img_feature_dim=2054, img_feature_type='faster_r-cnn',
                    use_img_layernorm=1, img_layer_norm_eps=1e-12, num_labels=2, loss_type='sfmx',
                    output_attentions=False, output_hidden_states=False)

    def __init__(self, **kw):
        for k, v in self.DEFAULTS.items():
            setattr(self, k, v)
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_pretrained(cls, path):
        """`path`: a checkpoint directory holding config.json, or the json file itself (alad_model.py:40)."""
        f = os.path.join(path, 'config.json') if os.path.isdir(path) else path
        with open(f) as fh:
            return cls(**json.load(fh))

    def to_dict(self):
        return dict(self.__dict__)


def gelu(x):
    """The erf form pytorch_transformers' BERT uses (`x * 0.5 * (1 + erf(x / sqrt(2)))`), not the tanh approximation."""
    return F.gelu(x)


class BertEmbeddings(nn.Module):
    """word + position + token-type embeddings -> LayerNorm -> dropout."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None, position_ids=None):
        if position_ids is None:
            position_ids = torch.arange(input_ids.size(1), dtype=torch.long, device=input_ids.device).unsqueeze(0).expand_as(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        x = self.word_embeddings(input_ids) + self.position_embeddings(position_ids) + self.token_type_embeddings(token_type_ids)
        return self.dropout(self.LayerNorm(x))


class CaptionBertSelfAttention(nn.Module):
    """oscar/modeling/modeling_bert.py:23-70: softmax(Q K^T / sqrt(d_head) + mask) V per head, dropout on the
    probabilities.  One fused SDPA call unless the probabilities themselves are asked for (output_attentions) or a
    head mask is given."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads:
            raise ValueError('hidden size %d is not a multiple of the number of heads %d'
                             % (config.hidden_size, config.num_attention_heads))
        self.output_attentions = config.output_attentions
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def transpose_for_scores(self, x):
        return x.view(x.size(0), x.size(1), self.num_attention_heads, self.attention_head_size).permute(0, 2, 1, 3)

    def forward(self, hidden_states, attention_mask, head_mask=None, history_state=None):
        kv_in = hidden_states if history_state is None else torch.cat([history_state, hidden_states], dim=1)     # :32-36
        q = self.transpose_for_scores(self.query(hidden_states))
        k = self.transpose_for_scores(self.key(kv_in))
        v = self.transpose_for_scores(self.value(kv_in))
        if not self.output_attentions and head_mask is None:
            p = self.dropout.p if self.training else 0.0
            ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask.to(q.dtype), dropout_p=p)
            probs = None
        else:
            scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self.attention_head_size) + attention_mask   # :47-50
            probs = self.dropout(torch.softmax(scores, dim=-1))                                                      # :53-57
            if head_mask is not None:
                probs = probs * head_mask
            ctx = torch.matmul(probs, v)
        ctx = ctx.permute(0, 2, 1, 3).contiguous().view(hidden_states.size(0), hidden_states.size(1), self.all_head_size)
        return (ctx, probs) if self.output_attentions else (ctx,)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dropout(self.dense(hidden_states)) + input_tensor)


class CaptionBertAttention(nn.Module):
    """oscar/modeling/modeling_bert.py:72-86."""

    def __init__(self, config):
        super().__init__()
        self.self = CaptionBertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask, head_mask=None, history_state=None):
        self_outputs = self.self(input_tensor, attention_mask, head_mask, history_state)
        return (self.output(self_outputs[0], input_tensor),) + self_outputs[1:]


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        if config.hidden_act not in ('gelu', 'relu'):
            raise ValueError('unsupported hidden_act %r' % (config.hidden_act,))
        self.act = gelu if config.hidden_act == 'gelu' else F.relu

    def forward(self, hidden_states):
        return self.act(self.dense(hidden_states))


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dropout(self.dense(hidden_states)) + input_tensor)


class CaptionBertLayer(nn.Module):
    """oscar/modeling/modeling_bert.py:129-147 (post-LayerNorm BERT layer)."""

    def __init__(self, config):
        super().__init__()
        self.attention = CaptionBertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask, head_mask=None, history_state=None):
        att = self.attention(hidden_states, attention_mask, head_mask, history_state)
        return (self.output(self.intermediate(att[0]), att[0]),) + att[1:]


class CaptionBertEncoder(nn.Module):
    """oscar/modeling/modeling_bert.py:88-127: the layer loop, collecting hidden states / attentions on request."""

    def __init__(self, config):
        super().__init__()
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.layer = nn.ModuleList([CaptionBertLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states, attention_mask, head_mask=None, encoder_history_states=None):
        all_hidden, all_att = (), ()
        for i, layer in enumerate(self.layer):
            if self.output_hidden_states:
                all_hidden = all_hidden + (hidden_states,)
            hist = None if encoder_history_states is None else encoder_history_states[i]
            out = layer(hidden_states, attention_mask, None if head_mask is None else head_mask[i], hist)
            hidden_states = out[0]
            if self.output_attentions:
                all_att = all_att + (out[1],)
        if self.output_hidden_states:
            all_hidden = all_hidden + (hidden_states,)
        outputs = (hidden_states,)
        if self.output_hidden_states:
            outputs = outputs + (all_hidden,)
        if self.output_attentions:
            outputs = outputs + (all_att,)
        return outputs


class BertPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)

    def forward(self, hidden_states):
        return torch.tanh(self.dense(hidden_states[:, 0]))


def _init_bert_weights(module, std):
    """BertPreTrainedModel.init_weights: N(0, initializer_range) for Linear / Embedding weights, unit LayerNorm, zero biases."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=std)
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()


class BertImgModel(nn.Module):
    """oscar/modeling/modeling_bert.py:150-279 for region-feature inputs (`img_feature_type` other than the
    'dis_code*' code-book variants, which no VinVL / ALADIN configuration uses).

    forward(input_ids, token_type_ids, attention_mask, position_ids, head_mask, img_feats, encoder_history_states)
      -> (sequence_output (B, T [+ R], H), pooled_output (B, H) [, all hidden states] [, all attention maps])"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        if str(config.img_feature_type).startswith('dis_code'):
            raise NotImplementedError("aladin_amd.backbone: img_feature_type %r (discrete code books, modeling_bert.py:167-176) "
                                      "is not provided; VinVL uses region features" % (config.img_feature_type,))
        self.embeddings = BertEmbeddings(config)
        self.encoder = CaptionBertEncoder(config)
        self.pooler = BertPooler(config)
        self.img_dim = config.img_feature_dim
        self.img_feature_type = config.img_feature_type
        self.use_img_layernorm = getattr(config, 'use_img_layernorm', None)           # :162-165
        self.img_embedding = nn.Linear(self.img_dim, config.hidden_size, bias=True)  # :178
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        if self.use_img_layernorm:
            self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.img_layer_norm_eps)
        self.apply(lambda m: _init_bert_weights(m, config.initializer_range))

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, position_ids=None, head_mask=None,
                img_feats=None, encoder_history_states=None):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        if attention_mask.dim() == 2:                                  # :214-219  (B, 1, 1, L): one row broadcast over queries
            ext = attention_mask[:, None, None, :]
        elif attention_mask.dim() == 3:
            ext = attention_mask[:, None]
        else:
            raise NotImplementedError
        dtype = self.img_embedding.weight.dtype
        ext = (1.0 - ext.to(dtype)) * -10000.0                         # :226-227
        if head_mask is not None:                                      # :234-241
            if head_mask.dim() == 1:
                head_mask = head_mask[None, None, :, None, None].expand(self.config.num_hidden_layers, -1, -1, -1, -1)
            elif head_mask.dim() == 2:
                head_mask = head_mask[:, None, :, None, None]
            head_mask = head_mask.to(dtype)
        x = self.embeddings(input_ids, position_ids=position_ids, token_type_ids=token_type_ids)
        if encoder_history_states:
            assert img_feats is None, 'Cannot take image features while using encoder history states'
        if img_feats is not None:
            e = self.img_embedding(img_feats)                          # :258-263
            if self.use_img_layernorm:
                e = self.LayerNorm(e)
            x = torch.cat((x, self.dropout(e)), 1)                     # :265-266
        enc = self.encoder(x, ext, head_mask=head_mask, encoder_history_states=encoder_history_states)
        return (enc[0], self.pooler(enc[0])) + enc[1:]                 # :274-279


    def forward_pair(self, txt_input_ids, txt_token_type_ids, txt_attention_mask, img_input_ids, img_token_type_ids,
                     img_attention_mask, img_feats):
        """The two passes ALADIN makes per step (alad_model.py:124-140: captions alone, then tags + regions) as ONE pass over
        a batch of 2B sequences: the caption embeddings are zero-padded to the image pass's length and masked.  A training
        step of this model on an MI355X is bound by the NUMBER of kernels (thousands of launches of a few microseconds each,
        tools/experiments/bench_e2e_config4.py), not by their work, so one pass of twice the batch costs about what one of the two did.
        Masked positions get exp(-10000) = 0 attention weight exactly, every other operation is per position: the real
        positions' states equal those of the two separate passes up to GEMM summation order.
        -> (txt_sequence_output (B, T_txt, H), img_sequence_output (B, T_img + R, H))"""
        if self.encoder.output_attentions or self.encoder.output_hidden_states:
            raise NotImplementedError('forward_pair returns the last hidden states only')
        B, Tt = txt_input_ids.shape
        e_txt = self.embeddings(txt_input_ids, token_type_ids=txt_token_type_ids)
        e_img = self.embeddings(img_input_ids, token_type_ids=img_token_type_ids)
        f = self.img_embedding(img_feats)
        if self.use_img_layernorm:
            f = self.LayerNorm(f)
        e_img = torch.cat((e_img, self.dropout(f)), 1)
        L = e_img.size(1)
        if L < Tt:
            raise ValueError('forward_pair expects the image pass to be at least as long as the caption pass')
        x = torch.cat((F.pad(e_txt, (0, 0, 0, L - Tt)), e_img), 0)                                   # (2B, L, H)
        if txt_attention_mask is None:
            txt_attention_mask = torch.ones_like(txt_input_ids)
        m = torch.cat((F.pad(txt_attention_mask, (0, L - Tt)), img_attention_mask), 0)                 # (2B, L)
        ext = (1.0 - m[:, None, None, :].to(x.dtype)) * -10000.0
        h = self.encoder(x, ext)[0]
        return h[:B, :Tt], h[B:]


class ImageBertForSequenceClassification(nn.Module):
    """The shell the reference loads the VinVL checkpoint into (oscar/modeling/modeling_bert.py:290-320): ALADIN only calls
    `.bert(...)` (alad_model.py:131,140); `classifier` is kept so that the checkpoint's keys load with strict=True."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.num_labels = config.num_labels
        self.loss_type = getattr(config, 'loss_type', 'sfmx')
        self.bert = BertImgModel(config)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        kind = getattr(config, 'classifier', 'linear')
        if kind == 'mlp':
            scale = getattr(config, 'cls_hidden_scale', 2)
            self.classifier = nn.Sequential(nn.Linear(config.hidden_size, config.hidden_size * scale), nn.ReLU(),
                                            nn.Linear(config.hidden_size * scale, config.num_labels))
        else:
            self.classifier = nn.Linear(config.hidden_size, config.num_labels)
        self.classifier.apply(lambda m: _init_bert_weights(m, config.initializer_range))

    @classmethod
    def from_pretrained(cls, path, config=None, map_location='cpu'):
        """`path`: the checkpoint directory of the VinVL model zoo (config.json + pytorch_model.bin), as the reference
        passes it (alad_model.py:40-43, train.py --eval_model_dir).  Like pytorch_transformers' loader the reference goes
        through, keys outside the backbone are tolerated: a pre-training checkpoint carries `cls.*` and no `classifier.*`,
        newer exports add buffers such as `bert.embeddings.position_ids` -- those are reported with a warning.  A MISSING
        `bert.*` parameter is an error (the encoder would silently run on random weights)."""
        import warnings
        config = config or BertConfig.from_pretrained(path)
        model = cls(config)
        weights = os.path.join(path, 'pytorch_model.bin') if os.path.isdir(path) else path
        state = torch.load(weights, map_location=map_location)
        missing, unexpected = model.load_state_dict(state, strict=False)
        bad = [k for k in missing if k.startswith('bert.')]
        if bad:
            raise RuntimeError('aladin_amd.backbone: checkpoint %s lacks backbone parameters: %s%s'
                               % (weights, ', '.join(bad[:8]), ' ...' if len(bad) > 8 else ''))
        if missing:
            warnings.warn('aladin_amd.backbone: not in the checkpoint, left at their initial values: %s' % ', '.join(missing))
        if unexpected:
            warnings.warn('aladin_amd.backbone: checkpoint keys this model does not use: %s%s'
                          % (', '.join(unexpected[:8]), ' ...' if len(unexpected) > 8 else ''))
        return model

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, labels=None, position_ids=None, head_mask=None,
                img_feats=None):
        outputs = self.bert(input_ids, position_ids=position_ids, token_type_ids=token_type_ids, attention_mask=attention_mask,
                            head_mask=head_mask, img_feats=img_feats)
        logits = self.classifier(self.dropout(outputs[1]))
        return (logits,) + outputs[2:]
