"""TEST INFRASTRUCTURE for tests/golden/make_golden.py (gen_backbone) -- never imported by aladin_amd/.

The reference's oscar/modeling/modeling_bert.py subclasses BERT layers from `transformers.pytorch_transformers`
(huggingface/transformers @ 067923d3267325f525f4e46f357360c191ba562e), an EMPTY submodule in /root/reference.  To run the
reference's OWN BertImgModel (its image-embedding glue, attention arithmetic, layer loop) and record golden vectors, the
generator installs this module under that name: a plain restatement of the published BERT building blocks the
reference imports.  The arithmetic of these classes is therefore "parity unpinned" (DESIGN.md section 2); what the
fixture pins is everything the reference's file itself computes on top of them.
"""
import math

import torch
from torch import nn

BERT_PRETRAINED_MODEL_ARCHIVE_MAP = {}
BertLayerNorm = nn.LayerNorm


class BertConfig:
    def __init__(self, **kw):
        self.__dict__.update(dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                                  intermediate_size=3072, hidden_act='gelu', hidden_dropout_prob=0.1,
                                  attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                                  initializer_range=0.02, layer_norm_eps=1e-12, output_attentions=False,
                                  output_hidden_states=False, num_labels=2, torchscript=False, pruned_heads={}))
        self.__dict__.update(kw)


class BertPreTrainedModel(nn.Module):
    config_class = BertConfig
    base_model_prefix = 'bert'

    def __init__(self, config, *a, **k):
        super().__init__()
        self.config = config

    def init_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()


def _erf_gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None, position_ids=None):
        n = input_ids.size(1)
        if position_ids is None:
            position_ids = torch.arange(n, dtype=torch.long, device=input_ids.device)[None, :].expand_as(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        e = self.word_embeddings(input_ids)
        e = e + self.position_embeddings(position_ids)
        e = e + self.token_type_embeddings(token_type_ids)
        return self.dropout(self.LayerNorm(e))


class BertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.output_attentions = config.output_attentions
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def transpose_for_scores(self, x):
        x = x.view(*(x.size()[:-1] + (self.num_attention_heads, self.attention_head_size)))
        return x.permute(0, 2, 1, 3)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        h = self.dropout(self.dense(hidden_states))
        return self.LayerNorm(h + input_tensor)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        self.intermediate_act_fn = _erf_gelu if config.hidden_act == 'gelu' else torch.relu

    def forward(self, hidden_states):
        return self.intermediate_act_fn(self.dense(hidden_states))


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        h = self.dropout(self.dense(hidden_states))
        return self.LayerNorm(h + input_tensor)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])


class BertPooler(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        return self.activation(self.dense(hidden_states[:, 0]))


# imported by name in oscar/modeling/modeling_bert.py:10-16, used only by classes outside the ALADIN path
class BertPredictionHeadTransform(nn.Module):
    pass


class BertOnlyMLMHead(nn.Module):
    pass


class BertLMPredictionHead(nn.Module):
    pass


def load_tf_weights_in_bert(*a, **k):
    raise NotImplementedError
