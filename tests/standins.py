"""Random-init stand-ins for the parts of ALADIN that are outside this repo's scope (the VinVL / Oscar backbone), so that
the shape-level tests can drive ALADModel.forward end to end.  TEST INFRASTRUCTURE: nothing under aladin_amd/ imports this."""
import torch
from torch import nn

from aladin_amd.loss import l2norm


class StandInEncoder(nn.Module):
    """Random-init substitute for JointTextImageTransformerEncoder with the same OUTPUT contract
    (reference alad_model.py:121-247): region features (B,R,F) + box counts, token ids (B,T) +
    token counts -> the 7-tuple, sets L2-normalised (:237-238), globals l2norm'd (:240-241).
    It exists so that ALADModel.forward can be exercised end to end without the VinVL checkpoint;
    it is NOT a model of the backbone."""

    def __init__(self, feat_dim=2054, embed=768, vocab=30522, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.img_proj = nn.Linear(feat_dim, embed)
        self.tok_emb = nn.Embedding(vocab, embed)
        with torch.no_grad():
            self.img_proj.weight.copy_(torch.randn(embed, feat_dim, generator=g) / feat_dim ** 0.5)
            self.img_proj.bias.zero_()
            self.tok_emb.weight.copy_(torch.randn(vocab, embed, generator=g))

    def forward(self, example_imgs, example_txts):
        img_feat, img_len = example_imgs
        tok_ids, cap_len = example_txts
        i_emb = self.img_proj(img_feat)[:, :max(img_len)]             # slice to the batch maximum (:174-175)
        c_emb = self.tok_emb(tok_ids)[:, :max(cap_len)]
        img_glob = l2norm(i_emb.mean(1))
        cap_glob = l2norm(c_emb.mean(1))
        i_set = nn.functional.normalize(i_emb, p=2, dim=2).permute(1, 0, 2)     # (R,B,D)
        c_seq = nn.functional.normalize(c_emb, p=2, dim=2).permute(1, 0, 2)     # (T,B,D)
        return img_glob, cap_glob, i_set, c_seq, list(img_len), list(cap_len), 0


class StandInBackbone(nn.Module):
    """Random-init substitute with the call surface of `ImageBertForSequenceClassification.bert`
    (oscar/modeling/modeling_bert.py:150-279): word embeddings for the token ids, a linear map of the
    2054-wide region features appended after them, LayerNorm; returns (sequence_output,).  It exists so that
    the head and ALADModel.forward can be driven end to end at the shipped shapes without the VinVL checkpoint
    -- it is NOT a model of the backbone."""

    def __init__(self, hidden=768, feat_dim=2054, vocab=30522, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.word = nn.Embedding(vocab, hidden)
        self.img = nn.Linear(feat_dim, hidden)
        self.norm = nn.LayerNorm(hidden)
        with torch.no_grad():
            self.word.weight.copy_(torch.randn(vocab, hidden, generator=g))
            self.img.weight.copy_(torch.randn(hidden, feat_dim, generator=g) / feat_dim ** 0.5)
            self.img.bias.zero_()
        self.bert = self._bert

    def _bert(self, input_ids, attention_mask=None, token_type_ids=None, img_feats=None):
        x = self.word(input_ids)
        if img_feats is not None:
            x = torch.cat([x, self.img(img_feats)], dim=1)
        return (self.norm(x),)
