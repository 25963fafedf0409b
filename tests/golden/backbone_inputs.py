"""Inputs of tests/golden/backbone_bertimg.npz (shared by the generator, make_golden.gen_backbone, and the tests)."""
import numpy as np

from aladin_amd import synth


def backbone_inputs(seed, B=3, n_tok=12, n_reg=7, vocab=120, feat=22):
    """(input_ids, attention_mask for text only, attention_mask for text + regions, token_type_ids, img_feats): ragged
    captions padded with id 0, ragged region counts (dataset.py:212-238 layout: tokens first, regions appended)."""
    ids = synth.integers((B, n_tok), 1, vocab - 1, seed).astype(np.int64)
    cap_len = synth.integers((B,), 5, n_tok, seed + 1)
    reg_len = synth.integers((B,), 3, n_reg, seed + 2)
    cap_len[0], reg_len[1] = n_tok, n_reg
    tmask = (np.arange(n_tok)[None, :] < np.asarray(cap_len)[:, None]).astype(np.int64)
    ids = ids * tmask
    rmask = (np.arange(n_reg)[None, :] < np.asarray(reg_len)[:, None]).astype(np.int64)
    types = np.zeros((B, n_tok), np.int64)
    types[:, n_tok // 2:] = 1
    feats = (synth.normal((B, n_reg, feat), seed + 3) * rmask[:, :, None]).astype(np.float32)
    return ids, tmask, np.concatenate([tmask, rmask], 1), types, feats
