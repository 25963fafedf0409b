"""Matching-head retrieval over the C ABI: the stored similarity matrix, rank kernels, the fused screened retrieval, top-k lists
(reference alad/recall_auxiliary.py:30-56, alad/evaluation.py:196-223,285-309)."""
import ctypes as C

import torch

from . import _lib
from ._ops_common import _RAW_STREAM, _stream, _ptr, _require_gpu, _rows_inner_contig, _LEN_CACHE, lengths_tensor, _ld, _workspace


# ------------------------------------------------------------------------------------------------
# retrieval
# ------------------------------------------------------------------------------------------------
def sim_matrix(img, cap):
    """(n_img, n_cap) = img @ cap.T on the split-fp16 MFMA path (no autograd); replaces
    ims.mm(caps.t()), reference alad/recall_auxiliary.py:30 and alad/evaluation.py:196,285."""
    _require_gpu(img, cap)
    lib = _lib.load()
    img = img if img.stride(1) == 1 else img.contiguous()
    cap = cap if cap.stride(1) == 1 else cap.contiguous()
    n_img, D = img.shape
    n_cap = cap.shape[0]
    sim = torch.empty((n_img, n_cap), dtype=torch.float32, device=img.device)
    ws = _workspace(lib.aladin_sim_workspace_bytes(n_img, n_cap, D), img.device)
    _lib.check(lib.aladin_sim_matrix(_ptr(img), _ld(img), _ptr(cap), _ld(cap), n_img, n_cap, D, _ptr(sim),
                                     _ld(sim), _ptr(ws), _stream()), 'sim_matrix')
    return sim


def recall_ranks(sim, caps_per_img=5):
    """(rank_i2t, top1_i2t, rank_t2i, top1_t2i) int32 device tensors from a (n_img, 5*n_img) score
    matrix; replaces the argsort/where loops of reference alad/recall_auxiliary.py:34-56."""
    _require_gpu(sim)
    lib = _lib.load()
    sim = sim if sim.stride(1) == 1 else sim.contiguous()
    n_img, n_cap = sim.shape
    dev = sim.device
    r_i2t = torch.empty(n_img, dtype=torch.int32, device=dev)
    t_i2t = torch.empty(n_img, dtype=torch.int32, device=dev)
    r_t2i = torch.empty(n_cap, dtype=torch.int32, device=dev)
    t_t2i = torch.empty(n_cap, dtype=torch.int32, device=dev)
    ws = _workspace(lib.aladin_recall_workspace_bytes(n_cap), dev)
    _lib.check(lib.aladin_recall_ranks(_ptr(sim), _ld(sim), n_img, n_cap, caps_per_img, _ptr(r_i2t), _ptr(t_i2t),
                                       _ptr(r_t2i), _ptr(t_t2i), _ptr(ws), _stream()), 'recall_ranks')
    return r_i2t, t_i2t, r_t2i, t_t2i


def retrieval_ranks(img, cap, caps_per_img=5, exact=False, return_stats=False):
    """(rank_i2t, top1_i2t, rank_t2i, top1_t2i) straight from the (n_img, D) / (n_cap, D) embeddings:
    sim_matrix + recall_ranks fused, the (n_img, n_cap) score matrix is never written.  Same bits as
    the two-step path; replaces reference alad/recall_auxiliary.py:30-56 in one pass.
    The kernel screens with the hi.hi third of the split product and continues to the exact score only the pairs a
    rigorous per-pair bound leaves undecided (include/aladin_hip.h); exact=True forces the three-product path on
    every tile (same outputs).  return_stats=True appends {'exact_tiles', 'listed_pairs', 'rescored_pairs', 'skipped_tiles', 'tiles'}
    (one D2H copy): tiles continued in place, pairs listed, listed pairs whose chains were continued, tiles that skipped the screen.
    The four outputs are deterministic; these statistics (and the call's duration) are NOT -- a tile decides whether to skip its
    analysis from what earlier tiles of the same launch have reported so far, which depends on scheduling.  Only
    rescored_pairs <= listed_pairs and skipped_tiles <= exact_tiles <= tiles hold run to run."""
    _require_gpu(img, cap)
    if img.dim() != 2 or cap.dim() != 2 or img.shape[1] != cap.shape[1]:
        raise ValueError('aladin_amd: (n_img,D) and (n_cap,D) embeddings expected')
    lib = _lib.load()
    img = img if img.stride(1) == 1 else img.contiguous()
    cap = cap if cap.stride(1) == 1 else cap.contiguous()
    n_img, n_cap, D = img.shape[0], cap.shape[0], img.shape[1]
    dev = img.device
    r_i2t = torch.empty(n_img, dtype=torch.int32, device=dev)
    t_i2t = torch.empty(n_img, dtype=torch.int32, device=dev)
    r_t2i = torch.empty(n_cap, dtype=torch.int32, device=dev)
    t_t2i = torch.empty(n_cap, dtype=torch.int32, device=dev)
    ws = _workspace(lib.aladin_retrieval_workspace_bytes(n_img, n_cap, D), dev)
    fn = lib.aladin_retrieval_ranks_exact if exact else lib.aladin_retrieval_ranks
    _lib.check(fn(_ptr(img), img.stride(0), _ptr(cap), cap.stride(0), n_img, n_cap, D, caps_per_img,
                  _ptr(r_i2t), _ptr(t_i2t), _ptr(r_t2i), _ptr(t_t2i), _ptr(ws), _stream()), 'retrieval_ranks')
    if return_stats:
        off = lib.aladin_retrieval_stats_offset(n_img, n_cap, D)
        st = ws[off:off + 36].view(torch.int32).cpu().tolist()
        tiles = -(-n_img // 256) * -(-n_cap // 384)
        return r_i2t, t_i2t, r_t2i, t_t2i, {'exact_tiles': st[0], 'listed_pairs': st[1], 'rescored_pairs': st[5], 'skipped_tiles': st[8],
                                            'tiles': tiles}
    return r_i2t, t_i2t, r_t2i, t_t2i


def topk_indices(scores, k, dim=1):
    """(n_q, k) int32 indices of each query's k best candidates, best first, ties -> lower index; queries are
    the rows of `scores` (dim=1) or its columns (dim=0, read in place: no transpose).  Replaces the
    `inds[i][0:50]` slices of the descending argsorts in reference alad/evaluation.py:303-309 (-1 past the
    number of candidates)."""
    _require_gpu(scores)
    if scores.dim() != 2 or dim not in (0, 1):
        raise ValueError('aladin_amd: topk_indices expects a 2-D score matrix and dim 0 or 1')
    sc = scores if scores.stride(1) == 1 else scores.contiguous()
    n_q, n_c = (sc.shape[0], sc.shape[1]) if dim == 1 else (sc.shape[1], sc.shape[0])
    q_stride, c_stride = (_ld(sc), 1) if dim == 1 else (1, _ld(sc))
    out = torch.empty((n_q, int(k)), dtype=torch.int32, device=sc.device)
    _lib.check(_lib.load().aladin_topk(_ptr(sc), q_stride, c_stride, n_q, n_c, int(k), _ptr(out), _ptr(None), _stream()),
               'topk')
    return out
