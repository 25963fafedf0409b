"""Retrieval evaluation with the reference's call surface, scored and ranked on the GPU.

    compute_sim_matrix                       new name for the sites
                                             recall_auxiliary.py:30,51, evaluation.py:196,285 and
                                             the alignment_sim_fn closures train.py:495-498 / test.py:260-263
    recall / recall_test / compute_recall    <- alad/recall_auxiliary.py:8-149
    i2t / t2i                                <- alad/evaluation.py:158-327
    AverageMeter / LogCollector              <- alad/evaluation.py:22-77 (host bookkeeping)

The per-query Python loops, per-iteration host->device copies and numpy argsort of the reference
are replaced by one score-matrix launch plus one rank launch per direction; only the rank vectors
come back to the host.  Rank = number of strictly larger scores, which equals the reference's
argsort position except on exact ties (whose order numpy leaves unspecified).
"""
from collections import OrderedDict

import numpy as np
import torch

from . import ops

CAPS_PER_IMG = 5


class AverageMeter:
    """Last value and running weighted mean of one logged quantity (same fields and printing as the
    meter of reference alad/evaluation.py:22-47: val, avg, sum, count; weight 0 records without averaging)."""

    __slots__ = ('val', 'avg', 'sum', 'count')

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=0):
        self.val = val
        self.count += n
        self.sum += n * val
        self.avg = self.sum / (self.count + 1e-4)          # the reference's guard against an empty meter

    def __str__(self):
        return str(self.val) if not self.count else '{:.4f} ({:.4f})'.format(self.val, self.avg)


class LogCollector:
    """Named AverageMeters in insertion order (reference alad/evaluation.py:50-77), incl. the
    TensorBoard hook `tb_log` the training driver calls."""

    def __init__(self):
        self.meters = OrderedDict()

    def update(self, k, v, n=0):
        self.meters.setdefault(k, AverageMeter()).update(v, n)

    def __str__(self):
        return '  '.join('%s %s' % (name, meter) for name, meter in self.meters.items())

    def tb_log(self, tb_logger, prefix='', step=None):
        for name, meter in self.meters.items():
            tb_logger.add_scalar(prefix + name, meter.val, global_step=step)


def encode_data(model, data_loader, log_step=10, logging=print, max_len=71):
    """reference alad/evaluation.py:80-155 with the embedding store kept ON THE DEVICE.

    Same protocol (model.forward_emb per batch under no_grad, sets zero-padded to `max_len` positions,
    slot 0 overwritten with the global matching-head embedding, length lists) and same return
    signature, but `img_embs` / `cap_embs` are float32 CUDA tensors: the reference's per-batch
    device->host copy (:124-128) and the per-query host->device copies of i2t / t2i
    (:179,202,267,291) disappear, and compute_sim_matrix() consumes the buffers in place."""
    import time
    clear_eval_cache()                               # a new embedding store is coming: the last grid is stale
    batch_time = AverageMeter()
    val_logger = LogCollector()
    model.eval()
    end = time.time()
    img_embs = cap_embs = None
    img_lengths, cap_lengths = [], []
    ids_pointer = 0
    n_total = len(data_loader.dataset)
    for i, (example_imgs, example_txts) in enumerate(data_loader):
        model.logger = val_logger
        with torch.no_grad():
            img_glob, cap_glob, img_emb, cap_emb, img_length, cap_length, _ = model.forward_emb(example_imgs, example_txts)
            bs = img_glob.shape[0]
            if img_embs is None:
                dev = img_emb.device
                img_embs = torch.zeros((n_total, max_len, img_emb.size(2)), dtype=torch.float32, device=dev)
                cap_embs = torch.zeros((n_total, max_len, cap_emb.size(2)), dtype=torch.float32, device=dev)
            sl = slice(ids_pointer, ids_pointer + bs)
            img_embs[sl, :img_emb.size(0), :] = img_emb.permute(1, 0, 2)          # (S,B,D) -> (B,S,D)
            cap_embs[sl, :cap_emb.size(0), :] = cap_emb.permute(1, 0, 2)
            img_embs[sl, 0, :] = img_glob                                         # I-CLS / T-CLS slots (:127-128)
            cap_embs[sl, 0, :] = cap_glob
            img_lengths.extend(img_length)
            cap_lengths.extend(cap_length)
            ids_pointer += bs
        batch_time.update(time.time() - end)
        end = time.time()
        if logging is not None and i % log_step == 0:
            logging('Test: [{0}/{1}]\t{e_log}\tTime {bt.val:.3f} ({bt.avg:.3f})\t'.format(
                i, len(data_loader), bt=batch_time, e_log=str(model.logger)))
    return img_embs, cap_embs, img_lengths, cap_lengths


def encode_data_packed(model, data_loader, log_step=10, logging=print, precision='split', max_len=71):
    """encode_data with the embedding store kept on the device in packed 16-bit form (SURVEY.md
    section 8(f) row 2): returns (img_store, cap_store, img_lengths, cap_lengths) where the stores are
    aladin_amd.store.PackedSetStore objects that compute_sim_matrix / i2t / t2i accept in place of the
    (N, 71, D) tensors -- same protocol as alad/evaluation.py:80-155 otherwise.  precision 'split' (default:
    rank-exact alignment retrieval, fp16 hi/lo pair per value) or 'fp16' (half the bytes, ~1e-4 score error)."""
    import time
    from .store import PackedSetStore
    clear_eval_cache()
    batch_time = AverageMeter()
    val_logger = LogCollector()
    model.eval()
    end = time.time()
    img_store = cap_store = None
    for i, (example_imgs, example_txts) in enumerate(data_loader):
        model.logger = val_logger
        with torch.no_grad():
            img_glob, cap_glob, img_emb, cap_emb, img_length, cap_length, _ = model.forward_emb(example_imgs, example_txts)
            if img_store is None:
                img_store = PackedSetStore(img_emb.size(2), 0, img_emb.device, precision=precision, padded_len=max_len)
                cap_store = PackedSetStore(cap_emb.size(2), 2, cap_emb.device, precision=precision, padded_len=max_len)
            img_store.append(img_emb.permute(1, 0, 2), img_length, img_glob)      # (S,B,D) -> (B,S,D) view
            cap_store.append(cap_emb.permute(1, 0, 2), cap_length, cap_glob)
        batch_time.update(time.time() - end)
        end = time.time()
        if logging is not None and i % log_step == 0:
            logging('Test: [{0}/{1}]\t{e_log}\tTime {bt.val:.3f} ({bt.avg:.3f})\t'.format(
                i, len(data_loader), bt=batch_time, e_log=str(model.logger)))
    return img_store, cap_store, list(img_store.lengths), list(cap_store.lengths)


def _is_store(x):
    from .store import PackedSetStore, StoreView
    return isinstance(x, (PackedSetStore, StoreView))


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('aladin_amd: retrieval scoring runs in HIP kernels on an MI355X only (no GPU visible)')
    return torch.device('cuda', torch.cuda.current_device())


def compute_sim_matrix(img, cap, img_len=None, cap_len=None, mode='matching', precision=None):
    """(N_img, N_cap) score matrix on the GPU.

    mode='matching'  : img (N_img, D), cap (N_cap, D) global embeddings -> img @ cap.T
    mode='alignment' : img (N_img, R, D), cap (N_cap, T, D) sets with length lists -> MrSw scores
                       (trimmed to the real lengths, chunked, in the evaluation precision: ops._scores_nograd;
                       precision='fp16' | 'split' overrides ops.set_eval_precision())
    Both arguments may instead be PackedSetStore / StoreView objects (encode_data_packed).
    """
    dev = _device()
    if _is_store(img) != _is_store(cap):
        raise ValueError('compute_sim_matrix: pass two stores or two tensors')
    if _is_store(img):                            # packed 16-bit stores (encode_data_packed): lengths travel with them
        from .store import alignment_scores_from_stores
        with torch.no_grad():
            if mode == 'matching':
                return ops.sim_matrix(img.glob, cap.glob)
            if mode == 'alignment':
                return alignment_scores_from_stores(img, cap)
        raise ValueError("mode must be 'matching' or 'alignment'")
    img = torch.as_tensor(img).to(dev, torch.float32)
    cap = torch.as_tensor(cap).to(dev, torch.float32)
    with torch.no_grad():
        if mode == 'matching':
            return ops.sim_matrix(img, cap)
        if mode == 'alignment':
            if img_len is None or cap_len is None:
                raise ValueError("compute_sim_matrix(mode='alignment') needs img_len and cap_len")
            return ops.alignment_scores(img, cap, list(img_len), list(cap_len), 'MrSw', precision=precision)
    raise ValueError("mode must be 'matching' or 'alignment'")


def _metrics(ranks):
    ranks = np.asarray(ranks, dtype=np.float64)
    r1 = 100.0 * len(np.where(ranks < 1)[0]) / len(ranks)
    r5 = 100.0 * len(np.where(ranks < 5)[0]) / len(ranks)
    r10 = 100.0 * len(np.where(ranks < 10)[0]) / len(ranks)
    medr = np.floor(np.median(ranks)) + 1
    meanr = ranks.mean() + 1
    return r1, r5, r10, medr, meanr


def _ranks(sim):
    r_i2t, t_i2t, r_t2i, t_t2i = ops.recall_ranks(sim, CAPS_PER_IMG)
    both = torch.cat([r_i2t, t_i2t, r_t2i, t_t2i]).cpu().numpy().astype(np.float64)      # one D2H copy
    n_img, n_cap = sim.shape
    return both[:n_img], both[n_img:2 * n_img], both[2 * n_img:2 * n_img + n_cap], both[2 * n_img + n_cap:]


def _fused_ranks(img, cap):
    """Ranks of both directions straight from the global embeddings (ops.retrieval_ranks: the score
    matrix is never materialised); same numbers as _ranks(compute_sim_matrix(img, cap))."""
    dev = _device()
    img = torch.as_tensor(img).to(dev, torch.float32)
    cap = torch.as_tensor(cap).to(dev, torch.float32)
    n_img, n_cap = img.shape[0], cap.shape[0]
    with torch.no_grad():
        r_i2t, t_i2t, r_t2i, t_t2i = ops.retrieval_ranks(img, cap, CAPS_PER_IMG)
    both = torch.cat([r_i2t, t_i2t, r_t2i, t_t2i]).cpu().numpy().astype(np.float64)      # one D2H copy
    return both[:n_img], both[n_img:2 * n_img], both[2 * n_img:2 * n_img + n_cap], both[2 * n_img + n_cap:]


def recall(images, captions, model, mode='i2t', lenghts=None, return_ranks=False):
    """reference alad/recall_auxiliary.py:8-69: rows 0::5 of `images` are the distinct images."""
    if mode not in ('i2t', 't2i'):
        raise ValueError('mode not correct')
    r_i2t, t_i2t, r_t2i, t_t2i = _fused_ranks(torch.as_tensor(images)[0::CAPS_PER_IMG], captions)
    ranks, top1 = (r_i2t, t_i2t) if mode == 'i2t' else (r_t2i, t_t2i)
    m = _metrics(ranks)
    return (m, (ranks, top1)) if return_ranks else m


def recall_test(img_embs, cap_embs, tot_lengths, model):
    """reference alad/recall_auxiliary.py:72-86; both directions from one fused GEMM + rank pass."""
    r_i2t, _, r_t2i, _ = _fused_ranks(torch.as_tensor(img_embs)[0::CAPS_PER_IMG], cap_embs)
    r1, r5, r10, _, _ = _metrics(r_i2t)
    r1i, r5i, r10i, _, _ = _metrics(r_t2i)
    return r1, r5, r10, r1i, r5i, r10i, r1 + r5 + r10 + r1i + r5i + r10i


def compute_recall(img_embs, cap_embs, tot_lengths=None, model=None, verbose=True):
    """reference alad/recall_auxiliary.py:133-149."""
    r1, r5, r10, r1i, r5i, r10i, _ = recall_test(img_embs, cap_embs, tot_lengths, model)
    rsum = r1 + r5 + r10 + r1i + r5i + r10i
    if verbose:
        print("Recall Image to text: %.2f, %.2f, %.2f" % (r1, r5, r10))
        print("Recall Text to image: %.2f, %.2f, %.2f" % (r1i, r5i, r10i))
        print('Sum score: %.2f' % rsum)
    return r1, r5, r10, r1i, r5i, r10i, rsum


def recall_1k_5fold_test(img_embs, cap_embs, tot_lengths=None, model=None, verbose=True):
    """reference alad/recall_auxiliary.py:90-130: mean over five 5000-row folds (torch.split(.., 5000))."""
    img_folds = torch.split(torch.as_tensor(img_embs), 5000, dim=0)
    cap_folds = torch.split(torch.as_tensor(cap_embs), 5000, dim=0)
    res = []
    for i in range(5):
        if verbose:
            print('Computing Test recall... chunk %s of 5' % (i + 1))
        res.append(recall_test(img_folds[i], cap_folds[i], tot_lengths, model)[:6])
    r1, r5, r10, r1i, r5i, r10i = (float(v) for v in np.mean(np.array(res, dtype=np.float64), axis=0))
    rsum = r1 + r5 + r10 + r1i + r5i + r10i
    if verbose:
        print("Test 1K 5Folds - Recall Image to text: %.2f, %.2f, %.2f" % (r1, r5, r10))
        print("Test 1K 5Folds - Recall Text to image: %.2f, %.2f, %.2f" % (r1i, r5i, r10i))
        print('Test 1K 5Folds - Sum score: %.2f' % rsum)
    return r1, r5, r10, r1i, r5i, r10i, rsum


# The score grid of the last i2t / t2i call: validate() / test() call the two back to back on the same
# embeddings (reference train.py:504-509, test.py:271-276), so the second call re-uses the first one's grid.
# A hit needs the SAME input objects, still alive (weak references compared with `is`: an address alone says
# nothing -- the caching allocator hands the next validation's buffers the block the previous ones freed, with
# the same shape and the same number of in-place fills), unchanged (tensor version counter / store row count).
# The memo never keeps its inputs alive, and encode_data / encode_data_packed drop it before they start.
_GRID_MEMO = {'key': None, 'refs': None, 'sim': None}


def clear_eval_cache():
    """Drop the memoised score grid (it holds N_img x N_cap floats on the device: 500 MB at COCO-5k)."""
    _GRID_MEMO['key'] = _GRID_MEMO['refs'] = _GRID_MEMO['sim'] = None


def _memo_key(images, captions, img_lenghts, cap_lenghts, measure, sim_function):
    """-> (key, objects the key is only valid for) or (None, None) when an input cannot be tracked (then nothing
    is memoised)."""
    import weakref
    objs = []

    def ident(x):
        if isinstance(x, torch.Tensor):
            objs.append(x)
            return ('t', x.data_ptr(), tuple(x.shape), tuple(x.stride()), x._version, str(x.device))
        if _is_store(x):
            st = getattr(x, 'store', x)
            objs.extend([x, st])
            return ('s', st.n_rows, len(st), hash(tuple(getattr(x, 'ids', ()))), len(x))
        if isinstance(x, np.ndarray):
            return None                              # no version counter: an in-place refill would go unnoticed
        return None
    ki, kc = ident(images), ident(captions)
    if ki is None or kc is None:
        return None, None
    if sim_function is None or isinstance(sim_function, str):
        fn = sim_function
    else:
        objs.append(sim_function)
        fn = 'callable'
    try:
        refs = [weakref.ref(o) for o in objs]
    except TypeError:                                # e.g. a bound method: a new object per access, not trackable
        return None, None
    key = (ki, kc, tuple(int(v) for v in img_lenghts) if img_lenghts is not None else None,
           tuple(int(v) for v in cap_lenghts) if cap_lenghts is not None else None, measure, fn, ops._EVAL_PRECISION[0])
    return key, (refs, objs)


def _eval_scores(images, captions, img_lenghts, cap_lenghts, measure, sim_function):
    """(n_img, n_cap) scores of the de-duplicated images (rows 0::5, alad/evaluation.py:171,252) against every
    caption -- ONE grid instead of the reference's per-query loops."""
    key, tracked = _memo_key(images, captions, img_lenghts, cap_lenghts, measure, sim_function)
    if key is not None and _GRID_MEMO['key'] == key and _GRID_MEMO['refs'] is not None \
            and len(_GRID_MEMO['refs']) == len(tracked[1]) \
            and all(r() is o for r, o in zip(_GRID_MEMO['refs'], tracked[1])):
        return _GRID_MEMO['sim']
    clear_eval_cache()                               # release the old grid before the new one is allocated
    sim = _eval_scores_uncached(images, captions, img_lenghts, cap_lenghts, measure, sim_function)
    if key is not None:
        _GRID_MEMO['key'], _GRID_MEMO['refs'], _GRID_MEMO['sim'] = key, tracked[0], sim
    return sim


def _eval_scores_uncached(images, captions, img_lenghts, cap_lenghts, measure, sim_function):
    if _is_store(images):
        ims = images.view(slice(0, None, CAPS_PER_IMG))
        if measure == 'order':
            with torch.no_grad():
                return ops.order_scores(ims.glob, captions.glob)
        if sim_function is None:
            return compute_sim_matrix(ims, captions)
        if sim_function == 'alignment':
            return compute_sim_matrix(ims, captions, mode='alignment')
        raise ValueError("aladin_amd: with packed stores sim_function must be None or 'alignment'")
    images = torch.as_tensor(images)
    captions = torch.as_tensor(captions)
    ims = images[0::CAPS_PER_IMG]
    if measure == 'order':
        # alad/evaluation.py:184-192,272-281 score order embeddings with order_sim (alad/loss.py:20-26) in blocks
        # of 100 queries.  That branch predates the (N, 71, D) sets -- order_sim's expand() takes (n, D) matrices
        # and fails on 3-D input -- so here it scores the slot-0 global embeddings, which is what the same call
        # sites use for 'dot' (:196, :285); 2-D embedding matrices are taken as they are.
        dev = _device()
        a = (ims[:, 0, :] if ims.dim() == 3 else ims).to(dev, torch.float32)
        b = (captions[:, 0, :] if captions.dim() == 3 else captions).to(dev, torch.float32)
        with torch.no_grad():
            return ops.order_scores(a, b)
    if sim_function is None:                      # matching head on the slot-0 global embeddings (:196, :285)
        return compute_sim_matrix(ims[:, 0, :], captions[:, 0, :])
    ims_len = list(img_lenghts[0::CAPS_PER_IMG])
    if sim_function == 'alignment':
        return compute_sim_matrix(ims, captions, ims_len, list(cap_lenghts), mode='alignment')
    # a reference-style closure (train.py:495-498: AlignmentContrastiveLoss(...)(..., return_loss=False) under
    # no_grad): called ONCE on the whole grid; with aladin_amd's loss module inside, that call lands on the same
    # trimmed / chunked / split-precision path as sim_function='alignment'
    dev = _device()
    with torch.no_grad():
        return sim_function(ims.to(dev, torch.float32), captions.to(dev, torch.float32), ims_len,
                            list(cap_lenghts)).to(torch.float32)


def _npts(images, npts):
    n_img = len(images) // CAPS_PER_IMG
    return n_img if npts is None else min(int(npts), n_img)


def i2t(images, captions, img_lenghts, cap_lenghts, npts=None, return_ranks=False, ndcg_scorer=None, fold_index=0,
        measure='dot', sim_function=None, cap_batches=1):
    """reference alad/evaluation.py:158-241.  sim_function may be None (matching head), the string
    'alignment' (HIP alignment scores) or a callable (img, cap, img_len, cap_len) -> scores; it is
    called ONCE on the whole (n_img x n_cap) grid instead of once per query and caption chunk
    (cap_batches is accepted and ignored).  npts: only the first npts images are queries (:164-165); they are
    still ranked against every caption.  Returns the 7-tuple, plus (ranks, top1) with return_ranks."""
    if ndcg_scorer is not None:
        raise NotImplementedError('aladin_amd: ndcg_scorer is out of scope (None at every reference call site)')
    sim = _eval_scores(images, captions, img_lenghts, cap_lenghts, measure, sim_function)
    ranks, top1, _, _ = _ranks(sim)
    n = _npts(images, npts)
    ranks, top1 = ranks[:n], top1[:n]
    m = _metrics(ranks) + (0, 0)
    return (m, (ranks, top1)) if return_ranks else m


def t2i(images, captions, img_lenghts, cap_lenghts, npts=None, return_ranks=False, ndcg_scorer=None, fold_index=0,
        measure='dot', sim_function=None, im_batches=1):
    """reference alad/evaluation.py:244-327.  With return_ranks the second element is (ranks, top50) as in the
    reference (:262,309,324-325): top50[c] = the 50 best images of caption c, best first (HIP top-k kernel; -1
    where there are fewer than 50 images).  npts: only the captions of the first npts images are queries
    (:250-251), ranked against every image."""
    if ndcg_scorer is not None:
        raise NotImplementedError('aladin_amd: ndcg_scorer is out of scope (None at every reference call site)')
    sim = _eval_scores(images, captions, img_lenghts, cap_lenghts, measure, sim_function)
    _, _, ranks, _ = _ranks(sim)
    n = CAPS_PER_IMG * _npts(images, npts)
    ranks = ranks[:n]
    m = _metrics(ranks) + (0, 0)
    if not return_ranks:
        return m
    top50 = ops.topk_indices(sim[:, :n], 50, dim=0).cpu().numpy().astype(np.float64)     # float table like numpy.zeros((5*npts, 50))
    return m, (ranks, top50)
