"""Generate tests/golden/*.npz by running the REFERENCE (mesnico/ALADIN) on synthetic inputs.

Runs ONLY in the build container (it imports /root/reference read-only; the reference does not
exist on the GPU box).  The reference holds no tests or fixtures of its own, so these files are
what pins the oracle and the HIP path.  Each .npz stores the generator arguments of its inputs
(see aladin_amd/synth.py), a checksum of the inputs, and the reference's outputs -- data only.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
warnings.filterwarnings('ignore')

import numpy as np
import torch
import yaml

from aladin_amd import synth

import alad.loss as ref_loss                      # noqa: E402
import alad.recall_auxiliary as ref_recall        # noqa: E402
import alad.evaluation as ref_eval                # noqa: E402

torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def save(name, **kw):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **kw)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------ alignment cases
ALIGN_CASES = [
    # name, kind, B, Bc, R, T, D, seed, ragged, margin
    ('align_tiny',      'random',     2,  2,  3,  5,   8,  11, False, 0.2),
    ('align_b5_d64',    'random',     5,  5, 34, 50,  64,  12, True,  0.2),
    ('align_b12_struct', 'structured', 12, 12, 20, 30,  64,  13, True,  0.2),
    ('align_b32_d64',   'random',    32, 32, 34, 50,  64,  14, True,  0.3),
    ('align_b16_d768',  'random',    16, 16, 34, 50, 768, 1234, False, 0.2),   # BASELINE config 1
    ('align_b8_d768_rag', 'structured4', 8, 8, 34, 50, 768,  15, True,  0.2),
    ('align_rect',      'random',     3,  7, 34, 50,  64,  16, True,  0.2),
    ('align_r33',       'random',     6,  6, 33, 34, 128,  17, True,  0.2),    # R'=32 exactly
    # round 4: the tile classes added for the shipped data shape and short captions
    ('align_vinvl_b12', 'structured3', 12, 12, 51, 38, 768, 21, True,  0.2),   # 50 regions + 35 tokens: 48 rows + 2 side rows, 40-word class
    ('align_t27_b10',   'random',    10, 10, 36, 27, 128,  22, True,  0.2),    # 32 rows + 3 side rows, 24-word caption class
    ('align_t11_rect',  'random',     6, 20, 34, 11,  64,  23, True,  0.2),    # 8-word caption class, rectangular
]
AGG_MODES = ['MrSw', 'MrAVGw', 'MwSr', 'symm', 'sum', 'mean']


def make_inputs(kind, B, Bc, R, T, D, seed, ragged):
    if kind.startswith('structured'):
        noise = float(kind[len('structured'):] or 1.0)
        return synth.structured_alignment_batch(B, R, T, D, seed, noise=noise, ragged=ragged)
    return synth.alignment_batch(B, R, T, D, seed, ragged, Bc=Bc)


def gen_alignment():
    only = [n for n in os.environ.get('GOLDEN_ONLY', '').split(',') if n]      # GOLDEN_ONLY=name,name: add cases without rewriting the others
    for name, kind, B, Bc, R, T, D, seed, ragged, margin in ALIGN_CASES:
        if only and name not in only:
            continue
        im, s, im_len, s_len = make_inputs(kind, B, Bc, R, T, D, seed, ragged)
        out = dict(kind=kind, B=B, Bc=Bc, R=R, T=T, D=D, seed=seed, ragged=ragged, margin=margin,
                   im_len=np.array(im_len), s_len=np.array(s_len),
                   im_checksum=synth.checksum(im), s_checksum=synth.checksum(s))
        for mode in AGG_MODES:
            crit = ref_loss.AlignmentContrastiveLoss(margin=margin, measure='dot',
                                                     max_violation=True, aggregation=mode)
            with torch.no_grad():
                S = crit(t(im), t(s), im_len, s_len, return_loss=False, return_similarity_mat=True)
            out['S_' + mode] = S.numpy()
        # 'scan-sentences' (alad/loss.py:136-149): scores for every case; loss + gradients only where the
        # reference's autograd is finite, i.e. full-length batches (its -inf rows give 0 * NaN otherwise)
        crit = ref_loss.AlignmentContrastiveLoss(margin=margin, measure='dot', max_violation=False,
                                                 aggregation='scan-sentences')
        a = t(im).requires_grad_(True)
        b = t(s).requires_grad_(True)
        S = crit(a, b, im_len, s_len, return_loss=False, return_similarity_mat=True)
        out['S_scan-sentences'] = S.detach().numpy()
        if not ragged and B == Bc:
            w = synth.normal((B, Bc), seed + 555)
            (S * t(w)).sum().backward()
            assert torch.isfinite(a.grad).all() and torch.isfinite(b.grad).all()
            sc_stride = 16 if D >= 512 else 1
            out['scan_w'] = w
            out['scan_stride'] = sc_stride
            out['dim_scan'] = a.grad.numpy()[:, :, ::sc_stride]
            out['ds_scan'] = b.grad.numpy()[:, :, ::sc_stride]
            a2 = t(im).requires_grad_(True)
            b2 = t(s).requires_grad_(True)
            out['loss_scan_sum'] = crit(a2, b2, im_len, s_len).item()
        if B == Bc:
            stride = 16 if D >= 512 else 1
            out['grad_stride'] = stride
            for mv in (True, False):
                crit = ref_loss.AlignmentContrastiveLoss(margin=margin, measure='dot',
                                                         max_violation=mv, aggregation='MrSw')
                a = t(im).requires_grad_(True)
                b = t(s).requires_grad_(True)
                loss, S = crit(a, b, im_len, s_len, return_similarity_mat=True)
                loss.backward()
                tag = 'mv' if mv else 'sum'
                out['loss_' + tag] = loss.item()
                out['dim_' + tag] = a.grad.numpy()[:, :, ::stride]
                out['ds_' + tag] = b.grad.numpy()[:, :, ::stride]
                out['dim_cs_' + tag] = synth.checksum(a.grad.numpy())
                out['ds_cs_' + tag] = synth.checksum(b.grad.numpy())
                out['dim_abs_' + tag] = float(a.grad.abs().sum())
                out['ds_abs_' + tag] = float(b.grad.abs().sum())
                # dloss/dS, through a leaf copy of S
                Sl = S.detach().clone().requires_grad_(True)
                crit.compute_contrastive_loss(Sl).backward()
                out['dS_' + tag] = Sl.grad.numpy()
        save(name, **out)


# ------------------------------------------------------------------------- evaluation-shape case
def gen_eval():
    n_img, D, seed = 50, 64, 31
    images, captions, img_len, cap_len = synth.eval_sets(n_img, D, seed)
    out = dict(n_img=n_img, D=D, seed=seed, images_checksum=synth.checksum(images),
               captions_checksum=synth.checksum(captions),
               img_len=np.array(img_len), cap_len=np.array(cap_len))
    crit = ref_loss.AlignmentContrastiveLoss(aggregation='MrSw')
    with torch.no_grad():
        S = crit(t(images[0::5]), t(captions), img_len[0::5], cap_len, return_loss=False,
                 return_similarity_mat=True)
    out['S_eval'] = S.numpy()

    def sim_fn(img, cap, il, cl):
        with torch.no_grad():
            return crit(img, cap, il, cl, return_loss=False, return_similarity_mat=True)

    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # evaluation.py:179,202 call .cuda()
    try:
        for tag, fn in (('match', None), ('align', sim_fn)):
            m, (ranks, top1) = ref_eval.i2t(t(images), t(captions), img_len, cap_len,
                                            return_ranks=True, sim_function=fn, cap_batches=5)
            out['i2t_%s_metrics' % tag] = np.array(m, dtype=np.float64)
            out['i2t_%s_ranks' % tag] = ranks
            out['i2t_%s_top1' % tag] = top1
            m, (ranks, top50) = ref_eval.t2i(t(images), t(captions), img_len, cap_len,
                                             return_ranks=True, sim_function=fn, im_batches=5)
            out['t2i_%s_metrics' % tag] = np.array(m, dtype=np.float64)
            out['t2i_%s_ranks' % tag] = ranks
            out['t2i_%s_top1' % tag] = top50[:, 0]
            out['t2i_%s_top50' % tag] = top50.astype(np.int16)
    finally:
        torch.Tensor.cuda = saved_cuda
    save('eval_sets', **out)


COCO1K = dict(n_img=1000, D=64, seed=71, base_weight=0.36, img_len_range=(20, 60), cap_len_range=(8, 26), n_full=8)
# the same grid at the HEADLINE feature width (north_star's Recall@1 claim is for D=768 features): base_weight chosen so that
# recall stays non-degenerate at this width (the noise cosine shrinks with 1/sqrt(D)); ~15 min per direction on 8 cores
COCO1K_D768 = dict(n_img=1000, D=768, seed=73, base_weight=0.16, img_len_range=(20, 60), cap_len_range=(8, 26), n_full=8)


def gen_eval_coco1k_d768():
    gen_eval_coco1k('eval_coco1k_d768', COCO1K_D768)


def gen_eval_coco1k(name='eval_coco1k', cfg=None):
    """COCO-1k sized alignment-head retrieval (1000 images x 5000 captions, sets padded to 71 positions as
    encode_data leaves them) through the reference's own i2t / t2i loops with its alignment_sim_fn closure
    (alad/evaluation.py:158-327, train.py:493-509: cap_batches=5, im_batches=1).  Besides ranks / top
    lists the file keeps, per query, the smallest gap between a ground-truth score and any competitor in
    the reference's OWN fp32 scores: queries whose gap is at the level of fp32 rounding are the ones
    whose rank the reference itself does not resolve (a different summation order moves them)."""
    import time
    kw = dict(cfg or COCO1K)
    n_img, D, seed = kw.pop('n_img'), kw.pop('D'), kw.pop('seed')
    images, captions, img_len, cap_len = synth.eval_sets(n_img, D, seed, **kw)
    out = dict(n_img=n_img, D=D, seed=seed, images_checksum=synth.checksum(images), captions_checksum=synth.checksum(captions),
               img_len=np.array(img_len, dtype=np.int16), cap_len=np.array(cap_len, dtype=np.int16),
               **{'gen_' + k: np.array(v) for k, v in kw.items()})
    crit = ref_loss.AlignmentContrastiveLoss(aggregation='MrSw')
    N = 5 * n_img
    S_i2t = np.zeros((n_img, N), dtype=np.float32)           # rows as i2t sees them (1 image x N/5 captions per call)
    S_t2i = np.zeros((n_img, N), dtype=np.float32)           # columns as t2i sees them (n_img images x 5 captions per call)
    state = {'mode': None, 'row': 0, 'chunk': 0, 'col': 0}

    def sim_fn(img, cap, il, cl):
        with torch.no_grad():
            sc = crit(img, cap, il, cl, return_loss=False, return_similarity_mat=True)
        if state['mode'] == 'i2t':
            w = sc.shape[1]
            S_i2t[state['row'], state['chunk'] * w:(state['chunk'] + 1) * w] = sc.numpy()[0]
            state['chunk'] += 1
            if state['chunk'] == 5:
                state['chunk'] = 0
                state['row'] += 1
        else:
            S_t2i[:, state['col']:state['col'] + 5] = sc.numpy()
            state['col'] += 5
        return sc

    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        t0 = time.time()
        state['mode'] = 'i2t'
        m, (ranks, top1) = ref_eval.i2t(t(images), t(captions), img_len, cap_len, return_ranks=True, sim_function=sim_fn,
                                        cap_batches=5)
        print('reference i2t: %.0f s' % (time.time() - t0), m)
        out['i2t_metrics'] = np.array(m, dtype=np.float64)
        out['i2t_ranks'] = ranks.astype(np.int16)
        out['i2t_top1'] = top1.astype(np.int16)
        t0 = time.time()
        state['mode'] = 't2i'
        m, (ranks, top50) = ref_eval.t2i(t(images), t(captions), img_len, cap_len, return_ranks=True, sim_function=sim_fn,
                                         im_batches=1)
        print('reference t2i: %.0f s' % (time.time() - t0), m)
        out['t2i_metrics'] = np.array(m, dtype=np.float64)
        out['t2i_ranks'] = ranks.astype(np.int16)
        out['t2i_top10'] = top50[:, :10].astype(np.int16)
    finally:
        torch.Tensor.cuda = saved_cuda
    assert state['row'] == n_img and state['col'] == N
    print('max |S_i2t - S_t2i| of the reference\'s two call shapes: %.3g' % np.abs(S_i2t - S_t2i).max())
    # per-query resolution of the reference's own scores
    cols = np.arange(N)
    # i2t: rank = min over the 5 ground truths of their positions = #(scores above the LARGEST ground truth), so
    # only competitors of that one decide the rank
    gts = S_i2t[np.arange(n_img)[:, None], 5 * np.arange(n_img)[:, None] + np.arange(5)[None, :]]      # (n_img, 5)
    best = gts.argmax(1)
    d = np.abs(S_i2t.astype(np.float64) - gts.max(1)[:, None].astype(np.float64))
    d[np.arange(n_img), 5 * np.arange(n_img) + best] = np.inf
    gap_i2t = d.min(1)
    gt = S_t2i[cols // 5, cols]
    d = np.abs(S_t2i.astype(np.float64) - gt[None, :].astype(np.float64))
    d[cols // 5, cols] = np.inf
    gap_t2i = d.min(0)
    srt = np.sort(S_t2i.astype(np.float64), axis=0)[::-1][:11]               # (11, N)
    out['t2i_top10_gap'] = (srt[:-1] - srt[1:]).min(0).astype(np.float32)
    srt = np.sort(S_i2t.astype(np.float64), axis=1)[:, ::-1][:, :2]
    out['i2t_top1_gap'] = (srt[:, 0] - srt[:, 1]).astype(np.float32)
    out['i2t_gap'] = gap_i2t.astype(np.float32)
    out['t2i_gap'] = gap_t2i.astype(np.float32)
    # a sample of the reference's scores themselves (every 20th image, every 10th caption) for a direct comparison
    out['S_sample'] = S_i2t[0::20, 0::10].copy()
    out['S_diag'] = S_i2t[cols // 5, cols].copy()
    print('gaps below 1e-5: i2t %d, t2i %d; below 1e-6: %d, %d' % ((gap_i2t < 1e-5).sum(), (gap_t2i < 1e-5).sum(),
                                                                    (gap_i2t < 1e-6).sum(), (gap_t2i < 1e-6).sum()))
    save(name, **out)


# --------------------------------------------------------------------- matching / distillation
def gen_matching():
    for name, B, D, seed, noise in (('match_b16_d768', 16, 768, 21, 1.0), ('match_b7_d64', 7, 64, 22, 0.5)):
        img, cap = synth.global_embeddings(B, D, seed, noise)
        out = dict(B=B, D=D, seed=seed, noise=noise, img_checksum=synth.checksum(img))
        for mv in (True, False):
            crit = ref_loss.ContrastiveLoss(margin=0.2, measure='dot', max_violation=mv)
            a = t(img).requires_grad_(True)
            b = t(cap).requires_grad_(True)
            loss, M = crit(a, b, return_similarity_mat=True)
            loss.backward()
            tag = 'mv' if mv else 'sum'
            out['M'] = M.detach().numpy()
            out['loss_' + tag] = loss.item()
            out['dimg_' + tag] = a.grad.numpy()
            out['dcap_' + tag] = b.grad.numpy()
        save(name, **out)


def gen_distill():
    # name, B, seed, teacher offset, ctor kwargs (defaults of alad/loss.py:360 when empty)
    cases = (('distill_b16', 16, 41, 4.0, {}), ('distill_b5', 5, 42, 4.0, {}),
             ('distill_b24_thr', 24, 43, 0.0, dict(margin=0.15, threshold=0.3, stride=2)),
             ('distill_b40_neg', 40, 44, -2.5, dict(margin=0.25, threshold=-1.0, stride=5)))
    for name, B, seed, offset, kw in cases:
        img, cap = synth.global_embeddings(B, 64, seed, 1.0)
        teacher = (synth.normal((B, B), seed + 7) * 0.8 + 3.0 * np.eye(B, dtype=np.float32) + offset).astype(np.float32)
        student = (img @ cap.T).astype(np.float32)
        out = dict(B=B, seed=seed, teacher=teacher, student=student, margin=kw.get('margin', 0.2),
                   threshold=kw.get('threshold', 0.1), stride=kw.get('stride', 3))
        for mode in ('listnet', 'mse', 'ordinal', 'contrastive'):
            crit = ref_loss.DistillationLoss(mode=mode, **kw)
            if mode == 'mse' and kw:
                with torch.no_grad():
                    crit.wb.copy_(torch.tensor([0.8, -0.3]))
            st = t(student).requires_grad_(True)
            loss = crit(t(teacher.copy()), st)
            loss.backward()
            out['loss_' + mode] = loss.item()
            out['dstudent_' + mode] = st.grad.numpy()
            if mode == 'mse':
                out['wb_mse'] = crit.wb.detach().numpy().copy()
                out['dwb_mse'] = crit.wb.grad.numpy().copy()
        save(name, **out)


def gen_order_sim():
    for name, Bi, Bc, D, seed in (('order_b12', 12, 12, 64, 51), ('order_rect', 5, 9, 40, 52)):
        im = synth.normal((Bi, D), seed)
        s = synth.normal((Bc, D), seed + 1)
        a = t(im).requires_grad_(True)
        b = t(s).requires_grad_(True)
        scores = ref_loss.order_sim(a, b)
        w = synth.normal((Bi, Bc), seed + 2)
        (scores * t(w)).sum().backward()
        out = dict(Bi=Bi, Bc=Bc, D=D, seed=seed, scores=scores.detach().numpy(), w=w, dim=a.grad.numpy(), ds=b.grad.numpy())
        if Bi == Bc:
            crit = ref_loss.ContrastiveLoss(margin=0.2, measure='order', max_violation=True)
            a2 = t(im).requires_grad_(True)
            b2 = t(s).requires_grad_(True)
            loss = crit(a2, b2)
            loss.backward()
            out.update(loss_mv=loss.item(), dim_mv=a2.grad.numpy(), ds_mv=b2.grad.numpy())
        save(name, **out)


# ------------------------------------------------------------------------- ALADModel orchestration
def import_alad_model():
    import oscar, oscar.modeling                                      # noqa: F401
    stub = types.ModuleType('transformers.pytorch_transformers')
    stub.BertTokenizer = type('BertTokenizer', (), {})
    stub.BertConfig = type('BertConfig', (), {})
    import transformers as _tf
    sys.modules['transformers.pytorch_transformers'] = stub
    _tf.pytorch_transformers = stub
    stub2 = types.ModuleType('oscar.modeling.modeling_bert')
    stub2.ImageBertForSequenceClassification = type('ImageBertForSequenceClassification', (), {})
    sys.modules['oscar.modeling.modeling_bert'] = stub2
    import alad.alad_model as am
    return am


def gen_model():
    am = import_alad_model()
    cfg_dir = '/root/reference/alad/configs'
    B, R, T, D, seed = 8, 34, 50, 64, 51
    im, s, im_len, s_len = synth.structured_alignment_batch(B, R, T, D, seed, 1.0, True)
    img_emb, cap_emb = synth.global_embeddings(B, D, seed + 1, 1.0)
    out = dict(B=B, R=R, T=T, D=D, seed=seed, im_len=np.array(im_len), s_len=np.array(s_len))
    names = []
    for fn in sorted(os.listdir(cfg_dir)):
        if not fn.endswith('.yaml'):
            continue
        with open(os.path.join(cfg_dir, fn)) as f:
            config = yaml.safe_load(f)
        tr = config['training']
        m = am.ALADModel.__new__(am.ALADModel)
        torch.nn.Module.__init__(m)
        m.losses_types = tr['loss-type'].split('-')
        m.losses_weights = {k: v for k, v in zip(m.losses_types, tr['loss-weights'])}
        m.auto_weight = False
        m.config = config
        m.alignment_criterion = ref_loss.AlignmentContrastiveLoss(
            margin=tr['margin'], measure=tr['measure'], max_violation=tr['max-violation'],
            aggregation=tr['alignment-mode'])
        m.matching_criterion = ref_loss.ContrastiveLoss(
            margin=tr['margin'], measure=tr['measure'], max_violation=tr['max-violation'])
        m.distillation_loss = ref_loss.DistillationLoss(mode=tr['distillation-mode'])
        m.logger = ref_eval.LogCollector()
        m.Eiters = 0
        sets = (t(img_emb), t(cap_emb), t(im).permute(1, 0, 2), t(s).permute(1, 0, 2), im_len, s_len, 0)
        m.forward_emb = lambda a, b, _sets=sets: _sets
        key = fn[:-5].replace('-', '_').replace('.', '_')
        names.append(fn)
        out[key + '__loss_type'] = tr['loss-type']
        out[key + '__weights'] = np.array(tr['loss-weights'], dtype=np.float64)
        for epoch in (0, 5):
            loss, d = m.forward(None, None, epoch=epoch, distill_epoch=2)
            out['%s__e%d_total' % (key, epoch)] = float(loss)
            out['%s__e%d_keys' % (key, epoch)] = np.array(list(d.keys()))
            out['%s__e%d_vals' % (key, epoch)] = np.array([float(v) for v in d.values()])
        out[key + '__logged'] = np.array(list(m.logger.meters.keys()))
        out[key + '__eiters'] = m.Eiters
    out['configs'] = np.array(names)
    save('model_forward', **out)


# --------------------------------------------------------------------- encode_data -> i2t / t2i pipeline
def gen_eval_pipeline():
    """The reference's evaluation pipeline end to end (alad/evaluation.py:80-327): its own encode_data over a loader of
    encoder batches (a fake model whose forward_emb returns prepared 7-tuples), then its own i2t / t2i on the buffers it
    filled, matching head and alignment head."""
    batches = synth.encoder_batches()
    N = sum(len(b['img_len']) for b in batches)

    class FakeModel:
        logger = None

        def eval(self):
            pass

        def forward_emb(self, example_imgs, example_txts):
            b = batches[int(example_txts[0][0])]
            return (t(b['img_glob']), t(b['cap_glob']), t(b['img_set']), t(b['cap_seq']), list(b['img_len']), list(b['cap_len']), 0)

    class Loader(list):
        dataset = list(range(N))
    loader = Loader([((torch.zeros((len(b['img_len']), 1)),), (torch.full((len(b['img_len']),), k),)) for k, b in enumerate(batches)])
    img_embs, cap_embs, il, cl = ref_eval.encode_data(FakeModel(), loader, logging=lambda *_: None)
    out = dict(N=N, img_embs_checksum=synth.checksum(img_embs.numpy()), cap_embs_checksum=synth.checksum(cap_embs.numpy()),
               img_embs_s=img_embs.numpy()[:, :, ::8], cap_embs_s=cap_embs.numpy()[:, :, ::8], img_len=np.array(il), cap_len=np.array(cl))
    crit = ref_loss.AlignmentContrastiveLoss(aggregation='MrSw')

    def sim_fn(img, cap, a, b):
        with torch.no_grad():
            return crit(img, cap, a, b, return_loss=False, return_similarity_mat=True)
    saved_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for tag, fn in (('match', None), ('align', sim_fn)):
            m, (ranks, top1) = ref_eval.i2t(img_embs, cap_embs, il, cl, return_ranks=True, sim_function=fn, cap_batches=5)
            out['i2t_%s_metrics' % tag], out['i2t_%s_ranks' % tag], out['i2t_%s_top1' % tag] = np.array(m, dtype=np.float64), ranks, top1
            m, (ranks, top50) = ref_eval.t2i(img_embs, cap_embs, il, cl, return_ranks=True, sim_function=fn, im_batches=1)
            out['t2i_%s_metrics' % tag], out['t2i_%s_ranks' % tag], out['t2i_%s_top1' % tag] = np.array(m, dtype=np.float64), ranks, top50[:, 0]
    finally:
        torch.Tensor.cuda = saved_cuda
    save('eval_pipeline', **out)


# ------------------------------------------------------------ matching head + encoder hand-off
def gen_matching_head():
    """The reference's OWN JointTextImageTransformerEncoder.forward (alad/alad_model.py:119-247) driven with a fake
    backbone that returns prepared hidden states: pins the slicing to the batch maxima, the key-padding masks, the
    2-layer nn.TransformerEncoder matching head (d 768, nhead 4, ffn 768), slot 0, F.normalize / l2norm -- forward
    values and gradients w.r.t. the backbone's outputs and the head's parameters (eval mode: dropout off)."""
    am = import_alad_model()
    D, B, n_tok, n_reg, seed = 768, 5, 14, 20, 81
    cap_len = [14, 9, 11, 6, 12]
    feat_len = [17, 20, 12, 20, 15]
    txt_seq = synth.normal((B, n_tok, D), seed)
    img_seq = synth.normal((B, n_tok + n_reg, D), seed + 1)
    enc = am.JointTextImageTransformerEncoder.__new__(am.JointTextImageTransformerEncoder)
    torch.nn.Module.__init__(enc)
    enc.freeze_teran = False
    enc.depth_aggregation_alignment = enc.depth_aggregation_matching = False
    enc.text_aggregation_type = enc.img_aggregation_type = None
    enc.l1_regularization = False
    enc.post_oscar_transformer = None
    enc.shared_transformer = True
    layer = torch.nn.TransformerEncoderLayer(d_model=D, nhead=4, dim_feedforward=D, dropout=0.1)         # alad_model.py:104-106
    enc.final_projection_net = torch.nn.TransformerEncoder(layer, num_layers=2)
    named = [(n, tuple(p.shape)) for n, p in enc.final_projection_net.named_parameters()]
    vals = synth.module_parameters(named, seed + 7)
    with torch.no_grad():
        for n, p in enc.final_projection_net.named_parameters():
            p.copy_(t(vals[n]))
    a, b = t(txt_seq).requires_grad_(True), t(img_seq).requires_grad_(True)

    class FakeBert:
        def bert(self, input_ids, attention_mask, token_type_ids, img_feats):
            return (a,) if img_feats is None else (b,)
    enc.oscar_model = FakeBert()
    enc.eval()
    ids = torch.zeros((B, n_tok), dtype=torch.long)
    examples_txts = (ids, None, None, None, cap_len)
    examples_imgs = (ids, None, None, torch.zeros((B, n_reg, 4)), None, feat_len)
    img_glob, cap_glob, img_set, cap_seq, fl, cl, reg = enc(examples_imgs, examples_txts)
    assert fl == feat_len and cl == cap_len and reg == 0
    w = [synth.normal(tuple(x.shape), seed + 20 + k) for k, x in enumerate((img_glob, cap_glob, img_set, cap_seq))]
    (img_glob * t(w[0])).sum().add((cap_glob * t(w[1])).sum()).add(0.05 * (img_set * t(w[2])).sum()) \
        .add(0.05 * (cap_seq * t(w[3])).sum()).backward()
    out = dict(D=D, B=B, n_tok=n_tok, n_reg=n_reg, seed=seed, cap_len=np.array(cap_len), feat_len=np.array(feat_len),
               param_names=np.array([n for n, _ in named]), img_glob=img_glob.detach().numpy(), cap_glob=cap_glob.detach().numpy(),
               img_set_shape=np.array(img_set.shape), cap_seq_shape=np.array(cap_seq.shape),
               img_set_s=img_set.detach().numpy()[:, :, ::16], cap_seq_s=cap_seq.detach().numpy()[:, :, ::16],
               d_txt_seq_s=a.grad.numpy()[:, :, ::16], d_img_seq_s=b.grad.numpy()[:, :, ::16],
               d_txt_seq_abs=float(a.grad.abs().sum()), d_img_seq_abs=float(b.grad.abs().sum()))
    for n, p in enc.final_projection_net.named_parameters():
        g = p.grad.numpy()
        key = n.replace('.', '__')
        out['dp_abs__' + key] = float(np.abs(g).sum())
        out['dp_s__' + key] = g.reshape(-1)[::max(1, g.size // 512)][:512].copy()
    save('matching_head', **out)



# ------------------------------------------------------------------ VinVL / Oscar backbone (last piece)
BACKBONE_CFG = dict(vocab_size=120, hidden_size=64, num_hidden_layers=3, num_attention_heads=4, intermediate_size=160,
                    hidden_act='gelu', hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=48,
                    type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12, img_feature_dim=22,
                    img_feature_type='faster_r-cnn', use_img_layernorm=1, img_layer_norm_eps=1e-12, num_labels=2, loss_type='sfmx')


from backbone_inputs import backbone_inputs            # noqa: E402  (tests/golden/backbone_inputs.py, shared with the tests)


def gen_backbone():
    """The reference's OWN BertImgModel (oscar/modeling/modeling_bert.py:150-279: image embedding + LayerNorm, concatenation,
    extended mask, CaptionBertEncoder loop, CaptionBertSelfAttention arithmetic, pooler hand-off) run over
    tests/golden/hf_bert_layers.py standing in for the un-vendored `transformers.pytorch_transformers` layers; plus, as an
    independent check of those layers, the installed transformers' BertModel on the text-only input with the same weights."""
    import importlib.util
    saved = {k: sys.modules.get(k) for k in ('transformers.pytorch_transformers', 'transformers.pytorch_transformers.modeling_bert',
                                             'oscar.modeling.modeling_bert', 'oscar.modeling.modeling_utils', 'oscar.utils.cbs')}
    import transformers as _tf
    spec = importlib.util.spec_from_file_location('transformers.pytorch_transformers.modeling_bert', os.path.join(HERE, 'hf_bert_layers.py'))
    layers = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(layers)
    pkg = types.ModuleType('transformers.pytorch_transformers')
    pkg.modeling_bert = layers
    mu = types.ModuleType('oscar.modeling.modeling_utils')
    mu.CaptionPreTrainedModel = type('CaptionPreTrainedModel', (layers.BertPreTrainedModel,), {})
    mu.ImgPreTrainedModel = type('ImgPreTrainedModel', (layers.BertPreTrainedModel,), {})
    cbs = types.ModuleType('oscar.utils.cbs')
    cbs.ConstrainedBeamSearch = cbs.select_best_beam_with_constraints = None
    try:
        sys.modules.update({'transformers.pytorch_transformers': pkg, 'transformers.pytorch_transformers.modeling_bert': layers,
                            'oscar.modeling.modeling_utils': mu, 'oscar.utils.cbs': cbs})
        sys.modules.pop('oscar.modeling.modeling_bert', None)
        import oscar, oscar.modeling                                   # noqa: F401,E401
        import importlib
        ref_mb = importlib.import_module('oscar.modeling.modeling_bert')
        assert ref_mb.__file__.startswith('/root/reference/')
        seed = 1301
        cfg = layers.BertConfig(output_attentions=True, output_hidden_states=True, **BACKBONE_CFG)   # as alad_model.py:41-42 sets them
        torch.manual_seed(0)
        ref = ref_mb.BertImgModel(cfg).eval()
        named = [(n, tuple(p.shape)) for n, p in ref.named_parameters()]
        vals = synth.module_parameters(named, seed, scale=0.08)
        with torch.no_grad():
            for n, p in ref.named_parameters():
                p.copy_(t(vals[n]))
        ids, tmask, fmask, types_, feats = backbone_inputs(seed + 50)
        out = dict(seed=seed, cfg_json=np.array(__import__('json').dumps(BACKBONE_CFG)), param_names=np.array([n for n, _ in named]),
                   ids_checksum=synth.checksum(ids), feats_checksum=synth.checksum(feats))
        # text-only pass (alad_model.py:124-131) and tags + regions pass (:133-140)
        o_txt = ref(t(ids), token_type_ids=t(types_), attention_mask=t(tmask), img_feats=None)
        f = t(feats).requires_grad_(True)
        o_img = ref(t(ids), token_type_ids=t(types_), attention_mask=t(fmask), img_feats=f)
        w = synth.normal(tuple(o_img[0].shape), seed + 60)
        ref.zero_grad()
        (o_img[0] * t(w)).sum().add(o_img[1].sum()).backward()
        out.update(txt_seq=o_txt[0].detach().numpy(), txt_pooled=o_txt[1].detach().numpy(),
                   img_seq=o_img[0].detach().numpy(), img_pooled=o_img[1].detach().numpy(),
                   img_hidden_1=o_img[2][1].detach().numpy(), n_hidden=len(o_img[2]),
                   img_att_last=o_img[3][-1].detach().numpy(), n_att=len(o_img[3]),
                   d_feats=f.grad.numpy(), d_img_embedding_weight=ref.img_embedding.weight.grad.numpy(),
                   d_word_embeddings_abs=float(ref.embeddings.word_embeddings.weight.grad.abs().sum()),
                   d_q0=ref.encoder.layer[0].attention.self.query.weight.grad.numpy())
        # independent check of the restated layers: today's transformers BertModel, same weights, text-only path
        from transformers.models.bert.modeling_bert import BertModel as HFBert, BertConfig as HFConfig
        hc = HFConfig(**{k: v for k, v in BACKBONE_CFG.items() if not k.startswith('img_') and k not in ('use_img_layernorm', 'loss_type')})
        try:
            hc._attn_implementation = 'eager'
        except Exception:
            pass
        hf = HFBert(hc, add_pooling_layer=True).eval()
        sd = {k: v for k, v in ref.state_dict().items() if not k.startswith('img_embedding') and k not in ('LayerNorm.weight', 'LayerNorm.bias')}
        missing = hf.load_state_dict(sd, strict=False)
        assert not [k for k in missing.unexpected_keys], missing
        assert all('position_ids' in k or 'token_type_ids' in k for k in missing.missing_keys), missing
        with torch.no_grad():
            ho = hf(input_ids=t(ids), attention_mask=t(tmask), token_type_ids=t(types_))
        d = (ho.last_hidden_state - o_txt[0]).abs().max().item()
        # the additive masks differ (-10000 in the reference, dtype-min in today's HF): identical on attended positions
        print('text-only path vs installed transformers %s BertModel: max |diff| %.3g (pooled %.3g)'
              % (_tf.__version__, d, (ho.pooler_output - o_txt[1]).abs().max().item()))
        assert d < 5e-5
        out['hf_version'] = np.array(_tf.__version__)
        out['hf_txt_seq_maxdiff'] = d
        save('backbone_bertimg', **out)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


# --------------------------------------------------------------------------------------- recall
def gen_recall():
    for name, n_img, D, seed, sigma in (('recall_n500', 100, 64, 61, 3.5),
                                        ('recall_n5000', 1000, 768, 62, 9.0)):
        img, cap = synth.retrieval_embeddings(n_img, D, seed, sigma)
        out = dict(n_img=n_img, D=D, seed=seed, sigma=sigma, img_checksum=synth.checksum(img),
                   cap_checksum=synth.checksum(cap))
        out['compute_recall'] = np.array(ref_recall.compute_recall(t(img), t(cap)), dtype=np.float64)
        for mode in ('i2t', 't2i'):
            m, (ranks, top1) = ref_recall.recall(t(img), t(cap), None, mode=mode, return_ranks=True)
            out[mode + '_metrics'] = np.array(m, dtype=np.float64)
            out[mode + '_ranks'] = ranks
            out[mode + '_top1'] = top1
        save(name, **out)


def gen_recall_5fold():
    """recall_1k_5fold_test + recall_test (alad/recall_auxiliary.py:72-130) on 25 000 rows = five 5000-row folds."""
    n_img, D, seed, sigma = 5000, 64, 63, 3.2
    img, cap = synth.retrieval_embeddings(n_img, D, seed, sigma)
    out = dict(n_img=n_img, D=D, seed=seed, sigma=sigma, img_checksum=synth.checksum(img), cap_checksum=synth.checksum(cap))
    out['recall_1k_5fold_test'] = np.array(ref_recall.recall_1k_5fold_test(t(img), t(cap)), dtype=np.float64)
    out['recall_test_fold0'] = np.array(ref_recall.recall_test(t(img[:5000]), t(cap[:5000]), None, None), dtype=np.float64)
    save('recall_5fold', **out)


# ---------------------------------------------------------------- call signatures
SIGNATURE_TARGETS = [
    # (key, module alias, dotted attribute) -- the drop-in surface SURVEY 8(b) lists
    ('AlignmentContrastiveLoss.__init__', 'loss', 'AlignmentContrastiveLoss.__init__'),
    ('AlignmentContrastiveLoss.forward', 'loss', 'AlignmentContrastiveLoss.forward'),
    ('ContrastiveLoss.__init__', 'loss', 'ContrastiveLoss.__init__'),
    ('ContrastiveLoss.forward', 'loss', 'ContrastiveLoss.forward'),
    ('DistillationLoss.__init__', 'loss', 'DistillationLoss.__init__'),
    ('DistillationLoss.forward', 'loss', 'DistillationLoss.forward'),
    ('Contrastive.__init__', 'loss', 'Contrastive.__init__'),
    ('Contrastive.compute_contrastive_loss', 'loss', 'Contrastive.compute_contrastive_loss'),
    ('dot_sim', 'loss', 'dot_sim'), ('cosine_sim', 'loss', 'cosine_sim'), ('order_sim', 'loss', 'order_sim'),
    ('ALADModel.forward', 'model', 'ALADModel.forward'),
    ('ALADModel.forward_emb', 'model', 'ALADModel.forward_emb'),
    ('ALADModel.forward_loss', 'model', 'ALADModel.forward_loss'),
    ('recall', 'recall', 'recall'), ('recall_test', 'recall', 'recall_test'), ('compute_recall', 'recall', 'compute_recall'),
    ('recall_1k_5fold_test', 'recall', 'recall_1k_5fold_test'),
    ('i2t', 'eval', 'i2t'), ('t2i', 'eval', 't2i'), ('encode_data', 'eval', 'encode_data'),
    ('l2norm', 'utils', 'l2norm'),
]


def signature_record(fn):
    """[[name, kind, default-or-null], ...]: what inspect.signature says, as plain JSON data."""
    import inspect
    out = []
    for p_ in inspect.signature(fn).parameters.values():
        default = None if p_.default is inspect.Parameter.empty else repr(p_.default)
        out.append([p_.name, p_.kind.name, default])
    return out


def gen_signatures():
    """tests/golden/signatures.json: inspect.signature of every callable of the reference's drop-in surface (names,
    parameter kinds, defaults as repr strings).  Data about the reference, not its text."""
    import json
    import alad.utils as ref_utils
    am = import_alad_model()
    mods = {'loss': ref_loss, 'model': am, 'recall': ref_recall, 'eval': ref_eval, 'utils': ref_utils}
    rec = {}
    for key, alias, dotted in SIGNATURE_TARGETS:
        obj = mods[alias]
        for part in dotted.split('.'):
            obj = getattr(obj, part)
        rec[key] = signature_record(obj)
    path = os.path.join(HERE, 'signatures.json')
    with open(path, 'w') as f:
        json.dump(rec, f, indent=1, sort_keys=True)
        f.write('\n')
    print('signatures.json: %d callables' % len(rec))


if __name__ == '__main__':
    only = sys.argv[1:]
    if only:                                   # e.g. `make_golden.py gen_eval_coco1k` regenerates one family
        for name in only:
            globals()[name]()
        sys.exit(0)
    gen_alignment()
    gen_eval()
    gen_eval_coco1k()
    gen_eval_coco1k_d768()
    gen_eval_pipeline()
    gen_matching()
    gen_distill()
    gen_order_sim()
    gen_model()
    gen_matching_head()
    gen_backbone()
    gen_recall()
    gen_recall_5fold()
    gen_signatures()
