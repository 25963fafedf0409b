"""CPU tier: the oracle (oracle/alad_oracle.py, oracle/faithful_torch.py) against the golden
fixtures produced by the reference itself (tests/golden/make_golden.py).  This is what pins the
oracle; the -m gpu tests then compare the HIP path with the pinned oracle and the same fixtures."""
import numpy as np
import pytest

from conftest import (ALIGN_GOLDENS, SQUARE_ALIGN_GOLDENS, golden_alignment_inputs, load_golden)
import alad_oracle as O
import faithful_torch as FT
import torch

RTOL = 1e-3          # north_star tolerance; the oracle itself lands around 1e-6


def close(a, b, rtol=2e-5, atol=2e-5):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize('name', ALIGN_GOLDENS)
@pytest.mark.parametrize('mode', ['MrSw', 'MrAVGw', 'MwSr', 'symm', 'sum', 'mean'])
def test_alignment_scores(name, mode):
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    S = O.alignment_scores(im, s, il, sl, mode)
    close(S, g['S_' + mode])


@pytest.mark.parametrize('name', SQUARE_ALIGN_GOLDENS)
@pytest.mark.parametrize('tag', ['mv', 'sum'])
def test_hinge_and_grad_wrt_scores(name, tag):
    g = load_golden(name)
    loss, dS = O.hinge_loss(g['S_MrSw'], float(g['margin']), tag == 'mv', return_grad=True)
    close(loss, g['loss_' + tag], rtol=1e-5, atol=1e-5)
    close(dS, g['dS_' + tag], rtol=0, atol=0)


@pytest.mark.parametrize('name', ['align_tiny', 'align_b5_d64', 'align_b12_struct', 'align_r33',
                                  'align_b8_d768_rag'])
@pytest.mark.parametrize('tag', ['mv', 'sum'])
def test_alignment_backward(name, tag):
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    dim, ds = O.alignment_scores_backward(im, s, il, sl, g['dS_' + tag])
    st = int(g['grad_stride'])
    scale = max(1e-6, float(np.abs(g['dim_' + tag]).max()))
    close(dim[:, :, ::st], g['dim_' + tag], rtol=1e-4, atol=1e-5 * scale)
    scale = max(1e-6, float(np.abs(g['ds_' + tag]).max()))
    close(ds[:, :, ::st], g['ds_' + tag], rtol=1e-4, atol=1e-5 * scale)
    from aladin_amd import synth
    assert abs(np.abs(dim).sum() - float(g['dim_abs_' + tag])) <= 1e-4 * float(g['dim_abs_' + tag]) + 1e-6


@pytest.mark.parametrize('name', SQUARE_ALIGN_GOLDENS)
def test_faithful_torch_matches_reference(name):
    import torch
    import faithful_torch as FT
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    for tag, mv in (('mv', True), ('sum', False)):
        loss, S, dim, ds = FT.alignment_triplet_step(torch.from_numpy(im), torch.from_numpy(s), il, sl,
                                                     float(g['margin']), mv)
        close(S.numpy(), g['S_MrSw'])
        close(loss.item(), g['loss_' + tag], rtol=1e-5, atol=1e-5)
        st = int(g['grad_stride'])
        close(dim.numpy()[:, :, ::st], g['dim_' + tag], rtol=1e-4, atol=1e-6)
        close(ds.numpy()[:, :, ::st], g['ds_' + tag], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('name', ['align_b5_d64', 'align_rect', 'align_r33'])
@pytest.mark.parametrize('mode', ['MrSw', 'MrAVGw', 'MwSr', 'symm', 'sum', 'mean'])
def test_faithful_torch_all_modes(name, mode):
    import torch
    import faithful_torch as FT
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    S = FT.alignment_scores_faithful(torch.from_numpy(im), torch.from_numpy(s), il, sl, mode)
    close(S.numpy(), g['S_' + mode])


@pytest.mark.parametrize('name', ['match_b16_d768', 'match_b7_d64'])
def test_matching(name):
    g = load_golden(name)
    from aladin_amd import synth
    img, cap = synth.global_embeddings(int(g['B']), int(g['D']), int(g['seed']), float(g['noise']))
    M = O.dot_scores(img, cap)
    close(M, g['M'], rtol=1e-5, atol=1e-6)
    for tag, mv in (('mv', True), ('sum', False)):
        loss, dM = O.hinge_loss(M, 0.2, mv, return_grad=True)
        close(loss, g['loss_' + tag], rtol=1e-5, atol=1e-6)
        close(dM @ cap, g['dimg_' + tag], rtol=1e-4, atol=1e-6)
        close(dM.T @ img, g['dcap_' + tag], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('name', ['distill_b16', 'distill_b5'])
def test_listnet(name):
    g = load_golden(name)
    loss, dM = O.listnet_loss(g['teacher'], g['student'], return_grad=True)
    close(loss, g['loss_listnet'], rtol=1e-5, atol=1e-6)
    close(dM, g['dstudent_listnet'], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('name', ['distill_b16', 'distill_b5', 'distill_b24_thr', 'distill_b40_neg'])
def test_other_distillation_modes(name):
    """mse / contrastive / ordinal closed forms (alad/loss.py:371-425) vs the reference's autograd."""
    g = load_golden(name)
    Tm, M = g['teacher'], g['student']
    m, th, st = float(g['margin']), float(g['threshold']), int(g['stride'])
    loss, dM, dwb = O.distill_mse(Tm, M, g['wb_mse'], return_grad=True)
    close(loss, g['loss_mse'], rtol=2e-6, atol=1e-6)
    close(dM, g['dstudent_mse'], rtol=1e-5, atol=1e-8)
    close(dwb, g['dwb_mse'], rtol=1e-5, atol=1e-6)
    loss, dM = O.distill_contrastive(Tm, M, m, return_grad=True)
    close(loss, g['loss_contrastive'], rtol=2e-6, atol=1e-6)
    np.testing.assert_array_equal(dM, g['dstudent_contrastive'])
    loss, dM = O.distill_ordinal(Tm, M, m, th, st, return_grad=True)
    close(loss, g['loss_ordinal'], rtol=2e-6, atol=1e-6)
    close(dM, g['dstudent_ordinal'], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('name', ['order_b12', 'order_rect'])
def test_order_sim(name):
    from aladin_amd import synth
    g = load_golden(name)
    im = synth.normal((int(g['Bi']), int(g['D'])), int(g['seed']))
    s = synth.normal((int(g['Bc']), int(g['D'])), int(g['seed']) + 1)
    close(O.order_scores(im, s), g['scores'], rtol=2e-6, atol=1e-6)
    di, ds = O.order_scores_backward(im, s, g['w'])
    close(di, g['dim'], rtol=1e-5, atol=1e-6)
    close(ds, g['ds'], rtol=1e-5, atol=1e-6)
    if 'loss_mv' in g:
        S = O.order_scores(im, s)
        loss, dS = O.hinge_loss(S, 0.2, True, return_grad=True)
        close(loss, g['loss_mv'], rtol=1e-5, atol=1e-6)
        di, ds = O.order_scores_backward(im, s, dS)
        close(di, g['dim_mv'], rtol=1e-5, atol=1e-6)
        close(ds, g['ds_mv'], rtol=1e-5, atol=1e-6)


def test_model_forward_dicts():
    g = load_golden('model_forward')
    from aladin_amd import synth
    B, R, T, D, seed = (int(g[k]) for k in ('B', 'R', 'T', 'D', 'seed'))
    im, s, il, sl = synth.structured_alignment_batch(B, R, T, D, seed, 1.0, True)
    img_emb, cap_emb = synth.global_embeddings(B, D, seed + 1, 1.0)
    for fn in g['configs']:
        key = str(fn)[:-5].replace('-', '_').replace('.', '_')
        lt = str(g[key + '__loss_type'])
        w = dict(zip(lt.split('-'), g[key + '__weights']))
        d = O.forward_loss(img_emb, cap_emb, im.transpose(1, 0, 2), s.transpose(1, 0, 2), il, sl, lt)
        for epoch in (0, 5):
            total, kept = O.total_loss(d, w, epoch, 2)
            assert list(kept.keys()) == [str(k) for k in g['%s__e%d_keys' % (key, epoch)]]
            close(list(kept.values()), g['%s__e%d_vals' % (key, epoch)], rtol=1e-5, atol=1e-6)
            close(total, g['%s__e%d_total' % (key, epoch)], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', ['recall_n500', 'recall_n5000'])
def test_recall(name):
    g = load_golden(name)
    from aladin_amd import synth
    img, cap = synth.retrieval_embeddings(int(g['n_img']), int(g['D']), int(g['seed']), float(g['sigma']))
    close(O.compute_recall(img, cap), g['compute_recall'], rtol=0, atol=1e-9)
    for mode in ('i2t', 't2i'):
        m, (ranks, top1) = O.recall(img, cap, mode, return_ranks=True)
        close(m, g[mode + '_metrics'], rtol=0, atol=1e-9)
        assert np.array_equal(ranks, g[mode + '_ranks'])
        assert np.array_equal(top1, g[mode + '_top1'])


def test_eval_i2t_t2i():
    g = load_golden('eval_sets')
    from aladin_amd import synth
    images, captions, il, cl = synth.eval_sets(int(g['n_img']), int(g['D']), int(g['seed']))
    S = O.alignment_scores(images[0::5], captions, il[0::5], cl)
    close(S, g['S_eval'])
    for tag, sim in (('match', 'matching'), ('align', 'alignment')):
        m, (r, t1) = O.i2t(images, captions, il, cl, sim, return_ranks=True)
        close(m, g['i2t_%s_metrics' % tag], rtol=0, atol=1e-9)
        assert np.array_equal(r, g['i2t_%s_ranks' % tag]) and np.array_equal(t1, g['i2t_%s_top1' % tag])
        m, (r, t1) = O.t2i(images, captions, il, cl, sim, return_ranks=True)
        close(m, g['t2i_%s_metrics' % tag], rtol=0, atol=1e-9)
        assert np.array_equal(r, g['t2i_%s_ranks' % tag]) and np.array_equal(t1, g['t2i_%s_top1' % tag])


def test_oracle_on_the_eval_pipeline_fixture():
    """eval_pipeline.npz: the reference's encode_data buffers rebuilt from the generator (position 0 = global embedding,
    sets zero-padded to 71) and the oracle's i2t / t2i on them for both heads."""
    g = load_golden('eval_pipeline')
    from aladin_amd import synth
    batches = synth.encoder_batches()
    N = int(g['N'])
    img = np.zeros((N, 71, 64), np.float32)
    cap = np.zeros((N, 71, 64), np.float32)
    il, cl, k0 = [], [], 0
    for b in batches:
        n = len(b['img_len'])
        img[k0:k0 + n, :b['img_set'].shape[0]] = b['img_set'].transpose(1, 0, 2)          # alad/evaluation.py:124-125
        cap[k0:k0 + n, :b['cap_seq'].shape[0]] = b['cap_seq'].transpose(1, 0, 2)
        img[k0:k0 + n, 0], cap[k0:k0 + n, 0] = b['img_glob'], b['cap_glob']             # :127-128
        il += b['img_len']
        cl += b['cap_len']
        k0 += n
    assert np.array_equal(img[:, :, ::8], g['img_embs_s']) and np.array_equal(cap[:, :, ::8], g['cap_embs_s'])
    for tag, sim in (('match', 'matching'), ('align', 'alignment')):
        m, (r, t1) = O.i2t(img, cap, il, cl, sim, return_ranks=True)
        assert np.array_equal(r, g['i2t_%s_ranks' % tag]) and np.array_equal(t1, g['i2t_%s_top1' % tag])
        close(m, g['i2t_%s_metrics' % tag], rtol=0, atol=1e-9)
        m, (r, t1) = O.t2i(img, cap, il, cl, sim, return_ranks=True)
        assert np.array_equal(r, g['t2i_%s_ranks' % tag]) and np.array_equal(t1, g['t2i_%s_top1' % tag])
        close(m, g['t2i_%s_metrics' % tag], rtol=0, atol=1e-9)


def test_recall_1k_5fold():
    g = load_golden('recall_5fold')
    from aladin_amd import synth
    img, cap = synth.retrieval_embeddings(int(g['n_img']), int(g['D']), int(g['seed']), float(g['sigma']))
    assert abs(synth.checksum(cap) - float(g['cap_checksum'])) <= 1e-6 * abs(float(g['cap_checksum']))
    close(O.recall_1k_5fold(img, cap), g['recall_1k_5fold_test'], rtol=0, atol=1e-9)
    close(O.compute_recall(img[:5000], cap[:5000]), g['recall_test_fold0'], rtol=0, atol=1e-9)


def test_oracle_on_coco1k_sized_fixture_sample():
    """eval_coco1k.npz (1000 images x 5000 captions through the reference's own i2t / t2i loops): the oracle is
    checked on the stored sample of the reference's score matrix (every 20th image x every 10th caption, incl.
    images that fill the 71-position set) and on a block of its diagonal; at full size the fixture is the
    checker itself (GPU tier)."""
    g = load_golden('eval_coco1k')
    from aladin_amd import synth
    n_img = int(g['n_img'])
    images, captions, il, cl = synth.eval_sets(n_img, int(g['D']), int(g['seed']), base_weight=float(g['gen_base_weight']),
                                               img_len_range=tuple(int(v) for v in g['gen_img_len_range']),
                                               cap_len_range=tuple(int(v) for v in g['gen_cap_len_range']),
                                               n_full=int(g['gen_n_full']))
    assert il == [int(v) for v in g['img_len']] and cl == [int(v) for v in g['cap_len']]
    assert abs(synth.checksum(captions) - float(g['captions_checksum'])) <= 1e-6 * abs(float(g['captions_checksum']))
    assert sum(1 for v in il[0::5] if v == 71) == int(g['gen_n_full'])
    S = O.alignment_scores(images[0::100], captions[0::10], il[0::100], cl[0::10])
    close(S, g['S_sample'], rtol=2e-6, atol=4e-6)
    blk = O.alignment_scores(images[0:200:5], captions[0:200], il[0:200:5], cl[0:200])
    close(blk[np.arange(200) // 5, np.arange(200)], g['S_diag'][:200], rtol=2e-6, atol=4e-6)
    # the fixture is self-consistent: its metrics follow from its ranks
    for d in ('i2t', 't2i'):
        close(O._metrics(g[d + '_ranks'].astype(np.float64)), g[d + '_metrics'][:5], rtol=0, atol=1e-9)


@pytest.mark.parametrize('name', ALIGN_GOLDENS)
def test_scan_sentences_scores(name):
    """aggregation='scan-sentences' (alad/loss.py:136-149): scores for every case (ragged included)."""
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    S = O.scan_sentences_scores(im, s, il, sl)
    close(S, g['S_scan-sentences'], rtol=2e-5, atol=2e-6)
    Sf = FT.alignment_scores_faithful(torch.from_numpy(im), torch.from_numpy(s), il, sl, 'scan-sentences').numpy()
    close(Sf, g['S_scan-sentences'], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('name', ['align_tiny', 'align_b16_d768'])
def test_scan_sentences_gradients_full_length(name):
    """The analytic gradient equals the reference's autograd where the latter is finite (full-length batches)."""
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    st = int(g['scan_stride'])
    _, d_im, d_s = O.scan_sentences_scores(im, s, il, sl, dS=g['scan_w'])
    scale = float(np.abs(g['dim_scan']).max())
    close(d_im[:, :, ::st], g['dim_scan'], rtol=1e-4, atol=2e-6 * max(scale, 1.0))
    close(d_s[:, :, ::st], g['ds_scan'], rtol=1e-4, atol=2e-6 * max(scale, 1.0))


def test_scan_sentences_ragged_gradient_is_that_of_the_masked_expression():
    """Ragged batches: the reference's autograd is NaN; the oracle's closed form must equal autograd of
    the NaN-free torch restatement (oracle/faithful_torch.py) in float64."""
    from aladin_amd import synth
    im, s, il, sl = synth.alignment_batch(4, 20, 24, 32, seed=808, ragged=True, Bc=6)
    w = synth.normal((4, 6), 809)
    a = torch.from_numpy(im).double().requires_grad_(True)
    b = torch.from_numpy(s).double().requires_grad_(True)
    S = FT.alignment_scores_faithful(a, b, il, sl, 'scan-sentences')
    (S * torch.from_numpy(w).double()).sum().backward()
    So, d_im, d_s = O.scan_sentences_scores(im, s, il, sl, dS=w)
    close(So, S.detach().numpy(), rtol=1e-6, atol=1e-6)
    close(d_im, a.grad.numpy(), rtol=1e-5, atol=1e-7)
    close(d_s, b.grad.numpy(), rtol=1e-5, atol=1e-7)
