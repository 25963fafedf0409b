"""GPU tier (-m gpu): the HIP path, called through the C ABI (aladin_amd.ops -> ctypes ->
libaladin_hip.so), against the golden fixtures made from the reference and against the pinned
oracle on identical seeded inputs.

Tolerance: north_star asks for 1e-3 relative on scores/losses.  Scores come from fp16 MFMA
operands with fp32 accumulation (measured ~1e-4); everything downstream of the scores (hinge,
listnet, dot products, backward argmax) is fp32 and is held to 1e-5..1e-4.
"""
import numpy as np
import pytest
import torch

from conftest import ALIGN_GOLDENS, SQUARE_ALIGN_GOLDENS, golden_alignment_inputs, load_golden
import alad_oracle as O
import faithful_torch as FT

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev())


from aladin_amd import ops as _ops_at_import
LIBRARY_DEFAULT_BWD = _ops_at_import._BWD_PARTNERS[0]          # captured before any fixture touches it


@pytest.fixture(autouse=True)
def exact_backward():
    """Tests that do not name `bwd_mode` run under the EXACT row step (raw fp32 rows): their bit-equality / 1e-5 claims between
    code paths (dense table vs per-pair, graph vs eager, fused vs composed) are statements about the arithmetic, not about the
    operand precision.  Every test that compares gradient VALUES with the reference or the oracle takes `bwd_mode` instead and
    runs under BOTH row steps."""
    from aladin_amd import ops
    old = ops.set_backward_precision('exact')
    yield
    ops.set_backward_precision(old)


# The backward's row step, per mode: the absolute term of a gradient comparison as a fraction of the largest reference entry.
#   'exact'  raw fp32 rows normalised again: the test's own 2e-5 .. 5e-5 (measured ~3e-7);
#   'fp16'   the LIBRARY DEFAULT (what bench.py times): partner rows from the forward's packed fp16 operands -- one rounding of
#            <= 2^-11 per component.  Gate 5e-4 = half of north_star's 1e-3 for D >= 64 (the same scoping as the scores' bar,
#            DESIGN.md section 2); toy widths below that are held to 1e-3.
# Both on top of rtol 1e-3 (north_star).  The measured worst error / largest entry of every call goes to
# gpurun_out/score_err_stats.json ('gradients') next to the scores'.
FP16_BWD_GATE, FP16_BWD_GATE_TOY = 5e-4, 1e-3
GRAD_ERR_LOG = []


@pytest.fixture(params=['exact', 'fp16'])
def bwd_mode(request, exact_backward, eval_precision):
    from aladin_amd import ops
    if request.param == 'fp16' and eval_precision == 'split':
        pytest.skip('the differentiable path does not depend on the evaluation precision; the fp16 row step runs once')
    old = ops.set_backward_precision(request.param)
    yield request.param
    ops.set_backward_precision(old)


def assert_grads_close(got, ref, mode, exact_atol=2e-5, D=None, rtol=1e-3):
    """|got - ref| <= rtol |ref| + atol * max|ref| with the mode's absolute term (see FP16_BWD_GATE)."""
    import os
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = ref.detach().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
    D = ref.shape[-1] if D is None else D
    scale = max(1e-9, float(np.abs(ref).max()))
    gate = exact_atol if mode == 'exact' else (FP16_BWD_GATE if D >= 64 else FP16_BWD_GATE_TOY)
    GRAD_ERR_LOG.append((os.environ.get('PYTEST_CURRENT_TEST', '?'), {
        'mode': mode, 'D': int(D), 'gate': gate, 'max_err_over_max_ref': float('%.3g' % (np.abs(got - ref).max() / scale)),
        'max_err_beyond_rtol_over_max_ref': float('%.3g' % (np.maximum(np.abs(got - ref) - rtol * np.abs(ref), 0).max() / scale))}))
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=gate * scale)


def test_library_default_backward_is_the_fp16_row_step():
    assert LIBRARY_DEFAULT_BWD == 'fp16'


@pytest.fixture(autouse=True, params=['fp16', 'split'])
def eval_precision(request):
    """Every test runs under both precisions of the no-grad score path (ops.set_eval_precision): 'fp16' is the
    training operand (north_star's 1e-3 tolerance), 'split' the rank-exact evaluation default.  Differentiable
    scores are fp16 either way."""
    from aladin_amd import ops
    from aladin_amd import evaluation as E
    old = ops.set_eval_precision(request.param)
    E.clear_eval_cache()
    yield request.param
    ops.set_eval_precision(old)
    E.clear_eval_cache()


SCORE_HEADROOM = 0.5       # assert_scores_close fails when an error exceeds this fraction of its tolerance (toy-sized score matrices exempt)
SCORE_ERR_LOG = []          # (test id, worst |S - ref| / (rtol |ref| + atol)) per call; conftest writes it out on the GPU box


def assert_scores_close(S, ref, rtol=RTOL, atol_rel=3e-4, scale='max'):
    """|S - ref| <= rtol*|ref| + atol_rel*scale, scale = max (or mean) |ref|.  north_star's tolerance is 1e-3
    relative.  A score is a SUM of signed cosines, each carrying an absolute error of ~1e-4 from the fp16 rounding of
    its operands whatever the sum comes to, so an element that cancels to ~0 cannot be held to a relative bound: the
    absolute term judges it against the matrix's largest score."""
    import os
    S = np.asarray(S, np.float64)
    ref = np.asarray(ref, np.float64)
    mag = np.abs(ref).max() if scale == 'max' else np.abs(ref).mean()
    atol = atol_rel * max(1e-6, mag)
    if S.shape == ref.shape and S.size:
        err = np.abs(S - ref)
        SCORE_ERR_LOG.append((os.environ.get('PYTEST_CURRENT_TEST', '?'), {
            'frac_of_tol': round(float((err / (rtol * np.abs(ref) + atol)).max()), 4),
            'max_abs_err': float('%.3g' % err.max()), 'max_abs_ref': float('%.3g' % np.abs(ref).max()),
            'mean_abs_ref': float('%.3g' % np.abs(ref).mean()),
            'max_err_over_max_ref': float('%.3g' % (err.max() / max(1e-30, np.abs(ref).max()))),
            'max_rel_err_where_ref_ge_tenth_of_max': float('%.3g' % (err / np.maximum(np.abs(ref), 1e-30))[np.abs(ref) >= 0.1 * np.abs(ref).max()].max())}))
        # the bar above was re-stated in round 3 (max- instead of mean-scaled absolute term); the measured worst
        # case of the whole suite is 0.10 of it (profiles/r03_score_err_stats.json).  A regression has to surface long before
        # it eats the tolerance: fail at HALF of it.
        # (exempt: matrices whose largest score is below 1 -- the D = 8 toy fixture, 8 components of ~0.35 rounded to 11 bits, sits at 0.8)
        assert mag < 1.0 or SCORE_ERR_LOG[-1][1]['frac_of_tol'] <= SCORE_HEADROOM, SCORE_ERR_LOG[-1]
    np.testing.assert_allclose(S, ref, rtol=rtol, atol=atol)


def test_extension_is_loaded():
    from aladin_amd import _lib
    lib = _lib.load()
    assert lib.aladin_version() == _lib.ABI_VERSION


@pytest.mark.parametrize('name', ALIGN_GOLDENS)
def test_alignment_scores_vs_reference(name):
    from aladin_amd import ops
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    S = ops.alignment_scores(T(im), T(s), il, sl).cpu().numpy()
    assert_scores_close(S, g['S_MrSw'])
    # the rank-exact evaluation precision: hi/lo split operands sit at the reference's own fp32 rounding level
    S = ops.alignment_scores(T(im), T(s), il, sl, precision='split').cpu().numpy()
    np.testing.assert_allclose(S, g['S_MrSw'], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(S, O.alignment_scores(im, s, il, sl, dtype=np.float64), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('name', ALIGN_GOLDENS)
@pytest.mark.parametrize('mode', ['MrSw', 'MrAVGw', 'MwSr', 'symm', 'sum', 'mean'])
def test_alignment_module_modes(name, mode, eval_precision):
    """Every pooling mode of alad/loss.py:120-135 against the reference's own score matrices."""
    from aladin_amd.loss import AlignmentContrastiveLoss
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=True, aggregation=mode)
    S = crit(T(im), T(s), il, sl, return_loss=False, return_similarity_mat=True).cpu().numpy()
    ref = g['S_' + mode]
    if mode in ('sum', 'mean'):
        # sums of up to R' x T' signed cosines: a score can cancel to ~0, so the error is judged against the matrix's
        # magnitude; fp32 throughout (normsum + exact-fp32 dot), whatever the evaluation precision
        np.testing.assert_allclose(S, ref, rtol=1e-4, atol=2e-6 * max(1e-6, float(np.abs(ref).max())) + 1e-6)
    elif eval_precision == 'split':
        np.testing.assert_allclose(S, ref, rtol=3e-6, atol=3e-6)
    else:
        assert_scores_close(S, ref)                      # max-pooled modes have no cancellation: the default 1e-3 rel


@pytest.mark.parametrize('name', ALIGN_GOLDENS)
def test_scan_sentences_scores_match_reference(name):
    """aggregation='scan-sentences' (alad/loss.py:136-149), fp32 end to end."""
    from aladin_amd.loss import AlignmentContrastiveLoss
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    crit = AlignmentContrastiveLoss(margin=float(g['margin']), measure='dot', max_violation=False, aggregation='scan-sentences')
    S = crit(T(im), T(s), il, sl, return_loss=False, return_similarity_mat=True)
    np.testing.assert_allclose(S.cpu().numpy(), g['S_scan-sentences'], rtol=3e-5, atol=5e-6)


@pytest.mark.parametrize('name', ['align_tiny', 'align_b16_d768'])
def test_scan_sentences_loss_and_gradients_match_reference(name):
    from aladin_amd.loss import AlignmentContrastiveLoss
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    crit = AlignmentContrastiveLoss(margin=float(g['margin']), measure='dot', max_violation=False, aggregation='scan-sentences')
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    S = crit(a, b, il, sl, return_loss=False, return_similarity_mat=True)
    (S * T(g['scan_w'])).sum().backward()
    st = int(g['scan_stride'])
    scale = max(1.0, float(np.abs(g['dim_scan']).max()))
    np.testing.assert_allclose(a.grad.cpu().numpy()[:, :, ::st], g['dim_scan'], rtol=2e-4, atol=5e-6 * scale)
    np.testing.assert_allclose(b.grad.cpu().numpy()[:, :, ::st], g['ds_scan'], rtol=2e-4, atol=5e-6 * scale)
    a2, b2 = T(im).requires_grad_(True), T(s).requires_grad_(True)
    loss = crit(a2, b2, il, sl)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g['loss_scan_sum']), rtol=2e-5)
    assert torch.isfinite(a2.grad).all() and torch.isfinite(b2.grad).all()


@pytest.mark.parametrize('shape', [(4, 6, 20, 24, 32), (7, 3, 34, 50, 100), (3, 3, 71, 71, 64), (2, 5, 2, 4, 8)])
def test_scan_sentences_ragged_vs_oracle(shape):
    """Ragged / rectangular / minimum-size batches against the oracle's closed form (whose gradient is
    pinned to the reference on full-length batches and to the NaN-free torch restatement on ragged ones)."""
    from aladin_amd import ops, synth
    Bi, Bc, R, Tn, D = shape
    im, s, il, sl = synth.alignment_batch(Bi, R, Tn, D, seed=31 + Bi, ragged=True, Bc=Bc)
    w = synth.normal((Bi, Bc), 77)
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    S = ops.alignment_scan_scores(a, b, il, sl)
    (S * T(w)).sum().backward()
    So, d_im, d_s = O.scan_sentences_scores(im, s, il, sl, dS=w)
    np.testing.assert_allclose(S.detach().cpu().numpy(), So, rtol=3e-5, atol=5e-6)
    scale = max(1e-3, float(np.abs(d_im).max()), float(np.abs(d_s).max()))
    np.testing.assert_allclose(a.grad.cpu().numpy(), d_im, rtol=2e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(b.grad.cpu().numpy(), d_s, rtol=2e-4, atol=1e-5 * scale)
    assert np.all(a.grad.cpu().numpy()[:, 0] == 0) and np.all(b.grad.cpu().numpy()[:, 0] == 0)


def test_unsupported_aggregation_raises():
    from aladin_amd.loss import AlignmentContrastiveLoss
    g = load_golden('align_tiny')
    im, s, il, sl = golden_alignment_inputs(g)
    with pytest.raises(ValueError):                      # the reference dies with a NameError (alad/loss.py:151-159)
        AlignmentContrastiveLoss(aggregation='sum-max-sentences')(T(im), T(s), il, sl, return_loss=False,
                                                                  return_similarity_mat=True)


@pytest.mark.parametrize('name', ['align_b5_d64', 'align_b12_struct', 'align_r33'])
@pytest.mark.parametrize('mode', ['MwSr', 'symm', 'sum', 'mean', 'MrAVGw'])
def test_alignment_other_modes_loss_and_gradients(name, mode, bwd_mode):
    """Loss + autograd for the non-default pooling modes against the reference's dataflow restated in
    torch on the CPU (oracle/faithful_torch.py, pinned to the goldens), with the hinge's hardest
    negatives taken from the HIP scores so that the comparison is self-consistent."""
    import faithful_torch as FT
    from aladin_amd import ops
    from aladin_amd.loss import AlignmentContrastiveLoss
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    crit = AlignmentContrastiveLoss(margin=float(g['margin']), measure='dot', max_violation=True, aggregation=mode)
    loss, S = crit(a, b, il, sl, return_similarity_mat=True)
    loss.backward()
    _, dS = O.hinge_loss(S.detach().cpu().numpy(), float(g['margin']), True, return_grad=True)
    ra, rb = torch.from_numpy(im).requires_grad_(True), torch.from_numpy(s).requires_grad_(True)
    S_ref = FT.alignment_scores_faithful(ra, rb, il, sl, mode)
    (S_ref * torch.from_numpy(dS)).sum().backward()
    np.testing.assert_allclose(loss.item(), FT.hinge_faithful(S_ref.detach(), float(g['margin']), True).item(), rtol=RTOL, atol=1e-3)
    for got, ref in ((a.grad, ra.grad), (b.grad, rb.grad)):
        # 'sum' / 'mean' never touch the alignment row step (normsum + fp32 dot): exact in both modes
        assert_grads_close(got, ref, 'exact' if mode in ('sum', 'mean') else bwd_mode, exact_atol=3e-5)


@pytest.mark.parametrize('name', ['align_b5_d64', 'align_b12_struct', 'align_b16_d768'])
def test_fused_triplet_returns_a_differentiable_score_matrix(name, bwd_mode):
    """AlignmentContrastiveLoss(..., return_similarity_mat=True) returns (loss, S) with S carrying grad, as the
    reference's does (alad/loss.py:151-159): a second loss on S back-propagates through the same node."""
    from aladin_amd.loss import AlignmentContrastiveLoss
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    margin = float(g['margin'])
    from aladin_amd import synth
    w = 0.1 * synth.normal(g['S_MrSw'].shape, 909)
    crit = AlignmentContrastiveLoss(margin=margin, measure='dot', max_violation=True, aggregation='MrSw')
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    loss, S = crit(a, b, il, sl, return_similarity_mat=True)
    assert S.requires_grad
    (2.0 * loss + (S * T(w)).sum()).backward()
    _, dS = O.hinge_loss(S.detach().cpu().numpy(), margin, True, return_grad=True)
    d_im, d_s = O.alignment_scores_backward(im, s, il, sl, 2.0 * dS + w)
    for got, ref in ((a.grad, d_im), (b.grad, d_s)):
        assert_grads_close(got, ref, bwd_mode, exact_atol=3e-5)
    # only S used: the hinge's own gradient must not leak in
    a2, b2 = T(im).requires_grad_(True), T(s).requires_grad_(True)
    _, S2 = crit(a2, b2, il, sl, return_similarity_mat=True)
    (S2 * T(w)).sum().backward()
    d_im, d_s = O.alignment_scores_backward(im, s, il, sl, w.astype(np.float64))
    for got, ref in ((a2.grad, d_im), (b2.grad, d_s)):
        assert_grads_close(got, ref, bwd_mode, exact_atol=3e-5)
    # only the loss used (the training path): unchanged sparse backward
    a3, b3 = T(im).requires_grad_(True), T(s).requires_grad_(True)
    loss3, S3 = crit(a3, b3, il, sl, return_similarity_mat=True)
    loss3.backward()
    st = int(g['grad_stride'])
    assert_grads_close(a3.grad.cpu().numpy()[:, :, ::st], g['dim_mv'], bwd_mode, exact_atol=3e-5, D=im.shape[2])


@pytest.fixture
def fp16_partners(exact_backward):
    """The library's default row step: PARTNER rows from the forward's packed fp16 operands (ALADIN_BWD_PARTNERS_FP16; the output
    row's own unit vector still comes from the raw fp32 set -- that is 'fp16-own', ALADIN_BWD_OWN_ROW_FP16, an opt-in)."""
    from aladin_amd import ops
    old = ops.set_backward_precision('fp16')
    yield
    ops.set_backward_precision(old)


@pytest.mark.parametrize('name', SQUARE_ALIGN_GOLDENS)
@pytest.mark.parametrize('tag', ['mv', 'sum'])
def test_default_backward_row_step_vs_reference(name, tag, fp16_partners, eval_precision):
    """ops.set_backward_precision('fp16') (ALADIN_BWD_PARTNERS_FP16, the library default): the row kernel takes the PARTNER rows --
    the unit vectors an output row's gradient is a weighted sum of -- from the forward's packed fp16 operands; the output row's own
    unit vector and norm still come from the raw fp32 set.
    Against the REFERENCE's gradients for the reference's dS: rtol 1e-3 + 5e-4 of the largest entry, i.e. HALF of north_star's
    1e-3 (the evidence the default rests on), arg-maxima and zero pattern exactly the exact path's.
    'fp16-own' (ALADIN_BWD_OWN_ROW_FP16: the row's own vector and inverse norm from the packed operands too) reaches 5.8e-4 on
    align_b12_struct -- D = 64, cosines near 1 -- and is therefore NOT the default (tools/experiments/bwd_precision_probe.py,
    profiles/r05_bwd_precision_probe.txt); held to 1e-3 here.  The flag words of the three modes are asserted: 0 / 1 / 17."""
    if eval_precision != 'fp16':
        pytest.skip('differentiable path only; run once')
    from aladin_amd import ops
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    d = dev()
    a, b, ilt, slt = T(im), T(s), ops.lengths_tensor(il, d), ops.lengths_tensor(sl, d)
    geom = ops.align_geometry(a.shape[0], b.shape[0], a.shape[1], b.shape[1], a.shape[2])
    packed = ops.pack_sets(a, b, ilt, slt, geom)                    # (geom, xm, xe, y, rnorm): the inverse norms make the fp16 row step possible
    assert ops._bwd_flags(packed) == 1 and ops._bwd_flags(packed[:4]) == 0          # 'fp16': partners only; no inverse norms at hand: exact
    for m, want in (('exact', 0), ('fp16-own', 17), ('fp16', 1)):
        ops.set_backward_precision(m)
        assert ops._bwd_flags(packed) == want, (m, want)
    d_im, d_s = ops._align_backward(a, b, ilt, slt, T(g['dS_' + tag]), packed=packed)
    old = ops.set_backward_precision('exact')
    e_im, e_s = ops._align_backward(a, b, ilt, slt, T(g['dS_' + tag]), packed=packed)
    ops.set_backward_precision(old)
    st = int(g['grad_stride'])
    worst = 0.0
    for got, exact, key in ((d_im, e_im, 'dim_'), (d_s, e_s, 'ds_')):
        got, exact = got.cpu().numpy(), exact.cpu().numpy()
        ref = g[key + tag]
        scale = max(1e-9, float(np.abs(ref).max()))
        np.testing.assert_allclose(got[:, :, ::st], ref, rtol=1e-3, atol=5e-4 * scale)
        worst = max(worst, float(np.abs(got[:, :, ::st] - ref).max()) / scale)
        assert np.array_equal(got == 0, exact == 0) or np.abs(got[(got == 0) != (exact == 0)]).max() < 1e-6 * scale
        assert not np.array_equal(got, exact) or D_is_tiny(im)        # the opt-in really took the other path
    assert worst < 5e-4
    ops.set_backward_precision('fp16-own')
    o_im, o_s = ops._align_backward(a, b, ilt, slt, T(g['dS_' + tag]), packed=packed)
    for got, key in ((o_im, 'dim_'), (o_s, 'ds_')):
        ref = g[key + tag]
        np.testing.assert_allclose(got.cpu().numpy()[:, :, ::st], ref, rtol=1e-3, atol=1e-3 * max(1e-9, float(np.abs(ref).max())))


def D_is_tiny(im):
    return im.shape[2] < 16


def test_b256_structured_triplet_step_default_backward(fp16_partners, eval_precision):
    """The default row step on the fused training step at BASELINE size (aladin_align_triplet_fwd / _bwd) against the
    oracle: rtol 1e-3 + 5e-4 of the largest entry.  (The bench batch itself: test_b256_triplet_step_gradients_vs_oracle[fp16-False].)"""
    if eval_precision != 'fp16':
        pytest.skip('differentiable path only; run once')
    from aladin_amd import synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    B = 256
    im, s, il, sl = synth.structured_alignment_batch(B, 34, 50, 768, seed=77, noise=3.0, ragged=True)
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    loss, S = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(a, b, il, sl, return_similarity_mat=True)
    loss.backward()
    _, dS = O.hinge_loss(S.detach().cpu().numpy(), 0.2, True, return_grad=True)
    dim, ds = O.alignment_scores_backward(im, s, il, sl, dS)
    for got, ref in ((a.grad, dim), (b.grad, ds)):
        scale = max(1e-9, float(np.abs(ref).max()))
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-3, atol=5e-4 * scale)


def test_backward_limits_are_reported_at_forward_time():
    """The one shape limit of the differentiable path (D > 1024) fails when the differentiable forward is requested -- not later
    inside loss.backward() -- and still scores fine without autograd."""
    from aladin_amd import ops, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    for D in (1028, 1030):
        im, s, il, sl = synth.alignment_batch(4, 10, 12, D, seed=3, ragged=True)
        a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
        with pytest.raises(ValueError, match='differentiable'):
            ops.alignment_scores(a, b, il, sl)
        with pytest.raises(ValueError, match='differentiable'):
            AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(a, b, il, sl)
        with torch.no_grad():
            S = ops.alignment_scores(a, b, il, sl, precision='fp16')
        assert_scores_close(S.cpu().numpy(), O.alignment_scores(im, s, il, sl))


@pytest.mark.parametrize('D', [30, 65, 127])
def test_feature_sizes_that_are_not_multiples_of_four(D, bwd_mode):
    """The reference takes any feature size; the backward kernels move float4 columns.  The public entry points pad such sets with
    zero features outside their autograd nodes (ops._pad_features): scores, loss and gradients -- in the callers' own D -- against
    the oracle, through the score node, the fused triplet node and the loss heads."""
    from aladin_amd import ops, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    im, s, il, sl = synth.structured_alignment_batch(9, 20, 18, D, seed=40 + D, noise=3.0, ragged=True)
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    S = ops.alignment_scores(a, b, il, sl)
    assert_scores_close(S.detach().cpu().numpy(), O.alignment_scores(im, s, il, sl))
    w = synth.normal(S.shape, 77).astype(np.float32)
    (S * T(w)).sum().backward()
    dim, ds = O.alignment_scores_backward(im, s, il, sl, w.astype(np.float64))
    assert a.grad.shape == a.shape and b.grad.shape == b.shape
    assert_grads_close(a.grad, dim, bwd_mode, exact_atol=5e-5)
    assert_grads_close(b.grad, ds, bwd_mode, exact_atol=5e-5)
    a2, b2 = T(im).requires_grad_(True), T(s).requires_grad_(True)
    loss, S2 = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(a2, b2, il, sl, return_similarity_mat=True)
    loss.backward()
    ref_loss, dS = O.hinge_loss(S2.detach().cpu().numpy(), 0.2, True, return_grad=True)
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-5, atol=1e-6)
    dim, ds = O.alignment_scores_backward(im, s, il, sl, dS)
    assert_grads_close(a2.grad, dim, bwd_mode, exact_atol=5e-5)
    assert_grads_close(b2.grad, ds, bwd_mode, exact_atol=5e-5)


def test_alignment_scores_permuted_view_input():
    """forward_loss hands (S,B,D)->(B,S,D) permuted views (reference alad_model.py:377-378)."""
    from aladin_amd import ops
    g = load_golden('align_b5_d64')
    im, s, il, sl = golden_alignment_inputs(g)
    im_v = T(im.transpose(1, 0, 2).copy()).permute(1, 0, 2)
    s_v = T(s.transpose(1, 0, 2).copy()).permute(1, 0, 2)
    assert not im_v.is_contiguous()
    assert_scores_close(ops.alignment_scores(im_v, s_v, il, sl).cpu().numpy(), g['S_MrSw'])


@pytest.mark.parametrize('name', SQUARE_ALIGN_GOLDENS)
@pytest.mark.parametrize('tag', ['mv', 'sum'])
def test_hinge_on_reference_scores(name, tag):
    from aladin_amd import ops
    g = load_golden(name)
    S = T(g['S_MrSw']).requires_grad_(True)
    loss = ops.hinge_loss(S, float(g['margin']), tag == 'mv')
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g['loss_' + tag]), rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(S.grad.cpu().numpy(), g['dS_' + tag])


@pytest.mark.parametrize('name', SQUARE_ALIGN_GOLDENS)
@pytest.mark.parametrize('tag', ['mv', 'sum'])
@pytest.mark.parametrize('path', ['f32', 'packed'])
def test_alignment_backward_kernel_vs_reference(name, tag, path):
    """K1b in isolation: the reference's dS in, the reference's input gradients out.  'f32' is the
    stand-alone entry point (fp32 MFMA argmax recompute), 'packed' the one used by autograd (fp16
    MFMA block + exact fp32 re-decision of close calls); both must give the fp32 argmax."""
    from aladin_amd import ops
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    d = dev()
    a, b, ilt, slt = T(im), T(s), ops.lengths_tensor(il, d), ops.lengths_tensor(sl, d)
    packed = None
    if path == 'packed':
        geom = ops.align_geometry(a.shape[0], b.shape[0], a.shape[1], b.shape[1], a.shape[2])
        xm, xe = ops.pack_images(a, ilt, geom)
        packed = (geom, xm, xe, ops.pack_captions(b, slt, geom))
    d_im, d_s = ops._align_backward(a, b, ilt, slt, T(g['dS_' + tag]), packed=packed)
    st = int(g['grad_stride'])
    for got, key in ((d_im, 'dim_'), (d_s, 'ds_')):
        got = got.cpu().numpy()
        ref = g[key + tag]
        scale = max(1e-9, float(np.abs(ref).max()))
        np.testing.assert_allclose(got[:, :, ::st], ref, rtol=1e-3, atol=2e-5 * scale)
        np.testing.assert_allclose(np.abs(got).sum(), float(g[key + 'abs_' + tag]), rtol=1e-4)
    # dropped / padded positions get exactly zero gradient
    got = d_im.cpu().numpy()
    assert np.all(got[:, 0, :] == 0)
    for i, L in enumerate(il):
        assert np.all(got[i, L:, :] == 0)
    got = d_s.cpu().numpy()
    assert np.all(got[:, 0, :] == 0)
    for j, L in enumerate(sl):
        assert np.all(got[j, max(L - 2, 1):, :] == 0)


@pytest.mark.parametrize('name', ['align_b5_d64', 'align_b12_struct', 'align_b16_d768'])
def test_alignment_loss_end_to_end(name, bwd_mode):
    """Module forward + autograd backward against the oracle chained on the HIP scores (the hardest
    negative is an argmax over scores, so the check is made self-consistent with the fp16 S)."""
    from aladin_amd.loss import AlignmentContrastiveLoss
    g = load_golden(name)
    im, s, il, sl = golden_alignment_inputs(g)
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    crit = AlignmentContrastiveLoss(margin=float(g['margin']), measure='dot', max_violation=True, aggregation='MrSw')
    loss, S = crit(a, b, il, sl, return_similarity_mat=True)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g['loss_mv']), rtol=RTOL, atol=1e-3)
    S_np = S.detach().cpu().numpy()
    _, dS = O.hinge_loss(S_np, float(g['margin']), True, return_grad=True)
    dim, ds = O.alignment_scores_backward(im, s, il, sl, dS)
    assert_grads_close(a.grad, dim, bwd_mode)
    assert_grads_close(b.grad, ds, bwd_mode)


@pytest.mark.parametrize('name', ['match_b16_d768', 'match_b7_d64'])
def test_matching_head(name):
    from aladin_amd import synth
    from aladin_amd.loss import ContrastiveLoss
    g = load_golden(name)
    img, cap = synth.global_embeddings(int(g['B']), int(g['D']), int(g['seed']), float(g['noise']))
    for tag, mv in (('mv', True), ('sum', False)):
        a, b = T(img).requires_grad_(True), T(cap).requires_grad_(True)
        loss, M = ContrastiveLoss(margin=0.2, measure='dot', max_violation=mv)(a, b, return_similarity_mat=True)
        loss.backward()
        np.testing.assert_allclose(M.detach().cpu().numpy(), g['M'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(loss.item(), float(g['loss_' + tag]), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(a.grad.cpu().numpy(), g['dimg_' + tag], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(b.grad.cpu().numpy(), g['dcap_' + tag], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('name', ['distill_b16', 'distill_b5'])
def test_listnet(name):
    from aladin_amd.loss import DistillationLoss
    g = load_golden(name)
    st = T(g['student']).requires_grad_(True)
    loss = DistillationLoss(mode='listnet')(T(g['teacher']), st)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g['loss_listnet']), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st.grad.cpu().numpy(), g['dstudent_listnet'], rtol=1e-4, atol=1e-7)


DISTILL_CASES = ['distill_b16', 'distill_b5', 'distill_b24_thr', 'distill_b40_neg']


@pytest.mark.parametrize('mode', ['mse', 'ordinal', 'contrastive'])
@pytest.mark.parametrize('name', DISTILL_CASES)
def test_distillation_modes_match_reference(name, mode):
    """DistillationLoss 'mse' / 'ordinal' / 'contrastive' (alad/loss.py:371-425) vs the reference's
    loss and autograd gradients; the integer-valued gradients of the two hinge modes must be exact."""
    from aladin_amd.loss import DistillationLoss
    g = load_golden(name)
    crit = DistillationLoss(mode=mode, margin=float(g['margin']), threshold=float(g['threshold']),
                            stride=int(g["stride"])).to(dev())
    if mode == 'mse':
        with torch.no_grad():
            crit.wb.copy_(T(g['wb_mse']))
    st = T(g['student']).requires_grad_(True)
    loss = crit(T(g['teacher']), st)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g['loss_' + mode]), rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(st.grad.cpu().numpy(), g['dstudent_' + mode], rtol=1e-5, atol=1e-8)
    if mode == 'mse':
        np.testing.assert_allclose(crit.wb.grad.cpu().numpy(), g['dwb_mse'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('B', [3, 64, 257, 1000])
def test_distillation_modes_vs_oracle_random(B):
    """Sizes the goldens do not reach (non power-of-two sort lengths, multi-pass lines), incl. upstream
    gradient scaling and strided (non-contiguous) inputs."""
    from aladin_amd import ops
    rng = np.random.default_rng(B)
    teacher = rng.standard_normal((B, B)).astype(np.float32)
    student = rng.standard_normal((B, 2 * B)).astype(np.float32)
    st_full = T(student).requires_grad_(True)
    st = st_full[:, ::2]                                            # stride(1) == 2
    wb = T(np.array([0.7, -0.2], np.float32)).requires_grad_(True)
    stride = 1 if B < 8 else 3
    for mode, ref in (('mse', lambda: O.distill_mse(teacher, student[:, ::2], [0.7, -0.2], True)),
                      ('contrastive', lambda: O.distill_contrastive(teacher, student[:, ::2], 0.2, True)),
                      ('ordinal', lambda: O.distill_ordinal(teacher, student[:, ::2], 0.2, 0.1, stride, True))):
        st_full.grad = None
        wb.grad = None
        loss = ops.distillation_loss(T(teacher), st, mode, 0.2, 0.1, stride, wb=wb if mode == 'mse' else None)
        (2.5 * loss).backward()
        out = ref()
        np.testing.assert_allclose(loss.item(), out[0], rtol=5e-6, atol=1e-6)
        got = st_full.grad.cpu().numpy()
        np.testing.assert_array_equal(got[:, 1::2], 0)
        np.testing.assert_allclose(got[:, ::2], 2.5 * out[1], rtol=1e-5, atol=1e-9)
        if mode == 'mse':
            np.testing.assert_allclose(wb.grad.cpu().numpy(), 2.5 * out[2], rtol=1e-5, atol=1e-6)


def test_ordinal_empty_selection_is_nan_loss_zero_grad():
    from aladin_amd import ops
    B = 12
    teacher = T(np.full((B, B), -5.0, np.float32) + np.arange(B * B, dtype=np.float32).reshape(B, B) * 1e-3)
    st = T(np.random.default_rng(0).standard_normal((B, B)).astype(np.float32)).requires_grad_(True)
    loss = ops.distillation_loss(teacher, st, 'ordinal', 0.2, 0.1, 3)      # no teacher score reaches 0.1
    loss.backward()
    assert np.isnan(loss.item())
    np.testing.assert_array_equal(st.grad.cpu().numpy(), 0)


@pytest.mark.parametrize('name', ['order_b12', 'order_rect'])
def test_order_sim_matches_reference(name):
    """order_sim (alad/loss.py:20-26) and ContrastiveLoss(measure='order') vs the reference."""
    from aladin_amd import synth
    from aladin_amd.loss import ContrastiveLoss, order_sim
    g = load_golden(name)
    im = synth.normal((int(g['Bi']), int(g['D'])), int(g['seed']))
    s = synth.normal((int(g['Bc']), int(g['D'])), int(g['seed']) + 1)
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    scores = order_sim(a, b)
    (scores * T(g['w'])).sum().backward()
    np.testing.assert_allclose(scores.detach().cpu().numpy(), g['scores'], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(a.grad.cpu().numpy(), g['dim'], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(b.grad.cpu().numpy(), g['ds'], rtol=1e-4, atol=2e-6)
    if 'loss_mv' in g:
        a2, b2 = T(im).requires_grad_(True), T(s).requires_grad_(True)
        loss = ContrastiveLoss(margin=0.2, measure='order', max_violation=True)(a2, b2)
        loss.backward()
        np.testing.assert_allclose(loss.item(), float(g['loss_mv']), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(a2.grad.cpu().numpy(), g['dim_mv'], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(b2.grad.cpu().numpy(), g['ds_mv'], rtol=1e-4, atol=2e-6)


def test_order_sim_vs_oracle_odd_shapes():
    from aladin_amd import ops
    rng = np.random.default_rng(5)
    for Bi, Bc, D in ((1, 1, 1), (33, 70, 100), (300, 65, 768)):
        im = rng.standard_normal((Bi, D)).astype(np.float32)
        s = rng.standard_normal((Bc, D)).astype(np.float32) + 0.5
        G = rng.standard_normal((Bi, Bc)).astype(np.float32)
        a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
        sc = ops.order_scores(a, b)
        (sc * T(G)).sum().backward()
        np.testing.assert_allclose(sc.detach().cpu().numpy(), O.order_scores(im, s), rtol=3e-6, atol=1e-6)
        di, ds = O.order_scores_backward(im, s, G)
        scale = max(1.0, float(np.abs(di).max()))
        np.testing.assert_allclose(a.grad.cpu().numpy(), di, rtol=1e-4, atol=2e-5 * scale, equal_nan=True)
        np.testing.assert_allclose(b.grad.cpu().numpy(), ds, rtol=1e-4, atol=2e-5 * scale, equal_nan=True)


def test_model_forward_matches_reference_configs():
    import yaml  # noqa: F401
    from aladin_amd import synth
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.evaluation import LogCollector
    g = load_golden('model_forward')
    B, R, Tn, D, seed = (int(g[k]) for k in ('B', 'R', 'T', 'D', 'seed'))
    im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, D, seed, 1.0, True)
    img_emb, cap_emb = synth.global_embeddings(B, D, seed + 1, 1.0)
    sets = (T(img_emb), T(cap_emb), T(im).permute(1, 0, 2), T(s).permute(1, 0, 2), il, sl, 0)
    for fn in g['configs']:
        key = str(fn)[:-5].replace('-', '_').replace('.', '_')
        lt = str(g[key + '__loss_type'])
        config = {'training': {'loss-type': lt, 'loss-weights': [float(w) for w in g[key + '__weights']],
                               'margin': 0.2, 'measure': 'dot', 'max-violation': True, 'alignment-mode': 'MrSw',
                               'distillation-mode': 'listnet'}}
        m = ALADModel(config)
        m.logger = LogCollector()
        m.forward_emb = lambda a, b, _s=sets: _s
        for epoch in (0, 5):
            loss, d = m.forward(None, None, epoch=epoch, distill_epoch=2)
            assert list(d.keys()) == [str(k) for k in g['%s__e%d_keys' % (key, epoch)]]
            np.testing.assert_allclose([float(v) for v in d.values()], g['%s__e%d_vals' % (key, epoch)], rtol=RTOL, atol=1e-4)
            np.testing.assert_allclose(float(loss), float(g['%s__e%d_total' % (key, epoch)]), rtol=RTOL, atol=1e-4)
        assert list(m.logger.meters.keys()) == [str(k) for k in g[key + '__logged']]
        assert m.Eiters == int(g[key + '__eiters'])


@pytest.mark.parametrize('name', ['recall_n500', 'recall_n5000'])
def test_recall_vs_reference(name):
    from aladin_amd import synth
    from aladin_amd import evaluation as E
    g = load_golden(name)
    img, cap = synth.retrieval_embeddings(int(g['n_img']), int(g['D']), int(g['seed']), float(g['sigma']))
    sim = E.compute_sim_matrix(img[0::5], cap).cpu().numpy()
    np.testing.assert_allclose(sim, img[0::5].astype(np.float64) @ cap.astype(np.float64).T, rtol=0, atol=2e-6)
    np.testing.assert_allclose(E.compute_recall(img, cap, verbose=False), g['compute_recall'], rtol=0, atol=1e-9)
    for mode in ('i2t', 't2i'):
        m, (ranks, top1) = E.recall(img, cap, None, mode=mode, return_ranks=True)
        np.testing.assert_allclose(m, g[mode + '_metrics'], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(ranks, g[mode + '_ranks'])
        np.testing.assert_array_equal(top1, g[mode + '_top1'])


def test_recall_1k_5fold_and_recall_test_vs_reference():
    """recall_1k_5fold_test / recall_test (alad/recall_auxiliary.py:72-130) against the reference's tuples."""
    from aladin_amd import synth
    from aladin_amd import evaluation as E
    g = load_golden('recall_5fold')
    img, cap = synth.retrieval_embeddings(int(g['n_img']), int(g['D']), int(g['seed']), float(g['sigma']))
    np.testing.assert_allclose(E.recall_1k_5fold_test(img, cap, verbose=False), g['recall_1k_5fold_test'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(E.recall_test(img[:5000], cap[:5000], None, None), g['recall_test_fold0'], rtol=0, atol=1e-9)
    np.testing.assert_allclose(E.recall_test(T(img[:5000]), T(cap[:5000]), None, None), g['recall_test_fold0'], rtol=0, atol=1e-9)


def _alignment_sim_fn():
    """The reference's evaluation closure (train.py:493-500 / test.py:259-264) over THIS package's loss module."""
    from aladin_amd.loss import AlignmentContrastiveLoss
    crit = AlignmentContrastiveLoss(aggregation='MrSw')

    def alignment_sim_fn(img, cap, img_len, cap_len):
        with torch.no_grad():
            return crit(img, cap, img_len, cap_len, return_loss=False, return_similarity_mat=True)
    return alignment_sim_fn


def test_eval_i2t_t2i_vs_reference(eval_precision):
    from aladin_amd import synth
    from aladin_amd import evaluation as E
    g = load_golden('eval_sets')
    images, captions, il, cl = synth.eval_sets(int(g['n_img']), int(g['D']), int(g['seed']))
    S = E.compute_sim_matrix(images[0::5], captions, il[0::5], cl, mode='alignment').cpu().numpy()
    assert_scores_close(S, g['S_eval'])
    if eval_precision == 'split':
        np.testing.assert_allclose(S, g['S_eval'], rtol=2e-6, atol=2e-6)
    # matching head (sim_function=None): exact ranks, and the (ranks, top50) return of t2i (:262,309,324-325)
    m, (r, t1) = E.i2t(images, captions, il, cl, return_ranks=True)
    np.testing.assert_allclose(m, g['i2t_match_metrics'], atol=1e-9)
    np.testing.assert_array_equal(r, g['i2t_match_ranks'])
    np.testing.assert_array_equal(t1, g['i2t_match_top1'])
    m, (r, top50) = E.t2i(images, captions, il, cl, return_ranks=True)
    np.testing.assert_allclose(m, g['t2i_match_metrics'], atol=1e-9)
    np.testing.assert_array_equal(r, g['t2i_match_ranks'])
    assert top50.shape == (len(cl), 50) and top50.dtype == np.float64
    np.testing.assert_array_equal(top50, g['t2i_match_top50'])
    if eval_precision != 'split':
        return            # fp16 operands: near-ties may swap (measured in test_alignment_head_retrieval_coco1k)
    # alignment head through the reference-style closure and through the string form: the reference's ranks
    for fn in (_alignment_sim_fn(), 'alignment'):
        m, (r, t1) = E.i2t(images, captions, il, cl, return_ranks=True, sim_function=fn, cap_batches=5)
        np.testing.assert_array_equal(r, g['i2t_align_ranks'])
        np.testing.assert_array_equal(t1, g['i2t_align_top1'])
        np.testing.assert_allclose(m, g['i2t_align_metrics'], atol=1e-9)
        m, (r, top50) = E.t2i(images, captions, il, cl, return_ranks=True, sim_function=fn, im_batches=5)
        np.testing.assert_array_equal(r, g['t2i_align_ranks'])
        np.testing.assert_allclose(m, g['t2i_align_metrics'], atol=1e-9)
        # the whole descending order of every caption's 50 images, wherever the reference's own scores separate
        # neighbours by more than its fp32 rounding (S_eval is the reference's matrix)
        srt = -np.sort(-g['S_eval'].astype(np.float64), axis=0)                    # (50, n_cap)
        clear = np.concatenate([np.ones((1, srt.shape[1]), bool), (srt[:-1] - srt[1:]) > 1e-5], 0)
        clear = clear & np.concatenate([clear[1:], np.ones((1, srt.shape[1]), bool)], 0)
        ref50 = g['t2i_align_top50'].astype(np.float64)
        assert clear.mean() > 0.99
        np.testing.assert_array_equal(top50[clear.T], ref50[clear.T])


@pytest.mark.parametrize('fixture', ['eval_coco1k', 'eval_coco1k_d768'])
def test_alignment_head_retrieval_coco1k(eval_precision, fixture):
    """SURVEY 8(f) row 1 at the size north_star quotes (COCO-1k: 1000 images x 5000 captions, sets padded to 71
    positions): i2t / t2i with the reference's own alignment_sim_fn protocol against the ranks the REFERENCE's
    loops produced on the same inputs (tests/golden/eval_coco1k.npz at D = 64 and eval_coco1k_d768.npz at the HEADLINE
    feature width D = 768 -- north_star's Recall@1 claim is for 768-d features; made by tests/golden/make_golden.py, 17 min of
    the reference's loops for the latter).
    In the evaluation precision the ranks are the reference's: equal for every query whose ground-truth score
    the reference's own fp32 arithmetic separates from its competitors (gap > 2e-5; the fixture stores the gaps),
    within one place otherwise -- hence Recall@K identical."""
    from aladin_amd import synth
    from aladin_amd import evaluation as E
    g = load_golden(fixture)
    n_img = int(g['n_img'])
    images, captions, il, cl = synth.eval_sets(n_img, int(g['D']), int(g['seed']), base_weight=float(g['gen_base_weight']),
                                               img_len_range=tuple(int(v) for v in g['gen_img_len_range']),
                                               cap_len_range=tuple(int(v) for v in g['gen_cap_len_range']),
                                               n_full=int(g['gen_n_full']))
    assert il == [int(v) for v in g['img_len']] and cl == [int(v) for v in g['cap_len']]
    assert abs(synth.checksum(images) - float(g['images_checksum'])) <= 1e-6 * abs(float(g['images_checksum']))
    assert max(il) == 71                                             # some images fill the padded set: no zero-fill in their max
    images_d, captions_d = T(images), T(captions)
    fn = _alignment_sim_fn()
    m_i, (r_i, t1_i) = E.i2t(images_d, captions_d, il, cl, return_ranks=True, sim_function=fn, cap_batches=5)
    m_t, (r_t, top50) = E.t2i(images_d, captions_d, il, cl, return_ranks=True, sim_function=fn, im_batches=1)
    ref_ri, ref_rt = g['i2t_ranks'].astype(np.float64), g['t2i_ranks'].astype(np.float64)
    if eval_precision == 'fp16':
        # training operands: ~1e-4 on a score; near-ties swap.  Recorded, not the evaluation default.
        agree_i, agree_t = np.mean(r_i == ref_ri), np.mean(r_t == ref_rt)
        print('fp16 operands at COCO-1k: ranks equal i2t %.4f t2i %.4f; R@1 %.2f/%.2f (ref %.2f/%.2f)'
              % (agree_i, agree_t, m_i[0], m_t[0], g['i2t_metrics'][0], g['t2i_metrics'][0]))
        assert agree_i > 0.97 and agree_t > 0.97
        np.testing.assert_allclose(m_i[:3], g['i2t_metrics'][:3], atol=0.5)
        np.testing.assert_allclose(m_t[:3], g['t2i_metrics'][:3], atol=0.5)
        return
    # a query is "unresolved" when the reference's own fp32 scores separate its ground truth from a competitor by less
    # than TAU: 2e-5 on the D = 64 fixture (scores ~ 5), 1e-5 on the D = 768 one (scores ~ 1.5-2.5, fp32 ulp 2.4e-7, 768-term
    # dot products and ~20-word sums in the reference's bmm)
    TAU = {'eval_coco1k': 2e-5, 'eval_coco1k_d768': 1e-5}[fixture]
    amb_i, amb_t = g['i2t_gap'] < TAU, g['t2i_gap'] < TAU
    assert amb_i.sum() <= 0.03 * n_img and amb_t.sum() <= 0.01 * 5 * n_img, (amb_i.sum(), amb_t.sum())     # a few % of the queries at most
    np.testing.assert_array_equal(r_i[~amb_i], ref_ri[~amb_i])
    np.testing.assert_array_equal(r_t[~amb_t], ref_rt[~amb_t])
    assert np.all(np.abs(r_i - ref_ri)[amb_i] <= 1) and np.all(np.abs(r_t - ref_rt)[amb_t] <= 1)
    # Recall@{1,5,10}, medr, meanr: the reference's numbers (an unresolved query could move R@K by 100/n at most)
    slack_i, slack_t = 100.0 * amb_i.sum() / n_img + 1e-9, 100.0 * amb_t.sum() / (5 * n_img) + 1e-9
    np.testing.assert_allclose(m_i[:3], g['i2t_metrics'][:3], atol=slack_i)
    np.testing.assert_allclose(m_t[:3], g['t2i_metrics'][:3], atol=slack_t)
    if not amb_i.any():
        np.testing.assert_allclose(m_i, g['i2t_metrics'], atol=1e-9)
    if not amb_t.any():
        np.testing.assert_allclose(m_t, g['t2i_metrics'], atol=1e-9)
    # top lists
    ok = g['i2t_top1_gap'] > TAU
    np.testing.assert_array_equal(t1_i[ok], g['i2t_top1'].astype(np.float64)[ok])
    ok = g['t2i_top10_gap'] > TAU                                   # ten gaps per caption: ~1 % of the captions have a close pair
    assert ok.mean() > 0.97
    np.testing.assert_array_equal(top50[ok, :10], g['t2i_top10'].astype(np.float64)[ok])
    # the scores themselves against a sample of the reference's matrix and its diagonal
    S = E.compute_sim_matrix(images_d[0::5], captions_d, il[0::5], cl, mode='alignment').cpu().numpy()
    np.testing.assert_allclose(S[0::20, 0::10], g['S_sample'], rtol=0, atol=4e-6)
    np.testing.assert_allclose(S[np.arange(5 * n_img) // 5, np.arange(5 * n_img)], g['S_diag'], rtol=0, atol=4e-6)
    # the packed split store gives the same bits, hence the same ranks
    si, sc = _fill_stores(images, captions, il, cl, batch=500, precision='split')
    assert torch.equal(E.compute_sim_matrix(si.view(slice(0, None, 5)), sc, mode='alignment').cpu(), torch.from_numpy(S))


@pytest.mark.parametrize('img_range,cap_range,n_full', [((6, 70), (5, 66), 4), ((24, 56), (7, 30), 0)],
                         ids=['whole-range', 'coco-long-images'])
def test_length_bucketed_grid_equals_the_single_launch(eval_precision, monkeypatch, img_range, cap_range, n_full):
    """ops.bucket_plan: a large ragged evaluation grid is scored in length classes (each image / caption pays for the tile
    class of its own length).  Against the single launch over the whole grid (the planner switched off) and the oracle, for
    (N, 71, D) tensors and for packed stores; images that fill the padded set (no zero fill in their max) included.  The
    second case is a COCO-like shape with images of up to 55 boxes + the global slot: its long images sit in the 48-row classes
    (48 and 48 + side rows), the single launch scores all of them there."""
    from aladin_amd import evaluation as E, ops, synth
    n_img, D = 90, 128
    images, captions, il, cl = synth.eval_sets(n_img, D, seed=77, img_len_range=img_range, cap_len_range=cap_range, n_full=n_full)
    assert max(il) == (71 if n_full else img_range[1]) and min(il) < 30
    ims, ils = images[0::5], il[0::5]
    monkeypatch.setattr(ops, 'BUCKET_MIN_PAIRS', 1)
    monkeypatch.setattr(ops, 'BUCKET_MIN_SAMPLES', 8)
    ops._PLAN_CACHE.clear()
    plan = ops.bucket_plan(ops._needed_positions(ils, 0, 71, True), ops._needed_positions(cl, 2, 71, False))
    assert plan is not None and len(plan[0]) >= 3 and len(plan[1]) >= (3 if n_full else 2)
    assert sorted(k for g in plan[0] for k in g) == list(range(n_img)) and sorted(k for g in plan[1] for k in g) == list(range(5 * n_img))
    S_b = E.compute_sim_matrix(T(ims), T(captions), ils, cl, mode='alignment')
    si, sc = _fill_stores(images, captions, il, cl, batch=53)
    S_bs = E.compute_sim_matrix(si.view(slice(0, None, 5)), sc, mode='alignment')
    assert torch.equal(S_b, S_bs)                                   # same plan, same operand bits
    monkeypatch.setattr(ops, 'bucket_plan', lambda *a: None)
    ops._PLAN_CACHE.clear()
    S_1 = E.compute_sim_matrix(T(ims), T(captions), ils, cl, mode='alignment')
    ops._PLAN_CACHE.clear()
    ref = O.alignment_scores(ims, captions, ils, cl, dtype=np.float64)
    if eval_precision == 'split':
        np.testing.assert_allclose(S_b.cpu().numpy(), S_1.cpu().numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(S_b.cpu().numpy(), ref, rtol=2e-6, atol=3e-6)
    else:
        assert_scores_close(S_b.cpu().numpy(), S_1.cpu().numpy(), rtol=1e-6, atol_rel=1e-6)      # same operands: summation order only
        assert_scores_close(S_b.cpu().numpy(), ref)


def _fill_store(sets, lens, tail, precision, batch=4):
    from aladin_amd.store import PackedSetStore
    st = PackedSetStore(sets.shape[2], tail, dev(), capacity_rows=64, precision=precision)
    for k0 in range(0, sets.shape[0], batch):
        k1 = min(sets.shape[0], k0 + batch)
        st.append(T(sets[k0:k1, :max(lens[k0:k1])]), lens[k0:k1])
    return st


def test_eval_pipeline_encode_data_to_ranks_vs_reference(eval_precision):
    """The evaluation pipeline end to end against the REFERENCE's own (tests/golden/eval_pipeline.npz: its encode_data over a
    loader of encoder batches, then its i2t / t2i): our encode_data fills the same (N, 71, D) buffers (on the device) bit
    for bit, and both heads give the reference's ranks from them -- and from the packed stores of encode_data_packed."""
    from aladin_amd import synth
    from aladin_amd import evaluation as E
    g = load_golden('eval_pipeline')
    batches = synth.encoder_batches()
    N = int(g['N'])

    class FakeModel:
        logger = None

        def eval(self):
            pass

        def forward_emb(self, example_imgs, example_txts):
            b = batches[int(example_txts[0][0])]
            return (T(b['img_glob']), T(b['cap_glob']), T(b['img_set']), T(b['cap_seq']), list(b['img_len']), list(b['cap_len']), 0)

    class Loader(list):
        dataset = list(range(N))
    loader = Loader([((torch.zeros((len(b['img_len']), 1)),), (torch.full((len(b['img_len']),), k),)) for k, b in enumerate(batches)])
    img_embs, cap_embs, il, cl = E.encode_data(FakeModel(), loader, logging=None)
    assert img_embs.is_cuda and tuple(img_embs.shape) == (N, 71, 64)
    assert il == [int(v) for v in g['img_len']] and cl == [int(v) for v in g['cap_len']]
    np.testing.assert_array_equal(img_embs.cpu().numpy()[:, :, ::8], g['img_embs_s'])
    np.testing.assert_array_equal(cap_embs.cpu().numpy()[:, :, ::8], g['cap_embs_s'])
    assert abs(synth.checksum(cap_embs.cpu().numpy()) - float(g['cap_embs_checksum'])) <= 1e-9 * abs(float(g['cap_embs_checksum']))
    si, sc, il2, cl2 = E.encode_data_packed(FakeModel(), loader, logging=None, precision=eval_precision)
    assert il2 == il and cl2 == cl
    for tag, fn in (('match', None), ('align', 'alignment')):
        if tag == 'align' and eval_precision != 'split':
            continue                                   # fp16 operands: near-ties may swap (measured in the coco1k test)
        for srcs in ((img_embs, cap_embs), (si, sc)):
            m, (r, t1) = E.i2t(srcs[0], srcs[1], il, cl, return_ranks=True, sim_function=fn)
            np.testing.assert_array_equal(r, g['i2t_%s_ranks' % tag])
            np.testing.assert_array_equal(t1, g['i2t_%s_top1' % tag])
            np.testing.assert_allclose(m, g['i2t_%s_metrics' % tag], atol=1e-9)
            m, (r, top50) = E.t2i(srcs[0], srcs[1], il, cl, return_ranks=True, sim_function=fn)
            np.testing.assert_array_equal(r, g['t2i_%s_ranks' % tag])
            np.testing.assert_array_equal(top50[:, 0], g['t2i_%s_top1' % tag])
            np.testing.assert_allclose(m, g['t2i_%s_metrics' % tag], atol=1e-9)


def test_trimmed_grid_keeps_the_zero_fill_of_the_longest_image(eval_precision):
    """A word whose cosine with EVERY region of an image is negative contributes max(negatives, 0) = 0 when the
    image is shorter than the padded set (masked regions are zero-filled and take part in the max,
    alad/loss.py:116,124) -- also for the LONGEST image of the evaluation set, which loses its padding when the
    grid is trimmed to the longest length -- and the negative maximum itself only when the image fills the set."""
    from aladin_amd import evaluation as E
    L, D, n_img, n_cap = 71, 16, 6, 10
    rng = np.random.default_rng(5)
    images = np.zeros((n_img, L, D), np.float32)
    captions = np.zeros((n_cap, L, D), np.float32)
    il = [20, 33, 33, 12, 33, 25]                                  # three images share the maximum length
    cl = [9, 12, 5, 7, 12, 6, 4, 10, 8, 11]
    for i, n in enumerate(il):
        images[i, :n] = rng.standard_normal((n, D))
    for j, n in enumerate(cl):
        captions[j, :n] = rng.standard_normal((n, D))
    images[1, 1:33, 0] = -np.abs(images[1, 1:33, 0]) - 3.0         # image 1 (a longest one): all regions point to -e0 ...
    captions[4, 2] = 0
    captions[4, 2, 0] = 5.0                                        # ... and caption 4's word 1 is +e0: every cosine < 0
    ref = O.alignment_scores(images, captions, il, cl, dtype=np.float64)      # on the padded-71 sets, like the reference
    S = E.compute_sim_matrix(images, captions, il, cl, mode='alignment').cpu().numpy()
    tol = 2e-6 if eval_precision == 'split' else 2e-3
    np.testing.assert_allclose(S, ref, rtol=0, atol=tol)
    # ... and the same from packed stores
    si, sc = _fill_store(images, il, 0, eval_precision), _fill_store(captions, cl, 2, eval_precision)
    np.testing.assert_array_equal(E.compute_sim_matrix(si, sc, mode='alignment').cpu().numpy(), S)
    # an image that fills the padded set keeps its negative maximum
    images[1, 33:] = images[1, 1:39]
    images[1, :, 0] = -np.abs(images[1, :, 0]) - 3.0
    il[1] = L
    ref = O.alignment_scores(images, captions, il, cl, dtype=np.float64)
    assert ref[1, 4] < O.alignment_scores(images, captions, [20, 70, 33, 12, 33, 25], cl, dtype=np.float64)[1, 4] - 0.1
    S = E.compute_sim_matrix(images, captions, il, cl, mode='alignment').cpu().numpy()
    np.testing.assert_allclose(S, ref, rtol=0, atol=tol)
    si, sc = _fill_store(images, il, 0, eval_precision), _fill_store(captions, cl, 2, eval_precision)
    np.testing.assert_array_equal(E.compute_sim_matrix(si, sc, mode='alignment').cpu().numpy(), S)


def test_eval_npts_order_and_grid_memo(eval_precision):
    from aladin_amd import synth
    from aladin_amd import evaluation as E
    g = load_golden('eval_sets')
    images, captions, il, cl = synth.eval_sets(int(g['n_img']), int(g['D']), int(g['seed']))
    # npts (alad/evaluation.py:164-165,250-251): the first npts images / their captions are the queries
    m_all, (r_all, t_all) = E.i2t(images, captions, il, cl, return_ranks=True)
    m7, (r7, t7) = E.i2t(images, captions, il, cl, npts=7, return_ranks=True)
    np.testing.assert_array_equal(r7, r_all[:7])
    np.testing.assert_array_equal(t7, t_all[:7])
    assert m7[0] == 100.0 * np.sum(r_all[:7] < 1) / 7
    m_all, (r_all, top_all) = E.t2i(images, captions, il, cl, return_ranks=True)
    m7, (r7, top7) = E.t2i(images, captions, il, cl, npts=7, return_ranks=True)
    assert r7.shape == (35,) and top7.shape == (35, 50)
    np.testing.assert_array_equal(r7, r_all[:35])
    np.testing.assert_array_equal(top7, top_all[:35])
    # measure='order' (alad/evaluation.py:184-192,272-281): order_sim on the slot-0 embeddings
    ims, caps = images[0::5, 0, :], captions[:, 0, :]
    ro_i, _, ro_t, _ = O.ranks_from_scores(O.order_scores(ims, caps).astype(np.float64))
    _, (r, _) = E.i2t(images, captions, il, cl, return_ranks=True, measure='order')
    np.testing.assert_array_equal(r, ro_i)
    _, (r, _) = E.t2i(images, captions, il, cl, return_ranks=True, measure='order')
    np.testing.assert_array_equal(r, ro_t)
    # i2t followed by t2i on the same embeddings scores the grid once
    calls = []
    inner = _alignment_sim_fn()

    def counting(img, cap, a, b):
        calls.append(img.shape)
        return inner(img, cap, a, b)
    images_d, captions_d = T(images), T(captions)
    E.clear_eval_cache()
    E.i2t(images_d, captions_d, il, cl, sim_function=counting)
    E.t2i(images_d, captions_d, il, cl, sim_function=counting)
    assert len(calls) == 1
    captions_d[3, 1, 0] += 1.0                                       # an in-place update invalidates the memo
    E.t2i(images_d, captions_d, il, cl, sim_function=counting)
    assert len(calls) == 2


def _fill_stores(images, captions, il, cl, batch=37, precision=None):
    """Feed (N, 71, D) eval sets to PackedSetStores the way encode_data_packed does: batch by batch,
    each batch trimmed to ITS longest sample (the encoder's output length varies per batch)."""
    from aladin_amd.store import PackedSetStore
    from aladin_amd import ops
    D = images.shape[2]
    precision = precision or ops._EVAL_PRECISION[0]
    si = PackedSetStore(D, 0, dev(), capacity_rows=64, precision=precision)
    sc = PackedSetStore(D, 2, dev(), capacity_rows=64, precision=precision)
    for k0 in range(0, images.shape[0], batch):
        k1 = min(images.shape[0], k0 + batch)
        Li, Lc = max(il[k0:k1]), max(cl[k0:k1])
        # (S, B, D) -> (B, S, D) permuted views, as forward_emb hands them over
        si.append(T(np.ascontiguousarray(images[k0:k1, :Li].transpose(1, 0, 2))).permute(1, 0, 2), il[k0:k1])
        sc.append(T(np.ascontiguousarray(captions[k0:k1, :Lc].transpose(1, 0, 2))).permute(1, 0, 2), cl[k0:k1])
    return si, sc


def test_packed_store_scores_are_bit_identical_and_smaller():
    """the 16-bit length-packed store gives the SAME bits as the fp32 (N, 71, D)
    buffers for both heads, for whole stores and strided views, at a fraction of the memory."""
    from aladin_amd import evaluation as E, synth
    from aladin_amd.store import alignment_scores_from_stores
    g = load_golden('eval_sets')
    images, captions, il, cl = synth.eval_sets(int(g['n_img']), int(g['D']), int(g['seed']))
    cl = list(cl)
    cl[3], cl[4] = 3, 4                                   # a caption with no scored word, one with a single word
    captions[3, 3:] = 0
    captions[4, 4:] = 0
    si, sc = _fill_stores(images, captions, il, cl)
    assert len(si) == images.shape[0] and si.lengths == list(il) and sc.lengths == cl
    dense_bytes = images.nbytes + captions.nbytes
    assert si.nbytes() + sc.nbytes() < (0.25 if si.precision == 'fp16' else 0.5) * dense_bytes
    S_dense = E.compute_sim_matrix(images[0::5], captions, il[0::5], cl, mode='alignment')
    S_store = E.compute_sim_matrix(si.view(slice(0, None, 5)), sc, mode='alignment')
    assert torch.equal(S_dense, S_store)
    assert torch.equal(S_store[:, 3], torch.zeros_like(S_store[:, 3]))        # no scored word -> exact 0 column
    # arbitrary sub-grids through index views
    N = images.shape[0]
    pick_i, pick_c = [N - 5, 0, 15, 5], [7, 3, N - 1, 4, N // 2]
    sub = alignment_scores_from_stores(si.view(pick_i), sc.view(pick_c))
    ref = E.compute_sim_matrix(images[pick_i], captions[pick_c], [il[k] for k in pick_i], [cl[k] for k in pick_c],
                               mode='alignment')
    assert torch.equal(sub, ref)
    # caption-side chunking (bounds the side-row scratch on big grids) does not change a single bit
    from aladin_amd import ops
    limit = ops.E_SCRATCH_LIMIT
    ops.E_SCRATCH_LIMIT = 1 << 20
    try:
        assert torch.equal(E.compute_sim_matrix(images[0::5], captions, il[0::5], cl, mode='alignment'), S_dense)
        assert torch.equal(E.compute_sim_matrix(si.view(slice(0, None, 5)), sc, mode='alignment'), S_dense)
        assert torch.equal(alignment_scores_from_stores(si.view(pick_i), sc.view(pick_c)), ref)
    finally:
        ops.E_SCRATCH_LIMIT = limit
    # matching head reads the fp32 globals
    M_dense = E.compute_sim_matrix(images[0::5, 0, :], captions[:, 0, :])
    M_store = E.compute_sim_matrix(si.view(slice(0, None, 5)), sc)
    assert torch.equal(M_dense, M_store)
    # retrieval drivers accept the stores in place of the tensors
    for fn in (None, 'alignment'):
        for drv in (E.i2t, E.t2i):
            m1, (r1, t1) = drv(images, captions, il, cl, return_ranks=True, sim_function=fn)
            m2, (r2, t2) = drv(si, sc, si.lengths, sc.lengths, return_ranks=True, sim_function=fn)
            assert m1 == m2
            np.testing.assert_array_equal(r1, r2)
            np.testing.assert_array_equal(t1, t2)


def test_encode_data_packed_matches_encode_data():
    """The two embedding stores, filled by the same stand-in encoder, score identically."""
    from aladin_amd import evaluation as E
    from standins import StandInEncoder

    class _Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.enc = StandInEncoder(feat_dim=48, embed=64, vocab=500).to(dev())

        def forward_emb(self, imgs, txts):
            return self.enc(imgs, txts)

    class _Loader(list):
        dataset = list(range(30))

    rng = np.random.default_rng(3)
    batches = _Loader()
    for k in range(3):
        n = 10
        il = [int(v) for v in rng.integers(5, 20, n)]
        cl = [int(v) for v in rng.integers(4, 16, n)]
        imgs = (T(rng.standard_normal((n, max(il), 48)).astype(np.float32)), il)
        txts = (T(rng.integers(0, 500, (n, max(cl))).astype(np.int64)), cl)
        batches.append((imgs, txts))
    model = _Model()
    dense = E.encode_data(model, batches, logging=None)
    from aladin_amd import ops
    si, sc, il2, cl2 = E.encode_data_packed(model, batches, logging=None, precision=ops._EVAL_PRECISION[0])
    assert il2 == list(dense[2]) and cl2 == list(dense[3])
    S1 = E.compute_sim_matrix(dense[0], dense[1], dense[2], dense[3], mode='alignment')
    S2 = E.compute_sim_matrix(si, sc, mode='alignment')
    assert torch.equal(S1, S2)
    assert torch.equal(E.compute_sim_matrix(dense[0][:, 0], dense[1][:, 0]), E.compute_sim_matrix(si, sc))


# ------------------------------------------------------------------ BASELINE-size property tests
def test_b256_scores_vs_oracle_and_properties():
    from aladin_amd import ops, synth
    B = 256
    im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=1234, ragged=False)
    a, b = T(im), T(s)
    S = ops.alignment_scores(a, b, il, sl)
    ref = O.alignment_scores(im, s, il, sl)
    assert_scores_close(S.cpu().numpy(), ref)
    # permutation equivariance under a batch permutation of images and captions (bit exact:
    # every (i, j) block sees the same operands in the same order)
    pi = torch.from_numpy(np.random.RandomState(0).permutation(B)).to(dev())
    pj = torch.from_numpy(np.random.RandomState(1).permutation(B)).to(dev())
    Sp = ops.alignment_scores(a[pi].contiguous(), b[pj].contiguous(), il, sl)
    assert torch.equal(Sp, S[pi][:, pj])
    # ragged: content of masked positions is irrelevant
    im2, s2, il2, sl2 = synth.alignment_batch(B, 34, 50, 768, seed=99, ragged=True)
    S1 = ops.alignment_scores(T(im2), T(s2), il2, sl2)
    im3, s3 = im2.copy(), s2.copy()
    for i, L in enumerate(il2):
        im3[i, L:] = 1e3
    for j, L in enumerate(sl2):
        s3[j, L - 2:] = -7.0
    S2 = ops.alignment_scores(T(im3), T(s3), il2, sl2)
    assert torch.equal(S1, S2)
    assert_scores_close(S1.cpu().numpy(), O.alignment_scores(im2, s2, il2, sl2))


@pytest.mark.parametrize('ragged', [True, False])
def test_b256_triplet_step_gradients_vs_oracle(ragged, bwd_mode):
    """BASELINE configs[1] size: loss AND gradient VALUES of the B = 256 triplet step against the oracle (float64
    closed form of the autograd of alad/loss.py:79-159, chained on the HIP S like the shape sweep).  The hinge
    leaves <= 3B non-zero pairs, so the oracle's per-pair loop finishes in seconds.  [fp16-False] is the batch AND the
    backward mode bench.py times."""
    from aladin_amd import synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    B = 256
    if ragged:
        im, s, il, sl = synth.structured_alignment_batch(B, 34, 50, 768, seed=77, noise=3.0, ragged=True)
    else:
        im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=1234, ragged=False)       # the bench batch
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=True, aggregation='MrSw')
    loss, S = crit(a, b, il, sl, return_similarity_mat=True)
    loss.backward()
    S_np = S.detach().cpu().numpy()
    assert_scores_close(S_np, O.alignment_scores(im, s, il, sl))
    ref_loss, dS = O.hinge_loss(S_np, 0.2, True, return_grad=True)
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-5)
    assert 0 < (dS != 0).sum() <= 3 * B
    dim, ds = O.alignment_scores_backward(im, s, il, sl, dS)
    for got, ref in ((a.grad, dim), (b.grad, ds)):
        assert_grads_close(got, ref, bwd_mode)


@pytest.mark.parametrize('B,kind,R,Tn', [(128, 'random', 34, 50), (256, 'random', 34, 50), (256, 'structured', 34, 50), (192, 'ties', 34, 50),
                                         (128, 'random', 51, 38), (192, 'ties', 51, 38), (160, 'structured', 65, 20), (144, 'random', 38, 30), (128, 'ties', 41, 50),
                                         (128, 'random', 42, 50), (192, 'ties', 49, 38), (160, 'structured', 57, 38), (128, 'random', 58, 38)])
def test_dense_backward_table_equals_the_per_pair_path(B, kind, R, Tn):
    """ALADIN_BWD_DENSE (sum-of-violations hinge: every pair carries a gradient): the arg-max table written by the
    split-precision tile kernel + the per-pair kernel on the flagged near-ties must give EXACTLY the gradients of the
    per-pair kernel on every pair (same winners => same rows kernel input => bit-identical sums)."""
    from aladin_amd import ops, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    # (51, 38): VinVL's 50 regions + 35 tokens -- the 48-row class + 2 side rows (R' = 50; round 4); (42, 50) / (49, 38): 48 rows
    # with tile-filling copies (R' = 41) / exactly filled; (57, 38): 48 rows + 8 side rows, the class limit; (58, 38): two
    # 32-row tiles (R' = 57); (65, 20): R' = 64; (38, 30) / (41, 50): one region tile + 5 / 8 side rows
    if kind == 'random':
        im, s, il, sl = synth.alignment_batch(B, R, Tn, 768, seed=B + 5, ragged=True)
    else:
        im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, 768, seed=B + 9, noise=3.0, ragged=True)
    if kind == 'ties':
        im[:, 5] = im[:, 3]                               # exact duplicate regions: every word ties between r = 2 and r = 4
        im[1::2, 9] = im[1::2, 8] * (1 + 1e-7)           # and a near-tie below the fp16 operand resolution
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=False, aggregation='MrSw')
    grads = {}
    for dense in (False, True, 'gemm'):
        a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
        old, ops.DENSE_BACKWARD, old_g, ops.DENSE_ROWS_GEMM = ops.DENSE_BACKWARD, bool(dense), ops.DENSE_ROWS_GEMM, dense == 'gemm'
        old_f, ops.DENSE_MIN_FRACTION = ops.DENSE_MIN_FRACTION, 0.0      # whatever the density of this batch's dS
        ops.DENSE_GEMM_FORCE = True                                        # ... and however full its captions
        try:
            loss = crit(a, b, il, sl)
            loss.backward()
        finally:
            ops.DENSE_BACKWARD, ops.DENSE_ROWS_GEMM, ops.DENSE_MIN_FRACTION, ops.DENSE_GEMM_FORCE = old, old_g, old_f, False
        grads[dense] = (loss.item(), a.grad.clone(), b.grad.clone())
    assert grads[True][0] == grads[False][0]
    assert torch.equal(grads[True][1], grads[False][1])
    assert torch.equal(grads[True][2], grads[False][2])
    assert grads[True][1].abs().sum() > 0
    # the row step as two MFMA GEMMs over the same table (csrc/align_bwd_dense.hip): hi + lo split operands, another
    # summation order -- equal to the fp32 gather to rounding (measured <= 2e-6 of the largest entry)
    for k in (1, 2):
        ref, got = grads[True][k], grads['gemm'][k]
        assert not torch.equal(ref, got)                 # it really took the other path
        assert torch.equal(ref == 0, got == 0) or (got[(ref == 0) != (got == 0)].abs().max() < 1e-6 * ref.abs().max())
        assert (ref - got).abs().max() <= 1e-5 * ref.abs().max()


def test_dense_backward_follows_the_measured_density():
    """The dense path is taken while most pairs violate the margin; once the pair count of the step LAG steps earlier (copied
    out asynchronously) says they do not, the sum-of-violations hinge goes back to the list path -- same gradients either way,
    and the switch happens at a FIXED step (reproducible runs)."""
    from aladin_amd import ops, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    B = 128
    im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=41, ragged=True)
    probe = ops._density_probe
    LAG = probe.LAG
    for margin, dense_expected in ((0.2, True), (-50.0, False)):
        crit = AlignmentContrastiveLoss(margin=margin, measure='dot', max_violation=False, aggregation='MrSw')
        probe.__init__()
        grads, flags = [], []
        for step in range(LAG + 2):
            a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
            crit(a, b, il, sl).backward()
            grads.append((a.grad.clone(), b.grad.clone()))
            flags.append(ops._LAST_BWD_FLAGS[0] != 0)
        assert flags == [True] * LAG + [dense_expected] * 2, flags          # unknown -> dense; then the count of step n - LAG decides
        assert (probe.newest() > 0.5) == dense_expected
        for g in grads[1:]:
            assert torch.allclose(grads[0][0], g[0], rtol=0, atol=1e-5 * float(grads[0][0].abs().max()) + 1e-30)
            assert torch.allclose(grads[0][1], g[1], rtol=0, atol=1e-5 * float(grads[0][1].abs().max()) + 1e-30)
    probe.__init__()


def test_model_heads_take_the_dense_backward_too():
    """ALADModel's single-node heads (ops._BigHeads, B > 64) with 'max-violation': False: dense and list backward agree."""
    from aladin_amd import ops, synth
    from aladin_amd.alad_model import ALADModel
    B = 128
    im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=43, ragged=True)
    gi, gc = synth.global_embeddings(B, 768, seed=44, noise=3.0)
    model = ALADModel({'training': {'loss-type': 'alignment-distillation-matching', 'loss-weights': [1, 1, 0.5], 'margin': 0.2,
                                    'measure': 'dot', 'max-violation': False, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}})
    grads = {}
    for dense in (False, True):
        x, y = T(gi).requires_grad_(True), T(gc).requires_grad_(True)
        a = T(im).permute(1, 0, 2).contiguous().requires_grad_(True)
        b = T(s).permute(1, 0, 2).contiguous().requires_grad_(True)
        old, ops.DENSE_BACKWARD, old_f, ops.DENSE_MIN_FRACTION = ops.DENSE_BACKWARD, dense, ops.DENSE_MIN_FRACTION, 0.0
        ops.DENSE_GEMM_FORCE = True
        try:
            loss, _ = model.forward_loss_total(x, y, a, b, il, sl, 0, epoch=5)
            loss.backward()
        finally:
            ops.DENSE_BACKWARD, ops.DENSE_MIN_FRACTION, ops.DENSE_GEMM_FORCE = old, old_f, False
        grads[dense] = [t.grad.clone() for t in (x, y, a, b)]
    for g0, g1 in zip(grads[False], grads[True]):
        assert (g0 - g1).abs().max() <= 1e-5 * g0.abs().max()
    assert not torch.equal(grads[False][2], grads[True][2])            # the GEMM row step really ran


def test_generic_score_gradient_learns_its_density():
    """ops.alignment_scores + a separate loss: the backward counts dS's non-zeros and, LAG steps later, takes the dense path
    while dS is dense -- and stays on the list path for the hardest-negative hinge's sparse dS."""
    from aladin_amd import ops, synth
    B = 128
    im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=47, ragged=True)
    LAG = ops._DensityProbe.LAG
    for mv, dense_expected in ((False, True), (True, False)):
        ops._generic_probes.clear()
        grads, flags = [], []
        for step in range(LAG + 2):
            a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
            ops.hinge_loss(ops.alignment_scores(a, b, il, sl), 0.2, mv).backward()
            grads.append((a.grad.clone(), b.grad.clone()))
            flags.append(ops._LAST_BWD_FLAGS[0] != 0)
        (probe,) = ops._generic_probes.values()
        assert flags == [False] * LAG + [dense_expected] * 2, flags         # unknown -> list path
        assert (probe.newest() > 0.5) == dense_expected
        for k in (0, 1):
            if dense_expected:
                assert not torch.equal(grads[0][k], grads[-1][k])          # list path vs dense table + GEMM row step
                assert (grads[0][k] - grads[-1][k]).abs().max() <= 1e-5 * grads[0][k].abs().max()
            else:
                assert torch.equal(grads[0][k], grads[-1][k])
    ops._generic_probes.clear()


def test_dense_row_step_follows_the_caption_fill():
    """Short captions (COCO: ~12 of 35 tokens): the gather row step costs ~ real words, the GEMM one the padded tiles --
    the choice follows the host-side lengths (ops._caption_fill / _gemm_rows_pay)."""
    from aladin_amd import _lib, ops, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    B = 128
    im, s, il, _ = synth.alignment_batch(B, 34, 50, 768, seed=53, ragged=False)
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=False, aggregation='MrSw')
    ops._density_probe.__init__()
    old_f, ops.DENSE_MIN_FRACTION = ops.DENSE_MIN_FRACTION, 0.0
    try:
        for sl, gather in (([50] * B, False), ([13] * B, True)):
            a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
            crit(a, b, il, sl).backward()
            assert ops._LAST_BWD_FLAGS[0] & _lib.BWD_DENSE
            assert bool(ops._LAST_BWD_FLAGS[0] & _lib.BWD_DENSE_GATHER) == gather
    finally:
        ops.DENSE_MIN_FRACTION = old_f
        ops._density_probe.__init__()


def test_dense_backward_through_the_score_matrix():
    """The other dense caller: a gradient arriving on the returned S (listnet on top of the alignment scores)."""
    from aladin_amd import ops, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    B = 128
    im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=31, ragged=True)
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=True, aggregation='MrSw')
    w = torch.randn(B, B, device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    grads = {}
    for dense in (False, True, 'gemm'):
        a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
        old, ops.DENSE_BACKWARD, old_g, ops.DENSE_ROWS_GEMM = ops.DENSE_BACKWARD, bool(dense), ops.DENSE_ROWS_GEMM, dense == 'gemm'
        try:
            loss, S = crit(a, b, il, sl, return_similarity_mat=True)
            (loss + (S * w).sum()).backward()
        finally:
            ops.DENSE_BACKWARD, ops.DENSE_ROWS_GEMM = old, old_g
        grads[dense] = (a.grad.clone(), b.grad.clone())
    assert torch.equal(grads[True][0], grads[False][0])
    assert torch.equal(grads[True][1], grads[False][1])
    for k in (0, 1):                                      # arbitrary real dS through the GEMM row step (dS = hi + lo in fp16)
        assert (grads[True][k] - grads['gemm'][k]).abs().max() <= 1e-5 * grads[True][k].abs().max()


def test_b256_triplet_step_gradient_sparsity():
    from aladin_amd import synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    B = 256
    im, s, il, sl = synth.alignment_batch(B, 34, 50, 768, seed=7, ragged=True)
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=True, aggregation='MrSw')
    loss, S = crit(a, b, il, sl, return_similarity_mat=True)
    loss.backward()
    S_np = S.detach().cpu().numpy()
    ref_loss, dS = O.hinge_loss(S_np, 0.2, True, return_grad=True)
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-5)
    assert (dS != 0).sum() <= 3 * B
    ga, gb = a.grad.cpu().numpy(), b.grad.cpu().numpy()
    assert np.isfinite(ga).all() and np.isfinite(gb).all()
    assert np.all(ga[:, 0] == 0) and np.all(gb[:, 0] == 0)
    # gradients are tangent to the unit sphere: <x, dx> = 0 for every row (normalise backward)
    dots = np.abs((ga * im).sum(-1))
    assert dots.max() <= 1e-3 * max(1e-6, np.abs(ga).sum(-1).max() * np.abs(im).max())


def test_large_batch_equals_its_blocks():
    """A score depends only on its own image and caption, not on where its tile sits: the 1024 x 768 matrix
    must equal, bit for bit, the 256 x 256 blocks scored one by one (64-bit indexing, tile order, XCD remap,
    side-GEMM strides at sizes beyond the headline batch)."""
    from aladin_amd import ops, synth
    Bi, Bc, blk = 1024, 768, 256
    im, s, il, sl = synth.alignment_batch(Bi, 34, 50, 256, seed=515, ragged=True, Bc=Bc)
    il[0], sl[0] = 34, 50                                  # keep the batch maxima at R'=33, T'=47 in every block
    a, b = T(im), T(s)
    S = ops.alignment_scores(a, b, il, sl)
    for i0 in range(0, Bi, blk):
        for j0 in range(0, Bc, blk):
            part = ops.alignment_scores(a[i0:i0 + blk], b[j0:j0 + blk], il[i0:i0 + blk], sl[j0:j0 + blk])
            assert torch.equal(S[i0:i0 + blk, j0:j0 + blk], part), (i0, j0)


def test_fused_hinge_argmax_equals_the_list_path():
    """The training step derives the backward's pairs from the hinge statistics and runs the hinge's element-wise pass in
    the pair kernel's launch (aladin_align_triplet_fwd / _bwd: one library call per direction); the list-driven form (aladin_align_pack
    + aladin_align_scores + aladin_hinge_fused -> pair list -> aladin_align_bwd) must give the same loss, dS and gradients bit for bit -- including
    when a row's and a column's hardest negative are the same pair, inactive terms and ragged lengths."""
    from aladin_amd import ops, synth
    # (51, 38): the shipped data shape, 50 regions + 35 tokens -- two region tiles per image (the second is the pair kernel's segment)
    for B, seed, noise, R, Tn in ((40, 11, 1.0, 34, 50), (96, 12, 3.0, 34, 50), (256, 13, 1.0, 34, 50), (72, 14, 3.0, 51, 38), (256, 15, 1.0, 51, 38)):
        im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, 768, seed=seed, noise=noise, ragged=True)
        a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
        loss, S = ops.alignment_triplet_loss(a, b, il, sl, 0.2, True)
        loss.backward()
        ilt, slt = ops.lengths_tensor(il, a.device), ops.lengths_tensor(sl, a.device)
        with torch.no_grad():
            S2, packed = ops._align_forward(a.detach(), b.detach(), ilt, slt)
            loss2, dS2, pairs = ops._hinge_raw(S2, 0.2, True, True, want_pairs=True)
            one = torch.ones((), device=a.device)
            d_im, d_s = ops._align_backward(a.detach(), b.detach(), ilt, slt, dS2, gscale=one, packed=packed, pairs=pairs)
        assert torch.equal(S, S2) and float(loss) == float(loss2), (B, float(loss), float(loss2))
        assert torch.equal(a.grad, d_im) and torch.equal(b.grad, d_s), B
        assert int(pairs[1].item()) <= 3 * B


def test_small_grid_score_variant_is_bit_identical():
    """Grids of at most 64 of the 256 x 384 tiles (B <= 64 at the headline shape) run the two-wave 128 x 192 /
    three-stage variant of the score kernel, larger ones the eight-wave 128 x 96-wave-tile kernel: same MFMA shape, K
    order and epilogue arithmetic, so a 96 x 96 matrix must equal its 32 x 32 and 48 x 96 blocks bit for bit, with
    and without the side row (R' = 33 / 32), full and ragged."""
    from aladin_amd import ops, synth
    # (51, 38) / (49, 38): the 48-row class (96 x 192 two-wave tiles for small grids, 192 x 384 otherwise), with / without side rows
    for R, ragged, seed, Tn in ((34, False, 901, 50), (33, True, 902, 50), (34, True, 903, 50), (51, True, 904, 38), (49, False, 905, 38)):
        B = 96
        im, s, il, sl = synth.alignment_batch(B, R, Tn, 768, seed=seed, ragged=ragged)
        a, b = T(im), T(s)
        S = ops.alignment_scores(a, b, il, sl)
        for (bi, bj) in ((32, 32), (48, 96), (96, 40)):
            for i0 in range(0, B, bi):
                for j0 in range(0, B, bj):
                    ii, jj = slice(i0, min(B, i0 + bi)), slice(j0, min(B, j0 + bj))
                    il_b, sl_b = list(il[ii]), list(sl[jj])
                    part = ops.alignment_scores(a[ii], b[jj], il_b, sl_b)
                    assert torch.equal(S[ii, jj], part), (R, ragged, bi, bj, i0, j0)


@pytest.mark.parametrize('Bi,Bc,R,Tn', [(200, 208, 34, 26), (200, 200, 36, 40), (130, 140, 60, 27), (140, 130, 66, 39), (200, 200, 50, 24), (200, 600, 34, 10), (200, 500, 51, 11)])
def test_large_grid_half_caption_classes_are_bit_identical(Bi, Bc, R, Tn):
    """The 8- / 24- / 40-word caption classes on grids of more than 64 workgroup tiles (eight-wave kernels: the 128 x 96 wave tile with
    twelve captions of 8 or four of 24 per strip, the 128 x 80 one with two of 40; one or two region tiles per image, with side rows) against
    their own small-grid blocks (two-wave kernels) bit for bit, and against the oracle."""
    from aladin_amd import ops, synth
    im, s, il, sl = synth.alignment_batch(Bi, R, Tn, 192, seed=7100 + R, ragged=True, Bc=Bc)
    il[0], sl[0] = R, Tn
    g = ops.align_geometry(Bi, Bc, R, Tn, 192)
    assert g.trows in (8, 24, 40) and g.trows == 16 * g.tp16 - 8 and (g.xm_rows // 256) * (g.y_rows // (640 if g.trows == 40 else 384)) > 64
    a, b = T(im), T(s)
    S = ops.alignment_scores(a, b, il, sl)
    assert_scores_close(S.cpu().numpy(), O.alignment_scores(im, s, il, sl))
    for i0 in range(0, Bi, 64):
        for j0 in range(0, Bc, 64):
            ii, jj = slice(i0, min(Bi, i0 + 64)), slice(j0, min(Bc, j0 + 64))
            il_b, sl_b = list(il[ii]), list(sl[jj])
            il_b[0], sl_b[0] = R, Tn                                     # every block keeps the geometry of the whole
            part = ops.alignment_scores(a[ii].clone(), b[jj].clone(), il_b, sl_b)
            keep_i = slice(1 if il_b[0] != il[i0] else 0, None)          # the sample whose length was raised scores differently
            keep_j = slice(1 if sl_b[0] != sl[j0] else 0, None)
            assert torch.equal(S[ii, jj][keep_i, keep_j], part[keep_i, keep_j]), (i0, j0)


@pytest.mark.parametrize('Bi,Bc,R,Tn', [(254, 270, 51, 38), (250, 258, 49, 43), (256, 256, 50, 36), (262, 272, 54, 40), (254, 270, 50, 38)])
def test_large_grid_40_word_tile_is_bit_identical(Bi, Bc, R, Tn):
    """Large grids of the 48-row region class x 40-word caption class run the 288 x 320 workgroup tile (three images per wave; the
    last row tile hangs over the operand's end when the image count is not a multiple of six: 254 -> 256 images = 42.67 tiles)
    when its whole rounds of 256 workgroups come out ahead -- (254, 270), (250, 258), (262, 272) here; (256, 256) is exactly four
    rounds of the 192 x 320 tile and stays there -- smaller ones the 192 x 320 / 96 x 160 tiles.  Same MFMA shape, K order and
    epilogue arithmetic: the big matrix must equal its blocks bit for bit (no / one / two / five side rows per image; (254, 270, 50, 38): one side row on the big tile), and the
    oracle to the fp16 tolerance."""
    from aladin_amd import ops, synth
    im, s, il, sl = synth.alignment_batch(Bi, R, Tn, 256, seed=7000 + R, ragged=True, Bc=Bc)
    il[0], sl[0] = R, Tn                                                 # every block keeps the geometry of the whole
    il[-1], sl[-1] = R, Tn
    g = ops.align_geometry(Bi, Bc, R, Tn, 256)
    assert g.mrows == 48 and g.trows == 40 and (g.xm_rows // 192) * (g.y_rows // 320) >= 512
    r192, r288 = -(-(g.xm_rows // 192) * (g.y_rows // 320) // 256), -(-(-(-g.xm_rows // 288)) * (g.y_rows // 320) // 256)
    assert (1.37 * r288 < 0.97 * r192) == ((Bi, Bc) != (256, 256))          # which kernel the library picks (align_fwd.hip launch_scores16_r48)
    a, b = T(im), T(s)
    S = ops.alignment_scores(a, b, il, sl)
    assert_scores_close(S.cpu().numpy(), O.alignment_scores(im, s, il, sl))
    for bi, bj in ((64, 64), (128, Bc)):
        for i0 in range(0, Bi, bi):
            for j0 in range(0, Bc, bj):
                ii, jj = slice(i0, min(Bi, i0 + bi)), slice(j0, min(Bc, j0 + bj))
                il_b, sl_b = list(il[ii]), list(sl[jj])
                il_b[0], sl_b[0] = R, Tn
                aa, bb = a[ii].clone(), b[jj].clone()
                part = ops.alignment_scores(aa, bb, il_b, sl_b)
                keep_i = slice(1 if il_b[0] != il[i0] else 0, None)      # the sample whose length was raised scores differently
                keep_j = slice(1 if sl_b[0] != sl[j0] else 0, None)
                assert torch.equal(S[ii, jj][keep_i, keep_j], part[keep_i, keep_j]), (bi, bj, i0, j0)


def test_l2norm_and_cosine_measure():
    """l2norm (alad/utils.py:134-139: no eps, zero row -> NaN) forward / backward, and measure='cosine'."""
    from aladin_amd.loss import ContrastiveLoss, l2norm
    rng = np.random.default_rng(12)
    x = rng.standard_normal((37, 200)).astype(np.float32)
    xt = T(x).requires_grad_(True)
    xr = torch.from_numpy(x).requires_grad_(True)
    w = rng.standard_normal((37, 100)).astype(np.float32)
    out = l2norm(xt[:, ::2])                                           # strided view: made contiguous inside
    (out * T(w)).sum().backward()
    ref = xr[:, ::2] / torch.pow(xr[:, ::2], 2).sum(dim=1, keepdim=True).sqrt()
    (ref * torch.from_numpy(w)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-6)
    z = T(np.zeros((2, 8), np.float32))
    assert torch.isnan(l2norm(z)).all()
    a, b = rng.standard_normal((9, 64)).astype(np.float32) * 3, rng.standard_normal((9, 64)).astype(np.float32) * 0.2
    at, bt = T(a).requires_grad_(True), T(b).requires_grad_(True)
    loss = ContrastiveLoss(0.2, 'cosine', True)(at, bt)
    loss.backward()
    ar, br = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    sc = torch.nn.functional.normalize(ar, dim=1) @ torch.nn.functional.normalize(br, dim=1).t()
    ref_loss, dS = O.hinge_loss(sc.detach().numpy(), 0.2, True, return_grad=True)
    (sc * torch.from_numpy(dS)).sum().backward()
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-5)
    np.testing.assert_allclose(at.grad.cpu().numpy(), ar.grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(bt.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-4, atol=1e-6)


def test_error_behaviour():
    from aladin_amd import ops
    from aladin_amd.loss import AlignmentContrastiveLoss
    g = load_golden('align_rect')
    im, s, il, sl = golden_alignment_inputs(g)
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=True, aggregation='MrSw')
    with pytest.raises(ValueError):                      # non-square S into the hinge (reference: diag/expand_as fails)
        crit(T(im), T(s), il, sl)
    with pytest.raises(RuntimeError):                    # CPU tensors: no fallback
        ops.alignment_scores(torch.from_numpy(im), torch.from_numpy(s), il, sl)
    with pytest.raises(ValueError):
        ops.alignment_scores(T(im), T(s), il[:-1], sl)


# (33, 5, 3000) / (20, 40, 768): an image's rows + its captions' do not fit the packer's 64 KB of LDS -> separate ground-truth kernel
@pytest.mark.parametrize('n_img,cpi,D', [(1, 5, 8), (7, 1, 33), (77, 3, 100), (300, 5, 64), (1000, 5, 768), (257, 8, 50), (40, 13, 24), (33, 5, 3000),
                                         (20, 40, 768)])
def test_fused_retrieval_ranks_equal_two_step(n_img, cpi, D):
    """aladin_retrieval_ranks (prefix screening + exact continuation, no score matrix) must give the very ints of
    aladin_sim_matrix + aladin_recall_ranks: ragged tile edges, every captions-per-image count, odd D; so must the
    all-exact variant (every tile on the three-product path)."""
    from aladin_amd import ops
    rng = np.random.default_rng(n_img * 31 + cpi)
    img = rng.standard_normal((n_img, D)).astype(np.float32)
    cap = (np.repeat(img, cpi, axis=0) + 1.5 * rng.standard_normal((n_img * cpi, D))).astype(np.float32)
    img /= np.linalg.norm(img, axis=1, keepdims=True)
    cap /= np.linalg.norm(cap, axis=1, keepdims=True)
    a, b = T(img), T(cap)
    two = ops.recall_ranks(ops.sim_matrix(a, b), cpi)
    one = ops.retrieval_ranks(a, b, cpi)
    for x, y in zip(one, two):
        assert torch.equal(x, y)
    for x, y in zip(ops.retrieval_ranks(a, b, cpi, exact=True), two):
        assert torch.equal(x, y)
    # and against the oracle's argsort-free definition
    if n_img <= 300:
        r_i2t, _, r_t2i, _ = O.ranks_from_scores(img.astype(np.float64) @ cap.astype(np.float64).T, cpi)
        # fp64 vs device near-ties may swap neighbours
        assert np.mean(one[0].cpu().numpy() == r_i2t) >= 0.99 and np.mean(one[2].cpu().numpy() == r_t2i) >= 0.99


def _adversarial_retrieval(case, n_img, cpi, D, seed):
    """Inputs built to sit ON the screening kernel's decision boundaries (ops.retrieval_ranks)."""
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((n_img, D)).astype(np.float32)
    img /= np.linalg.norm(img, axis=1, keepdims=True)
    if case == 'bulk':                   # ground truths inside the bulk of the scores: captions unrelated to their images
        cap = rng.standard_normal((n_img * cpi, D)).astype(np.float32)
        cap /= np.linalg.norm(cap, axis=1, keepdims=True)
    elif case == 'near_duplicates':      # every caption has near-copies (relative 1e-4 .. 1e-7) filed under OTHER images
        base = np.repeat(img, cpi, axis=0) + 0.6 * rng.standard_normal((n_img * cpi, D)).astype(np.float32)
        cap = base.copy()
        src = rng.integers(0, n_img * cpi, size=n_img * cpi // 2)
        dst = rng.permutation(n_img * cpi)[:src.size]
        eps = (10.0 ** rng.uniform(-7, -4, size=(src.size, 1))).astype(np.float32)
        cap[dst] = base[src] * (1 + eps * rng.standard_normal((src.size, D)).astype(np.float32))
        cap /= np.linalg.norm(cap, axis=1, keepdims=True)
    elif case == 'exact_ties':           # bit-identical caption rows under different images, duplicated images too
        cap = np.repeat(img, cpi, axis=0) + 0.8 * rng.standard_normal((n_img * cpi, D)).astype(np.float32)
        cap /= np.linalg.norm(cap, axis=1, keepdims=True)
        k = n_img * cpi
        src = rng.integers(0, k, size=k // 3)
        dst = rng.permutation(k)[:src.size]
        cap[dst] = cap[src]
        isrc = rng.integers(0, n_img, size=n_img // 4)
        idst = rng.permutation(n_img)[:isrc.size]
        img[idst] = img[isrc]
    elif case == 'norms':                # wildly different row norms (the band is per pair, not per matrix), some tiny rows
        cap = np.repeat(img, cpi, axis=0) + 1.2 * rng.standard_normal((n_img * cpi, D)).astype(np.float32)
        cap *= (10.0 ** rng.uniform(-3, 1, size=(n_img * cpi, 1))).astype(np.float32)
        img *= (10.0 ** rng.uniform(-3, 1, size=(n_img, 1))).astype(np.float32)
        img[::17] = 0.0
        cap[::29] = 0.0
    elif case == 'mixed':                # half the images retrieve cleanly, half have their ground truths in the bulk
        cap = np.repeat(img, cpi, axis=0) + 0.3 * rng.standard_normal((n_img * cpi, D)).astype(np.float32)
        bad = np.repeat(rng.random(n_img) < 0.5, cpi)
        cap[bad] = rng.standard_normal((int(bad.sum()), D)).astype(np.float32)
        cap /= np.linalg.norm(cap, axis=1, keepdims=True)
    else:
        raise ValueError(case)
    return img, cap.astype(np.float32)


@pytest.mark.parametrize('case', ['bulk', 'near_duplicates', 'exact_ties', 'norms', 'mixed'])
@pytest.mark.parametrize('n_img,cpi,D', [(700, 5, 768), (1300, 3, 96), (520, 1, 40)])
def test_fused_retrieval_adversarial_equals_two_step(case, n_img, cpi, D):
    """ground truths in the bulk, near-duplicate captions, exact ties, mixed norms -- the screened
    retrieval equals the two-step split path (stored scores + rank kernels) int for int, and so does the all-exact
    variant; the statistics say which mechanism ran (lists vs tiles continued in place)."""
    from aladin_amd import ops
    img, cap = _adversarial_retrieval(case, n_img, cpi, D, seed=n_img + 7 * cpi)
    a, b = T(img), T(cap)
    two = ops.recall_ranks(ops.sim_matrix(a, b), cpi)
    *one, stats = ops.retrieval_ranks(a, b, cpi, return_stats=True)
    for x, y in zip(one, two):
        assert torch.equal(x, y), (case, stats)
    for x, y in zip(ops.retrieval_ranks(a, b, cpi, exact=True), two):
        assert torch.equal(x, y)
    assert 0 <= stats['exact_tiles'] <= stats['tiles']
    if case == 'bulk' and D >= 96 and cpi > 1:
        assert stats['exact_tiles'] > 0                  # thousands of undecided pairs per tile: continued in place
    # ranks against float64 on the host: only near-ties may differ
    r_i2t, _, r_t2i, _ = O.ranks_from_scores(img.astype(np.float64) @ cap.astype(np.float64).T, cpi)
    if case in ('bulk', 'mixed'):
        assert np.mean(one[0].cpu().numpy() == r_i2t) >= 0.98 and np.mean(one[2].cpu().numpy() == r_t2i) >= 0.98


def test_fused_retrieval_clean_data_uses_the_screen():
    """Ground truths clear of the bulk (what a trained matching head produces): no tile needs the exact path and only a
    few pairs are continued through lists."""
    from aladin_amd import ops
    rng = np.random.default_rng(5)
    n_img, cpi, D = 1500, 5, 768
    img = rng.standard_normal((n_img, D)).astype(np.float32)
    cap = np.repeat(img, cpi, axis=0) + 0.7 * rng.standard_normal((n_img * cpi, D)).astype(np.float32)
    img /= np.linalg.norm(img, axis=1, keepdims=True)
    cap /= np.linalg.norm(cap, axis=1, keepdims=True)
    a, b = T(img), T(cap)
    *one, stats = ops.retrieval_ranks(a, b, cpi, return_stats=True)
    two = ops.recall_ranks(ops.sim_matrix(a, b), cpi)
    for x, y in zip(one, two):
        assert torch.equal(x, y)
    assert stats['exact_tiles'] == 0 and stats['listed_pairs'] < 2000, stats


def _host_ranks_float64(img, cap, n_img, sim_h=None):
    """COCO-protocol ranks (alad/recall_auxiliary.py:30-56: #scores strictly above the ground truth; i2t against the best of
    the image's 5 captions) from float64 scores on the host, in chunks.  -> (ref_i2t, ref_t2i, max |float64 - sim_h|)."""
    ims = img[0::5].astype(np.float64)
    capd = cap.astype(np.float64)
    ref_i2t = np.empty(n_img, np.int64)
    ref_t2i = np.empty(5 * n_img, np.int64)
    max_err = 0.0
    for i0 in range(0, n_img, 500):
        d = ims[i0:i0 + 500] @ capd.T                                   # (500, 25000) float64
        if sim_h is not None:
            max_err = max(max_err, float(np.abs(d - sim_h[i0:i0 + 500]).max()))
        for k in range(d.shape[0]):
            i = i0 + k
            gt = d[k, 5 * i:5 * i + 5]
            ref_i2t[i] = (d[k][None, :] > gt[:, None]).sum(1).min()
    for c0 in range(0, 5 * n_img, 2500):
        d = ims @ capd[c0:c0 + 2500].T                                  # (5000, 2500)
        gt = d[np.arange(c0, c0 + 2500) // 5, np.arange(2500)]
        ref_t2i[c0:c0 + 2500] = (d > gt[None, :]).sum(0)
    return ref_i2t, ref_t2i, max_err


# sigma 8: Recall@1 75.4 / 40.8 % -- the input SURVEY 8(d) config 3 specifies (R@1 in 40-80 %) and the one bench.py times;
# 6: 99 / 81 % (lists only); 12: 19 / 9 % (ground truths deep in the bulk: the screen gives up)
@pytest.mark.parametrize('sigma', [8.0, 6.0, 12.0])
def test_config3_full_size_retrieval_ranks(sigma):
    """BASELINE configs[2]: 5000 images x 25000 captions, D=768 -- the size bench.py times, through the kernel it times.
    The fused retrieval (screened, and with every tile forced exact) must equal the two-step path (stored split-fp16 scores
    + rank kernels) INT FOR INT on all four outputs; the two-step ranks must equal ranks computed on the host in float64
    from the same embeddings (near-ties aside), Recall@K within 0.1."""
    from aladin_amd import evaluation as E, ops, synth
    n_img = 5000
    img, cap = synth.retrieval_embeddings(n_img, 768, seed=303, sigma=sigma)
    a, b = T(img[0::5]), T(cap)
    sim = E.compute_sim_matrix(img[0::5], cap)
    two = ops.recall_ranks(sim)
    *one, stats = ops.retrieval_ranks(a, b, 5, return_stats=True)
    for k, (x, y) in enumerate(zip(one, two)):
        assert torch.equal(x, y), (sigma, k, stats, int((x != y).sum()))
    for k, (x, y) in enumerate(zip(ops.retrieval_ranks(a, b, 5, exact=True), two)):
        assert torch.equal(x, y), (sigma, k, 'exact')
    assert stats['tiles'] == 20 * 66
    if sigma == 8.0:
        # the specified data must run on the screen: in-register counts + lists, not the exact path (round 4: 1300 of 1320 tiles)
        assert stats['exact_tiles'] <= stats['tiles'] // 10, stats
        assert stats['rescored_pairs'] <= stats['listed_pairs']
    r_i2t, t_i2t, r_t2i, t_t2i = (x.cpu().numpy() for x in two)
    ref_i2t, ref_t2i, max_err = _host_ranks_float64(img, cap, n_img, sim.cpu().numpy())
    assert max_err < 3e-6
    # float64 host ranks vs the ~fp32-accurate device scores: only near-ties (|delta| < 3e-6) may move, by a few places
    assert np.mean(r_i2t == ref_i2t) > 0.998 and np.mean(r_t2i == ref_t2i) > 0.998
    assert np.abs(r_i2t - ref_i2t).max() <= 3 and np.abs(r_t2i - ref_t2i).max() <= 3
    for K in (1, 5, 10):
        assert abs(100.0 * np.mean(r_i2t < K) - 100.0 * np.mean(ref_i2t < K)) <= 0.1
        assert abs(100.0 * np.mean(r_t2i < K) - 100.0 * np.mean(ref_t2i < K)) <= 0.1
    m = E.compute_recall(img, cap, verbose=False)                       # the fused path behind the drop-in name
    assert m[0] == pytest.approx(100.0 * np.mean(r_i2t < 1)) and m[3] == pytest.approx(100.0 * np.mean(r_t2i < 1))
    if sigma == 8.0:
        assert 40.0 <= m[0] <= 80.0 and 40.0 <= m[3] <= 80.0, m       # non-degenerate ranks in both directions


def test_sharded_loss_under_rccl_world1():
    """aladin_amd.distributed on the real backend ("nccl" == RCCL).  A 1-GPU box only allows
    world_size 1, which still runs every collective call and the autograd wrappers on device; the
    result must equal the single-device fused loss bit for bit (world 2 is covered under gloo in
    tests/test_distributed_cpu.py)."""
    import os
    import torch.distributed as dist
    from aladin_amd import synth
    from aladin_amd.distributed import sharded_alignment_loss
    from aladin_amd.loss import AlignmentContrastiveLoss
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29611')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    created = False
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev())
        created = True
    try:
        im, s, il, sl = synth.alignment_batch(32, 34, 50, 768, seed=77, ragged=True)
        a1, b1 = T(im).requires_grad_(True), T(s).requires_grad_(True)
        loss1, S1 = sharded_alignment_loss(a1, b1, il, sl, 0.2, True)
        loss1.backward()
        a2, b2 = T(im).requires_grad_(True), T(s).requires_grad_(True)
        loss2, S2 = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(a2, b2, il, sl, return_similarity_mat=True)
        loss2.backward()
        assert torch.equal(S1.detach(), S2) and torch.equal(loss1.detach(), loss2.detach())
        np.testing.assert_allclose(a1.grad.cpu().numpy(), a2.grad.cpu().numpy(), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(b1.grad.cpu().numpy(), b2.grad.cpu().numpy(), rtol=1e-6, atol=1e-9)
    finally:
        if created:
            dist.destroy_process_group()


def test_many_sharded_forwards_before_one_backward_under_rccl_world1():
    """micro-batch losses summed, ONE backward.  Every sparse-exchange plan keeps its own pinned counts buffer until
    its backward resolved it: with more than 8 plans outstanding (round 3's rotation depth) the gradients of every micro-batch
    must still equal the single-device ones, and the buffers are handed back afterwards."""
    import os
    import torch.distributed as dist
    from aladin_amd import distributed as AD, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29613')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    created = False
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev())
        created = True
    try:
        n_micro = 11
        batches = [synth.structured_alignment_batch(24, 34, 50, 256, seed=300 + k, noise=3.0, ragged=True) for k in range(n_micro)]
        leaves = [(T(im).requires_grad_(True), T(s).requires_grad_(True)) for im, s, _, _ in batches]
        total = None
        for (a, b), (_, _, il, sl) in zip(leaves, batches):
            loss, _ = AD.sharded_alignment_loss_fast(a, b, il, sl, 0.2, True, exchange='sparse')
            total = loss if total is None else total + loss
        outstanding = AD._PINNED.allocated - sum(len(v) for v in AD._PINNED.free.values())
        assert outstanding >= n_micro                       # one buffer per unresolved plan
        total.backward()
        crit = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')
        for (a, b), (im, s, il, sl) in zip(leaves, batches):
            a2, b2 = T(im).requires_grad_(True), T(s).requires_grad_(True)
            crit(a2, b2, il, sl).backward()
            np.testing.assert_allclose(a.grad.cpu().numpy(), a2.grad.cpu().numpy(), rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(b.grad.cpu().numpy(), b2.grad.cpu().numpy(), rtol=1e-5, atol=1e-8)
        assert AD._PINNED.allocated == sum(len(v) for v in AD._PINNED.free.values())      # all handed back
    finally:
        if created:
            dist.destroy_process_group()


SWEEP = [
    # Bi, Bc, R,  T,  D    -> tiling class exercised
    (3, 5, 2, 4, 8),        # one region, one word
    (9, 4, 17, 9, 40),      # mtiles 1 padded, tp16 1, D not a multiple of 64 nor of 8
    (7, 7, 33, 20, 64),     # R' = 32 exactly, tp16 2
    (10, 6, 34, 35, 96),    # 32 + 1 side row, tp16 2
    (5, 9, 34, 67, 64),     # side row, tp16 4
    (6, 3, 50, 50, 128),    # 48-row class + 1 side row (R' = 49), tp16 3
    (5, 7, 42, 20, 64),     # 48-row class with tile-filling copies (R' = 41), tp16 2
    (9, 5, 49, 9, 40),      # 48-row class exactly filled (R' = 48), tp16 1, odd D
    (6, 6, 51, 38, 768),    # the shipped data shape: 50 regions + 35 tokens -> 48 rows + 2 side rows, 40-word caption class
    (6, 5, 51, 36, 64),     # 40-word class, T' = 33 (its lower edge)
    (5, 19, 45, 43, 128),   # 40-word class exactly filled (T' = 40), no side rows, captions past one 16-caption unit
    (4, 33, 57, 44, 64),    # T' = 41: back to three whole tiles; 8 side rows
    (7, 7, 33, 20, 64),     # 24-word class (T' = 17: its lower edge), 32-row region class, small grid
    (6, 50, 34, 11, 64),    # 8-word class exactly filled (T' = 8), past one 48-caption unit, side row
    (5, 7, 51, 9, 40),      # 8-word class on the 48-row region class (T' = 6)
    (4, 9, 60, 12, 64),     # T' = 9: one whole tile; two region tiles per image
    (9, 21, 36, 27, 96),    # 24-word class exactly filled (T' = 24), 3 side rows, captions past one 16-caption unit
    (6, 18, 34, 43, 128),   # 40-word class on the 32-row region class + side row (two-wave 128 x 160 tile)
    (5, 9, 60, 25, 64),     # 24-word class, two region tiles per image (R' = 59)
    (5, 6, 66, 36, 64),     # 40-word class, two region tiles + side row (R' = 65)
    (7, 40, 44, 22, 72),    # 24-word class on the 48-row region class
    (40, 37, 51, 38, 72),   # 40-word class, several workgroup tiles in both directions, ragged
    (5, 4, 57, 90, 64),     # 48 rows + 8 side rows (R' = 56: the class limit), tp16 6
    (4, 4, 58, 38, 64),     # R' = 57: two 32-row tiles
    (7, 3, 51, 60, 64),     # R' = 50 with 64-word captions (tp16 4 does not tile a 96-column strip): two 32-row tiles
    (4, 5, 66, 40, 64),     # two 32-row tiles + side row (R' = 65), tp16 3
    (3, 4, 71, 71, 64),     # evaluation shape: mtiles 3, tp16 5 -> 6
    (2, 2, 97, 99, 32),     # maximum supported regions / tokens
    (40, 24, 34, 50, 72),   # several workgroup tiles, ragged
]


@pytest.mark.parametrize('shape', SWEEP)
def test_alignment_scores_shape_sweep(shape, eval_precision):
    from aladin_amd import ops, synth
    Bi, Bc, R, Tn, D = shape
    im, s, il, sl = synth.alignment_batch(Bi, R, Tn, D, seed=1000 + Bi * 7 + R, ragged=True, Bc=Bc)
    il = [max(2, v) for v in il]
    sl = [max(4, v) for v in sl]
    S = ops.alignment_scores(T(im), T(s), il, sl).cpu().numpy()
    ref = O.alignment_scores(im, s, il, sl)
    assert S.shape == (Bi, Bc)
    if eval_precision == 'split':
        np.testing.assert_allclose(S, ref, rtol=3e-6, atol=3e-6)
    else:
        # tiny D (8..40): few, large vector components, so the fp16 operand rounding (2^-11 relative per
        # component) is not averaged down as at D=768; still within 1e-3 of the score magnitude
        assert_scores_close(S, ref)


@pytest.mark.parametrize('R', [35, 36, 37, 38, 39, 40, 41, 42, 49, 50, 51, 54, 57, 58, 66])
def test_leftover_regions_as_side_rows(R, bwd_mode):
    """R' = 32 + rem (rem = 2..8 side rows per image through the side GEMM), the 48-row class (R' 41..56: R = 42 tile-filling
    copies, 49 exactly 48 rows, 50 / 51 / 54 / 57 with 1 / 2 / 5 / 8 side rows), R' = 57 (two 32-row tiles) and
    R' = 65 (two tiles + one side row): scores vs the oracle, gradients vs the fp32 restatement, ragged lengths
    that put the longest image exactly at R'."""
    from aladin_amd import ops, synth
    Bi, Bc, Tn, D = 24, 18, 40, 256
    im, s, il, sl = synth.alignment_batch(Bi, R, Tn, D, seed=4000 + R, ragged=True, Bc=Bc)
    il[0], il[1], il[2] = R, R - 1, 33                     # full length, one short of it, exactly the main tile
    if R > 49:
        il[3], il[4] = 49, 50                              # exactly the 48 main rows; one side row in use
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    S = ops.alignment_scores(a, b, il, sl)
    assert_scores_close(S.detach().cpu().numpy(), O.alignment_scores(im, s, il, sl))
    w = np.random.default_rng(R).standard_normal((Bi, Bc)).astype(np.float32)
    w *= np.random.default_rng(R + 1).random((Bi, Bc)) < 0.25
    (S * T(w)).sum().backward()
    ra, rb = torch.from_numpy(im).double().requires_grad_(True), torch.from_numpy(s).double().requires_grad_(True)
    (FT.alignment_scores_faithful(ra, rb, il, sl) * torch.from_numpy(w).double()).sum().backward()
    for got, want in ((a.grad, ra.grad), (b.grad, rb.grad)):
        assert_grads_close(got, want, bwd_mode, exact_atol=5e-5)
    # and through the fused triplet node (packed backward with the pair list)
    if Bi == Bc:
        return
    n = min(Bi, Bc)
    a2, b2 = T(im[:n]).requires_grad_(True), T(s[:n]).requires_grad_(True)
    loss, S2 = ops.alignment_triplet_loss(a2, b2, il[:n], sl[:n], 0.2, True)
    loss.backward()
    ref_loss, dS = O.hinge_loss(S2.detach().cpu().numpy(), 0.2, True, return_grad=True)
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-5)
    ra, rb = torch.from_numpy(im[:n]).double().requires_grad_(True), torch.from_numpy(s[:n]).double().requires_grad_(True)
    (FT.alignment_scores_faithful(ra, rb, il[:n], sl[:n]) * torch.from_numpy(dS).double()).sum().backward()
    for got, want in ((a2.grad, ra.grad), (b2.grad, rb.grad)):
        assert_grads_close(got, want, bwd_mode, exact_atol=5e-5)


@pytest.mark.parametrize('shape', [(6, 6, 17, 9, 40), (5, 5, 34, 35, 96), (4, 4, 50, 50, 128), (3, 3, 71, 71, 64),
                                   (12, 12, 34, 50, 100), (10, 10, 51, 38, 768), (7, 7, 65, 20, 256), (9, 9, 42, 30, 64), (8, 8, 57, 38, 128),
                                   (6, 6, 58, 38, 64)])
@pytest.mark.parametrize('mv', [True, False])
def test_alignment_backward_shape_sweep(shape, mv, bwd_mode):
    """Autograd through the differentiable scores + hinge for every backward code path (fp16 pair
    kernel, fp32 fallback for multi-tile / long shapes), against the oracle chained on the HIP S."""
    from aladin_amd import ops, synth
    Bi, Bc, R, Tn, D = shape
    im, s, il, sl = synth.structured_alignment_batch(Bi, R, Tn, D, seed=2000 + R, noise=3.0, ragged=True)
    il = [max(2, v) for v in il]
    sl = [max(4, v) for v in sl]
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    S = ops.alignment_scores(a, b, il, sl)
    loss = ops.hinge_loss(S, 0.2, mv)
    loss.backward()
    S_np = S.detach().cpu().numpy()
    ref_loss, dS = O.hinge_loss(S_np, 0.2, mv, return_grad=True)
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-5, atol=1e-6)
    dim, ds = O.alignment_scores_backward(im, s, il, sl, dS)
    for got, ref in ((a.grad, dim), (b.grad, ds)):
        assert_grads_close(got, ref, bwd_mode)


@pytest.mark.parametrize('R,Tn', [(34, 50), (51, 38)])
def test_sharded_fast_path_rank_logic_emulated_world3(R, Tn):
    """The per-rank pieces of aladin_amd.distributed's fast path, driven for rank 0 and rank 1 on
    ONE GPU (three emulated ranks) with hand-made 'gathered' tensors (concatenation == all-gather, sum == reduce-scatter):
    global loss, score matrix and both gradients must equal the single-device result on the
    concatenated batch.  (51, 38): the shipped data shape -- the 48-row region class with two side rows per image, whose
    packed operands must concatenate across ranks like the headline class's."""
    from aladin_amd import distributed as DD, ops, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    W, B, D = 3, 64, 768
    im, s, il, sl = synth.alignment_batch(W * B, R, Tn, D, seed=4242, ragged=True)
    d = dev()
    g_loc, g_glob, ok = DD._local_and_global_geometry(B, W, R, Tn, D)
    assert ok
    ims = [T(im[r * B:(r + 1) * B]) for r in range(W)]
    caps = [T(s[r * B:(r + 1) * B]) for r in range(W)]
    ilt = [ops.lengths_tensor(il[r * B:(r + 1) * B], d) for r in range(W)]
    slt = [ops.lengths_tensor(sl[r * B:(r + 1) * B], d) for r in range(W)]
    packs = [ops.pack_images(ims[r], ilt[r], g_loc) for r in range(W)]
    xm_all = torch.cat([p[0] for p in packs])
    xe_all = torch.cat([p[1] for p in packs])
    il_all = torch.cat(ilt)
    im_all = torch.cat(ims)
    blocks = [DD.rank_scores_block(xm_all, xe_all, caps[r], slt[r], g_glob) for r in range(W)]
    # the overlapped form (local block first, then the rank ranges before / after) gives the same bits
    for r in range(W):
        y = ops.pack_captions(caps[r], slt[r], g_glob)
        S_blk = torch.full((W * B, B), float('nan'), device=d)
        ops.scores_from_packed(packs[r][0], packs[r][1], y, g_loc, out=S_blk[r * B:(r + 1) * B])
        DD.rank_scores_rows(xm_all, xe_all, y, S_blk, 0, r, B, R, Tn, D)
        DD.rank_scores_rows(xm_all, xe_all, y, S_blk, r + 1, W - 1 - r, B, R, Tn, D)
        assert torch.equal(S_blk, blocks[r][0])
    S_full = torch.cat([b[0] for b in blocks], dim=1)
    loss, dS_full, _ = ops._hinge_raw(S_full, 0.2, True, True)
    d_im_all = torch.zeros_like(im_all)
    d_caps = []
    for r in range(W):
        gi, gs = DD.rank_backward_block(im_all, il_all, caps[r], slt[r], dS_full, r, g_glob, xm_all, xe_all, blocks[r][1])
        d_im_all += gi
        d_caps.append(gs)
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    ref_loss, ref_S = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(a, b, il, sl, return_similarity_mat=True)
    ref_loss.backward()
    assert torch.equal(S_full, ref_S) and torch.equal(loss, ref_loss.detach())
    np.testing.assert_allclose(d_im_all.cpu().numpy(), a.grad.cpu().numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(torch.cat(d_caps).cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-5, atol=1e-8)
    # Round 6: the DEFAULT row step in the sharded node -- every rank packs its images with their inverse norms ([its xm rows | its xe
    # rows], local geometry), the node lays the gathered norms out for the global geometry (distributed.place_image_rnorms) and the
    # caption packer adds the y part.  With three emulated ranks: the assembled buffer must be, bit for bit, what ONE pack of the
    # concatenated batch writes, and the rank contributions under 'fp16' must sum to the single-device step's fp16 gradients.
    n_m, n_e = int(g_loc.xm_rows), int(g_loc.xe_rows)
    rn_ranks = torch.empty((W, n_m + n_e), dtype=torch.float32, device=d)
    for r in range(W):
        ops.pack_images(ims[r], ilt[r], g_loc, rnorm=rn_ranks[r])
    old = ops.set_backward_precision('fp16')
    try:
        d_im16 = torch.zeros_like(im_all)
        d_caps16 = []
        for r in range(W):
            rn_glob = torch.empty(int(g_glob.rnorm_bytes) // 4, dtype=torch.float32, device=d)
            y_r = ops.pack_captions(caps[r], slt[r], g_glob, rnorm=rn_glob)
            DD.place_image_rnorms(rn_glob, rn_ranks, n_m, n_e)
            whole = ops.pack_sets(im_all, caps[r], il_all, slt[r], g_glob)             # (geom, xm, xe, y, rnorm) of the concatenated images
            assert torch.equal(rn_glob, whole[4]) and torch.equal(y_r, whole[3]) and torch.equal(xm_all, whole[1])
            gi, gs = DD.rank_backward_block(im_all, il_all, caps[r], slt[r], dS_full, r, g_glob, xm_all, xe_all, y_r, rnorm=rn_glob)
            d_im16 += gi
            d_caps16.append(gs)
        a16, b16 = T(im).requires_grad_(True), T(s).requires_grad_(True)
        AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(a16, b16, il, sl).backward()
        assert not torch.equal(a16.grad, a.grad)                                        # the fp16 row step really is another arithmetic
        scale = float(a16.grad.abs().max())
        np.testing.assert_allclose(d_im16.cpu().numpy(), a16.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(torch.cat(d_caps16).cpu().numpy(), b16.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)
    finally:
        ops.set_backward_precision(old)


def test_config3_per_rank_block_and_eight_emulated_ranks(eval_precision):
    """BASELINE configs[3] as far as one GPU goes: per-GPU B=256, world 8 -> global 2048 x 2048.
    (1) the block a rank really scores -- all 2048 images x its 256 captions, D=768 -- against the oracle;
    (2) every one of the 8 ranks' pure pieces of the fast sharded step (aladin_amd.distributed), driven on this
        one GPU with concatenation standing in for the all-gather and a sum for the reduce-scatter: the eight
        column blocks are BIT-EQUAL to the single-device 2048 x 2048 matrix, the loss is bit-equal, and the
        summed gradients (dense exchange and pair-driven compact exchange) equal the single-device ones."""
    if eval_precision != 'fp16':
        pytest.skip('differentiable path: fp16 operands either way; run once')
    from aladin_amd import distributed as DD, ops, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    W, B, R, Tn, D = 8, 256, 34, 50, 768
    d = dev()
    parts = [synth.alignment_batch(B, R, Tn, D, seed=1234 + 17 * r, ragged=True) for r in range(W)]     # bench.py's per-rank seeds
    im = np.concatenate([p[0] for p in parts])
    s = np.concatenate([p[1] for p in parts])
    il = [v for p in parts for v in p[2]]
    sl = [v for p in parts for v in p[3]]
    g_loc, g_glob, ok = DD._local_and_global_geometry(B, W, R, Tn, D)
    assert ok and (g_glob.Bi, g_glob.Bc) == (W * B, B)
    ims = [T(p[0]) for p in parts]
    caps = [T(p[1]) for p in parts]
    ilt = [ops.lengths_tensor(p[2], d) for p in parts]
    slt = [ops.lengths_tensor(p[3], d) for p in parts]
    packs = [ops.pack_images(ims[r], ilt[r], g_loc) for r in range(W)]
    xm_all, xe_all = torch.cat([p[0] for p in packs]), torch.cat([p[1] for p in packs])
    il_all, im_all = torch.cat(ilt), torch.cat(ims)
    blocks = [DD.rank_scores_block(xm_all, xe_all, caps[r], slt[r], g_glob) for r in range(W)]
    # (1) rank 3's real block against the oracle (1.25 TFLOP on the host), fp16 tolerance; and in split precision
    r0 = 3
    ref_blk = O.alignment_scores(im, s[r0 * B:(r0 + 1) * B], il, sl[r0 * B:(r0 + 1) * B])
    assert_scores_close(blocks[r0][0].cpu().numpy(), ref_blk)
    with torch.no_grad():
        S_split = ops.alignment_scores(im_all, caps[r0], il, sl[r0 * B:(r0 + 1) * B], precision='split')
    np.testing.assert_allclose(S_split.cpu().numpy(), ref_blk, rtol=0, atol=3e-5)       # the oracle itself is fp32 here
    # (2a) overlapped form == one launch, for every rank
    for r in range(W):
        y = blocks[r][1]
        S_blk = torch.full((W * B, B), float('nan'), device=d)
        ops.scores_from_packed(packs[r][0], packs[r][1], y, g_loc, out=S_blk[r * B:(r + 1) * B])
        DD.rank_scores_rows(xm_all, xe_all, y, S_blk, 0, r, B, R, Tn, D)
        DD.rank_scores_rows(xm_all, xe_all, y, S_blk, r + 1, W - 1 - r, B, R, Tn, D)
        assert torch.equal(S_blk, blocks[r][0])
    # (2b) the eight column blocks are the single-device matrix, bit for bit; same loss
    S_full = torch.cat([b[0] for b in blocks], dim=1)
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    ref_loss, ref_S = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(a, b, il, sl, return_similarity_mat=True)
    ref_loss.backward()
    assert torch.equal(S_full, ref_S.detach())
    loss, dS_full, _ = ops._hinge_raw(S_full, 0.2, True, True)
    assert torch.equal(loss, ref_loss.detach())
    assert int((dS_full != 0).sum()) <= 3 * W * B
    # (2c) dense exchange: every rank's contribution to d(all image sets), summed
    d_im_all = torch.zeros_like(im_all)
    d_caps = []
    for r in range(W):
        gi, gs = DD.rank_backward_block(im_all, il_all, caps[r], slt[r], dS_full, r, g_glob, xm_all, xe_all, blocks[r][1])
        d_im_all += gi
        d_caps.append(gs)
    scale = float(a.grad.abs().max())
    np.testing.assert_allclose(d_im_all.cpu().numpy(), a.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)
    np.testing.assert_allclose(torch.cat(d_caps).cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)
    # (2d) pair-driven exchange: each rank differentiates its block against ONLY the image sets it pairs with
    d_im_sp = torch.zeros_like(im_all)
    needed = []
    for r in range(W):
        blk = dS_full[:, r * B:(r + 1) * B]
        need = torch.nonzero((blk != 0).any(dim=1)).flatten()
        needed.append(int(need.numel()))
        gi, gs = ops._align_backward(im_all.index_select(0, need), caps[r], il_all.index_select(0, need), slt[r],
                                     blk.index_select(0, need).contiguous())
        d_im_sp.index_add_(0, need, gi)
        np.testing.assert_allclose(gs.cpu().numpy(), d_caps[r].cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)
    assert max(needed) < 0.5 * W * B, needed                      # ~600 of 2048 sets per caption block
    np.testing.assert_allclose(d_im_sp.cpu().numpy(), a.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.parametrize('W,mv', [(4, True), (2, False)])
def test_sharded_sparse_exchange_compact_backward_emulated(W, mv):
    """What each rank does under SparseImageExchange, emulated on one GPU: differentiate its caption
    block against ONLY the image sets its dS block touches (compact problem), scatter the result
    back by global index.  Summed over ranks it must equal the single-device gradients."""
    from aladin_amd import ops, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    B, R, Tn, D = 16, 34, 50, 256
    im, s, il, sl = synth.alignment_batch(W * B, R, Tn, D, seed=9090 + W, ragged=True)
    d = dev()
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    ref_loss, S = AlignmentContrastiveLoss(0.2, 'dot', mv, 'MrSw')(a, b, il, sl, return_similarity_mat=True)
    ref_loss.backward()
    _, dS_full, _ = ops._hinge_raw(S, 0.2, mv, True)
    il_all, sl_all = ops.lengths_tensor(il, d), ops.lengths_tensor(sl, d)
    im_all, s_all = T(im), T(s)
    d_im_all = torch.zeros_like(im_all)
    d_caps = []
    touched = 0
    for r in range(W):
        blk = dS_full[:, r * B:(r + 1) * B]
        need = torch.nonzero((blk != 0).any(dim=1)).flatten()
        touched += need.numel()
        gi, gs = ops._align_backward(im_all.index_select(0, need), s_all[r * B:(r + 1) * B], il_all.index_select(0, need),
                                     sl_all[r * B:(r + 1) * B].contiguous(), blk.index_select(0, need).contiguous())
        d_im_all.index_add_(0, need, gi)
        d_caps.append(gs)
    if mv:
        assert touched < W * W * B * 0.8            # the exchange really is sparse
    scale = float(a.grad.abs().max())
    np.testing.assert_allclose(d_im_all.cpu().numpy(), a.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)
    np.testing.assert_allclose(torch.cat(d_caps).cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)


def test_sharded_fast_path_under_rccl_world1(bwd_mode):
    """The fast sharded node on the real backend (one rank) under BOTH backward row steps: since round 6 the node carries the packed
    rows' inverse norms (dense exchange) / packs its compact problem (pair-driven exchange), so the library's default fp16 row step
    runs in it as in the single-GPU node -- same scores and loss bit for bit, gradients equal to the single-device step's and
    within the mode's gate of the oracle's."""
    import os
    import torch.distributed as dist
    from aladin_amd import synth
    from aladin_amd.distributed import sharded_alignment_loss_fast
    from aladin_amd.loss import AlignmentContrastiveLoss
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29612')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    created = False
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev())
        created = True
    try:
        im, s, il, sl = synth.alignment_batch(64, 34, 50, 768, seed=78, ragged=True)
        a1, b1 = T(im).requires_grad_(True), T(s).requires_grad_(True)
        loss1, S1 = sharded_alignment_loss_fast(a1, b1, il, sl, 0.2, True)
        loss1.backward()
        a2, b2 = T(im).requires_grad_(True), T(s).requires_grad_(True)
        loss2, S2 = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(a2, b2, il, sl, return_similarity_mat=True)
        loss2.backward()
        assert torch.equal(S1, S2) and torch.equal(loss1.detach(), loss2.detach())
        assert torch.equal(a1.grad, a2.grad) and torch.equal(b1.grad, b2.grad)
        _, dS = O.hinge_loss(S2.detach().cpu().numpy(), 0.2, True, return_grad=True)
        dim, ds = O.alignment_scores_backward(im, s, il, sl, dS)
        assert_grads_close(a1.grad, dim, bwd_mode)
        assert_grads_close(b1.grad, ds, bwd_mode)
        # the pair-driven exchange (what world > 1 runs under max_violation), forced: RCCL all-to-all + compact backward
        a3, b3 = T(im).requires_grad_(True), T(s).requires_grad_(True)
        loss3, S3 = sharded_alignment_loss_fast(a3, b3, il, sl, 0.2, True, exchange='sparse')
        loss3.backward()
        assert torch.equal(S3, S2) and torch.equal(loss3.detach(), loss2.detach())
        scale = float(a2.grad.abs().max())
        np.testing.assert_allclose(a3.grad.cpu().numpy(), a2.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(b3.grad.cpu().numpy(), b2.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)
        assert_grads_close(a3.grad, dim, bwd_mode)
        assert_grads_close(b3.grad, ds, bwd_mode)
        with pytest.raises(ValueError):
            sharded_alignment_loss_fast(T(im[:10]), T(s[:10]), il[:10], sl[:10])
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize('world,shape,exchanges,bwd_mode', [(2, (64, 34, 50, 768, True), ('dense', 'sparse', 'auto'), 'fp16'),
                                                            (2, (64, 34, 50, 768, True), ('dense', 'sparse'), 'exact'),
                                                            (4, (32, 51, 38, 256, True), ('auto', 'dense'), 'fp16'),
                                                            (3, (64, 34, 50, 128, False), ('sparse',), 'fp16'),
                                                            (8, (64, 34, 50, 256, False), ('auto',), 'fp16')])
def test_fast_sharded_node_across_processes_on_one_gpu(world, shape, exchanges, bwd_mode, eval_precision):
    """The fast sharded step with the REAL kernels at W = 2, 3, 4: the ranks are separate processes sharing this GPU, the collectives go
    through gloo (RCCL refuses two ranks per device; no multi-GPU node exists for this build) -- tests/helpers/gpu_shard_worker.py.
    Against the single-device step on the concatenated batch: the global score matrix and the loss BIT for bit on every rank, the
    gradients of every rank's own samples to 1e-5 (dense and pair-driven exchange, both backward row steps), 'auto' taking the
    pair-driven exchange from W = 4."""
    if eval_precision != 'fp16':
        pytest.skip('training step; run once')
    import os
    import sys
    import torch.multiprocessing as mp
    from aladin_amd import ops, synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    ops.set_backward_precision(bwd_mode)                   # (restored by the autouse fixture) the single-device reference below runs in the ranks' mode
    helpers = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers')
    if helpers not in sys.path:
        sys.path.insert(0, helpers)
    import gpu_shard_worker
    B, R, Tn, D, ragged = shape
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 38500 + (os.getpid() % 2000) + world
    mp.spawn(gpu_shard_worker.worker, args=(world, port, shape, exchanges, bwd_mode, ret), nprocs=world, join=True)
    im, s, il, sl = synth.structured_alignment_batch(B * world, R, Tn, D, seed=4321 + world, noise=3.0, ragged=ragged)
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    ref_loss, ref_S = AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(a, b, il, sl, return_similarity_mat=True)
    (ref_loss * 0.5).backward()
    ga, gb = a.grad.cpu().numpy(), b.grad.cpu().numpy()
    scale = float(np.abs(ga).max())
    for exchange in exchanges:
        want_sparse = exchange == 'sparse' or (exchange == 'auto' and world >= 4)
        assert torch.equal(ret[0][exchange][1], ref_S.detach().cpu())                       # the gathered global matrix = the single-device one, bit for bit
        for r in range(world):
            loss_r, _, S_sum, d_a, d_b, took_sparse = ret[r][exchange]
            assert took_sparse == want_sparse
            assert torch.equal(loss_r, ref_loss.detach().cpu()) and S_sum == ret[0][exchange][2]
            np.testing.assert_allclose(d_a.numpy(), ga[r * B:(r + 1) * B], rtol=1e-5, atol=1e-6 * scale)
            np.testing.assert_allclose(d_b.numpy(), gb[r * B:(r + 1) * B], rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.parametrize('world,loss_type,weights,bwd_mode', [(2, 'alignment-distillation', [1, 1], 'fp16'),
                                                              (4, 'alignment-distillation-matching', [1, 1, 0.1], 'exact')])
def test_sharded_model_heads_across_processes_on_one_gpu(world, loss_type, weights, bwd_mode, eval_precision):
    """ALADModel(config, shard_group=None) -- matching hinge, alignment hinge and ListNet of the GLOBAL batch (the shipped distillation
    YAMLs across ranks) -- with the real kernels at W = 2 and 4: ranks as processes sharing this GPU over gloo.  Total, terms, logger
    entries (n = the global batch) and every rank's gradients against the single-device model on the concatenated batch, before and
    after the distillation epoch."""
    if eval_precision != 'fp16':
        pytest.skip('training step; run once')
    import os
    import sys
    import torch.multiprocessing as mp
    from aladin_amd import ops, synth
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.evaluation import LogCollector
    ops.set_backward_precision(bwd_mode)                   # (restored by the autouse fixture) the ranks are fresh processes: they are told the mode
    helpers = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers')
    if helpers not in sys.path:
        sys.path.insert(0, helpers)
    import gpu_shard_worker
    B, epochs = 64, (0, 5)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 39500 + (os.getpid() % 2000) + world
    mp.spawn(gpu_shard_worker.model_worker, args=(world, port, B, loss_type, weights, epochs, bwd_mode, ret), nprocs=world, join=True)
    config = {'training': {'loss-type': loss_type, 'loss-weights': weights, 'margin': 0.2, 'measure': 'dot',
                           'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}}
    im, s, il, sl = synth.structured_alignment_batch(B * world, 34, 50, 768, seed=977, noise=3.0, ragged=True)
    ie, ce = synth.global_embeddings(B * world, 768, seed=978, noise=3.0)
    for k, epoch in enumerate(epochs):
        m = ALADModel(config)
        m.logger = LogCollector()
        leaves = [T(x).requires_grad_(True) for x in (ie, ce, im, s)]
        loss, d = m.forward_loss_total(leaves[0], leaves[1], leaves[2].permute(1, 0, 2), leaves[3].permute(1, 0, 2), il, sl, 0, epoch=epoch, distill_epoch=2)
        loss.backward()
        for r in range(world):
            loss_r, d_r, log_r, grads_r = ret[r][k]
            np.testing.assert_allclose(loss_r, float(loss), rtol=2e-6)
            assert list(d_r) == list(d)
            for key in d:
                np.testing.assert_allclose(d_r[key], float(d[key]), rtol=2e-6)
            assert list(log_r) == list(m.logger.meters)
            for key, mm in m.logger.meters.items():
                np.testing.assert_allclose(log_r[key][0], mm.val, rtol=2e-6)
                assert log_r[key][1] == mm.count                                        # n = the global batch size on every rank
            for got, leaf in zip(grads_r, leaves):
                if leaf.grad is None or got is None:
                    assert (got is None or not bool(got.any())) and (leaf.grad is None or not bool(leaf.grad.any()))
                    continue
                want = leaf.grad[r * B:(r + 1) * B].cpu().numpy()
                np.testing.assert_allclose(got.numpy(), want, rtol=1e-5, atol=1e-6 * float(np.abs(want).max()) + 1e-12)


@pytest.mark.parametrize('loss_type,weights', [('alignment-distillation', [1, 1]), ('alignment-distillation-matching', [1, 1, 0.1]),
                                               ('matching', [1]), ('alignment', [1])])
def test_sharded_model_step_under_rccl_world1(loss_type, weights, eval_precision):
    """ALADModel(shard_group=None): forward_loss_total over aladin_amd.distributed.sharded_loss_heads on the real backend
    (RCCL, one rank: every collective and autograd wrapper runs on the device) must give the single-device model's
    terms, total, logger entries and gradients -- for the shipped distillation YAML's loss types, before and after
    distill_epoch, at the YAML's bs 32 and at bs 64 (the cross-rank logic is covered under gloo, world 2, in
    tests/test_distributed_cpu.py)."""
    if eval_precision != 'fp16':
        pytest.skip('differentiable path only; run once')
    import os
    import torch.distributed as dist
    from aladin_amd import synth
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.evaluation import LogCollector
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29613')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    created = False
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev())
        created = True
    try:
        config = {'training': {'loss-type': loss_type, 'loss-weights': weights, 'margin': 0.2, 'measure': 'dot',
                               'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}}
        for B in (64, 128):                             # the fast sharded path needs a per-rank batch of whole 64-row groups
            im, s, il, sl = synth.structured_alignment_batch(B, 34, 50, 768, seed=900 + B, noise=3.0, ragged=True)
            ie, ce = synth.global_embeddings(B, 768, seed=901 + B, noise=3.0)
            for epoch in (0, 5):
                res = []
                for shard in (False, None):
                    m = ALADModel(config, shard_group=shard)
                    m.logger = LogCollector()
                    leaves = [T(x).requires_grad_(True) for x in (ie, ce, im, s)]
                    sets = (leaves[0], leaves[1], leaves[2].permute(1, 0, 2), leaves[3].permute(1, 0, 2), il, sl, 0)
                    m.forward_emb = lambda a, b, _s=sets: _s
                    loss, d = m.forward(None, None, epoch=epoch, distill_epoch=2)
                    loss.backward()
                    res.append((loss.detach(), d, m.logger, leaves))
                (l0, d0, g0, x0), (l1, d1, g1, x1) = res
                assert list(d0.keys()) == list(d1.keys())
                np.testing.assert_allclose(float(l1), float(l0), rtol=1e-6)
                for k in d0:
                    np.testing.assert_allclose(float(d1[k]), float(d0[k]), rtol=1e-6)
                assert list(g0.meters.keys()) == list(g1.meters.keys())
                for k in g0.meters:
                    np.testing.assert_allclose(g1.meters[k].val, g0.meters[k].val, rtol=1e-6)
                    assert g1.meters[k].count == g0.meters[k].count              # world 1: global batch == local batch
                for a, b in zip(x0, x1):
                    if a.grad is None or b.grad is None:          # no active term depends on this input: None or zeros
                        assert all(t.grad is None or not bool(t.grad.any()) for t in (a, b))
                        continue
                    scale = float(a.grad.abs().max())
                    np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)
    finally:
        if created:
            dist.destroy_process_group()


def test_backbone_on_the_gpu_matches_the_reference_golden(eval_precision):
    """aladin_amd/backbone.py on PyTorch-ROCm against the outputs the reference's own BertImgModel.forward produced
    (tests/golden/backbone_bertimg.npz; the CPU tier runs the same comparison plus gradients): fused-SDPA and explicit
    attention paths, text-only and tags + regions passes."""
    if eval_precision != 'fp16':
        pytest.skip('no alignment scores involved; run once')
    from test_backbone_cpu import backbone_case
    for explicit in (False, True):
        g, model, (ids, tmask, fmask, types_, feats) = backbone_case(device=dev(), output_attentions=explicit,
                                                                     output_hidden_states=explicit)
        with torch.no_grad():
            o_txt = model(ids, token_type_ids=types_, attention_mask=tmask, img_feats=None)
            o_img = model(ids, token_type_ids=types_, attention_mask=fmask, img_feats=feats)
        tol = dict(rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(o_txt[0].cpu().numpy(), g['txt_seq'], **tol)
        np.testing.assert_allclose(o_img[0].cpu().numpy(), g['img_seq'], **tol)
        np.testing.assert_allclose(o_img[1].cpu().numpy(), g['img_pooled'], **tol)
        if explicit:
            np.testing.assert_allclose(o_img[3][-1].cpu().numpy(), g['img_att_last'], **tol)


def test_matching_head_and_encoder_handoff_vs_reference(eval_precision):
    """aladin_amd.encoder.JointTextImageTransformerEncoder -- slicing to the batch maxima,
    key-padding masks, the 2-layer transformer matching head evaluated for slot 0, F.normalize of the sets, HIP
    l2norm of the globals -- against the 7-tuple and the gradients the REFERENCE's own encoder forward produced
    on the same fake-backbone states and head weights (tests/golden/matching_head.npz)."""
    if eval_precision != 'fp16':
        pytest.skip('no alignment scores involved; run once')
    from conftest import matching_head_case
    g = load_golden('matching_head')
    enc, a, b, cap_len, feat_len, n_tok, w = matching_head_case(g, dev())
    B, n_reg = int(g['B']), int(g['n_reg'])
    ids = torch.zeros((B, n_tok), dtype=torch.long, device=dev())
    img_glob, cap_glob, img_set, cap_seq, fl, cl, reg = enc((ids, None, None, torch.zeros((B, n_reg, 4), device=dev()), None, feat_len),
                                                            (ids, None, None, None, cap_len))
    assert fl == feat_len and cl == cap_len and reg == 0
    assert list(img_set.shape) == list(g['img_set_shape']) and list(cap_seq.shape) == list(g['cap_seq_shape'])
    np.testing.assert_allclose(img_glob.detach().cpu().numpy(), g['img_glob'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(cap_glob.detach().cpu().numpy(), g['cap_glob'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(img_set.detach().cpu().numpy()[:, :, ::16], g['img_set_s'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cap_seq.detach().cpu().numpy()[:, :, ::16], g['cap_seq_s'], rtol=1e-5, atol=1e-6)
    ((img_glob * w[0]).sum() + (cap_glob * w[1]).sum() + 0.05 * (img_set * w[2]).sum() + 0.05 * (cap_seq * w[3]).sum()).backward()
    for got, key in ((a.grad, 'd_txt_seq'), (b.grad, 'd_img_seq')):
        ref = g[key + '_s']
        scale = float(np.abs(ref).max())
        np.testing.assert_allclose(got.cpu().numpy()[:, :, ::16], ref, rtol=1e-3, atol=2e-5 * scale)
        np.testing.assert_allclose(float(got.abs().sum()), float(g[key + '_abs']), rtol=1e-4)
    for n, p in enc.final_projection_net.named_parameters():
        np.testing.assert_allclose(float(p.grad.abs().sum()), float(g['dp_abs__' + n.replace('.', '__')]), rtol=2e-4)


def test_config4_shape_level_step_with_the_real_head(eval_precision):
    """BASELINE configs[4] as far as it goes offline: alad-alignment-and-matching-distill.yaml's sections (values
    copied from the YAML: they are configuration data), batch size 32, the VinVL-base BertImgModel of
    aladin_amd/backbone.py at its real size (12 layers, 768 wide, 2054-d region features) with RANDOM weights (the
    checkpoint cannot be downloaded here), the matching head and the HIP alignment / distillation heads: forward +
    backward of ALADModel.forward, gradients reach the head and the backbone; epoch gating of the distillation term as
    in the reference; evaluation hand-off through encode_data."""
    if eval_precision != 'fp16':
        pytest.skip('training step; run once')
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.backbone import BertConfig, ImageBertForSequenceClassification
    from aladin_amd.evaluation import LogCollector
    config = {'dataset': {'name': 'coco'},
              'model': {'name': 'teran', 'embed-size': 768, 'text-aggregation': 'first', 'image-aggregation': 'first',
                        'freeze-teran': False, 'teran-layers': 0, 'tern-layers': 2, 'post-layers': 0, 'exclude-stopwords': False,
                        'shared-transformer': True, 'depth-aggregation-alignment': False, 'depth-aggregation-matching': False,
                        'dropout': 0.1},
              'training': {'lr': 0.00001, 'grad-clip': 2.0, 'max-violation': True, 'loss-type': 'alignment-distillation',
                           'loss-weights': [1, 1], 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet',
                           'activate_distillation_after': 0, 'measure': 'dot', 'margin': 0.2, 'bs': 32}}
    torch.manual_seed(0)
    backbone = ImageBertForSequenceClassification(BertConfig(vocab_size=3000))       # VinVL base shapes otherwise
    model = ALADModel(config, backbone=backbone).to(dev())
    model.logger = LogCollector()
    model.train()
    bs, n_tok, n_reg = 32, 35, 50
    rng = np.random.default_rng(1)
    cap_len = [int(v) for v in rng.integers(6, n_tok + 1, bs)]
    feat_len = [int(v) for v in rng.integers(10, n_reg + 1, bs)]
    cap_len[0], feat_len[1] = n_tok, n_reg
    ids = torch.from_numpy(rng.integers(1, 3000, (bs, n_tok))).to(dev())
    feats = torch.from_numpy(rng.standard_normal((bs, n_reg, 2054)).astype(np.float32)).to(dev())
    # the collated tuples of alad/dataset.py: (input_ids, attention_mask, token_type_ids, [img_feats,] ..., lengths)
    tmask = (torch.arange(n_tok)[None, :] < torch.tensor(cap_len)[:, None]).long().to(dev())
    rmask = (torch.arange(n_reg)[None, :] < torch.tensor(feat_len)[:, None]).long().to(dev())
    types = torch.zeros_like(ids)
    examples_txts = (ids * tmask, tmask, types, None, cap_len)
    examples_imgs = (ids * tmask, torch.cat([tmask, rmask], 1), types, feats * rmask[:, :, None], None, feat_len)
    loss, d = model(examples_imgs, examples_txts, epoch=3, distill_epoch=2)
    assert list(d.keys()) == ['alignment', 'distillation'] and torch.isfinite(loss)
    loss.backward()
    head = model.img_txt_enc.final_projection_net
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0 for p in head.parameters())
    bert = model.img_txt_enc.oscar_model.bert
    for p in (bert.embeddings.word_embeddings.weight, bert.img_embedding.weight, bert.encoder.layer[0].attention.self.query.weight,
              bert.encoder.layer[11].output.dense.weight):
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0
    assert bert.pooler.dense.weight.grad is None                      # ALADIN never reads the pooled output
    assert list(model.logger.meters.keys()) == ['Eit', 'alignment_loss', 'distillation_loss']
    loss0, d0 = model(examples_imgs, examples_txts, epoch=0, distill_epoch=2)        # distillation popped before distill_epoch (:442-444)
    assert list(d0.keys()) == ['alignment']
    # evaluation hand-off: encode_data over a loader of such batches, then both retrieval heads
    from aladin_amd import evaluation as E

    class _Loader(list):
        dataset = list(range(bs))
    model.eval()
    img_embs, cap_embs, il, cl = E.encode_data(model, _Loader([(examples_imgs, examples_txts)]), logging=None)
    assert img_embs.shape == (bs, 71, 768) and il == feat_len and cl == cap_len


@pytest.mark.parametrize('log', ['deferred', 'sync'])
@pytest.mark.parametrize('loss_type,weights', [('alignment-distillation', [1, 1]), ('alignment-distillation-matching', [1, 1, 0.1])])
def test_graphed_loss_step_equals_eager(eval_precision, loss_type, weights, log):
    """aladin_amd.graphs.GraphedLossStep (HIP-graph replay of forward_loss + weighted sum + backward at the shipped
    batch size 32) gives the eager path's loss, terms, logger entries and input gradients bit for bit, on fresh data and
    fresh lengths of the captured shape, before and after the distillation epoch."""
    if eval_precision != 'fp16':
        pytest.skip('training step; run once')
    from aladin_amd import synth
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.evaluation import LogCollector
    from aladin_amd.graphs import GraphedLossStep
    config = {'training': {'loss-type': loss_type, 'loss-weights': weights, 'margin': 0.2, 'measure': 'dot',
                           'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}}
    B, R, Tn, D = 32, 34, 50, 768
    model = ALADModel(config)
    step = GraphedLossStep(model, log=log)
    for seed, epoch in ((1, 5), (2, 5), (3, 0), (4, 0), (5, 5)):
        im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, D, seed=seed, noise=3.0, ragged=True)
        ge, gc = synth.global_embeddings(B, D, seed=seed + 50, noise=1.0)
        outs = []
        for graphed in (False, True):
            t = [T(ge).requires_grad_(True), T(gc).requires_grad_(True), T(im.transpose(1, 0, 2).copy()).requires_grad_(True),
                 T(s.transpose(1, 0, 2).copy()).requires_grad_(True)]
            model.logger = LogCollector()
            if graphed:
                loss, d = step(t[0], t[1], t[2], t[3], il, sl, epoch=epoch, distill_epoch=2)
            else:
                d = model.forward_loss(t[0], t[1], t[2], t[3], il, sl, 0)
                loss = model.weighted_total(d, epoch, 2)
            (2.0 * loss).backward()                           # a power of two: scaling before or after the kernels gives the same bits
            if graphed and log == 'deferred':
                assert not model.logger.meters                # nothing has made the host wait for the device yet
                step.flush()
            outs.append((loss.detach().clone(), {k: v.detach().clone() for k, v in d.items()}, [None if x.grad is None else x.grad.clone() for x in t],
                         {k: m.val for k, m in model.logger.meters.items()}))
        (l0, d0, g0, log0), (l1, d1, g1, log1) = outs
        assert torch.equal(l0, l1) and list(d0) == list(d1) and log0 == log1
        assert all(torch.equal(d0[k], d1[k]) for k in d0)
        for a, b in zip(g0, g1):
            if a is None or b is None:                                 # no active term depends on that input
                assert (a is None or float(a.abs().max()) == 0.0) and (b is None or float(b.abs().max()) == 0.0)
            else:
                assert torch.equal(a, b)
    assert len(step._cache) == 2                                     # one graph per (shape, distillation active)


@pytest.mark.parametrize('loss_type,weights', [('alignment-distillation', [1, 1]), ('alignment-distillation-matching', [1, 1, 0.1])])
def test_model_graphed_flag_is_a_drop_in(eval_precision, loss_type, weights):
    """ALADModel(config, graphed=True) (or ALADIN_GRAPH_HEADS=1): the reference's train loop UNCHANGED -- `loss, loss_dict = model(imgs,
    txts, epoch=...)`, `loss.backward()`, `str(model.logger)` (alad/train.py:413-447) -- gets the graph replay.  Loss, terms, logger
    entries (current whenever `model.logger` is read), Eiters and the gradients reaching the encoder outputs equal the eager model's
    bit for bit; a no_grad call (validation) runs eagerly."""
    if eval_precision != 'fp16':
        pytest.skip('training step; run once')
    from aladin_amd import synth
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.evaluation import LogCollector
    config = {'training': {'loss-type': loss_type, 'loss-weights': weights, 'margin': 0.2, 'measure': 'dot',
                           'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}}
    B, R, Tn, D = 32, 51, 38, 768
    models = [ALADModel(config, graphed=False), ALADModel(config, graphed=True)]
    assert models[1].graphed and not models[0].graphed
    for m in models:
        m.logger = LogCollector()
    for seed, epoch in ((1, 5), (2, 5), (3, 0), (4, 5)):
        im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, D, seed=seed, noise=3.0, ragged=True)
        ge, gc = synth.global_embeddings(B, D, seed=seed + 50, noise=1.0)
        outs = []
        for m in models:
            t = [T(ge).requires_grad_(True), T(gc).requires_grad_(True), T(im.transpose(1, 0, 2).copy()).requires_grad_(True),
                 T(s.transpose(1, 0, 2).copy()).requires_grad_(True)]
            m.forward_emb = lambda a, b, _t=t: (_t[0], _t[1], _t[2], _t[3], il, sl, 0)
            loss, d = m(None, None, epoch=epoch, distill_epoch=2)            # the reference's call (alad/train.py:416)
            loss.backward()
            outs.append((loss.detach().clone(), {k: v.detach().clone() for k, v in d.items()}, [None if x.grad is None else x.grad.clone() for x in t],
                         {k: (mm.val, mm.count) for k, mm in m.logger.meters.items()}, str(m.logger)))
        (l0, d0, g0, log0, s0), (l1, d1, g1, log1, s1) = outs
        assert torch.equal(l0, l1) and list(d0) == list(d1) and log0 == log1 and s0 == s1 and 'Eit' in log1
        assert all(torch.equal(d0[k], d1[k]) for k in d0)
        for a, b in zip(g0, g1):
            if a is None or b is None:
                assert (a is None or float(a.abs().max()) == 0.0) and (b is None or float(b.abs().max()) == 0.0)
            else:
                assert torch.equal(a, b)
    assert models[1]._graph_step is not None and len(models[1]._graph_step._cache) == 2 and models[0]._graph_step is None
    # validation: no gradients wanted -> the eager path (nothing to capture), same numbers
    with torch.no_grad():
        t = [T(ge), T(gc), T(im.transpose(1, 0, 2).copy()), T(s.transpose(1, 0, 2).copy())]
        vals = []
        for m in models:
            m.forward_emb = lambda a, b, _t=t: (_t[0], _t[1], _t[2], _t[3], il, sl, 0)
            vals.append(m(None, None, epoch=5)[0])
        assert torch.equal(vals[0], vals[1]) and len(models[1]._graph_step._cache) == 2


def test_graphed_loss_step_runs_ahead_safely(eval_precision):
    """With no logger nothing makes the host wait: many steps of the same shape with DIFFERENT lengths are issued back to
    back (the pinned length buffers rotate behind events), every loss must be its own batch's eager loss; and a
    backward() that comes after a later step of the same shape is refused (the graph holds one set of gradient buffers)."""
    if eval_precision != 'fp16':
        pytest.skip('training step; run once')
    from aladin_amd import synth
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.graphs import GraphedLossStep
    config = {'training': {'loss-type': 'alignment-distillation', 'loss-weights': [1, 1], 'margin': 0.2, 'measure': 'dot',
                           'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}}
    B, R, Tn, D = 32, 34, 50, 768
    model = ALADModel(config)
    step = GraphedLossStep(model)
    batches = []
    for seed in range(12):
        im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, D, seed=300 + seed, noise=3.0, ragged=True)
        il[0], sl[0] = R, Tn                                           # same captured shape, different lengths elsewhere
        ge, gc = synth.global_embeddings(B, D, seed=400 + seed, noise=1.0)
        batches.append(([T(ge), T(gc), T(im.transpose(1, 0, 2).copy()), T(s.transpose(1, 0, 2).copy())], il, sl))
    losses = []
    for t, il, sl in batches:                                         # no sync anywhere in this loop
        t = [x.requires_grad_(True) for x in t]
        loss, _ = step(t[0], t[1], t[2], t[3], il, sl, epoch=5)
        loss.backward()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    for (t, il, sl), got in zip(batches, losses):
        d = model.forward_loss(t[0].detach(), t[1].detach(), t[2].detach(), t[3].detach(), il, sl, 0)
        assert torch.equal(model.weighted_total(d, 5, 2), got)
    t, il, sl = batches[0]
    l_a, _ = step(t[0], t[1], t[2], t[3], il, sl, epoch=5)
    l_b, _ = step(t[0], t[1], t[2], t[3], il, sl, epoch=5)
    with pytest.raises(RuntimeError, match='later step'):
        l_a.backward()
    l_b.backward()


@pytest.mark.parametrize('B,D', [(5, 64), (32, 768), (64, 768), (33, 100)])
@pytest.mark.parametrize('mv', [True, False])
def test_small_batch_fused_heads_equal_the_separate_kernels(eval_precision, B, D, mv):
    """ops.small_batch_match_distill (B <= 64: one forward + one backward launch) against the separate kernels
    (exact-fp32 sgemm + hinge + listnet, each pinned to the reference's goldens) and against the oracle."""
    if eval_precision != 'fp16':
        pytest.skip('matching / distillation heads; run once')
    from aladin_amd import ops, synth
    from aladin_amd.loss import ContrastiveLoss, DistillationLoss
    ge, gc = synth.global_embeddings(B, D, seed=700 + B, noise=0.7)
    teacher = (synth.normal((B, B), 800 + B) * 0.8 + 3.0 * np.eye(B, dtype=np.float32) + 4.0).astype(np.float32)
    w = 0.05 * synth.normal((B, B), 900 + B)
    res = []
    for fused in (True, False):
        a, b = T(ge).requires_grad_(True), T(gc).requires_grad_(True)
        if fused:
            lh, ll, M = ops.small_batch_match_distill(a, b, T(teacher), 0.2, mv)
        else:
            lh, M = ContrastiveLoss(0.2, 'dot', mv)(a, b, return_similarity_mat=True)
            ll = DistillationLoss('listnet')(T(teacher), M)
        (0.7 * lh + 1.3 * ll + (M * T(w)).sum()).backward()
        res.append((lh.item(), ll.item(), M.detach().cpu().numpy(), a.grad.cpu().numpy(), b.grad.cpu().numpy()))
    (h1, l1, M1, da1, db1), (h0, l0, M0, da0, db0) = res
    np.testing.assert_allclose(M1, M0, rtol=0, atol=1e-6)
    np.testing.assert_allclose([h1, l1], [h0, l0], rtol=2e-6, atol=1e-6)
    scale = max(float(np.abs(da0).max()), float(np.abs(db0).max()))
    np.testing.assert_allclose(da1, da0, rtol=1e-5, atol=2e-6 * scale)
    np.testing.assert_allclose(db1, db0, rtol=1e-5, atol=2e-6 * scale)
    # oracle: scores, both losses
    Mo = O.dot_scores(ge, gc)
    np.testing.assert_allclose(M1, Mo, rtol=0, atol=2e-6)
    np.testing.assert_allclose(h1, O.hinge_loss(Mo, 0.2, mv), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(l1, O.listnet_loss(teacher, Mo), rtol=1e-5, atol=1e-6)
    # heads switched off
    a, b = T(ge).requires_grad_(True), T(gc).requires_grad_(True)
    lh, ll, M = ops.small_batch_match_distill(a, b, None, 0.2, mv)
    assert ll.item() == 0.0 and abs(lh.item() - h0) <= 2e-6 * max(1.0, abs(h0))
    lh.backward()
    assert torch.isfinite(a.grad).all()
    a, b = T(ge).requires_grad_(True), T(gc).requires_grad_(True)
    lh, ll, M = ops.small_batch_match_distill(a, b, T(teacher), 0.2, mv, want_hinge=False)
    assert lh.item() == 0.0 and abs(ll.item() - l0) <= 2e-6 * max(1.0, abs(l0))
    with pytest.raises(ValueError):
        ops.small_batch_match_distill(T(np.zeros((65, 8), np.float32)), T(np.zeros((65, 8), np.float32)), None, 0.2, True)


@pytest.mark.parametrize('B,R,Tn', [(32, 34, 50), (96, 34, 50), (32, 51, 38)])
@pytest.mark.parametrize('heads', [('matching', 'alignment', 'distillation'), ('alignment', 'distillation'), ('alignment',),
                                   ('matching',), ('distillation',), ('matching', 'distillation')])
def test_small_batch_single_node_step_equals_the_composition(eval_precision, heads, B, R, Tn):
    """ops.small_batch_loss_heads (the whole loss-head step of a bs <= 64 batch as one autograd node, weights inside)
    against the same terms composed from the separate differentiable pieces."""
    if eval_precision != 'fp16':
        pytest.skip('training step; run once')
    from aladin_amd import ops, synth
    D = 768 if B == 32 else 256                               # B = 32: the three-launch heads; B = 96: the general kernels
    im, s, il, sl = synth.structured_alignment_batch(B, R, Tn, D, seed=321, noise=3.0, ragged=True)
    ge, gc = synth.global_embeddings(B, D, seed=322, noise=1.0)
    weights = {'matching': 0.1, 'alignment': 1.0, 'distillation': 0.75}
    ten = lambda: [T(ge).requires_grad_(True), T(gc).requires_grad_(True), T(im).requires_grad_(True), T(s).requires_grad_(True)]
    t1 = ten()
    total, terms, S, M = ops.small_batch_loss_heads(t1[0], t1[1], t1[2], t1[3], il, sl, 0.2, True, heads, weights)
    (2.0 * total).backward()
    t0 = ten()
    ref, vals = 0, [0.0, 0.0, 0.0]
    if 'alignment' in heads or 'distillation' in heads:
        la, S0 = ops.alignment_triplet_loss(t0[2], t0[3], il, sl, 0.2, True)
    if 'matching' in heads or 'distillation' in heads:
        if B <= ops.SMALL_BATCH_MAX:
            lm, ld, M0 = ops.small_batch_match_distill(t0[0], t0[1], S0 if 'distillation' in heads else None, 0.2, True,
                                                       want_hinge='matching' in heads)
        else:
            M0 = ops.dot_scores(t0[0], t0[1])
            lm = ops.hinge_loss(M0, 0.2, True) if 'matching' in heads else None
            ld = ops.listnet_loss(S0, M0) if 'distillation' in heads else None
    if 'matching' in heads:
        ref, vals[0] = ref + lm * weights['matching'], float(lm.detach())
    if 'alignment' in heads:
        ref, vals[1] = ref + la * weights['alignment'], float(la.detach())
    if 'distillation' in heads:
        ref, vals[2] = ref + ld * weights['distillation'], float(ld.detach())
    (2.0 * ref).backward()
    np.testing.assert_allclose(float(total), float(ref), rtol=1e-6)
    got_terms = terms.cpu().numpy()
    for k, name in enumerate(('matching', 'alignment', 'distillation')):
        if name in heads:
            np.testing.assert_allclose(got_terms[k], vals[k], rtol=1e-6, atol=1e-7)
    for a, b in zip(t1, t0):
        if b.grad is None:
            assert a.grad is None or float(a.grad.abs().max()) == 0.0
        else:
            scale = float(b.grad.abs().max())
            np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)


def test_forward_loss_small_and_large_batch_paths_agree_with_the_modules(eval_precision):
    """ALADModel.forward_loss takes the fused small-batch path for B <= 64 and the separate kernels above it: both give
    the terms the criterion modules give on their own."""
    if eval_precision != 'fp16':
        pytest.skip('training step; run once')
    from aladin_amd import synth
    from aladin_amd.alad_model import ALADModel
    config = {'training': {'loss-type': 'alignment-distillation-matching', 'loss-weights': [1, 1, 0.1], 'margin': 0.2,
                           'measure': 'dot', 'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}}
    model = ALADModel(config)
    for B in (48, 80):
        im, s, il, sl = synth.structured_alignment_batch(B, 20, 24, 128, seed=60 + B, noise=2.0, ragged=True)
        ge, gc = synth.global_embeddings(B, 128, seed=70 + B)
        d = model.forward_loss(T(ge), T(gc), T(im).permute(1, 0, 2), T(s).permute(1, 0, 2), il, sl, 0)
        assert list(d) == ['matching', 'alignment', 'distillation']
        lm, M = model.matching_criterion(T(ge), T(gc), return_similarity_mat=True)
        la, S = model.alignment_criterion(T(im), T(s), il, sl, return_similarity_mat=True)
        ld = model.distillation_loss(S, M)
        np.testing.assert_allclose([float(d['matching']), float(d['alignment']), float(d['distillation'])],
                                   [float(lm), float(la), float(ld)], rtol=2e-6, atol=1e-6)


def test_degenerate_lengths_and_zero_vectors():
    """Edge cases the reference's masks define (alad/loss.py:89-116): a caption with s_len = 3 has no
    scored word (its column of S is exactly 0), s_len = 4 one word, im_len = 2 one region, im_len = 1
    none (row of S exactly 0); all-zero vectors inside the valid range normalise to 0 (eps 1e-12),
    not NaN; gradients stay finite and vanish on the dropped positions."""
    from aladin_amd import ops, synth
    B, R, Tn, D = 8, 34, 50, 64
    im, s, il, sl = synth.alignment_batch(B, R, Tn, D, seed=606, ragged=True)
    sl[0], sl[1], sl[2] = 3, 4, 50
    il[0], il[1], il[2] = 1, 2, 34
    im[3, 5] = 0.0                      # a zero region inside the valid range
    s[4, 3] = 0.0                       # a zero word inside the valid range
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    S = ops.alignment_scores(a, b, il, sl)
    ref = O.alignment_scores(im, s, il, sl)
    S_np = S.detach().cpu().numpy()
    assert np.isfinite(S_np).all()
    assert np.all(S_np[:, 0] == 0) and np.all(S_np[0, :] == 0)
    assert_scores_close(S_np, ref)
    loss = ops.hinge_loss(S, 0.2, True)
    loss.backward()
    ga, gb = a.grad.cpu().numpy(), b.grad.cpu().numpy()
    assert np.isfinite(ga).all() and np.isfinite(gb).all()
    assert np.all(ga[0] == 0) and np.all(gb[0] == 0)          # nothing of sample 0 takes part
    assert np.all(ga[1, 2:] == 0) and np.all(gb[1, 2:] == 0)  # one region / one word
    assert np.all(ga[3, 5] == 0) and np.all(gb[4, 3] == 0)    # zero vectors: normalise backward gives 0
    _, dS = O.hinge_loss(S_np, 0.2, True, return_grad=True)
    dim, ds = O.alignment_scores_backward(im, s, il, sl, dS)
    for got, want in ((ga, dim), (gb, ds)):
        scale = max(1e-9, float(np.abs(want).max()))
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=3e-5 * scale)


def test_backward_exact_ties_and_full_candidate_list():
    """Duplicated regions make exact fp32 ties: the argmax the backward records must be the FIRST maximal region
    (numpy / the oracle; alad/loss.py:117 `max` over regions), also when every region of an image is the same
    vector — 33 x 47 close calls for one pair, more than the pair kernel's candidate list holds, so its serial
    fall-back decides — and when two regions coincide."""
    from aladin_amd import ops, synth
    B, R, Tn, D = 6, 34, 50, 768
    im, s, il, sl = synth.alignment_batch(B, R, Tn, D, seed=818, ragged=False)
    im[0, 1:] = im[0, 1]                # every region of image 0 identical
    im[1, 9] = im[1, 5]                 # one duplicated region
    im[2, 1:] = im[2, 1] * np.linspace(1.0, 2.0, R - 1, dtype=np.float32)[:, None]   # same direction, different norms: cosines tie up to rounding
    a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
    S = ops.alignment_scores(a, b, il, sl)
    S_np = S.detach().cpu().numpy()
    assert_scores_close(S_np, O.alignment_scores(im, s, il, sl))
    dS = np.zeros((B, B), dtype=np.float32)
    dS[0, 0], dS[0, 3], dS[1, 1], dS[1, 4], dS[2, 2], dS[3, 0], dS[4, 5] = 1.0, -0.5, 1.0, 0.25, -1.0, 0.75, 2.0
    (S * T(dS)).sum().backward()
    ga, gb = a.grad.cpu().numpy(), b.grad.cpu().numpy()
    dim, ds = O.alignment_scores_backward(im, s, il, sl, dS)
    # image 0 / image 1: exact ties -> first region takes the gradient, exactly as the oracle
    assert np.all(ga[0, 2:] == 0) and np.abs(ga[0, 1]).max() > 0
    for k in (0, 1, 3, 4, 5):
        scale = max(1e-9, float(np.abs(dim[k]).max()))
        np.testing.assert_allclose(ga[k], dim[k], rtol=1e-3, atol=3e-5 * scale)
    scale = max(1e-9, float(np.abs(ds).max()))
    np.testing.assert_allclose(gb[[0, 1, 3, 4, 5]], ds[[0, 1, 3, 4, 5]], rtol=1e-3, atol=3e-5 * scale)
    # image 2: the fp32 cosines of parallel regions differ in the last bit; whichever the reference's rounding picks,
    # the caption gradient is the same vector (the regions normalise to one direction) and finite
    assert np.isfinite(ga[2]).all()
    np.testing.assert_allclose(gb[2], ds[2], rtol=2e-3, atol=1e-4 * scale)


def test_non_contiguous_and_unaligned_inputs():
    """Strided views with a contiguous feature axis are consumed in place; anything else is copied by
    the wrapper (results must not depend on the memory layout)."""
    from aladin_amd import ops, synth
    B, R, Tn, D = 6, 20, 25, 96
    im, s, il, sl = synth.alignment_batch(B, R, Tn, D, seed=707, ragged=True)
    ref = ops.alignment_scores(T(im), T(s), il, sl)
    big_i = torch.zeros((B, R, D + 4), device=dev())
    big_i[:, :, 1:D + 1] = T(im)
    big_s = torch.zeros((Tn, B, D), device=dev())
    big_s.copy_(T(s).permute(1, 0, 2))
    S = ops.alignment_scores(big_i[:, :, 1:D + 1], big_s.permute(1, 0, 2), il, sl)     # misaligned rows + permuted view
    assert torch.equal(S, ref)


def test_single_sample_sides_backward():
    """Regression (found by tools/fuzz_parity.py): with one caption (or one image) the upstream
    gradient is a one-row / one-column matrix whose reported leading stride is arbitrary."""
    import faithful_torch as FT
    from aladin_amd import ops, synth
    for (Bi, Bc, mode) in ((29, 1, 'MwSr'), (1, 9, 'MrSw'), (1, 1, 'symm'), (7, 1, 'MrSw')):
        im, s, il, sl = synth.alignment_batch(Bi, 20, 18, 64, seed=900 + Bi + Bc, ragged=True, Bc=Bc)
        a, b = T(im).requires_grad_(True), T(s).requires_grad_(True)
        S = ops.alignment_scores(a, b, il, sl, mode)
        w = T(np.random.RandomState(Bi).randn(Bi, Bc).astype(np.float32))
        (S * w).sum().backward()
        ra, rb = torch.from_numpy(im).requires_grad_(True), torch.from_numpy(s).requires_grad_(True)
        (FT.alignment_scores_faithful(ra, rb, il, sl, mode) * w.cpu()).sum().backward()
        for got, want in ((a.grad, ra.grad), (b.grad, rb.grad)):
            want = want.numpy()
            scale = max(1e-9, float(np.abs(want).max()))
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-3, atol=5e-5 * scale)
