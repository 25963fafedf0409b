"""CPU tier: the C-ABI library loads and exports every symbol include/aladin_hip.h declares, the
host-side geometry is sane, and the host logic of the drop-in surface behaves like the reference
(no compute kernels are launched here -- there is no GPU in this container)."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def test_library_exports_every_declared_symbol():
    from aladin_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'aladin_hip.h')).read()
    declared = set(re.findall(r'\b(aladin_[a-z0-9_]+)\s*\(', hdr))
    declared.discard('aladin_align_geom')
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.aladin_version() == _lib.ABI_VERSION


def test_library_is_built_from_the_sources_of_this_tree():
    """The Makefile stamps every library it links with a hash of the kernel sources, their Makefile and the header
    (tools/srchash.py); a library left behind by an experiment, or sources edited without a rebuild, fail here -- before a GPU
    run measures or tests the wrong binary.  (`__graft_entry__.build()` always leaves the two in step.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import srchash
    from aladin_amd import _lib
    stamp = srchash.library_stamp(os.path.join(ROOT, 'aladin_amd', 'lib', 'libaladin_hip.so'))
    assert stamp is not None, 'no source stamp next to the library: rebuild with make -C aladin_amd/csrc'
    assert stamp == srchash.csrc_hash(ROOT), 'the library was built from other sources than this tree holds: make -C aladin_amd/csrc'
    diag = os.path.join(ROOT, 'aladin_amd', 'lib', 'libaladin_hip_diag.so')
    if os.path.exists(diag):
        assert srchash.library_stamp(diag) == srchash.csrc_hash(ROOT), 'stale diagnostic library: make -C aladin_amd/csrc diag'
    del _lib


def test_source_hash_follows_every_input_of_the_build(tmp_path):
    """tools/srchash.py: one byte in a kernel source, a shared header, the Makefile or include/aladin_hip.h changes the hash; files
    outside the build (a stray .txt) do not; a library without a stamp file reads as None."""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import srchash
    root = tmp_path / 'tree'
    shutil.copytree(os.path.join(ROOT, 'aladin_amd', 'csrc'), root / 'aladin_amd' / 'csrc')
    os.makedirs(root / 'include')
    shutil.copy(os.path.join(ROOT, 'include', 'aladin_hip.h'), root / 'include' / 'aladin_hip.h')
    base = srchash.csrc_hash(str(root))
    assert base == srchash.csrc_hash(ROOT) and len(base) == 16
    for rel in ('aladin_amd/csrc/recall.hip', 'aladin_amd/csrc/common.hpp', 'aladin_amd/csrc/Makefile', 'include/aladin_hip.h'):
        f = root / rel
        keep = f.read_bytes()
        f.write_bytes(keep + b'\n')
        assert srchash.csrc_hash(str(root)) != base, rel
        f.write_bytes(keep)
        assert srchash.csrc_hash(str(root)) == base
    (root / 'aladin_amd' / 'csrc' / 'notes.txt').write_text('not a build input')
    assert srchash.csrc_hash(str(root)) == base
    assert srchash.library_stamp(str(root / 'no_such_lib.so')) is None
    (root / 'lib.so.srchash').write_text(base + '\n')
    assert srchash.library_stamp(str(root / 'lib.so')) == base


def test_library_exports_nothing_but_the_declared_symbols():
    """The converse: the product library's dynamic symbol table holds the header's entry points and nothing
    else -- no debug probes, no kernel handles, no C++ helpers (-fvisibility=hidden + csrc/exports.map) -- and
    reads no environment variable (tuning knobs and timing-only ablation kernels live in the separate
    libaladin_hip_diag.so, `make -C aladin_amd/csrc diag`)."""
    import subprocess
    from aladin_amd import _lib
    _lib.load()
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == set(_lib.SYMBOLS), exported ^ set(_lib.SYMBOLS)
    und = subprocess.run(['nm', '-D', '--undefined-only', _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert 'getenv' not in und


def test_geometry_headline_and_edges():
    from aladin_amd import _lib, ops
    g = ops.align_geometry(256, 256, 34, 50, 768)
    assert (g.Rq, g.Tq, g.mrows, g.rem, g.tp16, g.Dp) == (33, 47, 32, 1, 3, 768)
    assert g.xm_rows == 256 * 32 and g.xe_rows == 256 and g.y_rows == 256 * 48
    g = ops.align_geometry(1000, 5000, 71, 71, 768)         # evaluation shape: 70 regions x 68 words
    assert (g.mrows, g.rem, g.tp16) == (96, 0, 6)
    g = ops.align_geometry(3, 7, 3, 5, 8)
    assert (g.Rq, g.Tq, g.mrows, g.rem, g.tp16, g.trows, g.Dp) == (2, 2, 32, 0, 1, 8, 64)
    # the shipped data shape (50 regions + 35 tokens): 48 rows + 2 side rows per image, 40 rows per caption (two captions share
    # five 16-word tiles); split operands and the 32-row region class keep whole tiles
    g = ops.align_geometry(256, 250, 51, 38, 768)
    assert (g.Rq, g.Tq, g.mrows, g.rem, g.tp16, g.trows, g.cap_unit) == (50, 35, 48, 2, 3, 40, 16)
    assert g.Bc_pad == 256 and g.y_rows == 256 * 40 and g.y_bytes == g.y_rows * 768 * 2 and g.e_bytes == g.xe_rows * g.y_rows * 4
    for T_, rows in ((4, 8), (11, 8), (12, 16), (19, 16), (20, 24), (27, 24), (28, 32), (35, 32), (36, 40), (43, 40), (44, 48), (51, 48), (52, 64)):
        for R_ in (34, 40, 51, 60, 66):                              # 32 (+ side rows), 48 + 2, 64 and 64 + 1 main rows
            if R_ == 51 and T_ == 52:
                continue                                             # 64-word captions do not tile the 48-row class's strip
            g = ops.align_geometry(64, 64, R_, T_, 768)
            assert g.trows == rows and g.tp16 == -(-rows // 16) and g.y_rows == g.Bc_pad * rows, (R_, T_, g.trows)
            assert g.cap_unit == {8: 48, 24: 16, 40: 16}[rows] if rows in (8, 24, 40) else g.trows == 16 * g.tp16
    assert ops.align_geometry(256, 256, 51, 38, 768, precision='split').trows == 40       # split operands: the same layout, K x 3
    assert ops.align_geometry(9, 9, 71, 38, 64).trows == 48                              # three region tiles per image: whole tiles
    for g in (ops.align_geometry(256, 256, 34, 50, 768), ops.align_geometry(9, 9, 71, 71, 64), ops.align_geometry(9, 9, 51, 60, 64)):
        assert g.trows == 16 * g.tp16
    gs = ops.align_geometry(256, 256, 34, 50, 768, precision='split')          # hi/lo split operands: three K segments per row
    g = ops.align_geometry(256, 256, 34, 50, 768)
    assert gs.split == 1 and g.split == 0 and gs.Dp == 3 * g.Dp and gs.xm_bytes == 3 * g.xm_bytes and gs.e_bytes == g.e_bytes
    assert (gs.xm_rows, gs.xe_rows, gs.y_rows) == (g.xm_rows, g.xe_rows, g.y_rows)
    assert g.Bi_pad % g.img_unit == 0 and g.Bc_pad % g.cap_unit == 0 and g.Bi_pad >= 3 and g.Bc_pad >= 7
    for args in ((1, 1, 1, 50, 8), (1, 1, 34, 3, 8), (0, 1, 34, 50, 8), (1, 1, 200, 50, 8)):
        with pytest.raises(RuntimeError):
            ops.align_geometry(*args)
    assert _lib.load().aladin_last_error()


def test_workspace_queries():
    from aladin_amd import _lib
    lib = _lib.load()
    assert lib.aladin_hinge_workspace_bytes(256) >= 256 * 16
    assert lib.aladin_listnet_workspace_bytes(256) >= 256 * 48
    import ctypes as C
    from aladin_amd import ops
    g = ops.align_geometry(256, 256, 34, 50, 768)
    base = lib.aladin_align_bwd_workspace_bytes(C.byref(g), 0)
    assert base >= 256 * 256 * (4 + 47)
    assert lib.aladin_align_bwd_workspace_bytes(C.byref(g), _lib.BWD_DENSE) > base            # + the split operands and GEMM partials
    # the fused training node: side-GEMM scratch + hinge statistics + the base workspace
    assert lib.aladin_align_triplet_workspace_bytes(C.byref(g)) >= base + g.e_bytes + lib.aladin_hinge_workspace_bytes(256)
    assert g.rnorm_bytes == 4 * (g.xm_rows + g.xe_rows + g.y_rows)
    assert lib.aladin_sim_workspace_bytes(5000, 25000, 768) >= (5000 + 25000) * 2 * 768 * 2      # [hi | lo] rows
    assert lib.aladin_recall_workspace_bytes(25000) == 25000 * 8


def test_cpu_tensors_are_rejected_not_silently_computed():
    from aladin_amd import ops
    from aladin_amd.loss import AlignmentContrastiveLoss, ContrastiveLoss, DistillationLoss
    im, s = torch.randn(2, 5, 8), torch.randn(2, 7, 8)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.alignment_scores(im, s, [5, 5], [7, 7])
    with pytest.raises(RuntimeError):
        AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw')(im, s, [5, 5], [7, 7])
    with pytest.raises(RuntimeError):
        ContrastiveLoss(0.2, 'dot', True)(torch.randn(3, 8), torch.randn(3, 8))
    with pytest.raises(RuntimeError):
        DistillationLoss('listnet')(torch.randn(3, 3), torch.randn(3, 3))


def test_every_distillation_mode_and_measure_constructs_and_refuses_cpu():
    from aladin_amd.loss import Contrastive, ContrastiveLoss, DistillationLoss, order_sim
    assert Contrastive(measure='order').sim is order_sim
    for mode in ('mse', 'ordinal', 'contrastive', 'listnet'):
        crit = DistillationLoss(mode=mode)
        # the reference's 'mse' mode owns a learnable pair (alad/loss.py:366); the others nothing
        assert sorted(crit.state_dict()) == (['wb'] if mode == 'mse' else [])
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            crit(torch.randn(4, 4), torch.randn(4, 4))
    with pytest.raises(ValueError):
        DistillationLoss(mode='nope')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ContrastiveLoss(0.2, 'order', True)(torch.randn(3, 8), torch.randn(3, 8))


def test_loss_modules_hold_no_state():
    """State-dict compatibility with reference checkpoints."""
    from aladin_amd.loss import AlignmentContrastiveLoss, ContrastiveLoss, DistillationLoss
    for m in (AlignmentContrastiveLoss(0.2, 'dot', True, 'MrSw'), ContrastiveLoss(0.2, 'dot', True),
              DistillationLoss('listnet')):
        assert len(m.state_dict()) == 0


class _StubCriterion(torch.nn.Module):
    def __init__(self, loss, mat):
        super().__init__()
        self.loss, self.mat, self.calls = loss, mat, 0

    def forward(self, *a, return_similarity_mat=False, **k):
        self.calls += 1
        return (self.loss, self.mat) if return_similarity_mat else self.loss


def test_alad_model_orchestration_on_stubs():
    """Loss-type gating, insertion order, logger keys, distill-epoch pop and weighting
    (reference alad_model.py:371-454) -- pure host logic, checked against the golden dicts."""
    from aladin_amd.alad_model import ALADModel
    from aladin_amd.evaluation import LogCollector
    g = load_golden('model_forward')
    for fn in g['configs']:
        key = str(fn)[:-5].replace('-', '_').replace('.', '_')
        lt = str(g[key + '__loss_type'])
        weights = [float(w) for w in g[key + '__weights']]
        config = {'training': {'loss-type': lt, 'loss-weights': weights, 'margin': 0.2, 'measure': 'dot',
                               'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}}
        m = ALADModel(config)
        ref5 = dict(zip([str(k) for k in g[key + '__e5_keys']], g[key + '__e5_vals']))
        m.matching_criterion = _StubCriterion(torch.tensor(float(ref5.get('matching', 0.5))), torch.zeros(2, 2))
        m.alignment_criterion = _StubCriterion(torch.tensor(float(ref5.get('alignment', 0.7))), torch.zeros(2, 2))
        m.distillation_loss = _StubCriterion(torch.tensor(float(ref5.get('distillation', 0.9))), None)
        m.logger = LogCollector()
        B, D = 2, 4
        sets = (torch.zeros(B, D), torch.zeros(B, D), torch.zeros(3, B, D), torch.zeros(5, B, D), [3, 3], [5, 5], 0)
        m.forward_emb = lambda a, b, _s=sets: _s
        for epoch in (0, 5):
            loss, d = m.forward(None, None, epoch=epoch, distill_epoch=2)
            assert list(d.keys()) == [str(k) for k in g['%s__e%d_keys' % (key, epoch)]]
            np.testing.assert_allclose(float(loss), float(g['%s__e%d_total' % (key, epoch)]), rtol=1e-5)
        assert list(m.logger.meters.keys()) == [str(k) for k in g[key + '__logged']]
        assert m.Eiters == 2 and m.matching_criterion.calls == 2          # matching is ALWAYS computed (:380)


def test_alad_model_auto_weights_and_missing_encoder():
    from aladin_amd.alad_model import ALADModel
    config = {'training': {'loss-type': 'alignment', 'loss-weights': 'auto', 'margin': 0.2, 'measure': 'dot',
                           'max-violation': True, 'alignment-mode': 'MrSw', 'distillation-mode': 'listnet'}}
    m = ALADModel(config)
    assert m.auto_weight and set(m.losses_weights) == {'alignment'}
    assert len(list(m.parameters())) == 0            # the reference's 'auto' weights are NOT registered (:272)
    with pytest.raises(RuntimeError, match='no encoder'):
        m.forward_emb([], [])


def test_metrics_and_meters():
    from aladin_amd.evaluation import AverageMeter, LogCollector, _metrics
    ranks = np.array([0, 0, 3, 7, 12, 40], dtype=np.float64)
    r1, r5, r10, medr, meanr = _metrics(ranks)
    assert (r1, r5, r10) == (100.0 * 2 / 6, 100.0 * 3 / 6, 100.0 * 4 / 6)
    assert medr == np.floor(np.median(ranks)) + 1 and meanr == ranks.mean() + 1
    lc = LogCollector()
    lc.update('Eit', 3)
    lc.update('loss', 2.0, 4)
    lc.update('loss', 4.0, 4)
    assert str(lc) == 'Eit 3  loss 4.0000 (3.0000)'
    assert isinstance(lc.meters['loss'], AverageMeter)


def test_synth_is_deterministic():
    from aladin_amd import synth
    a = synth.normal((4, 5), 7)
    b = synth.normal((4, 5), 7)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert abs(float(synth.normal((200000,), 3).std()) - 1.0) < 0.01


def test_encode_data_store_layout():
    """encode_data mirror: (N, 71, D) zero-padded store, slot 0 = global embedding, lengths in order
    (reference alad/evaluation.py:119-130).  Runs on CPU tensors: it is pure tensor plumbing."""
    from aladin_amd.evaluation import encode_data

    class FakeModel:
        logger = None

        def eval(self):
            return self

        def forward_emb(self, imgs, txts):
            feats, lens = imgs
            toks, tl = txts
            B = feats.shape[0]
            R, Tn = max(lens), max(tl)
            return (feats[:, 0, :8] * 0 + 7.0, toks[:, 0, :8] * 0 - 7.0, feats[:, :R, :8].permute(1, 0, 2),
                    toks[:, :Tn, :8].permute(1, 0, 2), list(lens), list(tl), 0)

    class DS(list):
        pass

    batches = []
    rng = np.random.RandomState(0)
    for b in range(3):
        lens, tl = [5, 9], [4, 6]
        batches.append(((torch.from_numpy(rng.randn(2, 12, 8).astype(np.float32)), lens),
                        (torch.from_numpy(rng.randn(2, 10, 8).astype(np.float32)), tl)))

    class Loader(list):
        dataset = list(range(6))

    img, cap, il, cl = encode_data(FakeModel(), Loader(batches), logging=None)
    assert img.shape == (6, 71, 8) and cap.shape == (6, 71, 8)
    assert il == [5, 9] * 3 and cl == [4, 6] * 3
    assert torch.all(img[:, 0, :] == 7.0) and torch.all(cap[:, 0, :] == -7.0)
    assert torch.equal(img[2:4, 1:9, :], batches[1][0][0][:, 1:9, :8]) and torch.all(img[:, 9:, :] == 0)
    assert torch.equal(cap[4:6, 1:6, :], batches[2][1][0][:, 1:6, :8]) and torch.all(cap[:, 6:, :] == 0)


# ------------------------------------------------------------------------------------------------
# The i2t / t2i score-grid memo (aladin_amd/evaluation.py): a hit needs the same LIVE input objects.
# ------------------------------------------------------------------------------------------------
def test_eval_grid_memo_is_keyed_on_live_objects_not_addresses(monkeypatch):
    """Two validations in one process: encode_data returns fresh buffers of the same shape, filled by the same number of
    in-place writes, and the allocator hands them the block the previous ones freed.  The second validation must NOT be
    served the first one's grid (reference alad/evaluation.py:158-327 recomputes every call; train.py:504-509 calls
    i2t then t2i on the same embeddings, which is the one reuse the memo exists for)."""
    import gc
    import weakref
    import torch
    from aladin_amd import evaluation as E
    calls = []

    def fake_scores(images, captions, il, cl, measure, fn):
        calls.append(float(images.sum()))
        return torch.full((len(images) // 5, len(captions)), float(images.sum()))
    monkeypatch.setattr(E, '_eval_scores_uncached', fake_scores)
    E.clear_eval_cache()

    def buffers(fill):
        a, b = torch.zeros((10, 7, 4)), torch.zeros((10, 7, 4))
        a[:, :3] = fill                                  # same number of in-place fills whatever the content
        b[:, :3] = -fill
        return a, b
    il, cl = [5] * 10, [6] * 10
    a, b = buffers(1.0)
    key_before = E._memo_key(a, b, il, cl, 'dot', None)[0]
    s1 = E._eval_scores(a, b, il, cl, 'dot', None)
    s2 = E._eval_scores(a, b, il, cl, 'dot', None)          # i2t then t2i: one grid
    assert len(calls) == 1 and s1 is s2
    wa = weakref.ref(a)
    del a, b, s1, s2
    gc.collect()
    assert wa() is None, 'the memo must not keep the embedding buffers alive'
    a, b = buffers(2.0)                                  # the next validation: same shapes, versions, very likely same addresses
    assert E._memo_key(a, b, il, cl, 'dot', None)[0][0][2:] == key_before[0][2:]      # indistinguishable by shape / version
    s3 = E._eval_scores(a, b, il, cl, 'dot', None)
    assert len(calls) == 2 and float(s3[0, 0]) == calls[1] != calls[0]
    a[0, 0, 0] += 1.0                                    # in-place update: version counter moves
    E._eval_scores(a, b, il, cl, 'dot', None)
    assert len(calls) == 3
    E._eval_scores(a, b, il, [7] * 10, 'dot', None)      # other lengths
    assert len(calls) == 4

    class Crit:
        def sim(self, *args):
            return None
    c = Crit()
    E._eval_scores(a, b, il, cl, 'dot', c.sim)            # a bound method is a new object per access: never memoised
    E._eval_scores(a, b, il, cl, 'dot', c.sim)
    assert len(calls) == 6
    fn = lambda *args: None                              # noqa: E731  a plain function object is tracked
    E._eval_scores(a, b, il, cl, 'dot', fn)
    E._eval_scores(a, b, il, cl, 'dot', fn)
    assert len(calls) == 7
    E._eval_scores(a.numpy(), b.numpy(), il, cl, 'dot', None)     # numpy inputs have no version counter: not memoised
    E._eval_scores(a.numpy(), b.numpy(), il, cl, 'dot', None)
    assert len(calls) == 9
    E.clear_eval_cache()


def test_bucket_classes_follow_the_library_geometry():
    """ops.X_CLASS_BOUNDS / Y_CLASS_BOUNDS are the planner's cost model of the packed geometry: every bound must be the top
    of a class the library really builds (main rows + side rows == the bound, nothing padded past it), and a ragged COCO-like
    grid (up to 50 boxes + the global slot) must put its long images in the 48-row classes, not the 64-row one."""
    from aladin_amd import ops
    for b in ops.X_CLASS_BOUNDS:
        g = ops.align_geometry(512, 512, b + 1, 50, 768)         # R = b + 1 set positions -> b scored positions
        assert g.Rq == b and g.mrows + g.rem == b, (b, g.mrows, g.rem)
        if b < ops.X_CLASS_BOUNDS[-1]:
            g = ops.align_geometry(512, 512, b + 2, 50, 768)     # one more position leaves the class
            assert g.mrows + g.rem > b
    for b in ops.Y_CLASS_BOUNDS:
        g = ops.align_geometry(512, 512, 34, b + 3, 768)         # T = b + 3 set positions -> b scored words
        assert g.Tq == b and g.trows == b
    rng = np.random.RandomState(5)
    il = [int(v) for v in np.minimum(rng.randint(18, 70, size=1000), 51)]        # VinVL-like: most images clipped at 50 boxes
    cl = [int(v) for v in rng.randint(7, 30, size=5000)]
    x_need, y_need = ops._needed_positions(il, 0, 71, True), ops._needed_positions(cl, 2, 71, False)
    assert max(x_need) == 51
    plan = ops.bucket_plan(x_need, y_need)
    assert plan is not None
    tops = sorted(max(x_need[k] for k in g) for g in plan[0])
    assert tops[-1] == 51 and len(tops) >= 2
    assert sorted(k for g in plan[0] for k in g) == list(range(1000)) and sorted(k for g in plan[1] for k in g) == list(range(5000))


# where each callable of tests/golden/signatures.json (inspect.signature of the REFERENCE, tests/golden/make_golden.py
# gen_signatures) lives in this package
_SIGNATURE_HOMES = {
    'AlignmentContrastiveLoss.__init__': ('loss', 'AlignmentContrastiveLoss.__init__'),
    'AlignmentContrastiveLoss.forward': ('loss', 'AlignmentContrastiveLoss.forward'),
    'ContrastiveLoss.__init__': ('loss', 'ContrastiveLoss.__init__'),
    'ContrastiveLoss.forward': ('loss', 'ContrastiveLoss.forward'),
    'DistillationLoss.__init__': ('loss', 'DistillationLoss.__init__'),
    'DistillationLoss.forward': ('loss', 'DistillationLoss.forward'),
    'Contrastive.__init__': ('loss', 'Contrastive.__init__'),
    'Contrastive.compute_contrastive_loss': ('loss', 'Contrastive.compute_contrastive_loss'),
    'dot_sim': ('loss', 'dot_sim'), 'cosine_sim': ('loss', 'cosine_sim'), 'order_sim': ('loss', 'order_sim'),
    'l2norm': ('loss', 'l2norm'),
    'ALADModel.forward': ('alad_model', 'ALADModel.forward'),
    'ALADModel.forward_emb': ('alad_model', 'ALADModel.forward_emb'),
    'ALADModel.forward_loss': ('alad_model', 'ALADModel.forward_loss'),
    'recall': ('evaluation', 'recall'), 'recall_test': ('evaluation', 'recall_test'),
    'compute_recall': ('evaluation', 'compute_recall'), 'recall_1k_5fold_test': ('evaluation', 'recall_1k_5fold_test'),
    'i2t': ('evaluation', 'i2t'), 't2i': ('evaluation', 't2i'), 'encode_data': ('evaluation', 'encode_data'),
}


def test_drop_in_signatures_match_the_reference():
    """SURVEY 8(b) / VERDICT r4 item 7: every callable of the drop-in surface takes the reference's parameters -- same
    names, same order, same kinds, same defaults (compared as repr strings) -- so positional and keyword call sites of the
    reference (alad_model.py:380,386,405; train.py:479,493-509; test.py:259-276) bind identically.  The drop-in may add
    TRAILING parameters, each with a default (log=, verbose=, precision=...)."""
    import importlib
    import inspect
    import json
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'signatures.json')))
    assert set(ref) == set(_SIGNATURE_HOMES), set(ref) ^ set(_SIGNATURE_HOMES)
    for key, want in ref.items():
        mod, dotted = _SIGNATURE_HOMES[key]
        obj = importlib.import_module('aladin_amd.' + mod)
        for part in dotted.split('.'):
            obj = getattr(obj, part)
        got = [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
               for p in inspect.signature(obj).parameters.values()]
        assert got[:len(want)] == want, (key, got, want)
        for extra in got[len(want):]:
            assert extra[2] is not None or extra[1] in ('VAR_POSITIONAL', 'VAR_KEYWORD'), (key, 'extra parameter without a default', extra)


def test_binding_matches_the_header_argument_for_argument():
    """Every prototype of include/aladin_hip.h against the ctypes binding: same number of arguments, pointers where the header
    has pointers, 64-bit integers / floats / ints where it has them (a stale entry in the binding's table -- the same key twice --
    once shadowed a changed signature until the GPU tier ran), and no key listed twice in the table's source."""
    import ctypes as C
    from aladin_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'aladin_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    protos = re.findall(r'ALADIN_API\s+([\w\s\*]+?)\s*\b(aladin_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;', hdr, flags=re.S)
    assert {n for _, n, _ in protos} == set(_lib.SYMBOLS)

    def kind(decl):
        decl = decl.strip()
        if '*' in decl:
            return 'ptr'
        base = decl.split()[:-1] if len(decl.split()) > 1 else decl.split()
        base = ' '.join(b for b in base if b != 'const')
        return {'int': 'i32', 'int32_t': 'i32', 'int64_t': 'i64', 'float': 'f32', 'size_t': 'sz'}[base]

    for _, name, args in protos:
        args = args.strip()
        want = [] if args in ('', 'void') else [kind(a) for a in args.split(',')]
        got = []
        for t in getattr(lib, name).argtypes:
            if t in (C.c_void_p, C.c_char_p) or hasattr(t, 'contents') or issubclass(t, C._Pointer):
                got.append('ptr')
            else:
                got.append({C.c_int: 'i32', C.c_int64: 'i64', C.c_float: 'f32', C.c_size_t: 'sz'}[t])
        assert got == want, (name, got, want)
    src = open(os.path.join(ROOT, 'aladin_amd', '_lib.py')).read()
    table = src[src.index('    sig = {'):src.index('    for name, (res, args) in sig.items():')]
    keys = re.findall(r"^\s+'(aladin_[a-z0-9_]+)':", table, flags=re.M)
    assert len(keys) == len(set(keys)), sorted(k for k in set(keys) if keys.count(k) > 1)
