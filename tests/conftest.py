import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionfinish(session, exitstatus):
    """GPU tier: keep the measured score errors (tests/test_gpu_parity.py: assert_scores_close) next to the run's other
    outputs, so that the tolerance the tests state can be read against what was measured."""
    import json
    mod = sys.modules.get('test_gpu_parity')
    log = getattr(mod, 'SCORE_ERR_LOG', None)
    glog = getattr(mod, 'GRAD_ERR_LOG', None)
    grads = None
    if glog:
        # gradient VALUES vs reference / oracle per backward row mode (assert_grads_close): worst call per test, and the worst of each mode
        worst = {}
        for test, d in glog:
            cur = worst.get(test)
            if cur is None or d['max_err_over_max_ref'] > cur['max_err_over_max_ref']:
                worst[test] = d
        grads = {'note': 'gradients vs reference / oracle: max |err| / max |ref| per test; gate = the absolute term the test held on top of rtol 1e-3',
                 'worst_by_mode': {m: max((v['max_err_over_max_ref'] for v in worst.values() if v['mode'] == m), default=None) for m in ('exact', 'fp16')},
                 'worst_fp16_D_ge_64': max((v['max_err_over_max_ref'] for v in worst.values() if v['mode'] == 'fp16' and v['D'] >= 64), default=None),
                 'tests': worst}
    if log:
        out = os.path.join(ROOT, 'gpurun_out')
        try:
            os.makedirs(out, exist_ok=True)
            worst = {}
            for test, d in log:
                cur = worst.get(test)
                if cur is None or d['frac_of_tol'] > cur['frac_of_tol']:
                    worst[test] = d
            with open(os.path.join(out, 'score_err_stats.json'), 'w') as f:
                json.dump({'note': 'fp16-operand score matrices vs reference / oracle, worst call per test (assert_scores_close)',
                           'max_frac_of_tol': max(v['frac_of_tol'] for v in worst.values()),
                           'max_err_over_max_ref': max(v['max_err_over_max_ref'] for v in worst.values()),
                           'max_rel_err_where_ref_ge_tenth_of_max': max(v['max_rel_err_where_ref_ge_tenth_of_max'] for v in worst.values()),
                           'tests': worst, 'gradients': grads}, f, indent=1)
        except OSError:
            pass


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def golden_alignment_inputs(g):
    """Regenerate the inputs of an alignment golden from its stored generator arguments and
    check the stored checksums (pins aladin_amd.synth)."""
    from aladin_amd import synth
    kind = str(g['kind'])
    B, Bc, R, T, D, seed = (int(g[k]) for k in ('B', 'Bc', 'R', 'T', 'D', 'seed'))
    ragged = bool(g['ragged'])
    if kind.startswith('structured'):
        noise = float(kind[len('structured'):] or 1.0)
        im, s, il, sl = synth.structured_alignment_batch(B, R, T, D, seed, noise, ragged)
    else:
        im, s, il, sl = synth.alignment_batch(B, R, T, D, seed, ragged, Bc=Bc)
    assert il == [int(v) for v in g['im_len']] and sl == [int(v) for v in g['s_len']]
    assert abs(synth.checksum(im) - float(g['im_checksum'])) <= 1e-6 * max(1.0, abs(float(g['im_checksum'])))
    assert abs(synth.checksum(s) - float(g['s_checksum'])) <= 1e-6 * max(1.0, abs(float(g['s_checksum'])))
    return im, s, il, sl


ALIGN_GOLDENS = ['align_tiny', 'align_b5_d64', 'align_b12_struct', 'align_b32_d64',
                 'align_b16_d768', 'align_b8_d768_rag', 'align_rect', 'align_r33',
                 'align_vinvl_b12', 'align_t27_b10', 'align_t11_rect']          # round 4: 48-row x 40-word, 24-word and 8-word tile classes
SQUARE_ALIGN_GOLDENS = [n for n in ALIGN_GOLDENS if n not in ('align_rect', 'align_t11_rect')]


@pytest.fixture(scope='session')
def golden():
    return load_golden


def matching_head_case(g, device):
    """Inputs, weights and upstream gradients of tests/golden/matching_head.npz (see make_golden.gen_matching_head):
    -> (encoder with the golden's head weights and a fake backbone, txt_seq, img_seq, cap_len, feat_len, n_tok, w)."""
    import torch
    from aladin_amd import synth
    from aladin_amd.encoder import JointTextImageTransformerEncoder
    D, B, n_tok, n_reg, seed = (int(g[k]) for k in ('D', 'B', 'n_tok', 'n_reg', 'seed'))
    cap_len, feat_len = [int(v) for v in g['cap_len']], [int(v) for v in g['feat_len']]
    a = torch.from_numpy(synth.normal((B, n_tok, D), seed)).to(device).requires_grad_(True)
    b = torch.from_numpy(synth.normal((B, n_tok + n_reg, D), seed + 1)).to(device).requires_grad_(True)

    class FakeBackbone(torch.nn.Module):
        def bert(self, input_ids, attention_mask, token_type_ids, img_feats):
            return (a,) if img_feats is None else (b,)
    cfg = {'model': {'embed-size': D, 'teran-layers': 0, 'tern-layers': 2, 'post-layers': 0, 'dropout': 0.1,
                     'shared-transformer': True}, 'training': {'loss-type': 'alignment-distillation', 'measure': 'dot'}}
    enc = JointTextImageTransformerEncoder(cfg, FakeBackbone()).to(device).eval()
    named = [(n, tuple(p.shape)) for n, p in enc.final_projection_net.named_parameters()]
    vals = synth.module_parameters(named, seed + 7)
    with torch.no_grad():
        for n, p in enc.final_projection_net.named_parameters():
            p.copy_(torch.from_numpy(vals[n]))
    shapes = [(B, D), (B, D), (max(feat_len), B, D), (max(cap_len), B, D)]
    w = [torch.from_numpy(synth.normal(sh, seed + 20 + k)).to(device) for k, sh in enumerate(shapes)]
    return enc, a, b, cap_len, feat_len, n_tok, w
