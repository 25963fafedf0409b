"""per-variant timing of the dense backward (ALADIN_LIB selects the build): the GEMM kernels by rocprof-free event timing of the whole backward"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    from aladin_amd import synth, ops
    from aladin_amd.loss import AlignmentContrastiveLoss
    im, s, il, sl = synth.alignment_batch(256, 34, 50, 768, seed=1234)
    a = torch.from_numpy(im).cuda().requires_grad_(True); b = torch.from_numpy(s).cuda().requires_grad_(True)
    crit = AlignmentContrastiveLoss(0.2, 'dot', False, 'MrSw')
    if os.environ.get('P16'): ops.set_backward_precision('fp16')
    def step():
        a.grad = None; b.grad = None
        crit(a, b, il, sl).backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    print('RESULT %.4f' % (e0.elapsed_time(e1) / 20))
else:
    for v in sys.argv[1:]:
        env = dict(os.environ)
        if v != 'base': env['ALADIN_LIB'] = os.path.join(ROOT, 'aladin_amd', 'lib', 'libdr_%s.so' % v)
        for rep in range(2):
            out = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
            r = [l for l in out.stdout.splitlines() if l.startswith('RESULT')]
            print(v, r[-1] if r else out.stderr[-400:], flush=True)
