#!/usr/bin/env python3
"""bench.py -- alignment image-text pairs/sec on MI355X (BASELINE.json metric).

A step = ONE forward + backward of the alignment-head triplet loss
(AlignmentContrastiveLoss(margin=.2,'dot',max_violation=True,'MrSw'), reference
alad/configs/alad-alignment-triplet.yaml) on synthetic features with B=256 per GPU, R=34, T=50,
D=768 (BASELINE.json configs[1]; configs[3] when --gpus > 1: images all-gathered over RCCL, each
rank scores the (N*256 x 256) caption block, hinge on the global matrix).
Inputs are resident in HBM before the timed region.  value = (N*256)^2 pairs / step time.

    python bench.py [--gpus N] [--steps K] [--warmup W]      # N > 1: starts its own N ranks as child processes
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # or is started as one of them

Timing protocol (so that a short driver run and a long builder run report the same number):
  1. clock-settling pre-roll: the step is replayed for --preroll-s seconds (default 1.0) whatever --warmup says
     -- the chip needs ~0.5 s of load before it holds its working clock;
  2. W untimed warm-up steps;
  3. the K-step region (barrier + synchronize on both sides, MAX over ranks) is timed --repeats times
     (default 5); ms_per_step / value are the MEDIAN region, min / max are in `config`.
Prints ONE JSON line on rank 0 (see DESIGN.md section "Measurement" for every field).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, R, T, D = 256, 34, 50, 768
FLOPS_PER_PAIR = 2 * (R - 1) * (T - 3) * D           # 2,382,336
PEAK_TFLOPS = 2500.0                                 # dense 16-bit MFMA, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--repeats', type=int, default=5, help='how many times the --steps region is timed (median reported)')
    ap.add_argument('--preroll-s', type=float, default=1.0, help='clock-settling pre-roll before the warm-up, seconds')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-eval', action='store_true', help='skip the secondary timings (configs[2] retrieval, shipped shape, loss heads at bs 32, '
                                                            'configs[4] end to end, COCO-1k alignment grid)')
    ap.add_argument('--no-cpu-b256', action='store_true', help='skip the live B = 256 CPU baseline (~25 s, ~40 GB of host memory); the committed '
                                                               'one-off measurement is attached instead')
    ap.add_argument('--stub-step', action='store_true',
                    help='CPU self-test of the multi-rank protocol (tests/test_launch_cpu.py): gloo, a stub step, the same launcher, '
                         'argument parsing, barriers, MAX over ranks and JSON line -- no GPU, no kernels, no performance claim')
    ap.add_argument('--cpu-standin', action='store_true',
                    help='CPU self-test of the REAL multi-rank step loop (tests/test_launch_cpu.py): gloo, the fast sharded node of '
                         'aladin_amd/distributed.py with the HIP entry points replaced by tests/helpers/cpu_standins.py, at a reduced shape '
                         '(B=64/rank, D=16) -- exchange tuning, PhaseRecorder, watchdog and the JSON line as on the GPUs; no performance claim')
    ap.add_argument('--shared-gpu', action='store_true',
                    help='GPU self-test of the multi-rank step with the REAL kernels on a one-GPU box (tests/test_bench_gpu.py): every rank uses '
                         'cuda:0 and the collectives go through gloo (RCCL refuses two ranks per device), B=64/rank -- the real step loop, both '
                         'backward exchanges, PhaseRecorder, watchdog, JSON line; the ranks contend for one GPU, so no performance claim')
    ap.add_argument('--selftest-batch', type=int, default=64, help='per-rank batch of --shared-gpu (a multiple of 64; 256 = configs[3]\'s own size)')
    ap.add_argument('--eager', action='store_true', help='do not capture the step in a HIP graph')
    ap.add_argument('--graph', action='store_true', help='always replay the captured HIP graph (default: the faster of graph / eager in a short trial)')
    ap.add_argument('--force-sharded', action='store_true',
                    help='self-test: run the multi-GPU (sharded, RCCL) step even with one rank')
    ap.add_argument('--launch-timeout', type=float, default=1500.0,
                    help='seconds after which a self-launched multi-GPU run is ended (the whole process group) and reported as failed')
    ap.add_argument('--exchange', default='tune', choices=['tune', 'dense', 'sparse'],
                    help='multi-GPU backward exchange of d(image sets): dense reduce-scatter, pair-driven sparse '
                         'all-to-all, or time both during warm-up and keep the faster (default)')
    ap.add_argument('--bwd-partners', default='fp16', choices=['exact', 'fp16', 'fp16-own'],
                    help="backward row kernel's unit vectors (ops.set_backward_precision): 'fp16' (the library default: partner rows from "
                         "the forward's packed operands, gradients <= 4.3e-4 of their largest entry off the reference on every fixture, "
                         "gate 5e-4), 'fp16-own' (+ the row's own vector: 5.8e-4 on one D = 64 fixture) or 'exact' (raw fp32 rows: 3e-7)")
    return ap.parse_args()


def kernel_roofline(im, s, il, sl, groups=5, iters=100):
    """Average duration of the dominant kernel (align_scores16_tall_kernel) measured with HIP events on
    the stream it is launched on (torch's current stream), on the packed operands of the bench
    batch.  The side GEMM (33rd region of every image, 1/33 of the work, its own kernel) is run
    once and reused, so each timed launch contracts 32 regions x 47 words x 768 per pair.
    `groups` event-bracketed groups of `iters` launches; the median group is reported."""
    import torch
    from aladin_amd import ops
    dev = im.device
    geom = ops.align_geometry(B, B, R, T, D)
    xm, xe = ops.pack_images(im, ops.lengths_tensor(il, dev), geom)
    y = ops.pack_captions(s, ops.lengths_tensor(sl, dev), geom)
    out = torch.empty((B, B), dtype=torch.float32, device=dev)
    e_scr = torch.empty(geom.e_bytes, dtype=torch.uint8, device=dev)
    ops.scores_from_packed(xm, xe, y, geom, out, e_scr)                       # side GEMM + score kernel
    for _ in range(200):
        ops.scores_from_packed(xm, xe, y, geom, out, e_scr, reuse_side=True)  # score kernel alone
    torch.cuda.synchronize()
    ms_groups = []
    for _ in range(groups):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.scores_from_packed(xm, xe, y, geom, out, e_scr, reuse_side=True)
        e1.record()
        torch.cuda.synchronize()
        ms_groups.append(e0.elapsed_time(e1) / iters)
    ms = statistics.median(ms_groups)
    flops = B * B * 2 * 32 * (T - 3) * D
    achieved = flops / (ms * 1e-3) / 1e12
    traffic, src = pmc_traffic('align_scores16_tall_kernel')
    out = {'bound': 'mfma', 'traffic_source': src, 'kernel': 'align_scores16_tall_kernel<true,3,1> (256x384 tile, 8 waves of 128x96, v_mfma_f32_16x16x32_f16)',
           'achieved': round(achieved, 2), 'peak': PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_TFLOPS, 4),
           'traffic': traffic, 'mfma_busy_frac': PMC_EXTRA.get('mfma_busy_frac'), 'kernel_us': round(ms * 1e3, 2), 'kernel_us_min_max': [round(min(ms_groups) * 1e3, 2), round(max(ms_groups) * 1e3, 2)],
           'flops_per_launch': flops}
    # the same fraction for what surrounds the kernel: the forward chain pack + side GEMM + score kernel against ALL
    # of a pair's flops (33 regions), event-timed here; bench.py's main adds step_frac from the timed step itself
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import benchlib
        chain_ms = benchlib.forward_chain(im, s, il, sl)
        out['forward_chain_us'] = round(chain_ms * 1e3, 2)
        out['forward_chain_frac'] = round(B * B * FLOPS_PER_PAIR / (chain_ms * 1e-3) / 1e12 / PEAK_TFLOPS, 4)
    except Exception as exc:
        out['forward_chain_error'] = str(exc)
    return out


def csrc_hash():
    """sha256 over what the kernels are built from (tools/srchash.py: aladin_amd/csrc/*.hip, *.hpp, the Makefile with its flags,
    include/aladin_hip.h): what a committed PMC summary was collected on, and what the Makefile stamps next to every library it links."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import srchash
    return srchash.csrc_hash(ROOT)


def library_sources():
    """{'lib': path, 'built_from': stamp or None, 'tree': hash, 'current': bool}: is the loaded library the tree's sources?"""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import srchash
    from aladin_amd import _lib
    stamp, tree = srchash.library_stamp(_lib.LIB_PATH), srchash.csrc_hash(ROOT)
    return {'lib': os.path.relpath(_lib.LIB_PATH, ROOT), 'built_from': stamp, 'tree': tree, 'current': stamp == tree}


PMC_EXTRA = {}


def pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of the dominant kernel from the newest committed PMC summary
    (profiles/*_pmc.json, written by tools/materialise_profiles.py from separate rocprofv3 --pmc passes with
    the gfx950 FETCH_SIZE x2 correction).  bench.py cannot run the profiler on itself, so it reports the committed
    measurement and names its file -- but only while the kernel sources are the ones that were profiled: a summary stamped
    with another csrc hash (or with none) is refused, traffic is null and traffic_source says why."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json')))          # by name = by round tag (mtimes mean nothing after a checkout)
    now = csrc_hash()
    stale = None
    for f in reversed(files):
        try:
            d = json.load(open(f))
            ks = d['kernels']
        except Exception:
            continue
        for name, k in ks.items():
            if kernel_substr in name and 'hbm_bytes_corrected' in k:
                if d.get('csrc_hash') == now:
                    # matrix-pipe busy share from the same passes: SQ_VALU_MFMA_BUSY_CYCLES summed over the chip's 1024 SIMDs against
                    # GRBM_GUI_ACTIVE summed over the 8 XCDs -- separates pipe idleness from the clock the chip holds
                    if k.get('SQ_VALU_MFMA_BUSY_CYCLES') and k.get('GRBM_GUI_ACTIVE'):
                        PMC_EXTRA['mfma_busy_frac'] = round((k['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0) / (k['GRBM_GUI_ACTIVE'] / 8.0), 4)
                    return int(k['hbm_bytes_corrected']), os.path.relpath(f, ROOT)
                if stale is None:
                    stale = 'stale: %s was collected on kernel sources %s, these are %s (re-run tools/collect_pmc.sh)' % (
                        os.path.relpath(f, ROOT), d.get('csrc_hash', 'unstamped'), now)
    return None, stale


def cpu_baseline(live_b256=True):
    """BASELINE.json configs[0] verbatim: B=16, R=34, T=50, D=768 random tensors through the reference's
    AlignmentContrastiveLoss dataflow (oracle/faithful_torch.py restates alad/loss.py:79-159 op for op and is
    pinned to the reference's outputs by tests/test_oracle_golden.py), forward and forward+backward, on the
    host cores.  Thread counts {8, 32, all} are tried and the BEST is reported (B^2 tiny bmm's oversubscribe
    a big host).  `b256`: the same dataflow at the headline size, live (tools/benchlib.py: 1 warm-up + 2 timed steps,
    ~25 s, ~40 GB) -- or, when the host cannot hold it / --no-cpu-b256, the committed one-off measurement, labelled so."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import faithful_torch as FT
    from aladin_amd import synth
    nproc = os.cpu_count() or 1
    im, s, il, sl = synth.alignment_batch(16, R, T, D, seed=1234, ragged=False)
    a, b = torch.from_numpy(im), torch.from_numpy(s)
    sweep = {}
    for th in sorted({min(8, nproc), min(32, nproc), nproc}):
        torch.set_num_threads(th)
        res = {}
        for tag, bwd in (('fwd', False), ('fwd_bwd', True)):
            t0 = time.perf_counter()
            for _ in range(2):
                FT.alignment_triplet_step(a, b, il, sl, 0.2, True, backward=bwd)       # warm-ups
            slow = (time.perf_counter() - t0) / 2 > 0.5            # an oversubscribed thread count: seconds per step
            reps, t0 = 0, time.perf_counter()
            while reps < (2 if slow else 10) or (not slow and time.perf_counter() - t0 < 1.5 and reps < 200):
                FT.alignment_triplet_step(a, b, il, sl, 0.2, True, backward=bwd)
                reps += 1
            res[tag] = (time.perf_counter() - t0) / reps
        sweep[th] = res
    best = min(sweep, key=lambda k: sweep[k]['fwd_bwd'])
    out = {'value': round(256 / sweep[best]['fwd_bwd'], 1), 'unit': 'pairs/s', 'cores': best, 'kind': 'port',
           'fwd_only_value': round(256 / min(v['fwd'] for v in sweep.values()), 1), 'host_cores': nproc,
           'sample': 'BASELINE configs[0] verbatim: B=16 R=34 T=50 D=768 fp32 through the reference dataflow '
                     '(expand + bmm + masks, oracle/faithful_torch.py), fwd+bwd, >=10 reps after 2 warm-ups (2 reps where a step takes > 0.5 s), best of threads '
                     + str(sorted(sweep)) + ': %.1f ms/step at %d threads' % (sweep[best]['fwd_bwd'] * 1e3, best),
           'sweep_ms': {str(k): {t: round(v * 1e3, 2) for t, v in r.items()} for k, r in sweep.items()}}
    # B = 256 (the >= 10x target is stated against this size): live, at the best thread count found above
    committed = os.path.join(ROOT, 'profiles', 'r02_cpu_baseline_b256.json')
    if live_b256:
        try:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import benchlib
            out['b256'] = benchlib.cpu_baseline_b256(min(32, nproc))      # 32 threads: the best of the round-2 sweep {8, 32, all} at this size
        except Exception as exc:
            out['b256_live_error'] = '%s: %s' % (type(exc).__name__, exc)
    if 'b256' not in out and os.path.exists(committed):
        try:
            out['b256'] = dict(json.load(open(committed)), source='committed (profiles/r02_cpu_baseline_b256.json, a round-2 box)')
        except Exception:
            pass
    # SURVEY 8(d) judges the >= 10x target against the faithful dataflow AT B = 256: that is the headline `value` when it was
    # measured (live, or the committed one-off); the B = 16 sweep (BASELINE configs[0]) stays beside it as `b16`
    # ... but ONLY when it was measured live on THIS host: the committed file is a round-2 box's number -- it stays in the
    # line under `b256` with its `source`, and the live B = 16 sweep remains the headline
    b = out.get('b256')
    if b and b.get('pairs_per_s_fwd_bwd') and b.get('source') == 'live':
        out['b16'] = {'value': out['value'], 'cores': out['cores'], 'fwd_only_value': out.pop('fwd_only_value'), 'sample': out['sample'],
                      'sweep_ms': out.pop('sweep_ms')}
        out['value'] = b['pairs_per_s_fwd_bwd']
        out['cores'] = b.get('threads', out['cores'])
        out['sample'] = ('B=256 R=34 T=50 D=768 fp32 through the reference dataflow (expand + bmm + masks, oracle/faithful_torch.py), fwd+bwd, '
                         '%s: %.2f s/step at %d threads (1 warm-up + %d timed); BASELINE configs[0] (B=16) under b16'
                         % (b.get('source', '?'), b.get('s_per_step', float('nan')), out['cores'], b.get('steps_timed', 0)))
    return out


def _kernel_us(fn, name_substr, iters=5):
    """Average device time (us) of the kernels whose name contains `name_substr` over `iters` calls of fn, from torch's kernel trace."""
    import torch
    if any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return None                                          # bench.py itself is being profiled (tools/collect_pmc.sh): one tracer at a time
    try:
        from torch.profiler import ProfilerActivity, profile
        fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
        tot, cnt = 0.0, 0
        for e in prof.key_averages():
            if name_substr in e.key:
                tot += float(getattr(e, 'device_time_total', 0.0) or getattr(e, 'cuda_time_total', 0.0))
                cnt += int(e.count)
        return round(tot / cnt, 2) if cnt else None
    except Exception:
        return None


def eval_config3(dev):
    """Secondary field: BASELINE configs[2] -- 5000 img x 25000 cap x 768 matching-head retrieval, scores + ranks
    of both directions in one fused pass (aladin_retrieval_ranks; the 500 MB matrix is never written).
    The kernel screens with a third of the split product and continues only undecided pairs to the exact score, so its
    COST depends on where the ground truths sit among the scores (its result never does).  `ms` is timed on the input
    SURVEY 8(d) config 3 specifies -- synth.retrieval_embeddings(sigma=8): Recall@1 75.4 / 40.8 %, both directions inside
    the 40-80 % band (rounds 1-4 timed captions = image + 0.05 noise, every R@1 = 100: now `by_data[0]`).
    `by_data` adds that clean input, a near-clean one (sigma 6: 99 / 81 %) and a hard one (sigma 12: 19 / 9 %), each with
    the 256 x 384 tiles continued in place, the pairs listed and the listed pairs whose chains were continued;
    `all_exact_ms` is the three-product path on every tile (what round 3 ran), on the `ms` input."""
    import torch
    from aladin_amd import ops, synth

    def ev_ms(fn, iters=10, warm=3, preroll_s=0.3):
        # clock-settling pre-roll, as for the headline step: each data set is generated on the HOST for seconds while the GPU idles and
        # drops its clock; three warm-up calls (~1 ms) do not bring it back (a clean-input run once read 0.39 ms instead of 0.28)
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < preroll_s:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def measure(name, a, b):
        r_i2t, _, r_t2i, _, sd = ops.retrieval_ranks(a, b, return_stats=True)
        return {'data': name, 'R@1_i2t': round(float((r_i2t == 0).float().mean()) * 100, 1),
                'R@1_t2i': round(float((r_t2i == 0).float().mean()) * 100, 1), 'ms': round(ev_ms(lambda: ops.retrieval_ranks(a, b)), 4),
                'exact_tiles': sd['exact_tiles'], 'listed_pairs': sd['listed_pairs'], 'rescored_pairs': sd['rescored_pairs']}

    def synth_pair(sigma):
        i_np, c_np = synth.retrieval_embeddings(5000, D, seed=303, sigma=sigma)
        return torch.from_numpy(i_np[0::5]).to(dev), torch.from_numpy(c_np).to(dev)

    a8, b8 = synth_pair(8.0)
    head = measure('synth.retrieval_embeddings(sigma=8)', a8, b8)
    ms_exact = ev_ms(lambda: ops.retrieval_ranks(a8, b8, exact=True))
    del a8, b8
    g = torch.Generator(device='cpu').manual_seed(7)
    img = torch.nn.functional.normalize(torch.randn(5000, D, generator=g), dim=1).to(dev)
    cap = torch.nn.functional.normalize(img.repeat_interleave(5, 0) + 0.05 * torch.randn(25000, D, generator=g).to(dev), dim=1)
    by_data = [measure('caption = image + 0.05 noise (the timing input of rounds 1-4)', img, cap)]
    del img, cap
    for sigma in (6.0, 12.0):
        a, b = synth_pair(sigma)
        by_data.append(measure('synth.retrieval_embeddings(sigma=%g)' % sigma, a, b))
    ms = head['ms']
    # roofline-style fields for this path: the whole call's algorithmic flops (2 * 5000 * 25000 * 768: ONE product
    # per pair, what the reference's mm contracts) over the call's time against the 16-bit MFMA peak, and the screening GEMM kernel's
    # own duration (torch's kernel trace over a few calls; None when a profiler is already attached to this process)
    frac = round(2 * 5000 * 25000 * D / (ms * 1e-3) / 1e12 / PEAK_TFLOPS, 4)
    return {'workload': 'configs[2]: 5000x25000x768 matching-head retrieval, fused scores + i2t/t2i ranks', 'ms': ms, 'frac': frac,
            'screen_kernel_us': None, 'screen_kernel_frac': None,            # filled by eval_config3_screen_kernel(), LAST (see there)
            'data': head['data'], 'R@1_i2t': head['R@1_i2t'], 'R@1_t2i': head['R@1_t2i'],
            'all_exact_ms': round(ms_exact, 4), 'tiles': 20 * 66, 'exact_tiles': head['exact_tiles'], 'listed_pairs': head['listed_pairs'],
            'rescored_pairs': head['rescored_pairs'],
            'data_note': 'ms: the SURVEY 8(d) input (Recall@1 of both directions in 40-80 %); the screened kernel\'s cost depends on the data, see '
                         'by_data; ranks equal the two-step split path int for int at this size on the sigma 6 / 8 / 12 inputs '
                         '(tests/test_gpu_parity.py::test_config3_full_size_retrieval_ranks) and on adversarial inputs (test_fused_retrieval_*)',
            'by_data': by_data,
            'pairs_per_s': round(5000 * 25000 / (ms * 1e-3), 1), 'tflops_algorithmic': round(2 * 5000 * 25000 * D / (ms * 1e-3) / 1e12, 1)}


def eval_config3_screen_kernel(dev):
    """The screening GEMM kernel's own duration on the eval_config3.ms input, from torch's kernel trace.  Run AFTER every other
    timing of this process: once torch's profiler has been attached, every later kernel launch of the process costs the host more
    (measured: the bs-32 loss heads went 0.10 -> 0.52 ms per step behind it)."""
    import torch
    from aladin_amd import ops, synth
    i_np, c_np = synth.retrieval_embeddings(5000, D, seed=303, sigma=8.0)
    a8, b8 = torch.from_numpy(i_np[0::5]).to(dev), torch.from_numpy(c_np).to(dev)
    us = _kernel_us(lambda: ops.retrieval_ranks(a8, b8), 'sim_screen_kernel')
    return {'screen_kernel_us': us, 'screen_kernel_frac': None if not us else round(2 * 5000 * 25000 * D / (us * 1e-6) / 1e12 / PEAK_TFLOPS, 4)}


def shipped_shape_step(dev):
    """Secondary field: the same triplet step at the shipped DATA shape -- 50 regions + 35 tokens (R = 51, T = 38: two
    region tiles per image; configs/*.yaml's dataset section) -- eager launches, B = 256, full lengths."""
    import torch
    from aladin_amd import synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    im, s, il, sl = synth.alignment_batch(B, 51, 38, D, seed=1234, ragged=False)
    a = torch.from_numpy(im).to(dev).requires_grad_(True)
    b = torch.from_numpy(s).to(dev).requires_grad_(True)
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=True, aggregation='MrSw')

    seed = torch.ones((), dtype=torch.float32, device=dev)     # as the headline step: no ones_like fill per backward()

    def step():
        a.grad = None
        b.grad = None
        crit(a, b, il, sl).backward(gradient=seed)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    # the same step replayed as a HIP graph -- the launch mode the headline step is timed in when the short trial favours it (the eager
    # number above stays the field's `ms_per_step`: it is what rounds 3-5 reported)
    graph_ms = None
    try:
        a.grad = None
        b.grad = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            crit(a, b, il, sl).backward(gradient=seed)
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(200):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        graph_ms = round(e0.elapsed_time(e1) / 200, 4)
        del g
    except Exception:
        pass
    # the shape's score kernel alone (align_scores16_r48, 192 x 320 tile; the side GEMM's result reused), event-timed like the
    # headline's: ALGORITHMIC flops (50 regions x 35 words per pair, not the 48 + 2 side rows x 40 columns the tiles hold)
    from aladin_amd import ops
    geom = ops.align_geometry(B, B, 51, 38, D)
    packed = ops.pack_sets(a.detach(), b.detach(), ops.lengths_tensor(il, dev), ops.lengths_tensor(sl, dev), geom, norms=False)
    out = torch.empty((B, B), dtype=torch.float32, device=dev)
    e_scr = torch.empty(max(int(geom.e_bytes), 16), dtype=torch.uint8, device=dev)
    ops.scores_from_packed(packed[1], packed[2], packed[3], geom, out, e_scr)
    for _ in range(50):
        ops.scores_from_packed(packed[1], packed[2], packed[3], geom, out, e_scr, reuse_side=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200):
        ops.scores_from_packed(packed[1], packed[2], packed[3], geom, out, e_scr, reuse_side=True)
    e1.record()
    torch.cuda.synchronize()
    k_us = e0.elapsed_time(e1) / 200 * 1e3
    k_flops = 2 * 50 * 35 * D * B * B
    return {'workload': 'triplet loss forward+backward at B=256, R=51, T=38, D=768 (50 regions + 35 tokens), eager launches',
            'ms_per_step': round(ms, 4), 'graph_ms_per_step': graph_ms, 'pairs_per_s': round(B * B / (ms * 1e-3), 1),
            'flops_per_pair': 2 * 50 * 35 * D, 'tflops_algorithmic_fwd_equiv': round(2 * 50 * 35 * D * B * B / (ms * 1e-3) / 1e12, 1),
            'score_kernel': 'align_scores16_r48_kernel (48-row region class + 2 side rows, 40-word caption class)',
            'score_kernel_us': round(k_us, 2), 'score_kernel_frac': round(k_flops / (k_us * 1e-6) / 1e12 / PEAK_TFLOPS, 4)}


def stub_main(args, world, rank):
    """--stub-step: the multi-rank PROTOCOL on CPU (gloo): what the driver's `torch.distributed.run ... bench.py --gpus N` relies
    on besides the kernels -- rendezvous on 127.0.0.1, one rank per slot, barriers around the timed region, MAX over ranks,
    exactly one JSON line from rank 0 carrying config.collectives.  The step is a stand-in (a rank-sized all-gather + all-reduce
    of small CPU tensors), the numbers mean nothing and the line says so ("data": "stub")."""
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group('gloo')
    x = torch.full((1024,), float(rank + 1))
    parts = [torch.empty(1024) for _ in range(world)]

    def step():
        dist.all_gather(parts, x)                          # stands for the all-gather of the packed image operands
        t = torch.stack(parts).sum(0)
        dist.all_reduce(t)                                 # stands for the reduce-scatter of d(image sets)
        return t

    def agree_max(v):
        tt = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())
    for _ in range(min(args.warmup, 5)):
        step()
    steps = min(args.steps, 20)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    dist.barrier()
    ms = agree_max(time.perf_counter() - t0) / steps * 1e3
    want = world * sum(range(1, world + 1))
    ok = bool((out == want).all())
    if rank == 0:
        print(json.dumps({'metric': 'alignment image-text pairs/sec (BxB scores, triplet loss fwd+bwd) at B=256/GPU,R=34,T=50,D=768',
                          'value': 0.0, 'unit': 'pairs/s', 'n_gpus': world, 'steps': steps, 'warmup': min(args.warmup, 5), 'ms_per_step': round(ms, 4),
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'stub',
                          'config': {'workload': 'STUB STEP: protocol self-test on CPU, no kernels, no performance claim',
                                     'collectives': {'backend': dist.get_backend(), 'ranks': dist.get_world_size(),
                                                     'launcher': 'self (aladin_amd.launch)' if os.environ.get('ALADIN_SELF_LAUNCHED') else 'external'},
                                     'stub_result_ok': ok}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


class StallWatchdog:
    """A rank that makes no progress for `limit_s` seconds (a collective that never completes, a peer that died) ends ITSELF with
    exit code 86 -- a plain process exit from a daemon thread, no re-exec, nothing touching the GPU -- so that the launcher sees a
    failed rank, ends the others and bench.py's parent prints its error line instead of hanging until the driver's limit.
    tick() is called once per step and around every phase that may legitimately take long."""

    def __init__(self, limit_s, rank):
        import threading
        self.limit, self.rank, self.last = float(limit_s), rank, time.monotonic()
        self._stop = False
        threading.Thread(target=self._run, daemon=True).start()

    def tick(self):
        self.last = time.monotonic()

    def stop(self):
        self._stop = True

    def _run(self):
        while not self._stop:
            time.sleep(2.0)
            idle = time.monotonic() - self.last
            if idle > self.limit:
                sys.stderr.write('bench: rank %d made no progress for %.0f s (stalled collective?); exiting 86\n' % (self.rank, idle))
                sys.stderr.flush()
                os._exit(86)


WATCHDOG = [None]


def tick():
    if WATCHDOG[0] is not None:
        WATCHDOG[0].tick()


def main():
    global B, D
    args = parse()
    # before anything initialises the HIP runtime: the host driver only supports dmabuf IPC
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # `python bench.py --gpus N` outside torch.distributed.run: start the N ranks as CHILDREN (before torch or the
    # HIP runtime exist in this process), relay their output with the JSON line last, exit with their return code
    from aladin_amd import launch
    if launch.needs_self_launch(args.gpus):
        rc, line = launch.run_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, timeout=args.launch_timeout)
        if rc != 0 and line is None:
            # a failed or stalled multi-GPU run still leaves ONE parseable line, with no value in it: the ranks'
            # own watchdogs (StallWatchdog below, the process group's collective timeout) exit non-zero after 60 - 120 s without
            # progress; rc 124 = the launcher's own limit ended the process group
            print(json.dumps({'metric': 'alignment image-text pairs/sec (BxB scores) at B=%d,R=%d,T=%d,D=%d' % (B, R, T, D), 'value': None,
                              'unit': 'pairs/s', 'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None,
                              'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
                              'error': 'ranks failed or stalled (launcher rc %d%s); see stderr' % (rc, ': time limit' if rc == 124 else ''),
                              'config': {'workload': 'configs[3]: per-GPU B=%d, caption-block sharding' % B}}), flush=True)
        raise SystemExit(rc)
    # HIP-runtime setting for graph replay: with "graph packet capture" on (this ROCm's default) the replay of the (then) 7-kernel
    # step costs ~4 us more than with it off (0.2357 vs 0.2316 ms, alternated three times on one box,
    # profiles/r02_ab_experiments.txt).  Read once when the runtime starts; an exported value wins.
    os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('bench.py --gpus %d runs as rank %d of a world of %d: --gpus must equal the number of ranks started'
                         % (args.gpus, rank, world))
    if args.stub_step:
        return stub_main(args, world, rank)
    standin, shared = args.cpu_standin, args.shared_gpu and not args.cpu_standin
    selftest = standin or shared                            # reduced protocol, labelled line, value 0: never a measurement
    if standin:
        # the real step loop below on CPU tensors under gloo, the HIP entry points replaced by torch restatements that keep the packed
        # layout (tests/helpers/cpu_standins.py; nothing under oracle/): a protocol and bookkeeping self-test of the multi-GPU path at a
        # reduced shape, labelled as such in the line -- never a measurement
        B, D = 64, 16
        torch.set_num_threads(1)
        dev = torch.device('cpu')
        sync = lambda: None
        args.preroll_s, args.steps, args.warmup, args.repeats = 0.0, min(args.steps, 3), min(args.warmup, 1), min(args.repeats, 2)
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X (no GPU visible); there is no CPU fallback')
        if shared:
            # all ranks on cuda:0 (separate processes, separate HIP contexts), gloo for the exchange: the real kernels inside the real
            # node at W > 1 on the one-GPU boxes this build has ever had
            B = int(args.selftest_batch)
            local_rank = 0
            args.preroll_s, args.steps, args.warmup, args.repeats = 0.0, min(args.steps, 5), min(args.warmup, 2), min(args.repeats, 2)
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
        sync = torch.cuda.synchronize
    sharded = world > 1 or args.force_sharded or selftest
    dist = None
    if sharded:
        import torch.distributed as dist
        import datetime
        limit = datetime.timedelta(seconds=120)             # a collective that does not complete in two minutes aborts the rank (RCCL watchdog)
        if selftest:
            if world == 1:
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                os.environ.setdefault('MASTER_PORT', '29672')
                dist.init_process_group('gloo', rank=0, world_size=1, timeout=limit)
            else:
                dist.init_process_group('gloo', timeout=limit)
        elif world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29671')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev, timeout=limit)
        else:
            dist.init_process_group('nccl', device_id=dev, timeout=limit)
        WATCHDOG[0] = StallWatchdog(60.0, rank)

    from aladin_amd import synth, ops
    from aladin_amd.loss import AlignmentContrastiveLoss
    from aladin_amd import distributed as AD
    ops.set_backward_precision(args.bwd_partners)
    if standin:
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
        import cpu_standins
        cpu_standins.install()

    im_np, s_np, il, sl = synth.alignment_batch(B, R, T, D, seed=1234 + 17 * rank, ragged=False)
    im = torch.from_numpy(im_np).to(dev).requires_grad_(True)
    s = torch.from_numpy(s_np).to(dev).requires_grad_(True)
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=True, aggregation='MrSw')

    exchange = ['dense' if args.exchange == 'tune' else args.exchange]
    tuned = {}
    # dloss/dloss = 1, allocated once: `loss.backward()` with no argument makes autograd fill a fresh ones_like(loss)
    # every step (one more 4.5 us launch in a 0.23 ms step); the gradients are the same
    seed = torch.ones((), dtype=torch.float32, device=dev)

    def step():
        tick()                                             # StallWatchdog: a rank stuck in a collective for 60 s ends itself
        im.grad = None
        s.grad = None
        if sharded:
            loss, _ = AD.sharded_alignment_loss_fast(im, s, il, sl, 0.2, True, exchange=exchange[0])
        else:
            loss = crit(im, s, il, sl)
        loss.backward(gradient=seed)
        return loss

    def fence():
        if sharded:
            dist.barrier()
        sync()

    def agree_max(x):
        if not sharded:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # The step is 6 short launches; eager Python issue time (~0.22 ms) is close to the GPU time, so
    # the step is also captured once into a HIP graph (the C ABI neither allocates nor synchronises);
    # a trial below picks replay or eager issue.  Same kernels, same work; --eager / --graph force one.
    # Multi-GPU stays eager (collectives).
    launch = 'eager'
    run = step
    if not sharded and not args.eager:
        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            im.grad = None
            s.grad = None
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_loss = crit(im, s, il, sl)
                static_loss.backward(gradient=seed)

            def run():
                graph.replay()
                return static_loss
            launch = 'hipgraph'
        except Exception as exc:            # capture unsupported: fall back to eager, say so
            print('bench: graph capture failed (%s), running eager' % exc, file=sys.stderr)
            run = step
    if sharded and args.exchange == 'tune':
        # Both exchanges give the same gradients (tests/); which is faster depends on the world size and
        # the fabric.  Time a few untimed steps of each, agree on the MAX over ranks, keep the winner.
        n_try, n_tune = (1, 2) if selftest else (3, 8)
        for mode in ('dense', 'sparse'):
            exchange[0] = mode
            failed, elapsed = 0.0, float('inf')
            try:
                for _ in range(n_try):
                    step()
                sync()
            except Exception as exc:
                print('bench: exchange %r failed on rank %d (%s); not used' % (mode, rank, exc), file=sys.stderr)
                failed = 1.0
            # every rank learns whether ANY rank failed before the next collective is entered: a rank that
            # failed mid-step and one that did not would otherwise meet in different collectives and hang
            if agree_max(failed) == 0.0:
                fence()
                t0 = time.perf_counter()
                for _ in range(n_tune):
                    step()
                fence()
                elapsed = agree_max(time.perf_counter() - t0)
            tuned[mode] = elapsed / n_tune * 1e3
        if all(v == float('inf') for v in tuned.values()):
            raise SystemExit('bench: both backward exchanges failed; see stderr')
        exchange[0] = min(tuned, key=tuned.get)

    # Graph replay or eager issue?  Same kernels either way.  Replay has no host cost but the runtime spends a few us more
    # between the nodes of a graph than between kernels queued on a stream; eager issue is faster as long as the host
    # (0.216 ms of Python + launches per step, measured) stays ahead of the GPU.  A short interleaved trial decides.
    launch_trial = None
    if launch == 'hipgraph' and not args.graph:
        def trial(fn, n=300):
            fence()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            fence()
            return (time.perf_counter() - t0) / n * 1e3
        for fn in (run, step):
            trial(fn, 600)                                   # settle the clock before comparing
        launch_trial = {'hipgraph': min(trial(run), trial(run)), 'eager': min(trial(step), trial(step))}
        for _ in range(2):
            launch_trial['hipgraph'] = min(launch_trial['hipgraph'], trial(run))
            launch_trial['eager'] = min(launch_trial['eager'], trial(step))
        if launch_trial['eager'] < launch_trial['hipgraph']:
            launch, run = 'eager', step
        launch_trial = {k: round(v, 4) for k, v in launch_trial.items()}

    # 1. clock-settling pre-roll (independent of --warmup)
    t_pre, n_pre = time.perf_counter(), 0
    while time.perf_counter() - t_pre < args.preroll_s:
        for _ in range(50):
            run()
        sync()
        n_pre += 50
    # 2. warm-up
    for _ in range(args.warmup):
        run()
    # 3. the K-step region, `repeats` times
    region_ms = []
    loss = None
    for _ in range(max(1, args.repeats)):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = run()
        fence()
        region_ms.append(agree_max(time.perf_counter() - t0) / args.steps * 1e3)
    ms = statistics.median(region_ms)
    pairs = (B * world) ** 2
    value = pairs / (ms * 1e-3)

    # The same step with the EXACT backward row step (raw fp32 rows instead of fp16 partner rows), timed the same way (one region of
    # K steps under the launch mode chosen above), so that the precision trade behind the headline is visible in the line
    bwd_exact_ms = None
    if not sharded and args.bwd_partners != 'exact':
        old_mode = ops.set_backward_precision('exact')
        try:
            run_x = step
            if launch == 'hipgraph':
                for _ in range(3):
                    step()
                sync()
                im.grad = None
                s.grad = None
                graph_x = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_x):
                    loss_x = crit(im, s, il, sl)
                    loss_x.backward(gradient=seed)
                run_x = graph_x.replay
            for _ in range(max(args.warmup, 50)):
                run_x()
            reg = []
            for _ in range(max(1, min(args.repeats, 3))):
                fence()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    run_x()
                fence()
                reg.append((time.perf_counter() - t0) / args.steps * 1e3)
            bwd_exact_ms = round(statistics.median(reg), 4)
        except Exception as exc:
            bwd_exact_ms = 'error: %s' % exc
        finally:
            ops.set_backward_precision(old_mode)

    phases = None
    if sharded:
        rec = AD.PhaseRecorder()
        AD.set_phase_recorder(rec)
        for _ in range(2 if selftest else 10):
            rec.begin()
            step()
            rec.mark('autograd_tail')
            rec.end()
        AD.set_phase_recorder(None)
        phases = rec.summary()
        # RCCL prints a version banner through C stdio when the communicator comes up; on a pipe it would
        # surface at exit, AFTER the JSON line.  Push it out now so the JSON line is the last line of stdout.
        # Every rank does it, and rank 0 prints only after all of them have (barrier).
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        dist.barrier()
        if WATCHDOG[0] is not None:
            WATCHDOG[0].stop()                             # the measured part is over; what follows is rank 0's own (bounded) work
    # the row step that RAN (the multi-GPU line must not be labelled with a mode it does not run): since round 6 the sharded
    # step carries the packed rows' inverse norms (dense exchange: in its all-gather segment; pair-driven exchange: by packing the
    # compact problem in the backward), so ops.set_backward_precision applies to it as to the single-GPU node
    bwd_mode = args.bwd_partners
    if rank == 0:
        roof = kernel_roofline(im.detach(), s.detach(), il, sl) if not selftest else {'bound': 'mfma', 'note': 'not measured: self-test mode'}
        cfg = {'workload': ('CPU STAND-INS (tests/helpers/cpu_standins.py) under gloo at B=%d/rank, D=%d: the multi-rank step loop, not a measurement; ' % (B, D) if standin else '') +
                           ('SHARED-GPU SELF-TEST: %d ranks on ONE GPU (real kernels, gloo exchange) at B=%d/rank: the multi-rank step loop, not a measurement; ' % (world, B) if shared else '') +
                           'configs[1]: alignment-head triplet loss forward+backward, B=256 synthetic '
                           'features per GPU (R=34,T=50,D=768, full lengths)' +
                           ('' if world == 1 else '; configs[3]: global %dx%d matrix, images all-gathered '
                            'over %s, caption-block sharding' % (B * world, B * world, 'gloo (self-test)' if selftest else 'RCCL')),
               'global_pairs_per_step': pairs, 'loss': float(loss.detach()), 'launch': launch, 'bwd_partners': bwd_mode, 'bwd_exact_ms_per_step': bwd_exact_ms, 'backward_seed': 'preallocated ones',
               'launch_trial_ms': launch_trial, 'hip_env': {'DEBUG_CLR_GRAPH_PACKET_CAPTURE': os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE')},
               'timing': {'preroll_s': args.preroll_s, 'preroll_steps': n_pre, 'repeats': len(region_ms), 'statistic': 'median',
                          'ms_per_step_min': round(min(region_ms), 4), 'ms_per_step_max': round(max(region_ms), 4)},
               'step_tflops_algorithmic': round(value * FLOPS_PER_PAIR / 1e12, 2) if not selftest else None}
        if sharded:
            cfg['collectives'] = {'backend': dist.get_backend(), 'ranks': dist.get_world_size(),
                                  'launcher': 'self (aladin_amd.launch)' if os.environ.get('ALADIN_SELF_LAUNCHED') else 'external'}
            cfg['bwd_exchange'] = exchange[0]
            cfg['bwd_exchange_tuning_ms'] = {k: (round(v, 4) if v != float('inf') else None) for k, v in tuned.items()}
            cfg['phases_ms'] = phases          # rank 0's device timeline of one step (10-step mean), see PhaseRecorder
        if standin:
            cfg['standin_calls'] = dict(cpu_standins.CALLS)
        if world == 1 and not args.no_eval and not selftest:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import benchlib
            for key, fn in (('eval_config3', eval_config3), ('shipped_shape', shipped_shape_step), ('loss_heads_bs32', benchlib.loss_heads_bs32),
                            ('e2e_config4', benchlib.e2e_config4), ('alignment_retrieval_coco1k', benchlib.alignment_retrieval_coco1k)):
                try:
                    cfg[key] = fn(dev)
                except Exception as exc:
                    cfg[key] = {'error': '%s: %s' % (type(exc).__name__, exc)}
            if 'error' not in cfg['eval_config3']:
                try:
                    cfg['eval_config3'].update(eval_config3_screen_kernel(dev))          # last: attaches torch's profiler to the process
                except Exception:
                    pass
        # which binary was timed: the library's link-time source stamp against the tree (tools/srchash.py); `current` false = a
        # library left over from an experiment, or built by hand -- the numbers of this line are then not the committed code's
        cfg['library'] = library_sources()
        # the three fractions of the 16-bit MFMA peak side by side: the score kernel alone (`frac`), the forward chain
        # pack + side GEMM + score kernel (`forward_chain_frac`), the whole timed step forward + backward (`step_frac`)
        if not selftest:
            roof['step_frac'] = round(value * FLOPS_PER_PAIR / world / 1e12 / PEAK_TFLOPS, 4)
        out = {
            'metric': 'alignment image-text pairs/sec (BxB scores, triplet loss fwd+bwd) at B=256/GPU,R=34,T=50,D=768',
            'value': round(value, 1) if not selftest else 0.0, 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16', 'data': 'cpu-standin' if standin else ('shared-gpu-selftest' if shared else 'synthetic'), 'config': cfg, 'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline and not selftest:
            out['cpu_baseline'] = cpu_baseline(live_b256=not args.no_cpu_b256)
        print(json.dumps(out), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
