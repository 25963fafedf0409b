#!/usr/bin/env python3
"""bench.py -- alignment image-text pairs/sec on MI355X (BASELINE.json metric).

A step = ONE forward + backward of the alignment-head triplet loss
(AlignmentContrastiveLoss(margin=.2,'dot',max_violation=True,'MrSw'), reference
alad/configs/alad-alignment-triplet.yaml) on synthetic features with B=256 per GPU, R=34, T=50,
D=768 (BASELINE.json configs[1]; configs[3] when --gpus > 1: images all-gathered over RCCL, each
rank scores the (N*256 x 256) caption block, hinge on the global matrix).
Inputs are resident in HBM before the timed region.  value = (N*256)^2 pairs / step time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md section "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

B, R, T, D = 256, 34, 50, 768
FLOPS_PER_PAIR = 2 * (R - 1) * (T - 3) * D           # 2,382,336
PEAK_TFLOPS = 2500.0                                 # dense 16-bit MFMA, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='do not capture the step in a HIP graph')
    ap.add_argument('--force-sharded', action='store_true',
                    help='self-test: run the multi-GPU (sharded, RCCL) step even with one rank')
    ap.add_argument('--exchange', default='tune', choices=['tune', 'dense', 'sparse'],
                    help='multi-GPU backward exchange of d(image sets): dense reduce-scatter, pair-driven sparse '
                         'all-to-all, or time both during warm-up and keep the faster (default)')
    ap.add_argument('--cpu-batch', type=int, default=96, help='batch of the bounded CPU-baseline sample')
    return ap.parse_args()


def kernel_roofline(im, s, il, sl, iters=50):
    """Average duration of the dominant kernel (align_scores_kernel) measured with HIP events on
    the stream it is launched on (torch's current stream), on the packed operands of the bench
    batch.  The side GEMM (33rd region of every image, 1/33 of the work, its own kernel) is run
    once and reused, so each timed launch contracts 32 regions x 47 words x 768 per pair."""
    from aladin_amd import ops
    dev = im.device
    geom = ops.align_geometry(B, B, R, T, D)
    xm, xe = ops.pack_images(im, ops.lengths_tensor(il, dev), geom)
    y = ops.pack_captions(s, ops.lengths_tensor(sl, dev), geom)
    out = torch.empty((B, B), dtype=torch.float32, device=dev)
    e_scr = torch.empty(geom.e_bytes, dtype=torch.uint8, device=dev)
    ops.scores_from_packed(xm, xe, y, geom, out, e_scr)                       # side GEMM + score kernel
    for _ in range(5):
        ops.scores_from_packed(xm, xe, y, geom, out, e_scr, reuse_side=True)  # score kernel alone
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.scores_from_packed(xm, xe, y, geom, out, e_scr, reuse_side=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = B * B * 2 * 32 * (T - 3) * D
    achieved = flops / (ms * 1e-3) / 1e12
    traffic, src = pmc_traffic('align_scores16_kernel')
    return {'bound': 'mfma', 'traffic_source': src, 'kernel': 'align_scores16_kernel<true> (256x384 tile, v_mfma_f32_16x16x32_f16)', 'achieved': round(achieved, 2), 'peak': PEAK_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_TFLOPS, 4), 'traffic': traffic,
            'kernel_us': round(ms * 1e3, 2), 'flops_per_launch': flops}


def pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of the dominant kernel from the newest committed PMC summary
    (profiles/*_pmc.json, written by tools/collect_pmc.sh from separate rocprofv3 --pmc passes with
    the gfx950 FETCH_SIZE x2 correction).  bench.py cannot run the profiler on itself, so it
    reports the committed measurement and names its file; None if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json')), key=os.path.getmtime)
    for f in reversed(files):
        try:
            ks = json.load(open(f))['kernels']
        except Exception:
            continue
        for name, d in ks.items():
            if kernel_substr in name and 'hbm_bytes_corrected' in d:
                return int(d['hbm_bytes_corrected']), os.path.relpath(f, ROOT)
    return None, None


def cpu_baseline(batch):
    """The reference's dataflow (oracle/faithful_torch.py) on the host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import faithful_torch as FT
    from aladin_amd import synth
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    im, s, il, sl = synth.alignment_batch(batch, R, T, D, seed=1234, ragged=False)
    a, b = torch.from_numpy(im), torch.from_numpy(s)
    FT.alignment_triplet_step(a, b, il, sl, 0.2, True)            # warm-up
    reps, t0 = 0, time.time()
    while reps < 3 or (time.time() - t0 < 10.0 and reps < 20):
        FT.alignment_triplet_step(a, b, il, sl, 0.2, True)
        reps += 1
    dt = (time.time() - t0) / reps
    return {'value': round(batch * batch / dt, 1), 'unit': 'pairs/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'faithful expand+bmm+mask restatement of alad/loss.py:79-159 (oracle/faithful_torch.py), '
                      'fwd+bwd, B=%d R=34 T=50 D=768 fp32, %d reps, %.3f s/step' % (batch, reps, dt)}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d'
                             % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no GPU visible); there is no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29671')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group('nccl', device_id=dev)

    from aladin_amd import synth
    from aladin_amd.loss import AlignmentContrastiveLoss
    from aladin_amd.distributed import sharded_alignment_loss_fast

    im_np, s_np, il, sl = synth.alignment_batch(B, R, T, D, seed=1234 + 17 * rank, ragged=False)
    im = torch.from_numpy(im_np).to(dev).requires_grad_(True)
    s = torch.from_numpy(s_np).to(dev).requires_grad_(True)
    crit = AlignmentContrastiveLoss(margin=0.2, measure='dot', max_violation=True, aggregation='MrSw')

    exchange = ['dense' if args.exchange == 'tune' else args.exchange]
    tuned = {}

    def step():
        im.grad = None
        s.grad = None
        if sharded:
            loss, _ = sharded_alignment_loss_fast(im, s, il, sl, 0.2, True, exchange=exchange[0])
        else:
            loss = crit(im, s, il, sl)
        loss.backward()
        return loss

    def fence():
        if sharded:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # The step is ~10 short launches; eager Python issue time (~0.24 ms) is close to the GPU time, so
    # the step is captured once into a HIP graph (the C ABI neither allocates nor synchronises) and
    # replayed.  Same kernels, same work; --eager keeps the plain path.  Multi-GPU stays eager
    # (collectives).
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
                static_loss.backward()

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
        import torch.distributed as dist
        for mode in ('dense', 'sparse'):
            exchange[0] = mode
            try:
                for _ in range(3):
                    step()
                fence()
                t0 = time.perf_counter()
                for _ in range(8):
                    step()
                fence()
                elapsed = time.perf_counter() - t0
            except Exception as exc:                  # same code on every rank: they fail (or not) together
                print('bench: exchange %r failed on rank %d (%s); not used' % (mode, rank, exc), file=sys.stderr)
                elapsed = float('inf')
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tuned[mode] = float(tt.item()) / 8 * 1e3
        exchange[0] = min(tuned, key=tuned.get)
    for _ in range(args.warmup):
        run()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = run()
    fence()
    dt = time.perf_counter() - t0
    if sharded:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    pairs = (B * world) ** 2
    value = pairs / (ms * 1e-3)

    if sharded:
        # RCCL prints a version banner through C stdio when the communicator comes up; on a pipe it would
        # surface at exit, AFTER the JSON line.  Push it out now so the JSON line is the last line of stdout.
        # Every rank does it, and rank 0 prints only after all of them have (barrier).
        import ctypes
        import torch.distributed as dist
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        dist.barrier()
    if rank == 0:
        roof = kernel_roofline(im.detach(), s.detach(), il, sl)
        out = {
            'metric': 'alignment image-text pairs/sec (BxB scores, triplet loss fwd+bwd) at B=256/GPU,R=34,T=50,D=768',
            'value': round(value, 1), 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: alignment-head triplet loss forward+backward, B=256 synthetic '
                                   'features per GPU (R=34,T=50,D=768, full lengths)' +
                                   ('' if world == 1 else '; configs[3]: global %dx%d matrix, images all-gathered '
                                    'over RCCL, caption-block sharding' % (B * world, B * world)),
                       'global_pairs_per_step': pairs, 'loss': float(loss.detach()), 'launch': launch,
                       **({} if not sharded else {'bwd_exchange': exchange[0],
                                                  'bwd_exchange_tuning_ms': {k: (round(v, 4) if v != float('inf') else None)
                                                                             for k, v in tuned.items()}}),
                       'step_tflops_algorithmic': round(value * FLOPS_PER_PAIR / 1e12, 2)},
            'roofline': roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.cpu_batch)
        print(json.dumps(out), flush=True)
    if sharded:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
