#!/usr/bin/env python3
"""In-kernel clock of the score kernel's main loop (diagnostic build, ALADIN_ALIGN_SPREAD=6):
runs ~2 s of back-to-back launches on the bench batch, then reads the per-workgroup
s_memtime / s_memrealtime deltas.  clock = d(memtime) / d(memrealtime) * 100 MHz."""
import ctypes as C
import os
import sys
import time

os.environ['ALADIN_ALIGN_SPREAD'] = os.environ.get('PROBE_SCHED', '26')      # 26: 16x16x32 kernel, 6: 32x32x16 kernel
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# probes and knobs exist only in the diagnostic build (`make -C aladin_amd/csrc diag`), never in the product library
os.environ.setdefault('ALADIN_LIB', os.path.join(ROOT, 'aladin_amd', 'lib', 'libaladin_hip_diag.so'))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from aladin_amd import _lib, ops, synth

B, R, T, D = 256, 34, 50, 768
dev = torch.device('cuda:0')
im_np, s_np, il, sl = synth.alignment_batch(B, R, T, D, seed=1234)
im, s = torch.from_numpy(im_np).to(dev), torch.from_numpy(s_np).to(dev)
geom = ops.align_geometry(B, B, R, T, D)
xm, xe = ops.pack_images(im, ops.lengths_tensor(il, dev), geom)
y = ops.pack_captions(s, ops.lengths_tensor(sl, dev), geom)
out = torch.empty((B, B), dtype=torch.float32, device=dev)
e = torch.empty(geom.e_bytes, dtype=torch.uint8, device=dev)
ops.scores_from_packed(xm, xe, y, geom, out, e)
t0 = time.time()
n = 0
while time.time() - t0 < 2.0:
    for _ in range(200):
        ops.scores_from_packed(xm, xe, y, geom, out, e, reuse_side=True)
    n += 200
    torch.cuda.synchronize()
lib = _lib.load()
lib.aladin_debug_read_clock_probe.restype = C.c_int
nb = 1024
buf = (C.c_ulonglong * (4 * nb))()
assert lib.aladin_debug_read_clock_probe(buf, nb) == 0
raw = np.array(buf, dtype=np.uint64).reshape(nb, 4)
loop_rt = raw[:, 1].astype(np.float64)
r0, r1, rexit = raw[:, 2].astype(np.float64), raw[:, 3].astype(np.float64), raw[:, 0].astype(np.float64)
print('per workgroup (100 MHz ticks -> us): main loop %.2f, epilogue (loop end -> exit) %.2f' % (np.median(loop_rt) / 100, np.median(rexit - r1) / 100))
if os.environ['ALADIN_ALIGN_SPREAD'] == '26':
    lib.aladin_debug_read_clock_cycles.restype = C.c_int
    cyc = (C.c_ulonglong * nb)()
    assert lib.aladin_debug_read_clock_cycles(cyc, nb) == 0
    cyc = np.array(cyc, dtype=np.float64)
    clk = cyc / loop_rt * 100e6
    print('main-loop shader cycles per workgroup: median %.0f -> in-kernel clock %.3f GHz (p10 %.3f, p90 %.3f); MFMA pipe cycles per tile per SIMD 36864 -> pipe busy %.1f %% of the loop'
          % (np.median(cyc), np.median(clk) / 1e9, np.percentile(clk, 10) / 1e9, np.percentile(clk, 90) / 1e9, 100.0 * 36864 / np.median(cyc)))
start = r0.min()
print('kernel span: first loop start -> last exit %.2f us; loop start times per round (us, sorted sample): %s' % ((rexit.max() - start) / 100, np.round(np.sort((r0 - start) / 100)[::128], 1)))
sys.exit(0)
a = None
clk = None
print('launches %d; main-loop shader cycles per workgroup: median %.0f; real time %.2f us; in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f)'
      % (n, np.median(a[:, 0]), np.median(a[:, 1]) / 100.0, np.median(clk) / 1e9, np.percentile(clk, 10) / 1e9, np.percentile(clk, 90) / 1e9))
mfma_cycles = 12 * 4 * 24 * 32          # per SIMD per workgroup tile: K steps x kk x MFMAs of both waves x 32 cycles
print('MFMA pipe cycles per tile per SIMD: %d -> pipe busy %.1f %% of the loop cycles' % (mfma_cycles, 100.0 * mfma_cycles / np.median(a[:, 0])))
