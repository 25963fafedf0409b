"""Helpers shared by the ops modules: device checks, raw pointers, the current stream, length tensors, workspaces.
Every tensor function of this package requires float32 tensors on an AMD GPU ("cuda" device of PyTorch-ROCm) and raises
otherwise; there is no CPU or eager-PyTorch fallback."""
import ctypes as C

import torch

from . import _lib

_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of PyTorch's current stream (reference kernels run there too).  The raw getter is
    ~10x cheaper than torch.cuda.current_stream(), which matters for the small-batch, launch-bound step."""
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _require_gpu(*tensors):
    for t in tensors:
        if not isinstance(t, torch.Tensor):
            raise TypeError('aladin_amd: expected a torch.Tensor, got %r' % type(t))
        if not t.is_cuda:
            raise RuntimeError('aladin_amd: the alignment/matching path runs in HIP kernels on an MI355X only; '
                               'got a %s tensor (no CPU fallback exists)' % t.device)
        if t.dtype != torch.float32:
            raise TypeError('aladin_amd: float32 expected, got %s' % t.dtype)
        if t.device.index is not None and t.device.index != torch.cuda.current_device():
            # kernels are enqueued on the CURRENT device's current stream (one process per GPU)
            raise RuntimeError('aladin_amd: tensor on %s but the current device is cuda:%d; call torch.cuda.set_device() '
                               '(or use `with torch.cuda.device(...)`) first' % (t.device, torch.cuda.current_device()))


def _rows_inner_contig(t):
    """Keep permuted (S,B,D)->(B,S,D) views (reference alad/alad_model.py:377-378) without a copy
    as long as the feature axis is contiguous and rows stay 16-byte aligned."""
    if t.stride(-1) != 1 or any(st % 4 for st in t.stride()[:-1]) or t.data_ptr() % 16:
        return t.contiguous()
    return t


_LEN_CACHE = {}


def lengths_tensor(lens, device):
    """Python list / tensor of lengths -> int32 device tensor (the reference passes lists).
    Lists are cached by value so that a repeated batch shape costs no host->device copy."""
    if isinstance(lens, torch.Tensor):
        return lens.to(device=device, dtype=torch.int32, non_blocking=True)
    key = (tuple(lens), device)
    t = _LEN_CACHE.get(key)
    if t is None:
        if len(_LEN_CACHE) >= 256:
            _LEN_CACHE.clear()
        t = torch.tensor([int(x) for x in key[0]], dtype=torch.int32).to(device, non_blocking=True)
        _LEN_CACHE[key] = t
    return t


def _ld(t):
    """Leading dimension of a 2-D tensor whose rows are contiguous (stride(0) is arbitrary when there is one row)."""
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)



def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
