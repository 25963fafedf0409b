"""ctypes binding of libaladin_hip.so (the C ABI declared in include/aladin_hip.h).

This is the whole FFI surface: plain pointers, sizes and a hipStream_t.  There is no CPU fallback:
if the shared library is missing or an entry point fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ALADIN_LIB points at an alternative build of the same ABI (kernel A/B runs, tools/ab_bench.py)
LIB_PATH = os.environ.get('ALADIN_LIB') or os.path.join(_HERE, 'lib', 'libaladin_hip.so')

ABI_VERSION = 11
PRECISION_FP16, PRECISION_SPLIT = 0, 1      # ALADIN_PRECISION_* of include/aladin_hip.h
BWD_PARTNERS_FP16, BWD_DENSE, BWD_DENSE_GATHER, TRIPLET_BWD_BASE_WORKSPACE, BWD_OWN_ROW_FP16 = 1, 2, 4, 8, 16      # ALADIN_BWD_*, ALADIN_TRIPLET_BWD_BASE_WORKSPACE

# every symbol include/aladin_hip.h declares (tests check that the library exports all of them)
SYMBOLS = [
    'aladin_version', 'aladin_last_error', 'aladin_align_geometry', 'aladin_align_pack', 'aladin_align_scores',
    'aladin_align_bwd_workspace_bytes', 'aladin_align_bwd',
    'aladin_align_triplet_workspace_bytes', 'aladin_align_triplet_fwd', 'aladin_align_triplet_bwd', 'aladin_heads_small_fwd_argmax',
    'aladin_hinge_workspace_bytes', 'aladin_hinge_fwd_bwd', 'aladin_hinge_fused',
    'aladin_listnet_workspace_bytes', 'aladin_listnet_fwd_bwd',
    'aladin_distill_workspace_bytes', 'aladin_distill_mse_fwd_bwd', 'aladin_distill_contrastive_fwd_bwd',
    'aladin_distill_ordinal_fwd_bwd', 'aladin_order_sim_fwd', 'aladin_order_sim_bwd', 'aladin_sgemm_strided',
    'aladin_sim_workspace_bytes', 'aladin_sim_matrix', 'aladin_recall_workspace_bytes',
    'aladin_recall_ranks', 'aladin_normsum_fwd', 'aladin_normsum_bwd',
    'aladin_l2norm_fwd', 'aladin_l2norm_bwd',
    'aladin_retrieval_workspace_bytes', 'aladin_retrieval_ranks', 'aladin_retrieval_ranks_exact', 'aladin_retrieval_stats_offset',
    'aladin_scan_workspace_bytes', 'aladin_scan_fwd', 'aladin_scan_bwd',
    'aladin_store_row_width', 'aladin_store_append', 'aladin_align_pack_store_x', 'aladin_align_pack_store_y', 'aladin_topk',
    'aladin_loss_total', 'aladin_grad_combine', 'aladin_heads_small_workspace_bytes', 'aladin_heads_small_fwd', 'aladin_heads_small_bwd',
]


class AlignGeom(C.Structure):
    """struct aladin_align_geom."""
    _fields_ = [(n, C.c_int32) for n in ('Bi', 'Bc', 'R', 'T', 'D', 'Rq', 'Tq', 'mrows', 'rem', 'tp16', 'trows', 'Dp',
                                         'img_unit', 'cap_unit', 'Bi_pad', 'Bc_pad', 'x_tail', 'y_tail', 'split')] + \
               [(n, C.c_int64) for n in ('xm_rows', 'xe_rows', 'y_rows', 'xm_bytes', 'xe_bytes', 'y_bytes',
                                         'e_bytes', 'rnorm_bytes')]


class SetView(C.Structure):
    """struct aladin_set: a (B, N, D) fp32 batch of sets with its length tensor."""
    _fields_ = [('data', C.c_void_p), ('stride_b', C.c_int64), ('stride_r', C.c_int64), ('len', C.c_void_p)]


class GradView(C.Structure):
    """struct aladin_set_grad."""
    _fields_ = [('data', C.c_void_p), ('stride_b', C.c_int64), ('stride_r', C.c_int64)]


class Packed(C.Structure):
    """struct aladin_packed: the fp16 MFMA operands of one problem + the rows' inverse norms."""
    _fields_ = [('xm', C.c_void_p), ('xe', C.c_void_p), ('y', C.c_void_p), ('rnorm', C.c_void_p)]


_lib = None


def _declare(lib):
    p, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
    G = C.POINTER(AlignGeom)
    SV, GV, PK = C.POINTER(SetView), C.POINTER(GradView), C.POINTER(Packed)
    sig = {
        'aladin_version': (C.c_int, []),
        'aladin_last_error': (C.c_char_p, []),
        'aladin_align_geometry': (C.c_int, [i32, i32, i32, i32, i32, i32, i32, i32, G]),
        'aladin_align_pack': (C.c_int, [SV, SV, G, PK, p]),
        'aladin_align_scores': (C.c_int, [PK, G, p, p, i64, i32, p]),
        'aladin_align_bwd_workspace_bytes': (sz, [G, i32]),
        'aladin_align_bwd': (C.c_int, [SV, SV, G, PK, p, i64, p, p, p, GV, GV, p, i32, p]),
        'aladin_align_triplet_workspace_bytes': (sz, [G]),
        'aladin_align_triplet_fwd': (C.c_int, [SV, SV, G, f32, PK, p, i64, p, p, p, p]),
        'aladin_align_triplet_bwd': (C.c_int, [SV, SV, G, PK, p, p, GV, GV, p, i32, p]),
        'aladin_heads_small_fwd_argmax': (C.c_int, [p, i64, p, i64, p, i64, i32, f32, i32, f32, f32, f32, f32, f32, p, p, p, p, p, p, p,
                                                    SV, SV, G, PK, p, p]),
        'aladin_hinge_workspace_bytes': (sz, [i32]),
        'aladin_hinge_fwd_bwd': (C.c_int, [p, i64, i32, f32, i32, p, p, p, p]),
        'aladin_hinge_fused': (C.c_int, [p, i64, i32, f32, i32, p, p, p, p, p, p]),
        'aladin_listnet_workspace_bytes': (sz, [i32]),
        'aladin_listnet_fwd_bwd': (C.c_int, [p, i64, p, i64, i32, f32, f32, p, p, p, p]),
        'aladin_distill_workspace_bytes': (sz, [i32]),
        'aladin_distill_mse_fwd_bwd': (C.c_int, [p, i64, p, i64, i32, p, p, p, p, p, p]),
        'aladin_distill_contrastive_fwd_bwd': (C.c_int, [p, i64, p, i64, i32, f32, p, p, p, p]),
        'aladin_distill_ordinal_fwd_bwd': (C.c_int, [p, i64, p, i64, i32, f32, f32, i32, p, p, p, p]),
        'aladin_order_sim_fwd': (C.c_int, [p, i64, p, i64, i32, i32, i32, p, i64, p]),
        'aladin_order_sim_bwd': (C.c_int, [p, i64, p, i64, i32, i32, i32, p, i64, p, i64, p, i64, p, i64, p]),
        'aladin_sgemm_strided': (C.c_int, [i32, i32, i32, p, i64, i64, p, i64, i64, p, i64, p]),
        'aladin_sim_workspace_bytes': (sz, [i32, i32, i32]),
        'aladin_sim_matrix': (C.c_int, [p, i64, p, i64, i32, i32, i32, p, i64, p, p]),
        'aladin_normsum_fwd': (C.c_int, [p, i64, i64, p, i32, i32, i32, i32, p, p]),
        'aladin_normsum_bwd': (C.c_int, [p, i64, i64, p, i32, i32, i32, i32, p, p, p]),
        'aladin_l2norm_fwd': (C.c_int, [p, i64, i32, i32, p, p]),
        'aladin_l2norm_bwd': (C.c_int, [p, i64, p, i64, i32, i32, p, p]),
        'aladin_retrieval_workspace_bytes': (sz, [i32, i32, i32]),
        'aladin_retrieval_ranks': (C.c_int, [p, i64, p, i64, i32, i32, i32, i32, p, p, p, p, p, p]),
        'aladin_retrieval_ranks_exact': (C.c_int, [p, i64, p, i64, i32, i32, i32, i32, p, p, p, p, p, p]),
        'aladin_retrieval_stats_offset': (sz, [i32, i32, i32]),
        'aladin_scan_workspace_bytes': (sz, [i32, i32, i32, i32, i32, i32]),
        'aladin_scan_fwd': (C.c_int, [p, i64, i64, p, p, i64, i64, p, i32, i32, i32, i32, i32, p, i64, p, p]),
        'aladin_scan_bwd': (C.c_int, [p, i64, i64, p, p, i64, i64, p, i32, i32, i32, i32, i32, p, i64, p, p, p, p, p]),
        'aladin_store_row_width': (C.c_int, [i32, i32]),
        'aladin_store_append': (C.c_int, [p, i64, i64, p, i32, i32, i32, i32, p, p, i32, p]),
        'aladin_topk': (C.c_int, [p, i64, i64, i32, i32, i32, p, p, p]),
        'aladin_loss_total': (C.c_int, [p, f32, p, f32, p, f32, p, p]),
        'aladin_grad_combine': (C.c_int, [i64, p, f32, p, f32, p, p, f32, p, p]),
        'aladin_heads_small_workspace_bytes': (sz, [i32]),
        'aladin_heads_small_fwd': (C.c_int, [p, i64, p, i64, p, i64, i32, i32, f32, i32, i32, f32, f32, f32, f32, f32, p, p, p, p, p, p,
                                             p, p, p, p]),
        'aladin_heads_small_bwd': (C.c_int, [p, i64, p, i64, i32, i32, p, p, f32, p, p, f32, p, i64, p, f32, p, p, p, p]),
        'aladin_align_pack_store_x': (C.c_int, [p, p, p, p, G, p, p, p]),
        'aladin_align_pack_store_y': (C.c_int, [p, p, p, p, G, p, p]),
        'aladin_recall_workspace_bytes': (sz, [i32]),
        'aladin_recall_ranks': (C.c_int, [p, i64, i32, i32, i32, p, p, p, p, p, p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def _preload_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64.so (SONAME libamdhip64.so.7) and asks for it by the
    unversioned name; this library asks for libamdhip64.so.7.  If ours were loaded first the loader
    would map a SECOND HIP runtime for torch and streams/devices would not be shared.  Loading
    torch's copy first makes the loader reuse it for us (same SONAME)."""
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError if the extension is absent."""
    global _lib
    if _lib is None:
        _preload_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'aladin_amd: HIP extension %s is missing -- build it with `python -c "import __graft_entry__ as g; '
                'g.build()"` (or `make -C aladin_amd/csrc`).  There is no CPU fallback.' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        _declare(lib)
        if lib.aladin_version() != ABI_VERSION:
            raise RuntimeError('aladin_amd: ABI version mismatch (library %d, binding %d)'
                               % (lib.aladin_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().aladin_last_error()
        raise RuntimeError('aladin_hip %s failed (status %d): %s' % (what, rc, msg.decode() if msg else ''))
