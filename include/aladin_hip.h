/*
 * aladin_hip.h -- C ABI of libaladin_hip.so: ALADIN's fine-grained alignment-scoring and
 * matching-retrieval hot path as hand-written HIP kernels for gfx950 (MI355X, CDNA4).
 *
 * The reference (mesnico/ALADIN) is pure Python/PyTorch and has no FFI of its own; the boundary
 * this library slots under is the Python call surface of alad/loss.py, alad/recall_auxiliary.py
 * and alad/evaluation.py.  Each entry point below names the reference
 * lines it replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add there;
 * aladin_amd/_lib.py is that binding as shipped.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless a parameter says "host";
 *   - `stream` is a hipStream_t (PyTorch: torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued asynchronously on it, nothing synchronises, nothing is allocated: outputs and
 *     workspaces are caller-owned, and no pointer is retained after return;
 *   - return value: 0 = ok, 1 = bad argument, 2 = unsupported shape, 3 = HIP launch error;
 *     aladin_last_error() gives the message (thread-local).  Never aborts.
 *   - float = IEEE binary32.  16-bit packed operands are IEEE binary16 (fp16): the MFMA path
 *     multiplies fp16 operands and accumulates in fp32 (see DESIGN.md for why not bf16).
 */
#ifndef ALADIN_HIP_H
#define ALADIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALADIN_ABI_VERSION 11

/* The library is built with -fvisibility=hidden: the entry points declared here are its ONLY exports. */
#if defined(__GNUC__)
#define ALADIN_API __attribute__((visibility("default")))
#else
#define ALADIN_API
#endif

ALADIN_API int aladin_version(void);
ALADIN_API const char* aladin_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Alignment scores  S[i][j] = sum_w max_r <im^[i,r], s^[j,w]>      (aggregation 'MrSw')
 * replaces AlignmentContrastiveLoss.forward, reference alad/loss.py:79-125 (normalise :80-81,
 * drop region 0 / token 0 / last two tokens :87-90, B*B batched matmul :97-99, length masks
 * :103-116, max over regions + sum over words :124-125).
 *
 * Three steps so that the packed image operand can be all-gathered between GPUs (RCCL) before
 * scoring:  pack(images) -> [all-gather] -> scores  <- pack(captions).  The training step of one
 * GPU is ONE call per direction: aladin_align_triplet_fwd / _bwd below.
 *
 * One entry point per operation; the alignment family takes its arguments in three small structs
 * (aladin_set, aladin_set_grad, aladin_packed).
 * ------------------------------------------------------------------------------------------- */

/* A batch of sets (B, N, D) fp32 in HBM: element [b][n][d] at data[b * stride_b + n * stride_r + d] (strides in floats,
 * unit inner stride, rows 16-byte aligned when D % 4 == 0: the reference hands permuted (S,B,D)->(B,S,D) views,
 * alad/alad_model.py:377-378).  len: B int32 on the device, the reference's length lists. */
typedef struct aladin_set {
  const float* data;
  int64_t stride_b, stride_r;
  const int32_t* len;
} aladin_set;

/* Where a gradient of such a batch goes: the CALLER'S layout (the same permuted views), every row written exactly once. */
typedef struct aladin_set_grad {
  float* data;
  int64_t stride_b, stride_r;
} aladin_set_grad;

typedef struct aladin_align_geom {
  int32_t Bi, Bc, R, T, D;      /* inputs: im (Bi,R,D), s (Bc,T,D)                              */
  int32_t Rq, Tq;               /* R-1 regions and T-3 words take part (alad/loss.py:87-88)     */
  int32_t mrows;                /* rows per image in the main operand xm: 32, 48, 64 or 96 (rows past R' repeat region 0) */
  int32_t rem;                  /* leftover regions per image (R' - mrows when positive) that go through the side GEMM: <= 8 */
  int32_t tp16;                 /* 16-word column tiles a caption needs: ceil(trows / 16)       */
  int32_t trows;                /* rows per caption in y: 16 * tp16, or 16 * tp16 - 8 = 8 / 24 / 40 (T' <= 8 / 17..24 / 33..40
                                   with 32, 48 or 64 main rows: two captions share one / three / five 16-word tiles) */
  int32_t Dp;                   /* halfs per packed row: D rounded up to 64 (zero filled), x3 when split */
  int32_t img_unit, cap_unit;   /* images / captions per workgroup tile                         */
  int32_t Bi_pad, Bc_pad;       /* batch sizes rounded up to the units (zero rows)              */
  int32_t x_tail, y_tail;       /* positions dropped at the END of the max-side / sum-side sets: the
                                   set uses positions 1 .. N-1-tail and length len-1-tail.  Images 0
                                   (alad/loss.py:87,89), captions 2 (:88,90)                      */
  int32_t split;                /* 1: split-fp16 operands (ALADIN_PRECISION_SPLIT), Dp = 3 * round_up(D, 64) */
  int64_t xm_rows, xe_rows, y_rows;            /* rows of the packed fp16 operands              */
  int64_t xm_bytes, xe_bytes, y_bytes;         /* their sizes                                   */
  int64_t e_bytes;              /* fp32 scratch for the side GEMM, xe_rows x y_rows (0 if !rem) */
  int64_t rnorm_bytes;          /* fp32, one per packed row in the order [xm rows | xe rows | y rows]: 1 / max(|x|, 1e-12) of the
                                   raw vector a row holds (0 for a zero row) -- what the backward's normalise step divides by */
} aladin_align_geom;

/* The packed fp16 MFMA operands of one problem (caller-owned buffers of geom->xm_bytes, xe_bytes, y_bytes, rnorm_bytes).
 * xe may be NULL when !geom->rem; rnorm may be NULL (then the backward reads the raw fp32 rows instead). */
typedef struct aladin_packed {
  void* xm;
  void* xe;
  void* y;
  float* rnorm;
} aladin_packed;

/* Operand precision of the packed sets.
 *   ALADIN_PRECISION_FP16   one rounding of every unit vector to fp16: scores within ~1e-4 of the fp32
 *                           reference (the training path: north_star's 1e-3 tolerance, full MFMA rate).
 *   ALADIN_PRECISION_SPLIT  x^ * 2^14 = hi + lo (fp16 each); a packed row is three K segments
 *                           [hi | lo | hi] (max side) / [hi | hi | lo] (sum side), so the SAME pack / score entry
 *                           points contract hi.hi + lo.hi + hi.lo with fp32 accumulation: ~2^-22 relative operand
 *                           error, i.e. the rounding level of the reference's own fp32 bmm, at 3x the MFMA work.
 *                           This is what evaluation uses: Recall needs rank-exact scores (near-ties between
 *                           thousands of candidates flip at the 1e-4 level; alad/evaluation.py:213-223,303-308).
 *                           Forward only: the backward entry points reject split operands. */
#define ALADIN_PRECISION_FP16 0
#define ALADIN_PRECISION_SPLIT 1
#define ALADIN_PRECISION_SPLIT_TABLE 2   /* split operands in whole 16-word caption tiles (no 24- / 40-word classes): the layout
                                            the dense backward's arg-max table kernel reads */

/* Host-only: derive the packed layout.  The set on the MAX side (Bi x R) and the set on the SUM side (Bc x T) each state
 * how many trailing positions they drop: 'MrSw' = (images, x_tail 0) x (captions, y_tail 2); 'MwSr' (alad/loss.py:134-135,
 * max over words, sum over regions) = (captions, tail 2) x (images, tail 0), result transposed.  Every "image" / "caption"
 * argument below means max-side / sum-side set. */
ALADIN_API int aladin_align_geometry(int Bi, int Bc, int R, int T, int D, int x_tail, int y_tail, int precision,
                                     aladin_align_geom* out);

/* L2-normalise (eps 1e-12, F.normalize), slice, length-mask and convert sets to the packed fp16 MFMA operands: the image
 * sets into out->xm / xe, the caption sets into out->y, in ONE launch when both are given; either may be NULL (the members
 * of `out` it would fill are then not touched).  out->rnorm, when not NULL, receives the rows' inverse norms. */
ALADIN_API int aladin_align_pack(const aladin_set* im, const aladin_set* s, const aladin_align_geom* geom,
                                 const aladin_packed* out, void* stream);

/* S (Bi x Bc, row stride ldS floats) from packed operands.  e_scratch: geom->e_bytes.
 * ALADIN_SCORES_REUSE_SIDE: e_scratch already holds the side-GEMM result of a previous call on the same operands, launch
 * the score kernel alone (used by bench.py to time the dominant kernel in isolation). */
#define ALADIN_SCORES_REUSE_SIDE 1
ALADIN_API int aladin_align_scores(const aladin_packed* packed, const aladin_align_geom* geom, void* e_scratch, float* S,
                                   int64_t ldS, int flags, void* stream);

/* Backward of S w.r.t. the raw sets (autograd of alad/loss.py:80-125).
 * dS (Bi x Bc, stride ld_dS) is multiplied by *gscale (device float, may be NULL = 1).  Pairs with dS == 0 are skipped, so
 * the max_violation=True hinge (<= 3B non-zeros) costs O(B) pair blocks.  Three steps: (1) the non-zero pairs (compacted
 * here, or given: pairs / pair_count as aladin_hinge_fused emits them; both or neither), (2) per pair the arg-max region of
 * every word -- recomputed on the fp16 MFMA from `packed` when given (xm, y and, with side rows, xe), with every word whose
 * top candidates are closer than the fp16 error bound re-decided by exact fp32 dot products, or entirely in fp32 when
 * packed is NULL -- the recorded arg-max is the fp32 arg-max, as autograd's; (3) one wave per OUTPUT row gathers the partner
 * rows the table points at and applies the normalise backward.  d_im / d_s: fully written, no atomics.
 * workspace: aladin_align_bwd_workspace_bytes(geom, flags).  Split operands are rejected (forward only).
 *   ALADIN_BWD_PARTNERS_FP16  step 3 gathers the PARTNER rows (the unit vectors an output row's gradient is a weighted sum of)
 *       from the forward's packed fp16 operands instead of normalising the raw fp32 rows again: half the bytes per partner, no norm
 *       reduction.  One fp16 rounding of those vectors (<= 2^-11 relative per component) leaves the gradients within 5e-4 of their
 *       LARGEST entry of the reference's autograd for D >= 128 (3e-7 without the flag; element by element the error reaches
 *       ~3e-3 of an entry a tenth of the largest -- a max-normalised bound).  Needs packed->xm, y, rnorm
 *       (and xe with side rows) of THIS problem.  The Python layer sets it by default (ops.set_backward_precision('exact') clears
 *       it).  The arg-maxima are the exact fp32 ones either way.
 *   ALADIN_BWD_OWN_ROW_FP16  (with ALADIN_BWD_PARTNERS_FP16) the output row's OWN unit vector and inverse norm come from the packed
 *       operands and packed->rnorm too: the raw sets are not read by step 3 at all.  Up to ~7e-4 of the largest entry when the
 *       partners are aligned with the row (cosines near 1): an opt-in ('fp16-own').
 *   ALADIN_BWD_DENSE  the caller states that (almost) every pair carries a gradient -- the sum-of-violations hinge
 *       (max_violation = False, alad/loss.py:60-67), or a gradient arriving on the score matrix itself.  The arg-max table of
 *       ALL pairs then comes from the forward's own tile kernel run in split precision (64 pairs per workgroup sharing their
 *       operand panels) instead of one workgroup per pair; only the pairs with a word whose two best regions it cannot
 *       separate (a few per cent) go through the exact per-pair kernel.  Same table, same gradients.  Classes the tile kernel
 *       does not cover (R' > 64, small batches) silently take the list path.  `pairs` / `pair_count` are ignored.
 *       With the table of all pairs in hand step 3 runs as two MFMA GEMMs, dXh = P Yh and dYh = P^T Xh with
 *       P[(i,r),(c,w)] = dS[i,c] [argmax == r] generated in registers from the table (csrc/align_bwd_dense.hip), exact to
 *       ~2^-22 by hi + lo splitting (one product under ALADIN_BWD_PARTNERS_FP16) -- sums in a different order than the
 *       gather, so equal to it to rounding, not bit for bit.
 *   ALADIN_BWD_DENSE_GATHER  (with ALADIN_BWD_DENSE) keep the per-row gather as step 3: bit-identical to the list path. */
#define ALADIN_BWD_PARTNERS_FP16 1
#define ALADIN_BWD_DENSE 2
#define ALADIN_BWD_DENSE_GATHER 4
#define ALADIN_BWD_OWN_ROW_FP16 16
ALADIN_API size_t aladin_align_bwd_workspace_bytes(const aladin_align_geom* geom, int flags);
ALADIN_API int aladin_align_bwd(const aladin_set* im, const aladin_set* s, const aladin_align_geom* geom,
                                const aladin_packed* packed, const float* dS, int64_t ld_dS, const float* gscale,
                                const int32_t* pairs, const int32_t* pair_count, const aladin_set_grad* d_im,
                                const aladin_set_grad* d_s, void* workspace, int flags, void* stream);

/* The training step of AlignmentContrastiveLoss(max_violation=True, aggregation='MrSw') -- every shipped YAML,
 * alad/loss.py:79-159 through alad/alad_model.py:386 -- as ONE call per direction, so that an eager drop-in module costs one
 * FFI crossing per direction.
 *   fwd: pack both sets (-> packed, kept by the caller for the backward), side GEMM + score kernel (-> S, Bi x Bi), the
 *        hinge's row / column statistics, then ONE kernel whose workgroups either recompute a non-zero pair of dloss/dS --
 *        the pairs follow from the statistics: (q, q), (q, hardest caption of image q), (hardest image of caption q, q) --
 *        into the backward's arg-max table, or run the hinge's element-wise pass (-> *loss, dS: (B, B) contiguous, fully
 *        written).  Square problems (Bi == Bc) of the classes the fp16 pair kernel covers (geom->mrows 32 / 48 with side rows
 *        or 64 without, <= 64 padded words; D % 4 == 0, D <= 1024), fp16 operands; anything else: ALADIN_ERR_UNSUPPORTED,
 *        and the caller composes pack + scores + aladin_hinge_fused + aladin_align_bwd.
 *   bwd: step 3 of aladin_align_bwd alone (the table is in the workspace), dS scaled by *gscale (the upstream gradient of
 *        the loss: a device scalar, may be NULL = 1).  flags: ALADIN_BWD_PARTNERS_FP16 and / or
 *        ALADIN_TRIPLET_BWD_BASE_WORKSPACE (`workspace` is the bwd_workspace aladin_heads_small_fwd_argmax filled, not the
 *        triplet workspace).
 * workspace: aladin_align_triplet_workspace_bytes(geom) -- side-GEMM scratch, hinge statistics, arg-max table, dS^T;
 * written by fwd, read by bwd: the caller keeps it (and packed, dS) untouched in between.  Both calls are asynchronous,
 * allocation-free and capturable in a HIP graph. */
#define ALADIN_TRIPLET_BWD_BASE_WORKSPACE 8
ALADIN_API size_t aladin_align_triplet_workspace_bytes(const aladin_align_geom* geom);
ALADIN_API int aladin_align_triplet_fwd(const aladin_set* im, const aladin_set* s, const aladin_align_geom* geom, float margin,
                                        const aladin_packed* packed, float* S, int64_t ldS, float* loss, float* dS,
                                        void* workspace, void* stream);
ALADIN_API int aladin_align_triplet_bwd(const aladin_set* im, const aladin_set* s, const aladin_align_geom* geom,
                                        const aladin_packed* packed, const float* dS, const float* gscale,
                                        const aladin_set_grad* d_im, const aladin_set_grad* d_s, void* workspace, int flags,
                                        void* stream);

/* The same forward for the small-batch loss heads (B <= 64; aladin_heads_small_fwd below): S is given (aladin_align_scores on
 * `packed`), one statistics launch + ONE kernel running the element-wise pass of all heads next to the pair recompute.
 * Hardest-negative hinge only; flags must include ALADIN_HEAD_ALIGN_HINGE.  Arguments as aladin_heads_small_fwd (no pair
 * list) followed by the alignment problem; D_emb is the width of the matching embeddings.  bwd_workspace:
 * aladin_align_bwd_workspace_bytes(geom, 0) -- it receives the arg-max table and dS^T; the backward is
 * aladin_align_triplet_bwd on that workspace. */
ALADIN_API int aladin_heads_small_fwd_argmax(const float* img, int64_t ld_img, const float* cap, int64_t ld_cap, const float* S,
                                             int64_t ld_S, int D_emb, float margin, int flags, float temperature, float eps,
                                             float w_match, float w_align, float w_dist, float* M, float* terms, float* total,
                                             float* dM_hinge, float* dM_listnet, float* dS, void* heads_workspace,
                                             const aladin_set* im, const aladin_set* s, const aladin_align_geom* geom,
                                             const aladin_packed* packed, void* bwd_workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 'sum' / 'mean' pooling (alad/loss.py:120-123): sum_r sum_w <im^,s^> = <sum_r im^, sum_w s^>.
 * normsum: out[b][:] = sum over positions 1 .. len[b]-1-tail of the L2-normalised rows of x (B,N,D);
 * the scores are then aladin_sgemm_strided(out_img, out_cap^T).  bwd: d_x (B,N,D contiguous, fully
 * written) from d_out (B,D).
 * ------------------------------------------------------------------------------------------- */
ALADIN_API int aladin_normsum_fwd(const float* x, int64_t stride_b, int64_t stride_r, const int32_t* len, int B, int N, int D,
                       int tail, float* out, void* stream);
ALADIN_API int aladin_normsum_bwd(const float* x, int64_t stride_b, int64_t stride_r, const int32_t* len, int B, int N, int D,
                       int tail, const float* d_out, float* d_x, void* stream);

/* ---------------------------------------------------------------------------------------------
 * VSE++ hinge loss on a square score matrix -- Contrastive.compute_contrastive_loss,
 * reference alad/loss.py:42-67.  loss: 1 float.  dS (B x B, contiguous) may be NULL; it receives
 * dloss/dS (max_violation: +-1 at the hardest negatives and the diagonal, <= 3B non-zeros).
 * workspace: aladin_hinge_workspace_bytes(B).
 * ------------------------------------------------------------------------------------------- */
ALADIN_API size_t aladin_hinge_workspace_bytes(int B);
ALADIN_API int aladin_hinge_fwd_bwd(const float* S, int64_t ldS, int B, float margin, int max_violation,
                         float* loss, float* dS, void* workspace, void* stream);

/* Same, additionally emitting the list of non-zero pairs of dS (pairs: B*B int32 holding i*B + j in
 * arbitrary order, pair_count: 1 int32) that aladin_align_bwd can consume directly. */
ALADIN_API int aladin_hinge_fused(const float* S, int64_t ldS, int B, float margin, int max_violation, float* loss,
                       float* dS, int32_t* pairs, int32_t* pair_count, void* workspace, void* stream);

/* ListNet score distillation -- DistillationLoss(mode='listnet'), reference alad/loss.py:427-445
 * (teacher detached :370; temperature 6 on the student only; eps 1e-10 inside the log).
 * d_student (B x B contiguous) may be NULL. */
ALADIN_API size_t aladin_listnet_workspace_bytes(int B);
ALADIN_API int aladin_listnet_fwd_bwd(const float* teacher, int64_t ld_t, const float* student, int64_t ld_s, int B,
                           float temperature, float eps, float* loss, float* d_student,
                           void* workspace, void* stream);

/* The fixed-weight sum of the loss terms (alad_model.py:450-453) and its backward without element-wise glue launches
 * (any batch size; the B <= 64 heads below do this inside their own kernels).
 *   aladin_loss_total    *total = wa * *a + wb * *b + wc * *c over up to three DEVICE scalars (NULL = absent)
 *   aladin_grad_combine  out[e] = *g * (wa * A[e] + wb * B[e]), e < n (A / B may be NULL; out may be NULL), and
 *                        *scale_out = *g * w_scale (may be NULL): the upstream gradient of the total applied to the
 *                        dLoss/dM matrices of the matching hinge and of ListNet, and turned into the alignment
 *                        backward's gscale, in one launch. */
ALADIN_API int aladin_loss_total(const float* a, float wa, const float* b, float wb, const float* c, float wc, float* total,
                                 void* stream);
ALADIN_API int aladin_grad_combine(int64_t n, const float* g, float wa, const float* A, float wb, const float* B, float* out,
                                   float w_scale, float* scale_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loss heads at the shipped batch size (B <= 64; every YAML trains with bs 32): three launches instead of the
 * ~15 few-microsecond kernels of the general path (csrc/small_batch.hip).  `flags` selects the heads:
 *   ALADIN_HEAD_MATCH_HINGE  M = img . cap^T (dot_sim, alad/loss.py:8-11) and the VSE++ hinge on it (:42-67)
 *   ALADIN_HEAD_ALIGN_HINGE  the same hinge on the alignment scores S (B x B, from aladin_align_scores)
 *   ALADIN_HEAD_LISTNET      ListNet distillation of M from the teacher S (:427-445; also computes M)
 * fwd (2 launches) writes M (B x B contiguous; needed by MATCH_HINGE / LISTNET), terms[3] = {matching hinge,
 * alignment hinge, listnet} (0 for a head that is off), *total = sum of the selected terms times w_* (the
 * fixed-weight sum of alad_model.py:450-453; may be NULL), and -- each optional -- dM_hinge, dM_listnet (B x B),
 * dS (B x B) with the non-zero pair list pairs / pair_count of the alignment hinge, in the form
 * aladin_align_bwd takes.  workspace: aladin_heads_small_workspace_bytes(B).
 * bwd (1 launch): d_img = C . cap, d_cap = C^T . img, C = *g_hinge * w_hinge * dM_hinge + *g_listnet * w_listnet *
 * dM_listnet + g_M (every term optional: NULL = absent; g_* are DEVICE scalars, g_M a (B x B) upstream gradient of M);
 * align_scale_out (may be NULL) receives *g_align * w_align, the `gscale` of the alignment backward.
 * d_img / d_cap: (B x D) contiguous, either may be NULL.
 * ------------------------------------------------------------------------------------------- */
#define ALADIN_HEAD_MATCH_HINGE 1
#define ALADIN_HEAD_ALIGN_HINGE 2
#define ALADIN_HEAD_LISTNET 4
ALADIN_API size_t aladin_heads_small_workspace_bytes(int B);
ALADIN_API int aladin_heads_small_fwd(const float* img, int64_t ld_img, const float* cap, int64_t ld_cap, const float* S,
                                      int64_t ld_S, int B, int D, float margin, int max_violation, int flags, float temperature,
                                      float eps, float w_match, float w_align, float w_dist, float* M, float* terms, float* total,
                                      float* dM_hinge, float* dM_listnet, float* dS, int32_t* pairs, int32_t* pair_count,
                                      void* workspace, void* stream);
ALADIN_API int aladin_heads_small_bwd(const float* img, int64_t ld_img, const float* cap, int64_t ld_cap, int B, int D,
                                      const float* dM_hinge, const float* g_hinge, float w_hinge, const float* dM_listnet,
                                      const float* g_listnet, float w_listnet, const float* g_M, int64_t ld_gM,
                                      const float* g_align, float w_align, float* align_scale_out, float* d_img, float* d_cap,
                                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Device-resident evaluation store (replaces the (N, 71, D) fp32 host buffers that encode_data fills,
 * reference alad/evaluation.py:119-130, and the per-query H2D copies of i2t / t2i :179,202,267,291).
 * A store holds, per sample, only positions [1, len - tail) of its set, L2-normalised like the pack
 * kernels and rounded once to fp16, contiguous by true length:
 *   rows (total_rows x aladin_store_row_width(D)) fp16; sample k = rows [offsets[k], offsets[k]+counts[k]).
 * aladin_store_append normalises one encoder batch (B, L, D; strides in elements) into rows at the
 * given per-sample offsets (int64, device); counts are max(0, lens[k] - 1 - tail) clipped to L - 1.
 * aladin_align_pack_store_x / _y build the max-side (xm, xe) / sum-side (y) operands of
 * aladin_align_scores for the samples ids[0..Bi) / ids[0..Bc) (ids NULL = 0, 1, 2, ...) under a
 * geometry whose Rq / Tq bound the counts; the operands -- hence the scores -- are bit-identical to
 * aladin_align_pack on the fp32 sets (no inverse norms: the store is forward-only).
 * Split stores (precision ALADIN_PRECISION_SPLIT): a row is [hi | lo], 2 * round_up(D, 64) halfs;
 * aladin_align_pack_store_x / _y take such rows when (and only when) geom->split.
 * ------------------------------------------------------------------------------------------- */
ALADIN_API int aladin_store_row_width(int D, int precision);      /* halfs per store row */
ALADIN_API int aladin_store_append(const float* sets, int64_t stride_b, int64_t stride_r, const int32_t* lens, int B, int L,
                                   int D, int tail, const int64_t* offsets, void* rows, int precision, void* stream);
ALADIN_API int aladin_align_pack_store_x(const void* rows, const int64_t* offsets, const int32_t* counts, const int32_t* ids,
                              const aladin_align_geom* g, void* xm, void* xe, void* stream);
ALADIN_API int aladin_align_pack_store_y(const void* rows, const int64_t* offsets, const int32_t* counts, const int32_t* ids,
                              const aladin_align_geom* g, void* y, void* stream);

/* l2norm -- reference alad/utils.py:134-139: out = X / sqrt(sum_dim1 X^2) on (rows, D), NO eps (a zero row
 * gives NaN, as the reference; F.normalize would give 0), and its backward.  out / d_x are contiguous. */
ALADIN_API int aladin_l2norm_fwd(const float* x, int64_t row_stride, int rows, int D, float* out, void* stream);
ALADIN_API int aladin_l2norm_bwd(const float* x, int64_t row_stride, const float* d_out, int64_t d_out_stride, int rows, int D,
                      float* d_x, void* stream);

/* ---------------------------------------------------------------------------------------------
 * aggregation = 'scan-sentences' -- reference alad/loss.py:136-149 (no shipped config uses it).
 * Per pair: relu(cosines) L2-normalised over regions, softmax over the caption's valid words for each
 * valid region, attended sentence vector, S = sum over regions of cos(region, attended vector).
 * Everything in fp32 (exact-fp32 MFMA GEMMs).  Same set / length / stride conventions as
 * aladin_align_bwd; R - 1 <= 96, T - 3 <= 96.  A caption without scored words gives NaN like the
 * reference.  aladin_scan_bwd recomputes the forward state itself and returns the gradient of the
 * length-masked expression (the reference's autograd is NaN on ragged batches and equal to this on
 * full-length ones); gscale (device scalar, may be NULL) multiplies dS.
 * ------------------------------------------------------------------------------------------- */
ALADIN_API size_t aladin_scan_workspace_bytes(int Bi, int Bc, int R, int T, int D, int backward);
ALADIN_API int aladin_scan_fwd(const float* im, int64_t im_stride_b, int64_t im_stride_r, const int32_t* im_len, const float* s,
                    int64_t s_stride_b, int64_t s_stride_t, const int32_t* s_len, int Bi, int Bc, int R, int T, int D,
                    float* S, int64_t ldS, void* workspace, void* stream);
ALADIN_API int aladin_scan_bwd(const float* im, int64_t im_stride_b, int64_t im_stride_r, const int32_t* im_len, const float* s,
                    int64_t s_stride_b, int64_t s_stride_t, const int32_t* s_len, int Bi, int Bc, int R, int T, int D,
                    const float* dS, int64_t ld_dS, const float* gscale, float* d_im, float* d_s, void* workspace,
                    void* stream);

/* The other modes of DistillationLoss, reference alad/loss.py:359-425 (teacher detached :370).
 * One workspace query covers the three of them.  d_student (B x B contiguous) may be NULL.
 *   mse          :371-373  mean((student*wb[0] + wb[1] - teacher)^2); wb = the module's learnable
 *                          pair (:366), a DEVICE pointer; d_wb (2 floats, may be NULL) receives its gradient
 *   contrastive  :401-425  as written: teacher diagonal zeroed, its row argmax selects whole columns
 *                          of cost_s and its column argmax whole rows of cost_im, diagonals kept
 *   ordinal      :374-399  per-row and per-column teacher sort, strided hinge on the student in that
 *                          order where teacher_sorted[p + stride] >= threshold; a side with no selected
 *                          position makes the loss NaN and contributes a zero gradient (as torch). B <= 8192 */
ALADIN_API size_t aladin_distill_workspace_bytes(int B);
ALADIN_API int aladin_distill_mse_fwd_bwd(const float* teacher, int64_t ld_t, const float* student, int64_t ld_s, int B,
                               const float* wb, float* loss, float* d_student, float* d_wb, void* workspace,
                               void* stream);
ALADIN_API int aladin_distill_contrastive_fwd_bwd(const float* teacher, int64_t ld_t, const float* student, int64_t ld_s, int B,
                                       float margin, float* loss, float* d_student, void* workspace, void* stream);
ALADIN_API int aladin_distill_ordinal_fwd_bwd(const float* teacher, int64_t ld_t, const float* student, int64_t ld_s, int B,
                                   float margin, float threshold, int stride, float* loss, float* d_student,
                                   void* workspace, void* stream);

/* Order-embedding similarity -- order_sim, reference alad/loss.py:20-26 (measure='order'):
 * scores[i][j] = -|| max(s_j - im_i, 0) ||_2, and its backward given d_scores and the forward's scores
 * (a pair without any violation has 0/0 = NaN gradient, as autograd).  d_im / d_s may be NULL. */
ALADIN_API int aladin_order_sim_fwd(const float* im, int64_t ld_im, const float* s, int64_t ld_s, int Bi, int Bc, int D,
                         float* scores, int64_t ld_scores, void* stream);
ALADIN_API int aladin_order_sim_bwd(const float* im, int64_t ld_im, const float* s, int64_t ld_s, int Bi, int Bc, int D,
                         const float* d_scores, int64_t ld_g, const float* scores, int64_t ld_scores,
                         float* d_im, int64_t ld_dim, float* d_s, int64_t ld_ds, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dot-product scores  C[m][n] = sum_k A[m*a_rs + k*a_cs] * B[k*b_rs + n*b_cs]   (fp32 in/out,
 * exact-fp32 MFMA).  With a_cs = 1, b_rs = 1 it is dot_sim  im.mm(s.t()), reference
 * alad/loss.py:8-11; the strides also give the two backward products.
 * ------------------------------------------------------------------------------------------- */
ALADIN_API int aladin_sgemm_strided(int M, int N, int K, const float* A, int64_t a_rs, int64_t a_cs,
                         const float* B, int64_t b_rs, int64_t b_cs, float* C, int64_t ldc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Retrieval similarity matrix  sim = img @ cap.T  at evaluation scale (5000 x 25000 x 768) on the
 * 16-bit MFMA path with a hi/lo fp16 split (3 MFMAs per product, ~fp32 accuracy so that ranks
 * agree with the reference; operand rows are kept [hi | lo] and one accumulator chain runs hi.hi, lo.hi, hi.lo).  Replaces ims.mm(caps.t()), reference alad/recall_auxiliary.py:30,51
 * and torch.mm at alad/evaluation.py:196,285.
 * workspace: aladin_sim_workspace_bytes(n_img, n_cap, D).
 * ------------------------------------------------------------------------------------------- */
ALADIN_API size_t aladin_sim_workspace_bytes(int n_img, int n_cap, int D);
ALADIN_API int aladin_sim_matrix(const float* img, int64_t img_rs, const float* cap, int64_t cap_rs,
                      int n_img, int n_cap, int D, float* sim, int64_t ld_sim,
                      void* workspace, void* stream);

/* Ranks for both retrieval directions from sim (n_img x n_cap), COCO protocol: captions
 * caps_per_img*i .. caps_per_img*i + caps_per_img-1 belong to image i.  rank = number of strictly
 * larger scores (== argsort position, reference alad/recall_auxiliary.py:34-56 and
 * alad/evaluation.py:213-223,303-308, except on exact ties).
 *   rank_i2t[n_img]: best rank among the image's captions;  top1_i2t[n_img]: argmax caption
 *   rank_t2i[n_cap]: rank of the caption's image;           top1_t2i[n_cap]: argmax image */
ALADIN_API size_t aladin_recall_workspace_bytes(int n_cap);
ALADIN_API int aladin_recall_ranks(const float* sim, int64_t ld_sim, int n_img, int n_cap, int caps_per_img,
                        int32_t* rank_i2t, int32_t* top1_i2t, int32_t* rank_t2i, int32_t* top1_t2i,
                        void* workspace, void* stream);

/* Top-k lists: for each of n_q queries the indices (out_idx, n_q x k int32, -1 past n_c) and optionally the
 * scores (out_val, may be NULL) of its k largest candidates M[q * q_stride + c * c_stride], c < n_c, in
 * descending order, ties -> lower index first.  With M = sim (n_img x n_cap), q_stride = 1, c_stride = ld_sim,
 * k = 50 it is the `top50` table t2i returns, reference alad/evaluation.py:262,309.  n_c <= 36864. */
ALADIN_API int aladin_topk(const float* M, int64_t q_stride, int64_t c_stride, int n_q, int n_c, int k, int32_t* out_idx,
                float* out_val, void* stream);

/* Fused retrieval: the same four outputs as aladin_sim_matrix + aladin_recall_ranks straight from the
 * embeddings, without ever writing the (n_img x n_cap) score matrix (reference
 * alad/recall_auxiliary.py:30-56 / alad/evaluation.py:196-223,285-308 in one pass), bit-identical to the
 * two-step path whatever the data (integer and packed-max atomics: independent of the tile order).
 * A rank is a count of DECISIONS "score > ground truth", so the GEMM runs the hi.hi third of the split
 * product only and bounds, per pair and rigorously (Cauchy-Schwarz on the two dropped segments of the
 * actual fp16 operands + the fp32 rounding of their accumulation: 2^-14 |s| covers D <= 16384, larger D takes the
 * exact path), what the rest can add.  Scores that beat their ground truth by more than the bound are counted in
 * registers; pairs the bound does not decide, and possible arg-maxima, are continued to the exact score -- through a
 * list of up to 512 per 256 x 384 tile (a second kernel first drops the arg-max candidates that the certified lower
 * bounds of all tiles rule out), a whole tile in place when it holds more.  Cost: a third of the three-product GEMM
 * when ground truths stand clear of the bulk of the scores, ~0.6 of it at Recall@1 75 / 41 %, all of it when they
 * sit deep inside; the result never depends on it.
 * aladin_retrieval_ranks_exact: every tile takes the three-product path (for A/B runs and tests).
 * aladin_retrieval_stats_offset: byte offset, inside the workspace of the last call, of int32[9]: [0] tiles continued in
 * place, [1] pairs listed, [5] listed pairs whose chains were continued, [6] tiles that ran the analysis, [7] of those,
 * tiles that overflowed their list, [8] tiles that skipped it ([2..4]: diagnostic build only).
 * The four rank / arg-max OUTPUTS are deterministic.  The STATISTICS, and the time a call takes, are not: a tile decides
 * whether to skip its analysis from what the tiles that finished before it have reported so far (relaxed loads of [6], [7]
 * while other tiles of the same launch update them), so which tiles skip -- and with it [0], [1], [5], [8] -- depends on
 * scheduling.  Callers may rely on [5] <= [1], [7] <= [6], [0] >= [8] and on nothing else. */
ALADIN_API size_t aladin_retrieval_workspace_bytes(int n_img, int n_cap, int D);
ALADIN_API size_t aladin_retrieval_stats_offset(int n_img, int n_cap, int D);
ALADIN_API int aladin_retrieval_ranks(const float* img, int64_t img_row_stride, const float* cap, int64_t cap_row_stride, int n_img,
                           int n_cap, int D, int caps_per_img, int32_t* rank_i2t, int32_t* top1_i2t, int32_t* rank_t2i,
                           int32_t* top1_t2i, void* workspace, void* stream);
ALADIN_API int aladin_retrieval_ranks_exact(const float* img, int64_t img_row_stride, const float* cap, int64_t cap_row_stride, int n_img,
                           int n_cap, int D, int caps_per_img, int32_t* rank_i2t, int32_t* top1_i2t, int32_t* rank_t2i,
                           int32_t* top1_t2i, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALADIN_HIP_H */
