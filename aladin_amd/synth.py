"""Deterministic, version-independent synthetic inputs for tests, goldens and bench.

No dataset or checkpoint is reachable offline, so every input in this repo comes from
the counter-based generator below (SplitMix64 on an index, then Box-Muller).  It does not
depend on torch/numpy RNG stream stability, so the golden fixtures under ``tests/golden``
(made once from the reference, see ``tests/golden/make_golden.py``) stay reproducible.

Shapes follow the reference's data conventions:
  image sets    (B, R, D)  R <= 34 regions, ``im_len`` = number of boxes
  caption seqs  (B, T, D)  T <= 50 tokens,  ``s_len``  = tokens incl. CLS and SEP
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def uniform(shape, seed):
    """float64 uniforms in (0, 1), a pure function of (index, seed)."""
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        bits = _splitmix64(_splitmix64(idx))
    u = ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def normal(shape, seed):
    """float32 ~N(0,1) via Box-Muller on two independent uniform streams."""
    u1 = uniform(shape, 2 * seed + 1)
    u2 = uniform(shape, 2 * seed + 2)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32)


def integers(shape, lo, hi, seed):
    """int64 uniform in [lo, hi] inclusive."""
    u = uniform(shape, seed)
    return (lo + np.floor(u * (hi - lo + 1))).astype(np.int64).clip(lo, hi)


def alignment_batch(B, R=34, T=50, D=768, seed=1234, ragged=False, Bc=None):
    """(im, s, im_len, s_len) as in SURVEY.md section 8(d) configs 1/2.

    ``ragged`` draws im_len ~ U{10..R} and s_len ~ U{6..T} and pins one sample of each to the
    maximum, mirroring the encoder which slices sets to the batch maximum
    (reference alad/alad_model.py:148-150,174-175).
    """
    Bc = B if Bc is None else Bc
    im = normal((B, R, D), seed)
    s = normal((Bc, T, D), seed + 4444)
    if ragged:
        im_len = integers((B,), min(10, R), R, seed + 17)
        s_len = integers((Bc,), min(6, T), T, seed + 29)
        im_len[int(seed) % B] = R
        s_len[int(seed + 1) % Bc] = T
    else:
        im_len = np.full((B,), R, dtype=np.int64)
        s_len = np.full((Bc,), T, dtype=np.int64)
    return im, s, [int(v) for v in im_len], [int(v) for v in s_len]


def structured_alignment_batch(B, R=34, T=50, D=768, seed=77, noise=1.0, ragged=True):
    """Like alignment_batch but caption i correlates with image i (scores are not exchangeable,
    so diagonals dominate and hinge terms are sparse as in a trained model)."""
    im, s, im_len, s_len = alignment_batch(B, R, T, D, seed, ragged)
    base = normal((B, 1, D), seed + 99)
    im = (im * noise + base).astype(np.float32)
    s = (s * noise + base).astype(np.float32)
    return im, s, im_len, s_len


def global_embeddings(B, D=768, seed=4321, noise=1.0):
    """Unit-norm matching-head embeddings (reference alad/alad_model.py:240-241)."""
    base = normal((B, D), seed)
    img = base + noise * normal((B, D), seed + 1)
    cap = base + noise * normal((B, D), seed + 2)
    img /= np.linalg.norm(img, axis=1, keepdims=True)
    cap /= np.linalg.norm(cap, axis=1, keepdims=True)
    return img.astype(np.float32), cap.astype(np.float32)


def retrieval_embeddings(n_img, D=768, seed=2024, sigma=2.0, caps_per_img=5):
    """COCO-protocol retrieval rows.

    Returns (img_rows, cap_rows), both (caps_per_img*n_img, D) float32 unit-norm, image rows
    repeated caps_per_img times as ``encode_data`` lays them out (reference
    alad/evaluation.py:119-128), cap[k] = normalize(img[k//5] + sigma*normalize(eps_k)).
    """
    img = normal((n_img, D), seed)
    img /= np.linalg.norm(img, axis=1, keepdims=True)
    eps = normal((n_img * caps_per_img, D), seed + 1)
    eps /= np.linalg.norm(eps, axis=1, keepdims=True)
    cap = np.repeat(img, caps_per_img, axis=0) + sigma * eps
    cap /= np.linalg.norm(cap, axis=1, keepdims=True)
    return np.repeat(img, caps_per_img, axis=0).astype(np.float32), cap.astype(np.float32)


def eval_sets(n_img, D=64, seed=31, L=71, base_weight=0.25, sigma=3.0, img_len_range=(12, 34), cap_len_range=(7, 30),
              n_full=0):
    """(N, 71, D) zero-padded sets as ``encode_data`` lays them out (reference
    alad/evaluation.py:98-99,119-128): N = 5*n_img rows, image rows repeated 5x, slot 0 overwritten
    with the global (matching-head) embedding.  Returns (images, captions, img_len, cap_len).
    ``n_full`` images (spread over the set) are given the full length L: they have NO masked region, so
    the max over regions of alad/loss.py:124 does not see the zero fill -- the edge the trimmed grid must keep."""
    N = 5 * n_img
    img_len = integers((n_img,), img_len_range[0], img_len_range[1], seed + 1)
    cap_len = integers((N,), cap_len_range[0], cap_len_range[1], seed + 2)
    for k in range(n_full):
        img_len[(k * n_img) // n_full + (seed % max(n_img // n_full, 1))] = L
    base = base_weight * normal((n_img, 1, D), seed + 3)
    reg = normal((n_img, L, D), seed + 4) + base
    tok = normal((N, L, D), seed + 5) + np.repeat(base, 5, axis=0)
    for i in range(n_img):
        reg[i, img_len[i]:] = 0
    for k in range(N):
        tok[k, cap_len[k]:] = 0
    g_img, g_cap = retrieval_embeddings(n_img, D, seed + 6, sigma=sigma)
    images = np.repeat(reg, 5, axis=0).astype(np.float32)
    captions = tok.astype(np.float32)
    images[:, 0, :] = g_img
    captions[:, 0, :] = g_cap
    return images, captions, [int(v) for v in np.repeat(img_len, 5)], [int(v) for v in cap_len]


def encoder_batches(n_img=50, D=64, seed=91, batch=32, max_regions=40, max_tokens=30):
    """What an encoder hands `encode_data` batch by batch (reference alad/evaluation.py:104-130): for every batch the
    7-tuple of forward_emb -- global embeddings (b, D), sets (S_batch, b, D) sliced to THAT batch's longest sample, length
    lists -- over N = 5 * n_img (image, caption) rows in COCO order (each image repeated for its 5 captions).
    -> list of dicts with numpy arrays; a pure function of its arguments."""
    N = 5 * n_img
    img_len = np.repeat(integers((n_img,), 8, max_regions, seed + 1), 5)
    cap_len = integers((N,), 6, max_tokens, seed + 2)
    base = 0.3 * normal((n_img, 1, D), seed + 3)
    reg = np.repeat(normal((n_img, max_regions, D), seed + 4) + base, 5, axis=0)
    tok = normal((N, max_tokens, D), seed + 5) + np.repeat(base, 5, axis=0)
    g_img, g_cap = retrieval_embeddings(n_img, D, seed + 6, sigma=2.5)
    out = []
    for k0 in range(0, N, batch):
        k1 = min(N, k0 + batch)
        il, cl = [int(v) for v in img_len[k0:k1]], [int(v) for v in cap_len[k0:k1]]
        i_set = reg[k0:k1, :max(il)].astype(np.float32).copy()
        c_seq = tok[k0:k1, :max(cl)].astype(np.float32).copy()
        for r, n_ in enumerate(il):
            i_set[r, n_:] = 0
        for r, n_ in enumerate(cl):
            c_seq[r, n_:] = 0
        i_set /= np.maximum(np.linalg.norm(i_set, axis=2, keepdims=True), 1e-12)       # the encoder F.normalize's its sets
        c_seq /= np.maximum(np.linalg.norm(c_seq, axis=2, keepdims=True), 1e-12)
        out.append(dict(img_glob=g_img[k0:k1], cap_glob=g_cap[k0:k1], img_set=np.ascontiguousarray(i_set.transpose(1, 0, 2)),
                        cap_seq=np.ascontiguousarray(c_seq.transpose(1, 0, 2)), img_len=il, cap_len=cl))
    return out


def module_parameters(named_shapes, seed, scale=0.03):
    """Deterministic values for a module's parameters, keyed by name: {name: float32 array}.  Matrices ~N(0, scale^2),
    biases ~N(0, (scale/3)^2), LayerNorm gains (names ending in 'norm1.weight' / 'norm2.weight' / 'norm.weight') 1 + 0.1 N(0,1).
    Used to give the matching head identical weights in the golden generator and in the tests."""
    out = {}
    for k, (name, shape) in enumerate(named_shapes):
        z = normal(tuple(shape), seed + 131 * k)
        if name.endswith('weight') and 'norm' in name.split('.')[-2].lower():
            out[name] = (1.0 + 0.1 * z).astype(np.float32)
        elif name.endswith('bias'):
            out[name] = (scale / 3.0 * z).astype(np.float32)
        else:
            out[name] = (scale * z).astype(np.float32)
    return out


def checksum(a):
    """Order-sensitive float64 checksum used to pin the generator inside golden files."""
    a = np.asarray(a, dtype=np.float64).ravel()
    w = 1.0 + (np.arange(a.size) % 97) / 97.0
    return float(np.sum(a * w))
