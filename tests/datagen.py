"""Deterministic, libm-free synthetic inputs shared by the golden generator and the tests.

Values are produced from a counter-based integer hash (splitmix64) and exact float arithmetic
only, so the same (shape, seed) gives the same bits on every machine / numpy version.  Large
inputs (attention maps) are therefore NOT stored in the fixtures -- only their seeds are.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_u24(shape, seed):
    """uint32 array of 24-bit integers."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
        z = _splitmix64(_splitmix64(ctr))
    return (z >> np.uint64(40)).astype(np.uint32).reshape(shape)


def uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 uniform on a 2^-24 grid of [lo, hi)."""
    k = hash_u24(shape, seed).astype(np.float64)
    return (lo + (hi - lo) * (k / 16777216.0)).astype(np.float32)


def bellish(shape, seed, scale=1.0):
    """Sum of 4 uniforms, centred: a cheap bell-shaped stand-in for randn (std ~= scale)."""
    acc = np.zeros(shape, np.float64)
    for j in range(4):
        acc += hash_u24(shape, seed * 4 + j + 1000003).astype(np.float64) / 16777216.0
    return ((acc - 2.0) * (scale * 1.7320508)).astype(np.float32)


def integers(shape, seed, n):
    return hash_u24(shape, seed).astype(np.int64) % n


# ------------------------------------------------------------------ canonical test cases
def graph_case(B=4, L=196, M=256, seed=11):
    """Ingredients / attention logits with the edge cases SURVEY.md 8c asks for:
    image 0 random words + one fully clamped attention source row,
    image 1 every token the same word (n_i = 1, 38416-term edge sum),
    image 2 all tokens distinct (n_i = L, needs M >= L) or as distinct as M allows,
    image 3 five words with skewed counts + a fully clamped attn_cls row."""
    ing = integers((B, L), seed, M)
    if B > 1:
        ing[1, :] = 7 % M
    if B > 2:
        # descending, so the sorted vertex order is the reverse of the position order
        ing[2, :] = (M - 1 - np.arange(L)) if M >= L else (np.arange(L) % M)
    if B > 3:
        ing[3, :] = np.asarray([2, 2, 2, 9, 9, 40 % M, 2, 77 % M, 2, 5])[integers((L,), seed + 1, 10)]
    attn_cls = bellish((B, L), seed + 2, 1.5)
    attn = bellish((B, L, L), seed + 3, 1.5)
    attn[0, 17, :] = -2.5 - np.abs(attn[0, 17, :])       # whole source row below the clamp
    if B > 3:
        attn_cls[3, :] = -3.0 - np.abs(attn_cls[3, :])   # whole cls row below the clamp
        attn[3, 0, :] = -1.5 - np.abs(attn[3, 0, :])
    return ing.astype(np.int64), attn, attn_cls


def labelled_case(B, L, M, K, seed):
    """A labelled mini-batch for the training-trajectory fixture: image b of class c draws about two thirds of its
    tokens from the c-th block of M // K words and the rest from all words, so there is something to learn.
    -> ingredients i64 [B, L], attn f32 [B, L, L], attn_cls f32 [B, L], label i64 [B]"""
    label = integers((B,), seed, K)
    block = M // K
    own = label[:, None] * block + integers((B, L), seed + 1, block)
    anyw = integers((B, L), seed + 2, M)
    pick = integers((B, L), seed + 3, 3) < 2
    ing = np.where(pick, own, anyw).astype(np.int64)
    return ing, bellish((B, L, L), seed + 4, 1.5), bellish((B, L), seed + 5, 1.5), label.astype(np.int64)
