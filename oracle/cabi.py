"""ctypes bindings of oracle/liboracle.so (schemanet_oracle.c).  TEST INFRASTRUCTURE ONLY.

numpy in, numpy out.  Builds the library on first use if it is missing (gcc, seconds).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.sno_instance_v.restype = ctypes.c_int64
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def codebook_norms(cb):
    cb = _f32(cb)
    out = np.empty(cb.shape[0], np.float64)
    lib().sno_codebook_norms(_p(cb, _f32p), cb.shape[0], cb.shape[1], _p(out, _f64p))
    return out


def assign_words(x, cb, return_score=False):
    """x [n_tok, D] f32, cb [M, D] f32 -> idx i64 [n_tok] (and fp64 best score)."""
    x, cb = _f32(x), _f32(cb)
    assert x.ndim == 2 and cb.ndim == 2 and x.shape[1] == cb.shape[1]
    idx = np.empty(x.shape[0], np.int64)
    score = np.empty(x.shape[0], np.float64)
    lib().sno_assign_words(_p(x, _f32p), ctypes.c_int64(x.shape[0]), _p(cb, _f32p),
                           cb.shape[0], cb.shape[1], _p(idx, _i64p), _p(score, _f64p))
    return (idx, score) if return_score else idx


def kmeans_update(x, ids, K):
    """-> (sums f32 [K, D] added in observation order, counts i64 [K])   (SciPy's update_cluster_means before the division)"""
    x, ids = _f32(x), _i64(ids)
    sums = np.empty((K, x.shape[1]), np.float32)
    counts = np.empty(K, np.int64)
    lib().sno_kmeans_update(_p(x, _f32p), ctypes.c_int64(x.shape[0]), x.shape[1], _p(ids, _i64p), K, _p(sums, _f32p), _p(counts, _i64p))
    return sums, counts


def kmeans_distances(x, ids, centres):
    x, ids, centres = _f32(x), _i64(ids), _f32(centres)
    dist = np.empty(x.shape[0], np.float64)
    lib().sno_kmeans_distances(_p(x, _f32p), ctypes.c_int64(x.shape[0]), x.shape[1], _p(ids, _i64p), _p(centres, _f32p),
                               centres.shape[0], _p(dist, _f64p))
    return dist


def instance_v(ing, attn_cls, w, mean=True):
    """-> ids [sum n], attrs2 [sum n, 2], weights [sum n], num_v [B]  (concatenated like the
    reference's return value)."""
    ing, attn_cls = _i64(ing), _f32(attn_cls)
    w = _f32(w).reshape(-1)
    B, L = ing.shape
    ids = np.empty(B * L, np.int64)
    attrs2 = np.empty((B * L, 2), np.float32)
    weights = np.empty(B * L, np.float32)
    num_v = np.empty(B, np.int64)
    tot = lib().sno_instance_v(_p(ing, _i64p), _p(attn_cls, _f32p), B, L, int(mean), _p(w, _f32p),
                               _p(ids, _i64p), _p(attrs2, _f32p), _p(weights, _f32p), _p(num_v, _i64p))
    return ids[:tot].copy(), attrs2[:tot].copy(), weights[:tot].copy(), num_v


def instance_e(ing, attn, geo, dicts, w, mean=True, remove_self_loop=False):
    """dicts: list of B {word: row}.  -> (list of [n_b, n_b, 2], list of [n_b, n_b])."""
    ing, attn, geo = _i64(ing), _f32(attn), _f32(geo)
    w = _f32(w).reshape(-1)
    B, L = ing.shape
    keys, vals, off, ln = [], [], [], []
    o = 0
    for d in dicts:
        ks = sorted(d.keys())
        keys += ks
        vals += [d[k] for k in ks]
        off.append(o)
        ln.append(len(ks))
        o += len(ks)
    keys = np.asarray(keys + [0], np.int64)
    vals = np.asarray(vals + [0], np.int64)
    off, ln = np.asarray(off, np.int64), np.asarray(ln, np.int64)
    tot = int((ln * ln).sum())
    e2 = np.empty((max(tot, 1), 2), np.float32)
    e = np.empty(max(tot, 1), np.float32)
    lib().sno_instance_e(_p(ing, _i64p), _p(attn, _f32p), _p(geo, _f32p), B, L, int(mean),
                         int(remove_self_loop), _p(w, _f32p), _p(keys, _i64p), _p(vals, _i64p),
                         _p(off, _i64p), _p(ln, _i64p), _p(e2, _f32p), _p(e, _f32p))
    out2, out = [], []
    o = 0
    for n in ln.tolist():
        out2.append(e2[o:o + n * n].reshape(n, n, 2).copy())
        out.append(e[o:o + n * n].reshape(n, n).copy())
        o += n * n
    return out2, out


def v_attr(ing, attn_cls, n_vertices, mean=True, ingredients_only=False):
    ing, attn_cls = _i64(ing), _f32(attn_cls)
    B, L = ing.shape
    attr = np.empty((B, n_vertices, 2), np.float32)
    lib().sno_v_attr(_p(ing, _i64p), _p(attn_cls, _f32p), B, L, n_vertices, int(mean),
                     int(ingredients_only), _p(attr, _f32p))
    return attr


def dicts_to_slot_table(class_dicts, m_tab=None):
    """K {word: slot} dicts -> dense int32 [K, Mtab] table (-1 = word not in class)."""
    mx = max((max(d.keys()) for d in class_dicts if d), default=-1) + 1
    m_tab = max(m_tab or 0, mx, 1)
    tab = np.full((len(class_dicts), m_tab), -1, np.int32)
    for k, d in enumerate(class_dicts):
        for word, slot in d.items():
            tab[k, word] = slot
    return tab


def feat_to_e(ing, attn, geo, class_slot, label, n_max, mean=True):
    ing, attn, geo = _i64(ing), _f32(attn), _f32(geo)
    class_slot = np.ascontiguousarray(class_slot, np.int32)
    label = _i64(label)
    B, L = ing.shape
    attr = np.empty((B, n_max, n_max, 2), np.float32)
    lib().sno_feat_to_e(_p(ing, _i64p), _p(attn, _f32p), _p(geo, _f32p), B, L,
                        _p(class_slot, _i32p), class_slot.shape[0], class_slot.shape[1],
                        _p(label, _i64p), n_max, int(mean), _p(attr, _f32p))
    return attr
