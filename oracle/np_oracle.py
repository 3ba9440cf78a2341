"""Independent numpy restatement of the scoring semantics -- TEST INFRASTRUCTURE ONLY.

Written as a *set definition* (dense enumeration of all sigma^h candidates per tree node with
boolean survivor masks), deliberately unlike the list/sort/early-break structure of
oracle/ipk_oracle.c (which follows ipk/src/pk_compute.cpp:42-114 statement by statement), so that
agreement between the two is evidence neither mis-states the reference.  Small k only
(sigma^k candidates are materialised).  PARITY UNPINNED: see ipk_oracle.c.
"""
import numpy as np

f32 = np.float32


def prefix_max(m):
    """window.cpp:16-27 -- sequential float32 accumulation of per-column maxima."""
    m = np.asarray(m, dtype=f32)
    best = np.zeros(m.shape[0] + 1, dtype=f32)
    acc = f32(0.0)
    colmax = m.max(axis=1)
    for j in range(m.shape[0]):
        acc = f32(acc + colmax[j])
        best[j + 1] = acc
    return best


def _node(m, best, start, j, h, eps, bits):
    """Returns (keys[sigma^h] u64, scores[sigma^h] f32, mask[sigma^h]) for S(j, h, eps)."""
    sigma = m.shape[1]
    if h == 1:
        col = m[start + j]
        return np.arange(sigma, dtype=np.uint64), col.copy(), col > f32(eps)   # pk_compute.cpp:14-26
    hl, hr = h // 2, h - h // 2
    # pk_compute.cpp:54-55 ; window.cpp:134-137 ; window.cpp:69-72
    m_right = f32(best[start + j + hl + hr] - best[start + j + hl])
    m_left = f32(best[start + j + hl] - best[start + j])
    eps_l = f32(f32(eps) - m_right)
    eps_r = f32(f32(eps) - m_left)
    kl, sl, ml = _node(m, best, start, j, hl, eps_l, bits)
    kr, sr, mr = _node(m, best, start, j + hl, hr, eps_r, bits)
    score = (sl[:, None] + sr[None, :]).astype(f32)                           # :90
    keys = (kl[:, None] << np.uint64(hr * bits)) | kr[None, :]                # :96-104
    mask = ml[:, None] & mr[None, :] & (score > f32(eps))                     # :91 (strict >)
    return keys.reshape(-1), score.reshape(-1), mask.reshape(-1)


def window(m, k, start, eps, bits, best=None):
    m = np.asarray(m, dtype=f32)
    if best is None:
        best = prefix_max(m)
    keys, scores, mask = _node(m, best, start, 0, k, f32(eps), bits)
    keys, scores = keys[mask].astype(np.uint32), scores[mask]
    o = np.argsort(keys, kind="stable")
    return keys[o], scores[o]


def explore_group(mats, k, eps, bits):
    """db_builder.cpp:629-698 (RAM mode): max per key over all windows of all matrices."""
    table = {}
    emitted = 0
    for m in mats:
        m = np.asarray(m, dtype=f32)
        best = prefix_max(m)
        for start in range(0, m.shape[0] - k + 1):
            keys, scores = window(m, k, start, eps, bits, best)
            emitted += len(keys)
            for key, s in zip(keys.tolist(), scores.tolist()):
                old = table.get(key)
                if old is None or old < s:                                    # branch_group.cpp:88-101
                    table[key] = s
    keys = np.array(sorted(table), dtype=np.uint32)
    scores = np.array([table[int(x)] for x in keys], dtype=f32)
    return keys, scores, emitted


def explore_group_pos(mats, k, eps, bits):
    """KEEP_POSITIONS flavour (branch_group.cpp:73-86): first window wins ties; returns positions too."""
    table = {}
    for m in mats:
        m = np.asarray(m, dtype=f32)
        best = prefix_max(m)
        for start in range(0, m.shape[0] - k + 1):
            keys, scores = window(m, k, start, eps, bits, best)
            for key, s in zip(keys.tolist(), scores.tolist()):
                old = table.get(key)
                if old is None or old[0] < s:
                    table[key] = (s, start)
    keys = np.array(sorted(table), dtype=np.uint32)
    return keys, np.array([table[int(x)][0] for x in keys], dtype=f32), np.array([table[int(x)][1] for x in keys], dtype=np.uint32)
