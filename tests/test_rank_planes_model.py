"""The arithmetic k_walk_rows decodes index rows with (yunikorn-k8shim_amd/csrc/engine/kernels.hip.h: k_dim_sort's rank planes,
k_walk_rows' majority ripple), restated with numpy integers and held against the mask table it replaces: for every 64-node word,
pmask[j] = { node : valid and rank(node) >= j } with rank = position in the word's ascending (free, lane) order; bit-sliced, plane k
holds bit k of r' = valid ? rank + 1 : 0 and mask(j) = (r' > j) = gt after gt = maj(gt, r_k, not j_k) for k = 0..6 (gt starts 0)."""
import numpy as np

M64 = (1 << 64) - 1


def maj(a, b, c):
    return (a & b) | (a & c) | (b & c)


def word_tables(free, valid):
    order = sorted(range(64), key=lambda i: (free[i] if valid[i] else -(1 << 63), i))
    rank = [0] * 64
    for pos, lane in enumerate(order):
        rank[lane] = pos
    pmask = [sum(1 << i for i in range(64) if valid[i] and rank[i] >= j) for j in range(65)]
    planes = [sum(1 << i for i in range(64) if ((rank[i] + 1 if valid[i] else 0) >> k) & 1) for k in range(7)]
    sfree = [free[i] if valid[i] else -(1 << 63) for i in order]
    return pmask, planes, sfree


def decode(planes, j):
    gt = 0
    for k in range(7):
        not_jk = 0 if (j >> k) & 1 else M64
        gt = maj(gt, planes[k], not_jk)
    return gt


def test_majority_ripple_equals_the_mask_table():
    rng = np.random.default_rng(12)
    for trial in range(60):
        n_valid = [64, 64, 1, 0, 17, 63][trial % 6]
        valid = [i < n_valid for i in range(64)]
        if trial % 4 == 3:
            rng.shuffle(valid)
        spread = [1, 3, 1 << 40][trial % 3]  # many ties, some ties, none
        free = [int(v) for v in rng.integers(-spread, spread + 1, size=64)]
        pmask, planes, sfree = word_tables(free, valid)
        assert pmask[0] == sum(1 << i for i in range(64) if valid[i]) and pmask[64] == 0
        for j in range(65):
            assert decode(planes, j) == pmask[j], (trial, j)
        # the index byte k_dim_walk stores for a request value v: the number of entries of the sorted list below v
        for v in sorted(set(free))[:8] + [min(free) - 1, max(free) + 1]:
            j = sum(1 for x in sfree if x < v)
            want = sum(1 << i for i in range(64) if valid[i] and free[i] >= v)
            assert decode(planes, j) == want == pmask[j]
        assert decode(planes, 65) == 0 and decode(planes, 127) == 0  # a stray byte beyond 64 decodes to the empty mask
