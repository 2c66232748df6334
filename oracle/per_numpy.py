"""ORACLE (test infrastructure, not product code).

Numpy restatement of the reference's PrioritizedReplayBuffer arithmetic
(/root/reference/src/mctsq_config_dqn_per.py:238-321), in two forms:

1. `reference_sample` - the reference formulation itself: p = prio**alpha (fp64), p /= p.sum(),
   indices = inverse CDF of sequential fp64 cumsum with searchsorted(side='right') (what
   np.random.choice(N, B, p=p) does with the uniforms u it draws), w = (N p[idx])**-beta / max.
   PINNED bit-exact (indices AND weights) against tests/golden/per_cases.npz, which was produced
   by running the reference class itself (tests/golden/gen_golden.py).

2. `fixed_point_*` - the order-independent re-formulation the HIP sum-tree implements:
   p from a deterministic pow (`pow_det`, only IEEE +,-,*,/ in a fixed order, so numpy and the
   GPU produce identical bits), quantised to integer masses m_i = rint(p_i * 2**S) (uint64), and
   an exact integer inverse CDF: idx = first i with C_i > floor(k*M / 2**53), u = k / 2**53.
   Integer addition is associative, so a GPU sum-tree and this sequential cumsum agree
   bit-for-bit by construction.  It is pinned to form 1 (identical indices, weights within one
   fp32 ulp) on every golden case; the two can differ only for a draw that lies within ~2**-34
   (relative) of a CDF bin edge, where numpy's own result already depends on its pow build.

Logical order: index 0 is the OLDEST entry (deque(maxlen) drops from the left, :245-246).
"""
from __future__ import annotations

import math

import numpy as np

# ---------------------------------------------------------------- form 1: the reference


def reference_sample(prio, u, alpha=0.6, beta=0.4):
    """prio fp64[N] in logical (oldest-first) order, u fp64[B] uniforms in [0,1).
    Returns (idx int64[B], weights fp32[B]).  dqn_per.py:283-301."""
    prio = np.asarray(prio, dtype=np.float64)
    p = prio ** alpha
    p /= p.sum()
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    idx = cdf.searchsorted(u, side="right")
    w = (len(prio) * p[idx]) ** (-beta)
    w /= w.max()
    return idx.astype(np.int64), w.astype(np.float32)


class RefPER:
    """List-based restatement of push / update_priorities / max-priority rule (:249-267,:307-314)."""

    def __init__(self, capacity, alpha=0.6):
        self.capacity, self.alpha = capacity, alpha
        self.items, self.prio = [], []

    def push(self, item):
        mx = max(self.prio) if self.prio else 1.0          # true current max, may decrease
        self.items.append(item); self.prio.append(mx)
        if len(self.items) > self.capacity:                 # deque(maxlen): evict the oldest
            self.items.pop(0); self.prio.pop(0)

    def update_priorities(self, idx, pr):
        for i, p in zip(idx, pr):                           # sequential: last duplicate wins
            self.prio[int(i)] = float(p)

    def __len__(self):
        return len(self.items)


# ------------------------------------------------------- form 2: deterministic fixed point

LN2_HI = 6.93147180369123816490e-01      # fdlibm split of ln 2: low 32 bits of HI are zero
LN2_LO = 1.90821492927058770002e-10
INV_LN2 = 1.44269504088896338700e+00
SQRT_HALF = 0.70710678118654752440
# 1/(2k+1), k = 11..0, for atanh series ln(m) = 2 s (1 + s^2/3 + s^4/5 + ...)
_LOG_C = [1.0 / (2 * k + 1) for k in range(11, -1, -1)]
# 1/k!, k = 13..0 for exp(r), |r| <= ln2/2
_EXP_C = [1.0 / math.factorial(k) for k in range(13, -1, -1)]


def pow_det(x, a):
    """x**a for x >= 0 (fp64) using only IEEE-754 basic operations in a fixed order - the same
    sequence mctsq_pow_det() in mcts_q_amd/csrc/per_kernels.hip executes (contraction off), so
    results are bit-identical on CPU and GPU.  Accuracy ~2-4 ulp (not correctly rounded)."""
    x = np.asarray(x, dtype=np.float64)
    a = np.float64(a)
    m, e = np.frexp(x)                       # x = m 2^e, m in [0.5,1)
    lo = m < SQRT_HALF
    m = np.where(lo, m * 2.0, m)
    e = np.where(lo, e - 1, e).astype(np.float64)
    s = (m - 1.0) / (m + 1.0)
    s2 = s * s
    poly = np.full_like(s, _LOG_C[0])
    for c in _LOG_C[1:]:
        poly = poly * s2 + c
    lnm = (2.0 * s) * poly
    L = e * LN2_HI + (e * LN2_LO + lnm)
    y = a * L
    n = np.rint(y * INV_LN2)
    r = (y - n * LN2_HI) - n * LN2_LO
    ex = np.full_like(r, _EXP_C[0])
    for c in _EXP_C[1:]:
        ex = ex * r + c
    n = np.clip(n, -2000, 2000)
    out = np.ldexp(ex, n.astype(np.int64))
    if a == 0.0:
        return np.ones_like(x)
    return np.where(x == 0.0, 0.0 if a > 0 else np.inf, out)


def mass_shift(capacity):
    """S such that capacity * 2**(S+8) < 2**63: masses of priorities**alpha up to 256 never
    overflow the uint64 root.  Same formula as mctsq_per_mass_shift() in the C ABI."""
    bits = max(1, int(capacity - 1).bit_length())
    return min(44, 62 - bits - 8)


def masses(prio, alpha, shift):
    """uint64 integer masses m_i = clamp(rint(pow_det(prio_i, alpha) * 2**S), [p>0], 2**(S+8))."""
    p = pow_det(prio, alpha)
    m = np.rint(np.minimum(p, 256.0) * np.float64(2.0 ** shift))
    m = np.where((p > 0) & (m < 1), 1.0, m)
    return m.astype(np.uint64), p


def fixed_point_sample(prio, u, alpha=0.6, beta=0.4, capacity=None):
    """Exact-integer inverse CDF over logical order + IS weights.  Returns (idx, w fp32, w fp64)."""
    prio = np.asarray(prio, dtype=np.float64)
    n = len(prio)
    shift = mass_shift(capacity if capacity is not None else n)
    m, p = masses(prio, alpha, shift)
    cum = np.cumsum(m, dtype=np.uint64)
    total = int(cum[-1])
    k = np.rint(np.asarray(u, dtype=np.float64) * 2.0 ** 53).astype(np.int64)   # u = k / 2^53 exactly
    tgt = np.array([(int(ki) * total) >> 53 for ki in k], dtype=np.uint64)
    idx = cum.searchsorted(tgt, side="right").astype(np.int64)     # first i with C_i > t
    # N * p_i / sum(p), with sum(p) := M / 2^S and p_i the clamped value the masses were made from
    ratio = (np.float64(n) * np.minimum(p[idx], 256.0)) * np.float64(2.0 ** shift) / np.float64(total)
    w64 = pow_det(ratio, -beta)
    w64 = w64 / w64.max()
    return idx, w64.astype(np.float32), w64
