"""oracle/rl.py — CPU (NumPy) restatement of the PPO-loop pieces around the hot path: generalised advantage estimation
(Schulman et al. 2016, eq. 16), advantage normalisation, inverse-CDF categorical sampling driven by oracle/philox.py.
TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline); parity unpinned by the reference (the upstream repository
has no code, SURVEY.md finding zero).  fp32 arithmetic in the same operation order as csrc/rl.hip (no fused multiply-add)."""
import numpy as np

from .philox import philox4x32_10

f32 = np.float32


def gae(rewards, values, dones, gamma, lam):
    """rewards/dones [T,N], values [T+1,N] -> adv, ret [T,N] (float32)"""
    r, v, d = (np.asarray(a, dtype=f32) for a in (rewards, values, dones))
    T, N = r.shape
    g, l = f32(gamma), f32(lam)
    adv = np.zeros((T, N), f32)
    a = np.zeros(N, f32)
    for t in range(T - 1, -1, -1):
        nd = f32(1.0) - d[t]
        delta = (r[t] + (g * v[t + 1]) * nd) - v[t]
        a = delta + ((g * l) * nd) * a
        adv[t] = a
    return adv, adv + v[:-1]


def normalize(x, eps=1e-8):
    x = np.asarray(x, dtype=np.float64)
    return ((x - x.mean()) / (x.std(ddof=1) + eps)).astype(f32)


def sample_actions(logits, seed, step, return_margin=False):
    """logits [B,A] float32 -> (action int64 [B], logp float32 [B]); u from Philox counter (row, step_lo, step_hi, 2).
    return_margin: also the distance of u to the nearest CDF boundary (a sample closer than a few fp32 ulps to a boundary
    may legitimately land on the neighbouring action when exp() differs in the last bit)."""
    lg = np.asarray(logits, dtype=f32)
    B, A = lg.shape
    z = np.zeros(B, np.uint32)
    r0 = philox4x32_10(np.arange(B, dtype=np.uint32), z + np.uint32(step & 0xFFFFFFFF), z + np.uint32(step >> 32), z + np.uint32(2),
                       seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)[0]
    u = (r0.astype(f32) + f32(0.5)) * f32(2.3283064365386963e-10)
    mx = lg.max(1, keepdims=True)
    e = np.exp(lg - mx, dtype=f32)
    zsum = np.zeros(B, f32)
    for a in range(A):
        zsum = zsum + e[:, a]
    c = np.zeros(B, f32)
    pick = np.full(B, A - 1, np.int64)
    found = np.zeros(B, bool)
    margin = np.full(B, np.inf, f32)
    for a in range(A):
        c = c + e[:, a] / zsum
        margin = np.minimum(margin, np.abs(c - u))
        hit = (c > u) & ~found
        pick[hit] = a
        found |= hit
    logp = (lg[np.arange(B), pick] - mx[:, 0]) - np.log(zsum, dtype=f32)
    if return_margin:
        return pick, logp.astype(f32), margin
    return pick, logp.astype(f32)


def topk_rows(x, k):
    """x [rows, n] -> (idx int64 [rows,k], val float32 [rows,k]).  Order: value descending, index ascending on exact ties
    (0.0 and -0.0 tie), NaN after everything else (also after -inf) — a stable lexicographic sort on (isnan, -value, index);
    k = 1 is arg-max with the lowest index on ties (oracle.model.select_action on finite logits)."""
    x = np.asarray(x, dtype=f32)
    rows, n = x.shape
    nan = np.isnan(x)
    negv = np.where(nan, f32(0), -x) + f32(0)            # + 0 turns -0.0 into 0.0 so the two zeros compare equal as keys
    j = np.broadcast_to(np.arange(n), x.shape)
    order = np.lexsort((j, negv, nan), axis=-1)[:, :k]   # last key is the primary one
    return order.astype(np.int64), np.take_along_axis(x, order, axis=1)
