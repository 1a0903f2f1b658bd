"""oracle/philox.py — CPU (numpy) restatement of the counter-based generator the Channel kernel uses in
throughput mode.  TEST INFRASTRUCTURE ONLY (see oracle/model.py header); parity unpinned by the reference
(the upstream repository has no code).  Follows the published Philox4x32-10 definition (Salmon et al., SC'11:
multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 / 0xBB67AE85) and Box-Muller.
Known-answer vector from the Random123 distribution: counter=0, key=0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_normal(nquads: int, seed: int, stream_id: int, draw: int = 0) -> np.ndarray:
    """-> float32 [nquads*4]: block i gives 4 normals (two Box-Muller pairs), as msc_philox_normal does.
    key = seed (lo, hi); counter = (i lo, i hi, stream_id + 2*draw_hi, draw_lo): the draw counter of the Channel layer."""
    idx = np.arange(nquads, dtype=np.uint64)
    z = np.zeros(nquads, dtype=np.uint32)
    c2 = np.uint32((stream_id + 2 * (draw >> 32)) & 0xFFFFFFFF)
    r = philox4x32_10((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                      z + c2, z + np.uint32(draw & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    inv = np.float32(2.3283064365386963e-10)
    out = np.empty((nquads, 4), dtype=np.float32)
    for i in range(2):
        u1 = (r[2 * i].astype(np.float32) + np.float32(0.5)) * inv
        u2 = (r[2 * i + 1].astype(np.float32) + np.float32(0.5)) * inv
        rad = np.sqrt(np.float32(-2.0) * np.log(u1))
        ang = np.float32(6.283185307179586) * u2
        out[:, 2 * i] = rad * np.cos(ang)
        out[:, 2 * i + 1] = rad * np.sin(ang)
    return out.reshape(-1)
