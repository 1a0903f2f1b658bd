"""oracle/wire.py — CPU (NumPy) restatement of the semantic-symbol wire format (per frame: fp32 scale = amax/127, then k
signed bytes I0,Q0,I1,Q1,... with q = clamp(rne(x/scale), -127, 127)).  TEST INFRASTRUCTURE ONLY; parity unpinned by the
reference (the upstream repository has no code).  Same fp32 operations as csrc/wire.hip (division, round-half-even)."""
import numpy as np


def pack(x):
    x = np.asarray(x, dtype=np.float32)
    amax = np.abs(x).max(axis=1)
    scale = np.where(amax > 0, amax / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    q = np.clip(np.rint(x / scale[:, None]), -127, 127).astype(np.int8)
    return q, scale


def unpack(q, scale):
    return q.astype(np.float32) * np.asarray(scale, dtype=np.float32)[:, None]
