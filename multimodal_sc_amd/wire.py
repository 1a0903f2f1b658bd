"""Wire format of the semantic symbols: k real symbols = k/2 complex I/Q pairs per frame, sent as
one fp32 scale + k signed bytes.  Thin host wrappers over msc_symbols_pack / msc_symbols_unpack (csrc/wire.hip)."""
import torch

from .runtime import Runtime


def pack_symbols(rt: Runtime, sym: torch.Tensor):
    """sym [B,k] (rt dtype) -> (q int8 [B,k] interleaved I/Q, scale fp32 [B])"""
    B, k = sym.shape
    q = torch.empty(B, k, dtype=torch.int8, device=rt.device)
    scale = rt.f32(B)
    rt.call("msc_symbols_pack", sym.contiguous(), q, scale, B, k, rt.dt)
    return q, scale


def unpack_symbols(rt: Runtime, q: torch.Tensor, scale: torch.Tensor):
    B, k = q.shape
    y = rt.empty(B, k)
    rt.call("msc_symbols_unpack", q, scale, y, B, k, rt.dt)
    return y


def frame_bytes(k: int) -> int:
    """bytes on the wire per frame: 4 (scale) + k (I/Q bytes)"""
    return 4 + k
