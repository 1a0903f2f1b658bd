import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def knob():
    """Set a planner knob (msc_set_knob: csrc/gemm_plan.h `Knobs`) for the duration of one test; restored afterwards.
    knob("gemm_tile", "256x128") accepts the BMxBN spelling of the old MSC_GEMM_TILE environment variable."""
    from multimodal_sc_amd import lib as L
    saved = {}

    def _set(name, value):
        if name == "gemm_tile" and isinstance(value, str):
            bm, bn = value.split("x")
            value = int(bm) * 1000 + int(bn)
        key = name.encode()
        if name not in saved:
            saved[name] = L.lib.msc_get_knob(key)
        assert L.lib.msc_set_knob(key, int(value)) == 0, name

    yield _set
    for name, v in saved.items():
        L.lib.msc_set_knob(name.encode(), v)


def last_gemm():
    """(kind, bm, bn, threads, K slices, xcd_split, cls_interleave, extended epilogue) of the most recent contraction launch"""
    import ctypes
    from multimodal_sc_amd import lib as L
    info = (ctypes.c_int * 8)()
    assert L.lib.msc_debug_last_gemm(ctypes.cast(info, ctypes.c_void_p)) == 0
    return tuple(info)
