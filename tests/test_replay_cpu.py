"""The launch-replay module on CPU (no device call): program building, segment bookkeeping, the generated dispatch table's
coverage of include/msc.h, and error propagation (a recorded call whose argument check fails stops the segment and names itself)."""
import ctypes

import pytest


@pytest.fixture(scope="module")
def L():
    from multimodal_sc_amd import lib as L
    L.lib.load()
    return L


def test_dispatch_table_covers_every_int_entry_point(L):
    lib = L.lib.load()
    n = 0
    for name, (restype, argtypes, _) in L.lib.protos.items():
        fid = lib.msc_replay_fn_id(name.encode())
        if restype is ctypes.c_int and not name.startswith("msc_replay_"):
            assert fid >= 0, name
            assert lib.msc_replay_fn_nargs(fid) == len(argtypes), name
            assert len(argtypes) <= 24          # MSC_REPLAY_MAX_ARGS
            n += 1
        else:
            assert fid == -1, name               # size queries / string getters / the replay API itself are not replayable
    assert n >= 45
    assert lib.msc_replay_fn_id(b"msc_no_such_entry") == -1 and lib.msc_replay_fn_nargs(999) == -1


def test_program_segments_and_error_propagation(L):
    from multimodal_sc_amd.replay import Program, Recorder
    rec = Recorder()
    # argument validation happens before any HIP call, so these records are safe to "run" without a device:
    rec.add("msc_counter_add", [None, 1, None], [])                       # ctr == NULL -> MSC_ERR_BAD_ARG
    rec.cut("host acts here")
    rec.add("msc_linear_fwd", [None, None, None, None, None, None, 1, 1, 8, 0, 0, None], [])
    rec.add("msc_adam_step", [None, None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None, 1.0, 1, None], [])
    prog = Program(rec)
    assert prog.n_segments == 2 and prog.labels == ["host acts here", None]
    assert prog.count() == 3 and prog.count(0) == 1 and prog.count(1) == 2
    with pytest.raises(RuntimeError, match=r"record 0 \(msc_counter_add\).*bad argument"):
        prog.run(0)
    with pytest.raises(RuntimeError, match=r"record 1 \(msc_linear_fwd\)"):
        prog.run(1)
    lib = L.lib.load()
    assert lib.msc_replay_run(prog.handle, 7) == -1          # no such segment
    with pytest.raises(RuntimeError, match="not a replayable entry point"):
        bad = Recorder()
        bad.add("msc_wgrad_workspace_bytes", [1, 1, 1], [])
        Program(bad)
    # wrong argument count for the entry point is refused when the record is added
    ia, fa = (ctypes.c_int64 * 2)(), (ctypes.c_double * 2)()
    assert lib.msc_replay_add(prog.handle, lib.msc_replay_fn_id(b"msc_counter_add"), ia, fa, 2) == -1


def test_recorder_hooks_lib_call_and_restores(L):
    from multimodal_sc_amd.replay import Recorder
    with Recorder() as rec:
        assert L._REC is rec
        with pytest.raises(RuntimeError):
            L.call("msc_counter_add", None, 1, None)       # fails (NULL counter): failing calls are not recorded
        with pytest.raises(RuntimeError, match="already active"):
            Recorder().__enter__()
    assert L._REC is None and rec.recs == []


def test_recorder_ignores_opted_out_threads_and_blocking_waits(L):
    """lib._REC is process-global: the frame stager's filler thread calls lib.call beside a step.  A thread that called
    lib.no_record() is never recorded, a host-blocking msc_event_synchronize never is — and any OTHER thread is (autograd runs the
    backward of a step on its engine thread: those launches are the step)."""
    import threading
    from multimodal_sc_amd.replay import Recorder

    def helper():
        L.no_record()
        L.call("msc_set_knob", b"gemm_t128", 0)

    with Recorder() as rec:
        L.call("msc_set_knob", b"gemm_pp", 1)                       # (a harmless int-returning entry point that needs no device)
        for fn in (helper, lambda: L.call("msc_set_knob", b"conv_splitk", 1)):
            th = threading.Thread(target=fn)
            th.start()
            th.join()
        rec.add("msc_event_synchronize", [0], [])
    assert [c[0] for _, c in rec.recs] == [b"gemm_pp", b"conv_splitk"]
