"""Recording and C-side replay of one training step's launch list (csrc/replay.hip; include/msc.h "Launch replay").

An eager step is a fixed sequence of C-ABI calls; Python spends ~12 us on each (autograd node, marshalling, ctypes), which
makes the step host-bound below ~40 frames per GPU.  Recorder hooks lib.call during ONE eager step and keeps, per call, the
entry point and the raw argument values (device addresses, sizes, stream handles, events).  Program hands them to the C side,
which re-issues a whole segment from one ctypes call.  Conditions the Trainer guarantees:
  * every per-step buffer comes from the step's bump arena (runtime.Arena): the same addresses every step; buffers that do not
    (the caller's input batch, anything torch allocated) are kept alive by the program for as long as it exists;
  * step / noise-draw numbers are device-side counters (Runtime.host_counters = False), persistent workspaces did not grow
    during the recorded step, and every cross-stream dependency of the step is an msc_stream_wait_stream call (Runtime.stream_wait)
    — including those torch.autograd would otherwise supply.
"""
from __future__ import annotations

import ctypes

from . import lib as L


class Recorder:
    """context manager: records every lib.call between __enter__ and __exit__; cut(label) ends a segment (Python acts there)"""

    def __init__(self):
        self.recs = []          # (name, converted args) | ("__cut__", label)
        self.keep = []          # tensors seen as arguments: alive as long as the program
        self.labels = []

    def add(self, name, conv, raw):
        self.recs.append((name, conv))
        for a in raw:
            if hasattr(a, "data_ptr"):
                self.keep.append(a)

    def cut(self, label):
        self.recs.append(("__cut__", label))

    def __enter__(self):
        if L._REC is not None:
            raise RuntimeError("a launch recording is already active")
        L._REC = self
        return self

    def __exit__(self, *exc):
        L._REC = None
        return False


class Program:
    """the recorded launch list on the C side; run(segment) re-issues it"""

    def __init__(self, rec: Recorder):
        lib = L.lib.load()
        self._lib = lib
        self.handle = int(lib.msc_replay_create())
        if not self.handle:
            raise RuntimeError("msc_replay_create failed")
        self.keep = rec.keep
        self.labels = []            # labels[i] = the host action after segment i (None after the last)
        self.names = []
        ids = {}
        for name, conv in rec.recs:
            if name == "__cut__":
                lib.msc_replay_cut(self.handle)
                self.labels.append(conv)
                continue
            fid = ids.get(name)
            if fid is None:
                fid = ids[name] = lib.msc_replay_fn_id(name.encode())
                if fid < 0:
                    raise RuntimeError(f"{name} is not a replayable entry point")
            argtypes = L.lib.protos[name][1]
            n = len(conv)
            ia, fa = (ctypes.c_int64 * n)(), (ctypes.c_double * n)()
            for j, (v, t) in enumerate(zip(conv, argtypes)):
                if t is ctypes.c_float:
                    fa[j] = float(v)
                elif v is not None:
                    iv = int(v)
                    ia[j] = iv - (1 << 64) if iv >= (1 << 63) else iv
            rc = lib.msc_replay_add(self.handle, fid, ia, fa, n)
            if rc != 0:
                raise RuntimeError(f"msc_replay_add({name}) failed: code {rc}")
            self.names.append(name)
        self.labels.append(None)
        self.n_segments = len(self.labels)

    def run(self, segment: int = -1):
        rc = self._lib.msc_replay_run(self.handle, segment)
        if rc != 0:
            i = int(self._lib.msc_replay_last_failed(self.handle))
            name = self.names[i] if 0 <= i < len(self.names) else "?"
            raise RuntimeError(f"replay of segment {segment} failed at record {i} ({name}): {self._lib.msc_strerror(rc).decode()} (code {rc})")

    def count(self, segment: int = -1) -> int:
        return int(self._lib.msc_replay_count(self.handle, segment))

    def __del__(self):
        try:
            if self.handle:
                self._lib.msc_replay_destroy(self.handle)
                self.handle = 0
        except Exception:  # noqa: BLE001 — interpreter shutdown
            pass
