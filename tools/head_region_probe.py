#!/usr/bin/env python3
"""GPU time between the end of the forward and the first transformer-layer backward (the loss + head/codec backward region,
~50 tiny launches): if it is much longer than the kernels take, the host paces that region."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd import functional as F
from multimodal_sc_amd.model import MultimodalSC, named_config
from multimodal_sc_amd.trainer import Trainer
from oracle import model as O
cfg = named_config("C3")
m = MultimodalSC(cfg, "bf16", "cuda:0")
tr = Trainer(m)
cb = O.synthetic_batch(O.named_config("C3"), 64, seed=1)
b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in cb.items() if k != "ppo"}
b["ppo"] = {k: v.cuda() for k, v in cb["ppo"].items()}
b["noise"] = b["fading"] = None
marks = {}
orig_bwd = F.FusionLayerFn.backward
def hooked(ctx, *a):
    if "t1" in marks and "t2" not in marks:
        e = torch.cuda.Event(enable_timing=True); e.record(); marks["t2"] = e
    return orig_bwd(ctx, *a)
F.FusionLayerFn.backward = staticmethod(hooked)
orig_fwd = m.forward
def fwd(*a, **k):
    out = orig_fwd(*a, **k)
    if marks.get("arm"):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks["t1"] = e
    return out
m.forward = fwd
m.__class__.__call__ = lambda self, *a, **k: self.forward(*a, **k)
for _ in range(5):
    tr.step(b)
res = []
for _ in range(10):
    marks.clear(); marks["arm"] = True
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    tr.step(b)
    e3 = torch.cuda.Event(enable_timing=True); e3.record()
    torch.cuda.synchronize()
    res.append((e0.elapsed_time(marks["t1"]), marks["t1"].elapsed_time(marks["t2"]), marks["t2"].elapsed_time(e3)))
import statistics
for i, n in enumerate(("forward", "loss+heads backward region", "rest of backward + adam")):
    print(f"{n:32s} {statistics.median(r[i] for r in res):.3f} ms")
