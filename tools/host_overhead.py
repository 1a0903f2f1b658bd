#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one training step (no sync) vs the GPU to run it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.model import MultimodalSC, named_config
from multimodal_sc_amd.trainer import Trainer
from multimodal_sc_amd.synthetic import synthetic_batch
cfg = named_config("C3")
m = MultimodalSC(cfg, "bf16", "cuda:0")
tr = Trainer(m)
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
cb = synthetic_batch(cfg, B, seed=1)
b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in cb.items() if k != "ppo"}
b["ppo"] = {k: v.cuda() for k, v in cb["ppo"].items()}
b["noise"] = b["fading"] = None
for _ in range(5):          # (replay mode: two eager warm-ups + the recorded step come first)
    tr.step(b)
torch.cuda.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(10):
    t = time.perf_counter()
    tr.step(b)
    enq.append(time.perf_counter() - t)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 10
print(f"[{B} frames, {'launch replay' if tr.use_replay else 'eager python loop'}] host enqueue per step: {1e3*sum(enq)/10:.2f} ms   wall per step: {1e3*tot:.2f} ms")
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        tr.step(b)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
