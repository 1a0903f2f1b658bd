#!/usr/bin/env python3
"""Every C-ABI call of one training step with its scalar arguments and its ISOLATED time (bench.instrumented_pass: one stream, HIP events
around each call), grouped by (entry point, arguments): count, total us, us per call, and for GEMM-shaped calls TFLOP/s.
usage: call_table.py [entry-point substring]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodal_sc_amd.model import MultimodalSC, named_config
from multimodal_sc_amd.trainer import Trainer
from multimodal_sc_amd.synthetic import synthetic_batch

cfg = named_config("C3")
m = MultimodalSC(cfg, "bf16", "cuda:0")
tr = Trainer(m)
B = 64
cb = synthetic_batch(cfg, B, seed=1)
b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in cb.items() if k != "ppo"}
b["ppo"] = {k: v.cuda() for k, v in cb["ppo"].items()}
b["noise"] = b["fading"] = None
rt = m.rt
sel = sys.argv[1] if len(sys.argv) > 1 else ""
acc = {}
for rep in range(4):
    torch.cuda.synchronize(); rt.join_side()
    rt.profile = []
    br = rt.branch_streams; rt.branch_streams = False
    try:
        tr.forward_backward(b); m.params.adam_step(lr=1e-4); torch.cuda.synchronize()
    finally:
        rt.branch_streams = br
    prof, rt.profile = rt.profile, None
    if rep == 0:
        continue
    for name, sc, e0, e1 in prof:
        k = (name, sc)
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in acc.values()) / 3
print(f"isolated sum of all calls: {tot:.0f} us per step")
for (name, sc), (n, us) in rows:
    if sel and sel not in name:
        continue
    fl = bench.gemm_flops(name, sc) if hasattr(bench, "gemm_flops") else 0
    per = us / n
    print(f"{us / 3:8.1f} us/step  {n // 3:3d} x {per:7.1f} us  {('%5.0f TF' % (fl / per / 1e6)) if fl else '        '}  {name} {sc}")
