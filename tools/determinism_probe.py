#!/usr/bin/env python3
"""Run the C3 @ batch-64 training step several times and report WHICH parameter gradients differ between runs (bitwise).
Used to bisect a non-determinism: run under the A/B knobs (MSC_*) or an alternative library (MSC_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_sc_amd.model import MultimodalSC, named_config
from multimodal_sc_amd.synthetic import synthetic_batch

name, B, reps = (sys.argv[1] if len(sys.argv) > 1 else "C3"), int(sys.argv[2]) if len(sys.argv) > 2 else 64, 4
m = MultimodalSC(named_config(name), dtype="bf16", device="cuda:0", seed=3)
def to_dev(b):
    return {k: (to_dev(v) if isinstance(v, dict) else v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in b.items()}


gb = to_dev(synthetic_batch(named_config(name), B, seed=5))
res = []
for _ in range(reps):
    out = m(gb["image"], gb.get("points"), gb.get("state"), gb["snr_db"], gb["noise"], gb["fading"])
    m.loss_and_grads(out, gb["image"], gb["ppo"])
    torch.cuda.synchronize()
    res.append((out["head"].clone(), out["patches"].clone(), m.params.G.clone()))
bad = []
for r in range(1, reps):
    if not (torch.equal(res[0][0], res[r][0]) and torch.equal(res[0][1], res[r][1])):
        bad.append(("FORWARD", r))
    for nm, ref in m.params.refs.items():
        a, b = res[0][2][ref.off:ref.off + ref.numel], res[r][2][ref.off:ref.off + ref.numel]
        if not torch.equal(a, b):
            bad.append((nm, r, float((a - b).abs().max() / a.abs().max().clamp_min(1e-30))))
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("MSC_"))
print(f"[{tag}] {'DETERMINISTIC' if not bad else 'DIFFERS: ' + str(len(bad))}")
seen = set()
for b in bad:
    if b[0] not in seen:
        seen.add(b[0]); print("   ", b)
