"""CPU checks of the oracle pieces added for configs[4] and SURVEY §8(f): known answers and cross-checks against independent
implementations (torch's float8 cast, closed forms).  The reference repository holds no fixtures (README only), so these
known answers are what pins the oracle ("parity unpinned" by the reference itself)."""
import math

import numpy as np
import torch


def test_e4m3_rounding_matches_torch_float8_cast():
    from oracle.model import e4m3_rne, fp8_qdq
    g = torch.Generator().manual_seed(0)
    v = torch.cat([torch.randn(50000, generator=g) * 100, torch.randn(5000, generator=g) * 0.01,
                   torch.tensor([448.0, 447.9, 464.0, 0.0, -0.0019, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 2.5 * 2.0 ** -9, -448.0])])
    v = v.clamp(-448, 448)
    assert torch.equal(e4m3_rne(v), v.to(torch.float8_e4m3fn).float())
    # grid facts: 448 is the maximum, 2^-9 the subnormal step, ties go to even
    assert e4m3_rne(torch.tensor([1e6])).item() == 448.0
    assert e4m3_rne(torch.tensor([1.0 + 1 / 16])).item() == 1.0 and e4m3_rne(torch.tensor([1.0 + 3 / 16])).item() == 1.25
    x = torch.randn(64, 512, generator=g)
    q = fp8_qdq(x)
    assert (q - x).abs().max() <= x.abs().max() / 448 * 16 + 1e-6       # half a grid step at the top binade (step 32 of 448)
    assert fp8_qdq(torch.zeros(8)).abs().sum() == 0


def test_fp8_linear_straight_through_gradients():
    from oracle.model import FP8Linear, fp8_qdq
    torch.manual_seed(0)
    lin = FP8Linear(32, 16)
    x = torch.randn(8, 32, requires_grad=True)
    dy = torch.randn(8, 16)
    lin(x).backward(dy)
    xh, wh = fp8_qdq(x.detach()), fp8_qdq(lin.weight.detach())
    assert torch.allclose(x.grad, dy @ wh, atol=1e-6) and torch.allclose(lin.weight.grad, dy.t() @ xh, atol=1e-6)
    assert torch.allclose(lin.bias.grad, dy.sum(0), atol=1e-6)


def test_gae_known_answers():
    from oracle import rl as R
    # constant reward 1, zero values, no terminations: adv_t = sum_k (gamma*lam)^k
    T, N, g, l = 5, 3, 0.9, 0.8
    adv, ret = R.gae(np.ones((T, N)), np.zeros((T + 1, N)), np.zeros((T, N)), g, l)
    want = [sum((g * l) ** k for k in range(T - t)) for t in range(T)]
    assert np.allclose(adv[:, 0], want, rtol=1e-6) and np.allclose(ret, adv)
    # lambda = 0 -> one-step TD error; lambda = 1, gamma = 1 -> Monte-Carlo return minus value
    rng = np.random.default_rng(0)
    r, v = rng.normal(size=(T, N)).astype(np.float32), rng.normal(size=(T + 1, N)).astype(np.float32)
    adv0, _ = R.gae(r, v, np.zeros((T, N)), 0.97, 0.0)
    assert np.allclose(adv0, r + np.float32(0.97) * v[1:] - v[:-1], atol=1e-6)
    adv1, ret1 = R.gae(r, v, np.zeros((T, N)), 1.0, 1.0)
    mc = np.cumsum(r[::-1], 0)[::-1] + v[-1]
    assert np.allclose(ret1, mc, atol=1e-5)
    # a terminal step cuts the bootstrap and the recursion
    d = np.zeros((T, N), np.float32); d[2] = 1
    advd, _ = R.gae(r, v, d, 0.9, 0.9)
    assert np.allclose(advd[2], r[2] - v[2], atol=1e-6)
    x = R.normalize(adv1)
    assert abs(x.mean()) < 1e-6 and abs(x.std(ddof=1) - 1) < 1e-5


def test_sample_actions_distribution_and_determinism():
    from oracle import rl as R
    lg = np.tile(np.array([[2.0, 1.0, 0.0, -1.0]], np.float32), (20000, 1))
    a, lp = R.sample_actions(lg, seed=7, step=3)
    a2, _ = R.sample_actions(lg, seed=7, step=3)
    a3, _ = R.sample_actions(lg, seed=7, step=4)
    assert np.array_equal(a, a2) and not np.array_equal(a, a3)
    p = np.exp(lg[0]) / np.exp(lg[0]).sum()
    freq = np.bincount(a, minlength=4) / len(a)
    assert np.abs(freq - p).max() < 0.01
    assert np.allclose(lp, np.log(p)[a], atol=1e-6)


def test_wire_format_round_trip():
    from oracle import wire as W
    rng = np.random.default_rng(1)
    x = rng.normal(size=(9, 64)).astype(np.float32)
    x[2] = 0
    q, s = W.pack(x)
    assert q.dtype == np.int8 and np.abs(q).max() == 127 and s[2] == 1.0 and not q[2].any()
    assert np.all(np.abs(q).max(1)[[0, 1, 3]] == 127)            # the per-frame maximum always maps to +-127
    assert np.abs(W.unpack(q, s) - x).max() <= 0.5 * s.max() * (1 + 1e-6)
    assert math.isclose(float(s[0]), float(np.abs(x[0]).max() / np.float32(127.0)), rel_tol=1e-7)


# ---- committed golden vectors (tests/golden/make_golden.py): the oracle must keep reproducing them
def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name), allow_pickle=False)


def test_golden_small_kernels():
    from oracle import philox, rl, wire
    from oracle.model import e4m3_rne
    G = _golden("small_kernels.npz")
    r = philox.philox4x32_10(np.array([0, 1, 0xFFFFFFFF], np.uint32), np.array([0, 2, 0xFFFFFFFF], np.uint32),
                             np.array([0, 3, 0xFFFFFFFF], np.uint32), np.array([0, 4, 0xFFFFFFFF], np.uint32), 0x0, 0x0)
    assert np.array_equal(np.stack(r), G["philox"])
    assert [hex(int(v)) for v in G["philox"][:, 0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]   # Random123 KAT
    assert np.allclose(philox.philox_normal(4, 42, 1), G["normals"], atol=1e-6)
    adv, ret = rl.gae(G["gae_rew"], G["gae_val"], G["gae_done"], 0.99, 0.95)
    assert np.array_equal(adv, G["gae_adv"]) and np.array_equal(ret, G["gae_ret"])
    act, logp = rl.sample_actions(G["samp_logits"], 1234, 17)
    assert np.array_equal(act, G["samp_action"]) and np.allclose(logp, G["samp_logp"], atol=1e-6)
    q, s = wire.pack(G["wire_x"])
    assert np.array_equal(q, G["wire_q"]) and np.array_equal(s, G["wire_scale"])
    assert np.array_equal(e4m3_rne(torch.from_numpy(G["e4m3_in"])).numpy(), G["e4m3_out"])


def test_golden_oracle_c0():
    """the C0 oracle run (configs[0], the reference's CPU-runnable case) against its committed outputs; tolerance covers
    a different BLAS summation order on another host"""
    from oracle import model as O
    G = _golden("oracle_c0_b8.npz")
    torch.set_num_threads(1)
    cfg = O.named_config("C0")
    om = O.build_oracle(cfg, seed=1)
    b = O.synthetic_batch(cfg, 8, seed=2)
    out = om(b["image"], b.get("points"), b.get("state"), b["snr_db"], b["noise"], b["fading"])
    loss, terms = O.total_loss(out, b["image"], cfg, b["ppo"])
    loss.backward()
    assert np.allclose(out["logits"].detach().numpy(), G["logits"], atol=2e-4)
    assert np.allclose(out["symbols"].detach().numpy(), G["symbols"], atol=2e-4)
    assert np.allclose(O.psnr(out["recon"], b["image"]).detach().numpy(), G["psnr"], atol=1e-3)
    assert np.array_equal(O.select_action(out["logits"]).numpy(), G["action"])
    assert np.allclose([float(terms[k].detach()) for k in ("mse", "policy", "value", "entropy", "loss")], G["terms"], rtol=1e-4, atol=1e-5)
    g = {k: p.grad for k, p in om.named_parameters()}
    norms = np.array([float(g[k].double().norm()) for k in G["grad_names"]])
    assert np.allclose(norms, G["grad_norm"], rtol=2e-2, atol=1e-7)     # training-mode BN over 8 samples: ill-conditioned


def test_topk_rows_known_answers_and_golden():
    """top-k order = (value desc, index asc), NaN last: against a plain Python sort with that key, hand cases, and the committed
    fixture tests/golden/topk.npz (which the HIP kernel is held to bit-exactly in tests/test_kernels_gpu.py)."""
    import os
    from oracle import rl as R
    from oracle.model import select_action
    x = np.array([[1, 3, 3, 2], [5, 5, 5, 5], [0.0, -0.0, 1, 1], [np.nan, 2, np.nan, -np.inf]], np.float32)
    idx, val = R.topk_rows(x, 4)
    assert idx.tolist() == [[1, 2, 3, 0], [0, 1, 2, 3], [2, 3, 0, 1], [1, 3, 0, 2]]
    assert val[0].tolist() == [3, 3, 2, 1] and np.isnan(val[3, 2:]).all()
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "topk.npz"), allow_pickle=False)
    xs = G["x"]
    for k in (1, 3, 8):
        idx, val = R.topk_rows(xs, k)
        assert np.array_equal(idx, G[f"idx{k}"]) and np.array_equal(val, G[f"val{k}"], equal_nan=True)
        for r in range(xs.shape[0]):        # independent restatement: Python's stable sort on the same key
            key = lambda j: (bool(np.isnan(xs[r, j])), 0.0 if np.isnan(xs[r, j]) else -float(xs[r, j]), j)
            assert idx[r].tolist() == sorted(range(xs.shape[1]), key=key)[:k], r
    finite = np.isfinite(xs).all(1)
    assert np.array_equal(R.topk_rows(xs, 1)[0][finite, 0], select_action(torch.from_numpy(xs[finite])).numpy())
