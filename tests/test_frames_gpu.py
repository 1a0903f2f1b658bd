"""Host-resident frame source: a recorded drive on disk -> pinned ring -> msc_memcpy_h2d_async on the side
stream -> msc_frames_u8_to_nchw.  The staged batches must equal the source BYTES (camera bytes bit-exact, decoded planes bit-equal
to float32(u8) / float32(255), clouds / ego / actions bit-exact), batch after batch, while training steps run in between; and the
step fed from the stager must produce exactly the parameters of the step fed from device-resident copies of the same frames."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_u8_decode_kernel_bit_exact(cuda):
    from multimodal_sc_amd.runtime import Runtime
    rt = Runtime("bf16", "cuda:0")
    rng = np.random.default_rng(0)
    u8 = rng.integers(0, 256, size=(3, 32, 32, 3), dtype=np.uint8)
    u8[0, 0, :8] = np.arange(24, dtype=np.uint8).reshape(8, 3) * 10            # a recognisable corner
    u8[1] = np.arange(256, dtype=np.uint8).repeat(12).reshape(32, 32, 3)       # every byte value
    d = torch.from_numpy(u8).cuda()
    out = torch.empty(3, 3, 32, 32, device="cuda:0")
    rt.call("msc_frames_u8_to_nchw", d, out, 3, 32, 32)
    torch.cuda.synchronize()
    want = (u8.astype(np.float32) / np.float32(255)).transpose(0, 3, 1, 2)
    assert np.array_equal(out.cpu().numpy(), want)


def test_staged_batches_equal_the_source_bytes_and_feed_the_step(cuda, tmp_path):
    from multimodal_sc_amd.frames import FrameLog, FrameStager
    from multimodal_sc_amd.model import MultimodalSC, named_config
    from multimodal_sc_amd.trainer import Trainer
    cfg = named_config("C3")
    B, nb = 4, 6
    log = FrameLog.write_synthetic(str(tmp_path / "drive"), cfg, B * nb - 3, seed=7)       # not a multiple of B: the windows wrap
    rng = np.random.default_rng(1)
    ppo_host = [{"actions": rng.integers(0, cfg.n_actions, B).astype(np.int64), "old_logp": (-np.log(cfg.n_actions) + 0.1 * rng.standard_normal(B)).astype(np.float32),
                 "adv": rng.standard_normal(B).astype(np.float32), "ret": rng.standard_normal(B).astype(np.float32)} for _ in range(nb)]

    direct = [0]

    def run(staged):
        torch.manual_seed(0)
        m = MultimodalSC(cfg, dtype="bf16", device="cuda:0", seed=3)
        tr = Trainer(m, lr=1e-3)
        snr = torch.full((B,), 10.0, device="cuda:0")
        st = FrameStager(m.rt, log, B, extra_fn=lambda k: ppo_host[k % len(ppo_host)]) if staged else None
        seen = []
        batch = None
        for k in range(nb):
            idx = (np.arange(B) + k * B) % log.n
            if staged:
                # once the trainer has recorded its launch list, the camera bytes are decoded straight into the step's own image input
                own = tr.input_buffers(batch) if batch is not None else None
                b = st.next(image_out=own["image"] if own is not None else None)
                assert own is None or b["image"].data_ptr() == own["image"].data_ptr()
                direct[0] += own is not None
                st.prefetch()
                seen.append({kk: b[kk].clone() for kk in ("frames_u8", "image", "points", "state", "action")})
                batch = {"image": b["image"], "points": b["points"], "state": b["state"], "snr_db": snr, "noise": None, "fading": None,
                         "ppo": dict(b["extra"])}
            else:
                # (numpy's IEEE division; torch's `tensor / 255.0` on the device multiplies by the reciprocal: 1 ulp off)
                cam = np.asarray(log.arrays["camera"][idx]).astype(np.float32) / np.float32(255)
                batch = {"image": torch.from_numpy(np.ascontiguousarray(cam.transpose(0, 3, 1, 2))).cuda(),
                         "points": torch.from_numpy(np.ascontiguousarray(log.arrays["lidar"][idx])).cuda(),
                         "state": torch.from_numpy(np.ascontiguousarray(log.arrays["ego"][idx])).cuda(), "snr_db": snr, "noise": None,
                         "fading": None, "ppo": {kk: torch.from_numpy(v).cuda() for kk, v in ppo_host[k].items()}}
            tr.step(batch)
        torch.cuda.synchronize()
        if st is not None:
            st.close()
        return m.params.P.clone(), seen

    P_staged, seen = run(True)
    assert direct[0] >= 2, "the decode-into-the-input-buffer path was not exercised"
    for k, b in enumerate(seen):
        idx = (np.arange(B) + k * B) % log.n
        cam = np.asarray(log.arrays["camera"][idx])
        assert np.array_equal(b["frames_u8"].cpu().numpy(), cam), k
        assert np.array_equal(b["image"].cpu().numpy(), (cam.astype(np.float32) / np.float32(255)).transpose(0, 3, 1, 2)), k
        assert np.array_equal(b["points"].cpu().numpy(), np.asarray(log.arrays["lidar"][idx])), k
        assert np.array_equal(b["state"].cpu().numpy(), np.asarray(log.arrays["ego"][idx])), k
        assert np.array_equal(b["action"].cpu().numpy(), np.asarray(log.arrays["action"][idx])), k
    P_dev, _ = run(False)
    assert torch.equal(P_staged, P_dev), "the step fed from the host-resident source differs from the device-resident one"


def test_recorded_driving_env_through_the_ppo_loop(cuda, tmp_path):
    """the reset()/step() interface of ppo.py over a recorded drive: observations come through the stager, rewards from the logged actions"""
    from multimodal_sc_amd.frames import FrameLog, RecordedDrivingEnv
    from multimodal_sc_amd.model import MultimodalSC, MSCConfig
    from multimodal_sc_amd.ppo import PPOTrainer, RolloutBuffer
    cfg = MSCConfig(img_size=32, use_lidar=False, use_state=True)
    N, T = 8, 6
    log = FrameLog.write_synthetic(str(tmp_path / "d"), cfg, 256, seed=2)
    m = MultimodalSC(cfg, dtype="bf16", device="cuda:0")
    env = RecordedDrivingEnv(log, m.rt, N, episode_len=3)
    ppo = PPOTrainer(m, lr=3e-4, seed=1)
    buf = RolloutBuffer(T, N, "cuda:0")
    ppo.collect(env, buf)
    torch.cuda.synchronize()
    starts = (np.arange(N) * (256 // N))
    for t in range(T):
        want = np.asarray(log.arrays["camera"][(starts + t) % 256]).astype(np.float32) / np.float32(255)
        assert np.array_equal(buf.obs["image"][t].cpu().numpy(), want.transpose(0, 3, 1, 2)), t
        logged = torch.from_numpy(np.asarray(log.arrays["action"][(starts + t) % 256])).cuda()
        assert torch.equal(buf.rewards[t], (buf.actions[t] == logged).float()), t
    assert buf.dones[2].all() and not buf.dones[1].any()
    logs = ppo.update(buf, epochs=1, minibatch=16, generator=torch.Generator().manual_seed(0))
    assert len(logs) == T * N // 16 and all(np.isfinite(l["loss"]) for l in logs)
    env.stager.close()
