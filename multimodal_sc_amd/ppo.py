"""PPO loop around the hot path: rollout buffer, GAE, minibatch epochs.

The reference README names the method ("Reinforcement Learning") and nothing else, so the loop follows the published PPO-clip
recipe (Schulman et al. 2017): collect T steps from N parallel environments with the current policy, estimate advantages
with GAE(gamma, lambda), normalise them over the rollout, then run `epochs` passes of shuffled minibatches through
Trainer.step (the fused PPO + reconstruction loss of the hot path).  Device-side pieces are C-ABI kernels
(msc_sample_actions, msc_gae, msc_normalize); torch is used for storage and index gathers only.

An environment is any object with
    reset()            -> obs dict  {"image" [N,3,H,W], "points" [N,P,4] (opt), "state" [N,S] (opt)}  (fp32, on device)
    step(actions[N])   -> (obs, reward fp32 [N], done fp32 [N])
SyntheticDrivingEnv below produces seeded random frames of the benchmark shape (no simulator is installed here).
"""
from __future__ import annotations

import torch

from .model import MultimodalSC
from .trainer import Trainer


class SyntheticDrivingEnv:
    """N independent streams of random camera / LiDAR / ego frames; reward = +1 when the action equals a hidden
    per-environment target that is a function of the ego state (so a policy CAN learn it), episodes of fixed length."""

    def __init__(self, cfg, n_envs: int, device, episode_len: int = 16, seed: int = 0):
        self.cfg, self.N, self.device, self.episode_len = cfg, n_envs, torch.device(device), episode_len
        self.g = torch.Generator(device="cpu").manual_seed(seed)
        self.t = 0

    def _obs(self):
        cfg, N = self.cfg, self.N
        o = {"image": torch.rand(N, 3, cfg.img_size, cfg.img_size, generator=self.g)}
        if cfg.use_lidar:
            xy = (torch.rand(N, cfg.n_points, 2, generator=self.g) * 2 - 1) * cfg.bev_range
            z = torch.rand(N, cfg.n_points, 1, generator=self.g) * 4 - 3
            it = torch.rand(N, cfg.n_points, 1, generator=self.g)
            o["points"] = torch.cat([xy, z, it], -1)
        if cfg.use_state:
            o["state"] = torch.randn(N, cfg.state_dim, generator=self.g)
        self._target = (o["state"][:, 0] > 0).long() if cfg.use_state else (o["image"].mean((1, 2, 3)) > 0.5).long()
        return {k: v.to(self.device) for k, v in o.items()}

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, actions):
        reward = (actions.cpu() == self._target).float().to(self.device)
        self.t += 1
        done = torch.full((self.N,), 1.0 if self.t % self.episode_len == 0 else 0.0, device=self.device)
        return self._obs(), reward, done


class RolloutBuffer:
    """[T][N] storage on the device; observations are kept as given (fp32) and gathered per minibatch."""

    def __init__(self, T: int, N: int, device):
        self.T, self.N, self.device = T, N, torch.device(device)
        self.obs = None
        f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
        self.actions = torch.zeros(T, N, dtype=torch.int64, device=self.device)
        self.logp, self.rewards, self.dones = f(T, N), f(T, N), f(T, N)
        self.values = f(T + 1, N)
        self.adv, self.ret = f(T, N), f(T, N)

    def add(self, t, obs, action, logp, value, reward, done):
        if self.obs is None:
            self.obs = {k: torch.empty(self.T, *v.shape, dtype=v.dtype, device=self.device) for k, v in obs.items()}
        for k, v in obs.items():
            self.obs[k][t] = v
        self.actions[t], self.logp[t], self.values[t] = action, logp, value
        self.rewards[t], self.dones[t] = reward, done

    def minibatches(self, size: int, generator=None):
        """shuffled index sets over the T*N transitions (CPU generator -> reproducible)"""
        n = self.T * self.N
        perm = torch.randperm(n, generator=generator).to(self.device)
        for lo in range(0, n - size + 1, size):
            yield perm[lo:lo + size]

    def gather(self, idx):
        flat = lambda t: t.reshape(self.T * self.N, *t.shape[2:])
        b = {k: flat(v).index_select(0, idx) for k, v in self.obs.items()}
        b["ppo"] = {"actions": flat(self.actions).index_select(0, idx), "old_logp": flat(self.logp).index_select(0, idx),
                    "adv": flat(self.adv).index_select(0, idx), "ret": flat(self.ret).index_select(0, idx)}
        return b


class PPOTrainer:
    def __init__(self, model: MultimodalSC, lr=1e-4, gamma=0.99, lam=0.95, snr_db=10.0, seed=0):
        self.model, self.trainer = model, Trainer(model, lr=lr, use_ppo=True)
        self.gamma, self.lam, self.snr_db, self.seed = gamma, lam, snr_db, seed
        self.global_step = 0

    def _policy(self, obs):
        """sampled action, log-prob and value for a batch of observations (forward of the hot path, no backward)"""
        m, rt = self.model, self.model.rt
        N = obs["image"].shape[0]
        snr = torch.full((N,), float(self.snr_db), device=rt.device)
        with torch.no_grad():
            out = m(obs["image"], obs.get("points"), obs.get("state"), snr, None, None)
        act = torch.empty(N, dtype=torch.int64, device=rt.device)
        logp, val = rt.f32(N), rt.f32(N)
        head = out["head"]
        rt.call("msc_sample_actions", head, act, logp, val, N, m.cfg.n_actions, head.shape[1], self.seed, None, self.global_step, rt.dt)
        self.global_step += 1
        return act, logp, val

    def collect(self, env, buf: RolloutBuffer, obs=None):
        rt = self.model.rt
        obs = env.reset() if obs is None else obs
        for t in range(buf.T):
            act, logp, val = self._policy(obs)
            nxt, reward, done = env.step(act)
            buf.add(t, obs, act, logp, val, reward, done)
            obs = nxt
        _, _, last_val = self._policy(obs)
        buf.values[buf.T] = last_val
        rt.call("msc_gae", buf.rewards, buf.values, buf.dones, buf.adv, buf.ret, buf.T, buf.N, self.gamma, self.lam)
        rt.call("msc_normalize", buf.adv, buf.T * buf.N, 1e-8)
        return obs

    def update(self, buf: RolloutBuffer, epochs: int, minibatch: int, generator=None):
        """-> list of per-minibatch metric dicts"""
        rt, logs = self.model.rt, []
        for _ in range(epochs):
            for idx in buf.minibatches(minibatch, generator):
                b = buf.gather(idx)
                b["snr_db"] = torch.full((minibatch,), float(self.snr_db), device=rt.device)
                b["noise"] = b["fading"] = None
                terms = self.trainer.step(b)
                logs.append(self.model.metrics(terms))
        return logs
