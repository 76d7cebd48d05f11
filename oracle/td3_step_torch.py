"""Oracle: "rlkit-equivalent restatement" of one TD3 gradient step (TEST INFRASTRUCTURE).

Plain PyTorch fp32 eager + autograd on the CPU; the checker for the HIP TD3 path.
The product never calls it.

What it follows: rlkit's TD3Trainer is NOT vendored under /root/reference (pins b7f97b2 / d63dab7,
/root/reference/README.md:28) and no shipped run uses TD3, so the step is restated from the upstream
algorithm [R] and anchored on the reference's own call sites:
* /root/reference/util/rlkit_utils.py:107-135   TanhMlpPolicy(input_size, output_size, **policy_kwargs) x2
      (policy + target_policy), GaussianStrategy(max_sigma=0.1, min_sigma=0.1),
      PolicyWrappedWithExplorationStrategy, TD3Trainer(policy, qf1, qf2, target_qf1, target_qf2,
      target_policy, **trainer_kwargs)
* /root/reference/scripts/train.py:38-47        trainer kwargs: target_policy_noise, discount=0.99,
      reward_scale, policy_learning_rate, qf_learning_rate, policy_and_target_update_period, tau
* /root/reference/util/arguments.py:141-156     defaults: noise 0.2, period 2, tau 0.005
Upstream step [R] (rlkit/torch/td3/td3.py train_from_torch):
  a' = target_policy(s');  noise = clamp(N(0,1) * target_policy_noise, +-target_policy_noise_clip [0.5]);
  y = reward_scale r + (1 - d) discount min(tqf1, tqf2)(s', a' + noise)           (a' + noise is NOT re-clipped)
  qf_i loss = mean((qf_i(s, a) - y)^2); both critics are stepped (Adam) FIRST;
  if n_train_steps_total % policy_and_target_update_period == 0:
      policy loss = -mean(qf1(s, policy(s)))   (through the ALREADY UPDATED qf1); Adam step;
      Polyak(policy -> target_policy), (qf1 -> target_qf1), (qf2 -> target_qf2) with tau
  statistics (first step of an epoch): QF1/QF2 Loss, Policy Loss (recomputed without an update when this
  is not a policy step), Q1/Q2 Predictions, Q Targets, Bellman Errors 1/2, Policy Action (Mean Std Max Min).
TanhMlpPolicy = Mlp(hidden_sizes, output_size, input_size, init_w=1e-3 [R: TanhMlpPolicy default],
output_activation=tanh).

PARITY STATUS: **parity unpinned** -- nothing in the reference (no test, no run, no log) holds a TD3 number."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from .sac_step_torch import QNet, _Net, _stats, init_mlp_params


def init_td3_params(obs_dim, act_dim, hidden=(256, 256), seed=0):
    """Six independently initialised nets, as rlkit_utils.py:64-83,108-117 builds them."""
    rs = np.random.RandomState(seed)
    nets = OrderedDict()
    for name in ("qf1", "qf2", "target_qf1", "target_qf2"):
        nets[name] = init_mlp_params(rs, obs_dim + act_dim, hidden, [1], 3e-3)
    for name in ("policy", "target_policy"):
        nets[name] = init_mlp_params(rs, obs_dim, hidden, [act_dim], 1e-3)
    return nets


class TanhMlp(_Net):
    def forward(self, obs):
        h = obs
        for i in range(len(self.ws) - 1):
            h = torch.relu(torch.nn.functional.linear(h, self.ws[i], self.bs[i]))
        return torch.tanh(torch.nn.functional.linear(h, self.ws[-1], self.bs[-1]))


class RlkitEquivalentTD3:
    def __init__(self, nets, act_dim, target_policy_noise=0.2, target_policy_noise_clip=0.5, discount=0.99,
                 reward_scale=1.0, policy_learning_rate=1e-3, qf_learning_rate=1e-3,
                 policy_and_target_update_period=2, tau=0.005):
        self.policy, self.target_policy = TanhMlp(nets["policy"]), TanhMlp(nets["target_policy"])
        self.qf1, self.qf2 = QNet(nets["qf1"]), QNet(nets["qf2"])
        self.target_qf1, self.target_qf2 = QNet(nets["target_qf1"]), QNet(nets["target_qf2"])
        self.noise, self.clip = float(target_policy_noise), float(target_policy_noise_clip)
        self.discount, self.reward_scale = float(discount), float(reward_scale)
        self.period, self.tau = int(policy_and_target_update_period), float(tau)
        self.policy_opt = torch.optim.Adam(self.policy.parameters(), lr=policy_learning_rate)
        self.qf1_opt = torch.optim.Adam(self.qf1.parameters(), lr=qf_learning_rate)
        self.qf2_opt = torch.optim.Adam(self.qf2.parameters(), lr=qf_learning_rate)
        self.n_train_steps_total = 0
        self.last = {}

    def step(self, obs, act, rew, term, next_obs, eps):
        """float32 numpy inputs; eps (B,A) = the N(0,1) draw of the target-policy smoothing noise.
        Returns the statistics of this step (always computed)."""
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        obs, act, rew, term, next_obs, eps = map(t, (obs, act, rew.reshape(-1, 1), term.reshape(-1, 1), next_obs, eps))
        with torch.no_grad():
            a2 = self.target_policy(next_obs)
            noisy = a2 + torch.clamp(eps * self.noise, -self.clip, self.clip)
            tq1, tq2 = self.target_qf1(next_obs, noisy), self.target_qf2(next_obs, noisy)
            y = self.reward_scale * rew + (1.0 - term) * self.discount * torch.min(tq1, tq2)
        q1, q2 = self.qf1(obs, act), self.qf2(obs, act)
        be1, be2 = (q1 - y) ** 2, (q2 - y) ** 2
        qf1_loss, qf2_loss = be1.mean(), be2.mean()
        self.qf1_opt.zero_grad(); qf1_loss.backward(); self.qf1_opt.step()
        self.qf2_opt.zero_grad(); qf2_loss.backward(); self.qf2_opt.step()
        g_q1 = torch.cat([p.grad.reshape(-1) for pair in zip(self.qf1.ws, self.qf1.bs) for p in pair]).numpy().copy()
        g_q2 = torch.cat([p.grad.reshape(-1) for pair in zip(self.qf2.ws, self.qf2.bs) for p in pair]).numpy().copy()
        policy_step = self.n_train_steps_total % self.period == 0
        pa = self.policy(obs)
        q_pi = self.qf1(obs, pa)                       # through the already updated qf1
        policy_loss = -q_pi.mean()
        g_pol = None
        if policy_step:
            self.policy_opt.zero_grad(); policy_loss.backward(); self.policy_opt.step()
            g_pol = torch.cat([p.grad.reshape(-1) for pair in zip(self.policy.ws, self.policy.bs) for p in pair]).numpy().copy()
            with torch.no_grad():
                for src, dst in ((self.policy, self.target_policy), (self.qf1, self.target_qf1), (self.qf2, self.target_qf2)):
                    for ps, pd in zip(src.parameters(), dst.parameters()):
                        pd.mul_(1.0 - self.tau).add_(ps, alpha=self.tau)
        d = OrderedDict()
        d["QF1 Loss"], d["QF2 Loss"], d["Policy Loss"] = (float(x.detach()) for x in (qf1_loss, qf2_loss, policy_loss))
        d.update(_stats("Q1 Predictions", q1)); d.update(_stats("Q2 Predictions", q2)); d.update(_stats("Q Targets", y))
        d.update(_stats("Bellman Errors 1", be1)); d.update(_stats("Bellman Errors 2", be2))
        d.update(_stats("Policy Action", pa))
        self.last = dict(a2=a2, noisy=noisy, tq1=tq1, tq2=tq2, y=y, q1=q1, q2=q2, pa=pa, q_pi=q_pi, g_qf1=g_q1, g_qf2=g_q2,
                         g_policy=g_pol, policy_step=policy_step)
        self.n_train_steps_total += 1
        return d

    def export_nets(self):
        return OrderedDict((k, getattr(self, k).export()) for k in
                           ("qf1", "qf2", "target_qf1", "target_qf2", "policy", "target_policy"))
