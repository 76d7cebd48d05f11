"""Oracle: "rlkit-equivalent restatement" of one SAC gradient step (TEST INFRASTRUCTURE).

Plain PyTorch fp32 eager + autograd on the CPU.  It is the checker for the HIP path
and the timed ``cpu_baseline`` (kind "port") of bench.py; the product never calls it.

What it follows (rlkit itself is NOT vendored under /root/reference and cannot be
imported -- pinned commits b7f97b2 / d63dab7, /root/reference/README.md:28):

* call sites   /root/reference/util/rlkit_utils.py:64-106  (4x FlattenMlp, TanhGaussianPolicy,
               SACTrainer(**trainer_kwargs)), :139-142 (EnvReplayBuffer)
               /root/reference/util/rlkit_custom.py:233-240 (random_batch -> trainer.train)
               /root/reference/scripts/train.py:29-37 (trainer kwargs), :112-113 (seeding)
* embedded src /root/reference/runs/Lift-Panda-OSC-POSE-SEED17/*/params.pkl, pickle string
               @423 ``class TanhGaussianPolicy`` (forward: clamp, exp, TanhNormal rsample,
               log_prob sum), @8560 ``class FlattenMlp`` (cat(inputs, dim=1)); read as text
               with pickletools.genops -- never unpickled.
* normative    SURVEY.md Appendix A lines 1-18, ordering note O1, logging quirk Q1.

PARITY STATUS: **parity unpinned** at the bit/epsilon level -- the reference ships no
golden vector for a gradient step.  Pinned against the known answers its progress.csv
files hold (tests/test_oracle_known_answers.py): KA1 Alpha0 = fp32(exp(-policy_lr)),
Alpha Loss0 = -0.0; KA2 alpha-loss / log-alpha relation; KA3 logged "Policy Loss" =
mean(log_pi - q_new); KA4 log_std clamp at 2; KA6 E[log_pi]0 ~ -0.676*A; KA7 MSE mean.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

LOG_SIG_MAX = 2.0     # [D] progress.csv "Policy log std Max" never exceeds 2.0
LOG_SIG_MIN = -20.0   # [R] upstream rlkit constant
TANH_EPS = 1e-6       # [R] TanhNormal epsilon


# --------------------------------------------------------------------------------------
# parameter containers.  Layout of one net = list of (W (out,in), b (out,)) in order
#   Q nets : fc0, fc1, last_fc
#   policy : fc0, fc1, last_fc (mean), last_fc_log_std
# --------------------------------------------------------------------------------------
def init_mlp_params(rs: np.random.RandomState, in_dim: int, hidden, heads, init_w: float, b_init_value: float = 0.1):
    """Deterministic rlkit-style init ([R]: fanin_init uses size[0] of the (out,in) weight,
    hidden bias constant b_init_value -- 0.1 in the rlkit the reference pins (b7f97b2, before the networks
    refactor that made it 0; unpinned: rlkit is not vendored and no shipped artefact holds an initial bias) --
    heads uniform(+-init_w)).  Init never enters per-step parity."""
    params = []
    d = in_dim
    for h in hidden:
        bound = 1.0 / math.sqrt(h)
        params.append((rs.uniform(-bound, bound, size=(h, d)).astype(np.float32),
                       np.full(h, b_init_value, dtype=np.float32)))
        d = h
    for n_out in heads:
        params.append((rs.uniform(-init_w, init_w, size=(n_out, d)).astype(np.float32),
                       rs.uniform(-init_w, init_w, size=(n_out,)).astype(np.float32)))
    return params


def init_sac_params(obs_dim: int, act_dim: int, hidden=(256, 256), seed: int = 0, hidden_q=None):
    """Five independently initialised nets, as rlkit_utils.py:64-97 builds them (`hidden`: policy_kwargs hidden_sizes,
    `hidden_q`: qf_kwargs hidden_sizes, default the same)."""
    rs = np.random.RandomState(seed)
    nets = OrderedDict()
    for name in ("qf1", "qf2", "target_qf1", "target_qf2"):
        nets[name] = init_mlp_params(rs, obs_dim + act_dim, hidden_q or hidden, [1], 3e-3)
    nets["policy"] = init_mlp_params(rs, obs_dim, hidden, [act_dim, act_dim], 1e-3)
    return nets


class _Net(torch.nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.ws = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(np.array(w))) for w, _ in layers])
        self.bs = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(np.array(b))) for _, b in layers])

    def export(self):
        return [(w.detach().numpy().copy(), b.detach().numpy().copy()) for w, b in zip(self.ws, self.bs)]


class QNet(_Net):
    """FlattenMlp: cat(inputs, dim=1) -> relu(fc0) -> relu(fc1) -> last_fc (identity)."""

    def forward(self, obs, act):
        h = torch.cat([obs, act], dim=1)
        n = len(self.ws)
        for i in range(n - 1):
            h = torch.relu(torch.nn.functional.linear(h, self.ws[i], self.bs[i]))
        return torch.nn.functional.linear(h, self.ws[n - 1], self.bs[n - 1])


class PolicyNet(_Net):
    """TanhGaussianPolicy.forward(obs, reparameterize=True, return_log_prob=True) with the
    N(0,1) noise passed in (rlkit draws it from the torch global generator)."""

    def trunk(self, obs):
        h = obs
        for i in range(len(self.ws) - 2):
            h = torch.relu(torch.nn.functional.linear(h, self.ws[i], self.bs[i]))
        mean = torch.nn.functional.linear(h, self.ws[-2], self.bs[-2])
        log_std = torch.nn.functional.linear(h, self.ws[-1], self.bs[-1])
        log_std = torch.clamp(log_std, LOG_SIG_MIN, LOG_SIG_MAX)
        return mean, log_std

    def forward(self, obs, eps):
        mean, log_std = self.trunk(obs)
        std = torch.exp(log_std)
        # TanhNormal.rsample(return_pretanh_value=True)
        z = mean + std * eps
        action = torch.tanh(z)
        # TanhNormal.log_prob(action, pre_tanh_value=z) = Normal(mean,std).log_prob(z) - log(1-a^2+eps)
        var = std ** 2
        log_scale = torch.log(std)
        normal_lp = -((z - mean) ** 2) / (2 * var) - log_scale - math.log(math.sqrt(2 * math.pi))
        log_prob = normal_lp - torch.log(1 - action * action + TANH_EPS)
        log_prob = log_prob.sum(dim=1, keepdim=True)
        return action, mean, log_std, log_prob, z


def _stats(prefix, t):
    """eval_util.create_stats_ordered_dict: Mean / Std (population) / Max / Min."""
    a = t.detach().numpy().astype(np.float32).ravel()
    return OrderedDict([(prefix + " Mean", float(np.mean(a))), (prefix + " Std", float(np.std(a))),
                        (prefix + " Max", float(np.max(a))), (prefix + " Min", float(np.min(a)))])


class RlkitEquivalentSAC:
    """State + one-step transition of SACTrainer.train_from_torch."""

    def __init__(self, nets, act_dim, discount=0.99, reward_scale=1.0, policy_lr=1e-3, qf_lr=1e-3,
                 soft_target_tau=1e-2, target_update_period=1, use_automatic_entropy_tuning=True,
                 target_entropy=None):
        self.policy = PolicyNet(nets["policy"])
        self.qf1, self.qf2 = QNet(nets["qf1"]), QNet(nets["qf2"])
        self.target_qf1, self.target_qf2 = QNet(nets["target_qf1"]), QNet(nets["target_qf2"])
        self.discount, self.reward_scale = float(discount), float(reward_scale)
        self.tau, self.period = float(soft_target_tau), int(target_update_period)
        self.auto_alpha = bool(use_automatic_entropy_tuning)
        self.target_entropy = float(-act_dim if target_entropy is None else target_entropy)  # [D] = -prod(act shape)
        self.log_alpha = torch.zeros(1, requires_grad=True)
        self.alpha_opt = torch.optim.Adam([self.log_alpha], lr=policy_lr)
        self.policy_opt = torch.optim.Adam(self.policy.parameters(), lr=policy_lr)
        self.qf1_opt = torch.optim.Adam(self.qf1.parameters(), lr=qf_lr)
        self.qf2_opt = torch.optim.Adam(self.qf2.parameters(), lr=qf_lr)
        self.n_train_steps_total = 0
        self.last = {}

    # ---- one gradient step -------------------------------------------------------------
    def step(self, obs, act, rew, term, next_obs, eps1, eps2):
        """All inputs float32 numpy: obs (B,O) act (B,A) rew (B,1) term (B,1) next_obs (B,O)
        eps1/eps2 (B,A).  Returns the diagnostics of this step (always computed)."""
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        obs, act, rew, term, next_obs, eps1, eps2 = map(t, (obs, act, rew.reshape(-1, 1), term.reshape(-1, 1),
                                                           next_obs, eps1, eps2))
        # 1-3: policy on s
        a_new, mu, log_std, log_pi, z = self.policy(obs, eps1)
        # 4-6: alpha loss / step (post-step alpha used below)  [D]
        if self.auto_alpha:
            alpha_loss = -(self.log_alpha * (log_pi + self.target_entropy).detach()).mean()
            self.alpha_opt.zero_grad()
            alpha_loss.backward()
            self.alpha_opt.step()
            alpha = self.log_alpha.exp()
        else:
            alpha_loss = torch.zeros(())
            alpha = torch.ones(1)
        # 7-8: actor loss
        q1_new, q2_new = self.qf1(obs, a_new), self.qf2(obs, a_new)
        q_new = torch.min(q1_new, q2_new)
        policy_loss = (alpha * log_pi - q_new).mean()
        # 9-13: critic losses
        q1, q2 = self.qf1(obs, act), self.qf2(obs, act)
        a2, _, _, log_pi2, _ = self.policy(next_obs, eps2)
        tq = torch.min(self.target_qf1(next_obs, a2), self.target_qf2(next_obs, a2)) - alpha * log_pi2
        y = (self.reward_scale * rew + (1.0 - term) * self.discount * tq).detach()
        qf1_loss = torch.mean((q1 - y) ** 2)
        qf2_loss = torch.mean((q2 - y) ** 2)
        # 14-15: all grads at the pre-step parameters (O1), then Adam
        self.policy_opt.zero_grad()
        policy_loss.backward()
        g_pol = [p.grad.detach().numpy().copy() for p in self.policy.parameters()]
        self.qf1_opt.zero_grad()
        qf1_loss.backward()
        self.qf2_opt.zero_grad()
        qf2_loss.backward()
        g_q1 = [p.grad.detach().numpy().copy() for p in self.qf1.parameters()]
        g_q2 = [p.grad.detach().numpy().copy() for p in self.qf2.parameters()]
        self.policy_opt.step()
        self.qf1_opt.step()
        self.qf2_opt.step()
        # 16: Polyak, pre-increment counter
        if self.n_train_steps_total % self.period == 0:
            with torch.no_grad():
                for src, dst in ((self.qf1, self.target_qf1), (self.qf2, self.target_qf2)):
                    for ps, pd in zip(src.parameters(), dst.parameters()):
                        pd.copy_(pd * (1.0 - self.tau) + ps * self.tau)
        # 17: diagnostics (Q1 quirk: logged "Policy Loss" has no alpha)
        d = OrderedDict()
        d["QF1 Loss"] = float(qf1_loss.detach())
        d["QF2 Loss"] = float(qf2_loss.detach())
        d["Policy Loss"] = float((log_pi - q_new).mean().detach())
        d["Actor Loss"] = float(policy_loss.detach())          # the optimised one; not an rlkit column
        d.update(_stats("Q1 Predictions", q1))
        d.update(_stats("Q2 Predictions", q2))
        d.update(_stats("Q Targets", y))
        d.update(_stats("Log Pis", log_pi))
        d.update(_stats("Policy mu", mu))
        d.update(_stats("Policy log std", log_std))
        d["Alpha"] = float(alpha.detach())
        d["Alpha Loss"] = float(alpha_loss.detach())
        self.last = dict(a_new=a_new, mu=mu, log_std=log_std, log_pi=log_pi, z=z, q1=q1, q2=q2, y=y,
                         q1_new=q1_new, q2_new=q2_new, a2=a2, log_pi2=log_pi2,
                         g_policy=g_pol, g_qf1=g_q1, g_qf2=g_q2)
        # 18
        self.n_train_steps_total += 1
        return d

    def export_nets(self):
        return OrderedDict(qf1=self.qf1.export(), qf2=self.qf2.export(), target_qf1=self.target_qf1.export(),
                           target_qf2=self.target_qf2.export(), policy=self.policy.export())


# --------------------------------------------------------------------------------------
# reference-shaped host replay buffer (rlkit SimpleReplayBuffer): float64 per-field
# arrays, uint8 terminals, global np.random stream, fancy-index copies.
# --------------------------------------------------------------------------------------
class HostReplayBuffer:
    def __init__(self, max_size, obs_dim, act_dim):
        self._max = int(max_size)
        self._obs = np.zeros((self._max, obs_dim))
        self._next_obs = np.zeros((self._max, obs_dim))
        self._act = np.zeros((self._max, act_dim))
        self._rew = np.zeros((self._max, 1))
        self._term = np.zeros((self._max, 1), dtype="uint8")
        self._top = 0
        self._size = 0

    def add_sample(self, o, a, r, t, no):
        self._obs[self._top] = o
        self._act[self._top] = a
        self._rew[self._top] = r
        self._term[self._top] = t
        self._next_obs[self._top] = no
        self._top = (self._top + 1) % self._max
        self._size = min(self._size + 1, self._max)

    def add_block(self, o, a, r, t, no):
        for i in range(len(o)):
            self.add_sample(o[i], a[i], r[i], t[i], no[i])

    def fill_block(self, o, a, r, t, no):
        """Bulk equivalent of n add_sample calls on an empty/large buffer (test helper)."""
        n = len(o)
        idx = (self._top + np.arange(n)) % self._max
        self._obs[idx], self._act[idx], self._next_obs[idx] = o, a, no
        self._rew[idx, 0], self._term[idx, 0] = np.ravel(r), np.ravel(t)
        self._top = (self._top + n) % self._max
        self._size = min(self._size + n, self._max)

    def random_batch(self, batch_size, rng=np.random):
        idx = rng.randint(0, self._size, batch_size)
        return dict(observations=self._obs[idx], actions=self._act[idx], rewards=self._rew[idx],
                    terminals=self._term[idx], next_observations=self._next_obs[idx]), idx


def np_to_f32_batch(b):
    """rlkit np_to_pytorch_batch: every array -> float32 (uint8 terminals included)."""
    return {k: np.asarray(v).astype(np.float32) for k, v in b.items()}
