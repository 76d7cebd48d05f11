"""Host-side parameter holders with the constructor signatures of rlkit's networks.

Reference call sites: /root/reference/util/rlkit_utils.py:64-83 (4x ``FlattenMlp(input_size,
output_size, hidden_sizes)``), :92-97 (``TanhGaussianPolicy(obs_dim, action_dim, hidden_sizes)``,
``MakeDeterministic``).  Layer names (fc0, fc1, last_fc, last_fc_log_std) follow the shipped
params.pkl.  These objects own numpy parameters; ``SACTrainer`` uploads them to HBM and keeps
them in sync.  ``.to(device)`` / ``.train(mode)`` are no-ops (rlkit_custom.py:306-312)."""
from __future__ import annotations

import math
import sys
import zlib
from collections import OrderedDict

import numpy as np


# ---- where the shim's random numbers come from ---------------------------------------------------------------
# The reference seeds two generators per run (/root/reference/scripts/train.py:112-113): ``np.random.seed(args.seed)``
# -- consumed by random_batch (and by robosuite) -- and ``torch.manual_seed(args.seed)`` -- from which rlkit draws the
# initial weights, the rsample noise of the training step and the exploration noise of get_action.  The shim keeps that
# split: when torch is loaded in the process (any run of the reference's own scripts), initial weights and acting noise
# are drawn from torch's default generator and the device noise stream is keyed by ``torch.initial_seed()``; np.random
# is then consumed by the replay buffer ALONE, exactly as with rlkit.  Without torch (this package's own driver, which
# seeds np.random only and passes explicit seeds) the fall-back is np.random itself.
class TorchGlobalStream:
    """The draws of torch's default CPU generator behind the two RandomState methods the holders use (stateless:
    picklable; every draw advances the process-wide generator, like rlkit's own ``torch.randn`` / ``uniform_``)."""

    def uniform(self, low, high, size):
        import torch
        size = (size,) if isinstance(size, int) else tuple(size)
        return (torch.rand(size, dtype=torch.float64) * (high - low) + low).numpy()

    def standard_normal(self, size):
        import torch
        size = (size,) if isinstance(size, int) else tuple(size)
        return torch.randn(size, dtype=torch.float64).numpy()


def torch_is_loaded():
    return "torch" in sys.modules and hasattr(sys.modules["torch"], "initial_seed")


def process_stream():
    """The generator rlkit would have drawn from: torch's global one when torch is loaded, else np.random."""
    return TorchGlobalStream() if torch_is_loaded() else np.random


def process_seed():
    """The run's seed as the reference set it, WITHOUT consuming any stream: ``torch.initial_seed()`` when torch is
    loaded (train.py:113), else a digest of np.random's current state (train.py:112)."""
    if torch_is_loaded():
        return int(sys.modules["torch"].initial_seed()) & 0xFFFFFFFFFFFFFFFF
    st = np.random.get_state()
    return (zlib.crc32(np.ascontiguousarray(st[1], np.uint32).tobytes()) << 10) ^ int(st[2])


class _Mlp:
    def __init__(self, hidden_sizes, output_sizes, input_size, init_w, rs=None, b_init_value=0.1):
        # b_init_value: rlkit Mlp's constant hidden bias -- 0.1 at the commit the reference pins (b7f97b2,
        # /root/reference/README.md:28; later rlkit: 0).  Unpinned: rlkit is not vendored and no shipped artefact
        # holds an initial bias; it only shapes from-scratch learning curves, never per-step parity.
        rs = rs or process_stream()      # rlkit draws its init from the torch global generator; [R]
        self.input_size, self.hidden_sizes = int(input_size), list(hidden_sizes)
        self.layers = OrderedDict()
        d = self.input_size
        for i, h in enumerate(self.hidden_sizes):
            bound = 1.0 / math.sqrt(h)      # rlkit fanin_init: size[0] of the (out,in) weight  [R]
            self.layers[f"fc{i}"] = [rs.uniform(-bound, bound, (h, d)).astype(np.float32),
                                     np.full(h, b_init_value, np.float32)]
            d = h
        for name, n_out in output_sizes:
            self.layers[name] = [rs.uniform(-init_w, init_w, (n_out, d)).astype(np.float32),
                                 rs.uniform(-init_w, init_w, (n_out,)).astype(np.float32)]
        self._trainer = None

    # nn.Module look-alikes the epoch loop (rlkit_custom.py:306-312) and the rollout scripts (rlkit_utils.py:202,250:
    # ``policy.cuda()`` on a loaded snapshot) touch
    def to(self, device):
        return self

    def train(self, mode=True):
        return self

    def cuda(self, device=None):
        return self

    def cpu(self):
        return self

    def eval(self):
        return self

    # The reference pickles these objects every epoch: ``logger.save_itr_params(epoch, snapshot)`` with the trainer's
    # networks and both collectors' policies in the snapshot (rlkit_custom.py:54-56,68-82).  A holder bound to a trainer
    # first pulls its trained weights off the device and is saved WITHOUT the binding (a ctypes handle cannot be pickled):
    # the loaded object is a self-contained host network, which is all rlkit_utils.py:173-174,241-250 need.
    def __getstate__(self):
        tr = self._trainer
        if tr is not None and getattr(tr, "_h", None) is not None and tr.policy is self:
            tr.sync_holder_to_host(self)
        st = dict(self.__dict__)
        st["_trainer"] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)

    def flat(self):
        return np.concatenate([np.concatenate([w.ravel(), b.ravel()]) for w, b in self.layers.values()])

    def load_flat(self, vec):
        vec = np.asarray(vec, dtype=np.float32)
        off = 0
        for wb in self.layers.values():
            for j in range(2):
                n = wb[j].size
                wb[j] = vec[off:off + n].reshape(wb[j].shape).copy()
                off += n
        assert off == vec.size

    def state_dict(self):
        sd = OrderedDict()
        for name, (w, b) in self.layers.items():
            sd[name + ".weight"], sd[name + ".bias"] = w.copy(), b.copy()
        return sd

    def as_layer_list(self):
        return [(w, b) for w, b in self.layers.values()]


class FlattenMlp(_Mlp):
    """``FlattenMlp(input_size=, output_size=, hidden_sizes=)``: cat(inputs, dim=1) -> relu MLP."""

    def __init__(self, hidden_sizes, output_size, input_size, init_w=3e-3, **kwargs):
        super().__init__(hidden_sizes, [("last_fc", output_size)], input_size, init_w, kwargs.get("rs"),
                         kwargs.get("b_init_value", 0.1))

    def forward_np(self, *inputs):
        h = np.concatenate(inputs, axis=1).astype(np.float32)
        names = list(self.layers)
        for n in names[:-1]:
            w, b = self.layers[n]
            h = np.maximum(h @ w.T + b, 0)
        w, b = self.layers[names[-1]]
        return h @ w.T + b


class TanhGaussianPolicy(_Mlp):
    """``TanhGaussianPolicy(hidden_sizes=, obs_dim=, action_dim=)``; ``get_action(obs_np)`` returns
    ``(action, {})`` like rlkit (rlkit_custom.py:437).  Acting runs on the host from this object's
    parameters, which the trainer refreshes once per training block."""

    LOG_SIG_MAX, LOG_SIG_MIN = 2.0, -20.0

    def __init__(self, hidden_sizes, obs_dim, action_dim, std=None, init_w=1e-3, **kwargs):
        assert std is None, "fixed-std policies are not used by the benchmark"
        super().__init__(hidden_sizes, [("last_fc", action_dim), ("last_fc_log_std", action_dim)], obs_dim,
                         init_w, kwargs.get("rs"), kwargs.get("b_init_value", 0.1))
        self.obs_dim, self.action_dim = int(obs_dim), int(action_dim)
        # exploration noise of get_action: rlkit draws it from torch's global generator (seeded at train.py:113); without
        # torch a private stream keyed by the process seed -- two runs with different seeds explore differently either way
        self._noise = kwargs.get("noise") or (TorchGlobalStream() if torch_is_loaded()
                                              else np.random.RandomState(process_seed() & 0xFFFFFFFF))

    def _trunk(self, obs):
        h = np.asarray(obs, np.float32)
        for n in list(self.layers)[:-2]:
            w, b = self.layers[n]
            h = np.maximum(h @ w.T + b, 0)
        wm, bm = self.layers["last_fc"]
        ws, bs = self.layers["last_fc_log_std"]
        return h @ wm.T + bm, np.clip(h @ ws.T + bs, self.LOG_SIG_MIN, self.LOG_SIG_MAX)

    def get_actions(self, obs_np, deterministic=False):
        obs = np.ascontiguousarray(np.atleast_2d(obs_np), dtype=np.float32)
        eps = None if deterministic else self._noise.standard_normal((obs.shape[0], self.action_dim)).astype(np.float32)
        tr = self._trainer
        if tr is not None and getattr(tr, "_h", None) is not None and tr.policy is self:
            # the library's acting entry (sac_policy_act: host forward from the policy mirrored D2H once per
            # training block) -- ONE implementation of the acting path once a trainer owns the weights
            return tr.policy_act(obs, deterministic, eps)
        mean, log_std = self._trunk(obs)           # holder without a trainer (no device state yet)
        if deterministic:
            return np.tanh(mean)
        return np.tanh(mean + np.exp(log_std) * eps)

    def get_action(self, obs_np, deterministic=False):
        return self.get_actions(np.asarray(obs_np)[None], deterministic=deterministic)[0, :], {}

    def reset(self):
        pass


class MakeDeterministic:
    """``MakeDeterministic(stochastic_policy)``: action = tanh(mean)."""

    def __init__(self, stochastic_policy):
        self.stochastic_policy = stochastic_policy

    def get_action(self, observation):
        return self.stochastic_policy.get_action(observation, deterministic=True)

    def reset(self):
        pass

    def to(self, device):
        return self

    def train(self, mode=True):
        return self

    def cuda(self, device=None):
        return self

    def cpu(self):
        return self

    def eval(self):
        return self


class TanhMlpPolicy(_Mlp):
    """``TanhMlpPolicy(input_size=, output_size=, hidden_sizes=)`` (rlkit_utils.py:108-117, the TD3 actor and its
    target): relu MLP with a tanh output; ``get_action(obs_np)`` returns ``(action, {})``."""

    def __init__(self, hidden_sizes, output_size, input_size, init_w=1e-3, **kwargs):
        super().__init__(hidden_sizes, [("last_fc", output_size)], input_size, init_w, kwargs.get("rs"),
                         kwargs.get("b_init_value", 0.1))
        self.obs_dim, self.action_dim = int(input_size), int(output_size)

    def get_actions(self, obs_np):
        obs = np.ascontiguousarray(np.atleast_2d(obs_np), dtype=np.float32)
        tr = self._trainer
        if tr is not None and getattr(tr, "_h", None) is not None and tr.policy is self:
            return tr.policy_act(obs, True, None)  # sac_policy_act (TD3 handles: tanh(last_fc))
        h = obs
        names = list(self.layers)
        for n in names[:-1]:
            w, b = self.layers[n]
            h = np.maximum(h @ w.T + b, 0)
        w, b = self.layers[names[-1]]
        return np.tanh(h @ w.T + b)

    def get_action(self, obs_np):
        return self.get_actions(np.asarray(obs_np)[None])[0, :], {}

    def reset(self):
        pass


class GaussianStrategy:
    """rlkit GaussianStrategy(action_space, max_sigma, min_sigma) with constant sigma as the reference uses it
    (rlkit_utils.py:118-122: max_sigma = min_sigma = 0.1): action + N(0, sigma), clipped to the action box."""

    def __init__(self, action_space=None, max_sigma=1.0, min_sigma=None, decay_period=1000000, low=-1.0, high=1.0,
                 seed=None):
        self._max_sigma = float(max_sigma)
        self._min_sigma = float(max_sigma if min_sigma is None else min_sigma)
        self._decay_period = int(decay_period)
        self.low = getattr(action_space, "low", low)
        self.high = getattr(action_space, "high", high)
        # rlkit: np.random.randn -- the process-wide NumPy stream (seeded at train.py:112); an explicit seed = private stream
        self._rs = np.random if seed is None else np.random.RandomState(seed)

    def __getstate__(self):
        st = dict(self.__dict__)
        if st["_rs"] is np.random:
            st["_rs"] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        if self._rs is None:
            self._rs = np.random

    def get_action_from_raw_action(self, action, t=None):
        frac = min(1.0, (t or 0) * 1.0 / self._decay_period)
        sigma = self._max_sigma - (self._max_sigma - self._min_sigma) * frac
        return np.clip(action + self._rs.standard_normal(len(action)) * sigma, self.low, self.high).astype(np.float32)


class PolicyWrappedWithExplorationStrategy:
    """``PolicyWrappedWithExplorationStrategy(exploration_strategy=, policy=)`` (rlkit_utils.py:123-126)."""

    def __init__(self, exploration_strategy, policy):
        self.es, self.policy, self.t = exploration_strategy, policy, 0

    def get_action(self, *args, **kwargs):
        action, info = self.policy.get_action(*args, **kwargs)
        return self.es.get_action_from_raw_action(action, self.t), info

    def reset(self):
        self.policy.reset()

    def set_num_steps_total(self, t):
        self.t = t
