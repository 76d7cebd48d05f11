"""TD3Trainer with rlkit's constructor signature over the C ABI (td3_trainer_create + the shared sac_* entry points).

Reference assembly: /root/reference/util/rlkit_utils.py:107-135 (TanhMlpPolicy x2, GaussianStrategy,
PolicyWrappedWithExplorationStrategy, TD3Trainer(policy=, qf1=, qf2=, target_qf1=, target_qf2=, target_policy=,
**trainer_kwargs)); kwargs: /root/reference/scripts/train.py:38-47."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import TD3_DIAG_NAMES, TD3_NET_IDS, Td3Config
from .sac import SACTrainer


class TD3Trainer(SACTrainer):
    NETS = TD3_NET_IDS
    TRAINED = ("policy", "qf1", "qf2")

    def __init__(self, policy=None, qf1=None, qf2=None, target_qf1=None, target_qf2=None, target_policy=None,
                 target_policy_noise=0.2, target_policy_noise_clip=0.5, discount=0.99, reward_scale=1.0,
                 policy_learning_rate=1e-3, qf_learning_rate=1e-3, policy_and_target_update_period=2, tau=0.005,
                 qf_criterion=None, optimizer_class=None, batch_size=None, noise_seed=None, device=0):
        assert optimizer_class is None and qf_criterion is None, "only Adam / MSELoss (rlkit's defaults) are implemented"
        self.target_policy = target_policy
        self.target_policy_noise, self.target_policy_noise_clip = float(target_policy_noise), float(target_policy_noise_clip)
        self.policy_and_target_update_period, self.tau = int(policy_and_target_update_period), float(tau)
        self.policy_learning_rate, self.qf_learning_rate = float(policy_learning_rate), float(qf_learning_rate)
        super().__init__(env=None, policy=policy, qf1=qf1, qf2=qf2, target_qf1=target_qf1, target_qf2=target_qf2,
                         discount=discount, reward_scale=reward_scale, policy_lr=policy_learning_rate,
                         qf_lr=qf_learning_rate, soft_target_tau=tau, target_update_period=1,
                         use_automatic_entropy_tuning=False, target_entropy=0.0, batch_size=batch_size,
                         noise_seed=noise_seed, device=device)

    def _new_handle(self, batch):
        hp, hq = self._hidden("policy"), self._hidden("qf1")
        cfg = Td3Config(self.obs_dim, self.act_dim, 256, batch, self.discount, self.reward_scale,
                        self.policy_learning_rate, self.qf_learning_rate, self.tau, self.target_policy_noise,
                        self.target_policy_noise_clip, self.policy_and_target_update_period, self.noise_seed,
                        self.device, 0, (C.c_int32 * 2)(0, 0), (C.c_int32 * 2)(0, 0))
        h = C.c_void_p()
        _lib.check(self._lib.td3_trainer_create_mlp(C.byref(h), C.byref(cfg), (C.c_int32 * len(hp))(*hp), len(hp),
                                                    (C.c_int32 * len(hq))(*hq), len(hq)), "td3_trainer_create_mlp")
        return h

    def _create(self, batch):
        if self._hidden("target_policy") != self._hidden("policy"):
            raise RuntimeError("policy and target_policy must share their hidden_sizes (rlkit_utils.py:108-117 builds them so)")
        super()._create(batch)

    @property
    def networks(self):
        return [self.policy, self.qf1, self.qf2, self.target_qf1, self.target_qf2, self.target_policy]

    def train(self, np_batch, eps=None):
        """eps: the (B, A) N(0,1) draw of the target-policy smoothing noise (None: device stream)."""
        return super().train(np_batch, eps=None if eps is None else (None, eps))

    def _record(self, diag):
        if self._need_to_update_eval_statistics:
            self._need_to_update_eval_statistics = False
            for i, name in enumerate(TD3_DIAG_NAMES):
                if name != "Actor Loss":
                    self.eval_statistics[name] = float(diag[i])

    def get_snapshot(self):
        snap = super().get_snapshot()
        snap["target_policy"] = self.target_policy
        snap["trained_policy"] = self.policy            # rlkit TD3Trainer.get_snapshot key
        return snap
