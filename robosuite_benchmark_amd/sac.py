"""SACTrainer over libsac_hip.so (mirror of rlkit.torch.sac.sac.SACTrainer).

Reference call sites: /root/reference/util/rlkit_utils.py:98-106 (constructor kwargs =
variant['trainer_kwargs'], scripts/train.py:29-37), /root/reference/util/rlkit_custom.py:238
(``train(np_batch)``), :258 (``get_diagnostics``), :63,:70 (``get_snapshot``), :306-312
(``networks``), :291 (``reward_scale``).  Step semantics: SURVEY.md Appendix A."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _lib
from ._lib import DIAG_NAMES, NET_IDS, SacConfig
from .networks import process_seed


class SACTrainer:
    def __init__(self, env=None, policy=None, qf1=None, qf2=None, target_qf1=None, target_qf2=None,
                 discount=0.99, reward_scale=1.0, policy_lr=1e-3, qf_lr=1e-3, optimizer_class=None,
                 soft_target_tau=1e-2, target_update_period=1, plotter=None, render_eval_paths=False,
                 use_automatic_entropy_tuning=True, target_entropy=None, batch_size=None, noise_seed=None,
                 device=0):
        assert optimizer_class is None, "only Adam (rlkit's default) is implemented"
        self.env = env
        self.policy, self.qf1, self.qf2 = policy, qf1, qf2
        self.target_qf1, self.target_qf2 = target_qf1, target_qf2
        self.discount, self.reward_scale = float(discount), float(reward_scale)
        self.policy_lr, self.qf_lr = float(policy_lr), float(qf_lr)
        self.soft_target_tau, self.target_update_period = float(soft_target_tau), int(target_update_period)
        self.use_automatic_entropy_tuning = bool(use_automatic_entropy_tuning)
        self.obs_dim, self.act_dim = policy.obs_dim, policy.action_dim
        self.target_entropy = float(-self.act_dim if target_entropy is None else target_entropy)
        # rsample noise of the step: rlkit draws it from the generator torch.manual_seed(args.seed) seeded
        # (/root/reference/scripts/train.py:113); here a device counter-based stream keyed by that same process seed
        # unless the caller names one -- two runs with different seeds see different noise, as in the reference
        self.noise_seed = (process_seed() if noise_seed is None else int(noise_seed)) & 0xFFFFFFFFFFFFFFFF
        self.device = int(device)
        self.eval_statistics = OrderedDict()
        self._need_to_update_eval_statistics = True
        self._num_train_steps = 0
        self._lib = _lib.load()
        self._h, self._batch = None, None
        self._host_policy_stale = False
        self._saved_state = None
        for name in self.NETS:                   # (a holder finds its trainer: acting path, pickling)
            getattr(self, name)._trainer = self
        if batch_size is not None:
            self._create(int(batch_size))

    # ---- handle management -----------------------------------------------------------------
    NETS = NET_IDS                           # name -> C net id (TD3Trainer adds target_policy)

    def _new_handle(self, batch):
        hp, hq = self._hidden("policy"), self._hidden("qf1")
        cfg = SacConfig(self.obs_dim, self.act_dim, 256, batch, self.discount, self.reward_scale, self.policy_lr,
                        self.qf_lr, self.soft_target_tau, self.target_update_period,
                        int(self.use_automatic_entropy_tuning), self.target_entropy, self.noise_seed, self.device, 0,
                        (C.c_int32 * 2)(0, 0), (C.c_int32 * 2)(0, 0))
        h = C.c_void_p()
        _lib.check(self._lib.sac_trainer_create_mlp(C.byref(h), C.byref(cfg), (C.c_int32 * len(hp))(*hp), len(hp),
                                                    (C.c_int32 * len(hq))(*hq), len(hq)), "sac_trainer_create_mlp")
        return h

    MAX_HIDDEN_LAYERS, MAX_HIDDEN_UNITS = 7, 4096

    def _hidden(self, net):
        """variant['policy_kwargs'|'qf_kwargs']['hidden_sizes'] of a network family (arguments.py:98,104).  Two layers of at
        most 256 units (every shipped variant.json: [256, 256]) run on the fused kernels -- narrower layers exactly, as
        zero rows / columns of the 256-wide ones; any other depth / width runs the library's general step."""
        hs = [int(h) for h in getattr(self, net).hidden_sizes]
        if not 1 <= len(hs) <= self.MAX_HIDDEN_LAYERS or not all(1 <= h <= self.MAX_HIDDEN_UNITS for h in hs):
            raise RuntimeError(f"hidden_sizes {hs} unsupported: 1..{self.MAX_HIDDEN_LAYERS} hidden layers of "
                               f"1..{self.MAX_HIDDEN_UNITS} units")
        return hs

    def _create(self, batch):
        if not all(self._hidden(n) == self._hidden("qf1") for n in ("qf2", "target_qf1", "target_qf2")):
            raise RuntimeError("the four Q networks must share their hidden_sizes (rlkit_utils.py:64-83 builds them so)")
        h = self._new_handle(batch)
        state = self._export_state() if self._h else self._saved_state
        self._saved_state = None
        self._destroy()
        self._h, self._batch = h, batch
        if state is None:
            for name in self.NETS:
                self._set_params(name, getattr(self, name).flat())
        else:
            self._import_state(state)

    def _destroy(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.sac_trainer_destroy(h)

    def __del__(self):
        self._destroy()

    # A trainer can be pickled too (rlkit's snapshot never holds one, but callers copy / checkpoint whole algorithms):
    # the device state -- parameters, Adam moments, entropy coefficient, counters -- travels as host arrays and goes
    # back into a fresh handle on the first step after loading.
    def __getstate__(self):
        st = dict(self.__dict__)
        if self._h is not None:
            self.sync_networks_to_host()
            st["_saved_state"] = self._export_state()
        st["_h"], st["_lib"] = None, None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._lib = _lib.load()
        for name in self.NETS:
            getattr(self, name)._trainer = self

    def _set_params(self, name, flat):
        flat = _lib.f32(flat)
        _lib.check(self._lib.sac_set_params(self._h, self.NETS[name], _lib.ptr(flat), flat.size), "sac_set_params")

    def _get_params(self, name):
        n = int(self._lib.sac_param_count(self._h, self.NETS[name]))
        out = np.empty(n, np.float32)
        _lib.check(self._lib.sac_get_params(self._h, self.NETS[name], _lib.ptr(out), n), "sac_get_params")
        return out

    def _export_state(self):
        st = dict(params={k: self._get_params(k) for k in self.NETS}, opt={})
        for k in ("policy", "qf1", "qf2"):
            n = st["params"][k].size
            m, v = np.empty(n, np.float32), np.empty(n, np.float32)
            _lib.check(self._lib.sac_get_opt_state(self._h, NET_IDS[k], _lib.ptr(m), _lib.ptr(v), n),
                       "sac_get_opt_state")
            st["opt"][k] = (m, v)
        sc = np.zeros(6, np.float64)
        _lib.check(self._lib.sac_get_scalars(self._h, _lib.ptr(sc)), "sac_get_scalars")
        st["scalars"] = sc
        return st

    def _import_state(self, st):
        for k, v in st["params"].items():
            self._set_params(k, v)
        for k, (m, v) in st["opt"].items():
            _lib.check(self._lib.sac_set_opt_state(self._h, NET_IDS[k], _lib.ptr(_lib.f32(m)), _lib.ptr(_lib.f32(v)),
                                                   m.size), "sac_set_opt_state")
        sc = np.ascontiguousarray(st["scalars"], np.float64)
        _lib.check(self._lib.sac_set_scalars(self._h, _lib.ptr(sc)), "sac_set_scalars")
        self._host_policy_stale = True           # the acting copy on the host is refreshed at its next use

    state_dict, load_state_dict = _export_state, _import_state

    # ---- rlkit Trainer interface -------------------------------------------------------------
    @property
    def networks(self):
        return [self.policy, self.qf1, self.qf2, self.target_qf1, self.target_qf2]

    def train(self, np_batch, eps=None):
        """TorchTrainer.train: np_to_pytorch_batch + train_from_torch (one gradient step).

        A batch that random_batch() left on the device (DeviceBatch, not read by anyone) is trained on where it
        is: no host copy and no synchronisation; like rlkit the call then returns None, and the step's
        diagnostics are fetched only when the epoch's statistics are due.  Host batches return the diagnostics."""
        self._num_train_steps += 1
        if eps is None and getattr(np_batch, "on_device", False) and np_batch._buffer._h is not None:
            B = np_batch._batch_size
            if self._h is None or B != self._batch:
                self._create(B)
            diag = np.empty(_lib.SAC_DIAG_N, np.float32) if self._need_to_update_eval_statistics else None
            _lib.check(self._lib.sac_step_device(self._h, np_batch._buffer._h, np_batch._token, _lib.ptr(diag)),
                       "sac_step_device")
            self._host_policy_stale = True
            if diag is not None:
                self._record(diag)
            return diag
        obs = _lib.f32(np_batch["observations"])
        B = obs.shape[0]
        if self._h is None or B != self._batch:
            self._create(B)
        act, nobs = _lib.f32(np_batch["actions"]), _lib.f32(np_batch["next_observations"])
        rew = _lib.f32(np.asarray(np_batch["rewards"]).reshape(B))
        term = _lib.f32(np.asarray(np_batch["terminals"]).reshape(B))
        e1 = e2 = None
        if eps is not None:
            e1, e2 = (None if e is None else _lib.f32(e) for e in eps)
        diag = np.empty(_lib.SAC_DIAG_N, np.float32)
        _lib.check(self._lib.sac_step(self._h, _lib.ptr(obs), _lib.ptr(act), _lib.ptr(rew), _lib.ptr(term),
                                      _lib.ptr(nobs), _lib.ptr(e1), _lib.ptr(e2), _lib.ptr(diag)), "sac_step")
        self._host_policy_stale = True
        self._record(diag)
        return diag

    def train_loop(self, replay_buffer, n_steps, batch_size=None):
        """The fused hot loop of rlkit_custom.py:234-238 (n_steps x {random_batch; train})."""
        B = int(batch_size or self._batch)
        if self._h is None or B != self._batch:
            self._create(B)
        first, last = np.empty(_lib.SAC_DIAG_N, np.float32), np.empty(_lib.SAC_DIAG_N, np.float32)
        _lib.check(self._lib.sac_train_loop(self._h, replay_buffer._h, int(n_steps), _lib.ptr(first), _lib.ptr(last)),
                   "sac_train_loop")
        self._num_train_steps += int(n_steps)
        self._host_policy_stale = True
        self._record(first)
        return first, last

    def profile_loop(self, replay_buffer, n_steps, batch_size=None):
        """Instrumented pass: per-kernel mean launch duration (ms) from HIP events."""
        B = int(batch_size or self._batch)
        if self._h is None or B != self._batch:
            self._create(B)
        ms = np.zeros(9, np.float32)
        _lib.check(self._lib.sac_profile_loop(self._h, replay_buffer._h, int(n_steps), _lib.ptr(ms)),
                   "sac_profile_loop")
        self._num_train_steps += int(n_steps)
        self._host_policy_stale = True
        names = ["k_mt_randint", "k_gather", "k_fwd_a", "k_fwd_b", "k_bwd", "reserved", "k_dw_adam",
                 "event_pair", "steps_wall"]
        mode = self.fused_mode()
        if mode == 1:                # the fused step: k_abc = launches A + B + C in one (the next two slots read 0)
            names[2], names[3], names[4] = "k_fwd_abc", "fused_b", "fused_c"
        elif mode == 2:              # column split 1: k_chain = launches A + B in one (the next slot reads 0)
            names[2], names[3] = "k_chain", "chain_b"
        elif mode == 4:              # ... with the backward blocks inside too (the next two slots read 0)
            names[2], names[3], names[4] = "k_chain_bwd", "chain_b", "chain_c"
        return OrderedDict(zip(names, [float(x) for x in ms]))

    def fused_mode(self):
        """0: four launches per step; 1: the fused step, k_abc + k_dw_adam; 2: k_chain + k_bwd + k_dw_adam (batch >= 1024);
        3: the general step (hidden_sizes beyond two layers of at most 256 units); 4: k_chain8 with the backward blocks inside
        + k_dw_adam (batch 1024, SAC_CHAIN_BWD=1)."""
        return int(self._lib.sac_trainer_step_kind(self._h)) if self._h is not None else 0

    def is_fused(self):
        return self.fused_mode() == 1

    def loop_timing_ms(self):
        v = [C.c_float() for _ in range(4)]
        _lib.check(self._lib.sac_last_loop_ms(self._h, *[C.byref(x) for x in v]), "sac_last_loop_ms")
        return dict(total=v[0].value, sample=v[1].value, gather=v[2].value, steps=v[3].value)

    def _record(self, diag):
        if self._need_to_update_eval_statistics:
            self._need_to_update_eval_statistics = False
            for i, name in enumerate(DIAG_NAMES):
                if name != "Actor Loss":          # not an rlkit column
                    self.eval_statistics[name] = float(diag[i])

    def get_diagnostics(self):
        return self.eval_statistics

    def end_epoch(self, epoch):
        self._need_to_update_eval_statistics = True

    def policy_act(self, obs, deterministic, eps):
        """policy.get_actions through the C ABI: sac_policy_act per observation row (rlkit_custom.py:437 acts on one
        observation at a time).  The library mirrors the policy D2H on the first call after a training block."""
        obs = _lib.f32(obs)
        n, A = obs.shape[0], self.act_dim
        out = np.empty((n, A), np.float32)
        e = None if eps is None else _lib.f32(eps)
        for i in range(n):
            _lib.check(self._lib.sac_policy_act(self._h, obs[i].ctypes.data_as(C.c_void_p), int(bool(deterministic)),
                                                None if e is None else e[i].ctypes.data_as(C.c_void_p),
                                                out[i].ctypes.data_as(C.c_void_p)), "sac_policy_act")
        return out

    def refresh_host_policy(self):
        """Mirror the trained policy D2H once per training block (acting stays on the host)."""
        if self._h is not None and self._host_policy_stale:
            self.policy.load_flat(self._get_params("policy"))
            self._host_policy_stale = False

    def sync_holder_to_host(self, holder):
        """The trained weights of ONE network holder (whichever of this trainer's nets it is) into its host arrays."""
        for name in self.NETS:
            if getattr(self, name) is holder:
                holder.load_flat(self._get_params(name))
                if name == "policy":
                    self._host_policy_stale = False

    def sync_networks_to_host(self):
        for name in self.NETS:
            getattr(self, name).load_flat(self._get_params(name))
        self._host_policy_stale = False

    def get_snapshot(self):
        if self._h is not None:
            self.sync_networks_to_host()
        return dict(policy=self.policy, qf1=self.qf1, qf2=self.qf2, target_qf1=self.target_qf1,
                    target_qf2=self.target_qf2)

    def debug_fetch(self, name, n):
        out = np.empty(int(n), np.float32)
        got = self._lib.sac_debug_fetch(self._h, name.encode(), _lib.ptr(out), out.size)
        _lib.check(int(got), "sac_debug_fetch")
        return out[:got]
