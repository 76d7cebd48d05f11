"""Own counterpart of the reference's experiment assembly + epoch loop for boxes without
rlkit / robosuite (the GPU box): variant.json -> nets, trainer, HBM replay buffer -> epochs.

Mirrors /root/reference/util/rlkit_utils.py:31-165 (``experiment``) and
/root/reference/util/rlkit_custom.py:199-301 (``_train`` / ``_log_stats``): prefill, then per
epoch {eval collection, exploration collection, add_paths, num_trains_per_train_loop x
(random_batch; train), end_epoch + one progress.csv row with the reference's column names}.
Environment stepping stays on the host; with no robosuite here the env is a synthetic stand-in
with the task's observation/action sizes (SURVEY.md section 8d synthetic data)."""
from __future__ import annotations

import csv
import os
import time
from collections import OrderedDict

import numpy as np

from .checkpoint import checkpoint_exists, load_checkpoint, save_checkpoint
from .networks import (FlattenMlp, GaussianStrategy, MakeDeterministic, PolicyWrappedWithExplorationStrategy,
                       TanhGaussianPolicy, TanhMlpPolicy)
from .replay_buffer import EnvReplayBuffer
from .sac import SACTrainer
from .td3 import TD3Trainer
from .variant import env_dims, validate


class SyntheticEnv:
    """Host-side stand-in for NormalizedBoxEnv(GymWrapper(robosuite env)): shaped reward in [0,1],
    never terminates (ignore_done=True in every shipped variant)."""

    def __init__(self, obs_dim, action_dim, horizon=500, seed=0):
        self.obs_dim, self.action_dim, self.horizon = obs_dim, action_dim, horizon
        self._rs = np.random.RandomState(seed)
        self._goal = self._rs.normal(0, 0.5, action_dim)
        self._t = 0

    def reset(self):
        self._t = 0
        self._obs = self._rs.normal(0, 0.5, self.obs_dim)
        return self._obs

    def step(self, action):
        self._t += 1
        rew = float(np.exp(-np.sum((np.asarray(action) - np.tanh(self._goal)) ** 2)))
        self._obs = 0.9 * self._obs + self._rs.normal(0, 0.2, self.obs_dim)
        return self._obs, rew, False, {}


def rollout(env, policy, max_path_length):
    """rlkit rollout contract (rlkit_custom.py:487-495 path dict)."""
    obs_l, act_l, rew_l, nobs_l, term_l = [], [], [], [], []
    o = env.reset()
    policy.reset()
    for _ in range(max_path_length):
        a, _ = policy.get_action(o)
        no, r, d, _ = env.step(a)
        obs_l.append(o); act_l.append(a); rew_l.append(r); nobs_l.append(no); term_l.append(d)
        o = no
        if d:
            break
    n = len(obs_l)
    return dict(observations=np.array(obs_l), actions=np.array(act_l), rewards=np.array(rew_l).reshape(n, 1),
                next_observations=np.array(nobs_l), terminals=np.array(term_l).reshape(n, 1),
                agent_infos=[{}] * n, env_infos=[{}] * n)


class PathCollector:
    """MdpPathCollector.collect_new_paths: the last path is clipped to the remaining step budget and
    kept unless discard_incomplete_paths (pins 'exploration/num paths total' = 12 at epoch 0)."""

    def __init__(self, env, policy):
        self.env, self.policy = env, policy
        self.num_steps_total, self.num_paths_total, self.epoch_paths = 0, 0, []

    def collect_new_paths(self, max_path_length, num_steps, discard_incomplete_paths):
        paths, collected = [], 0
        while collected < num_steps:
            mpl = min(max_path_length, num_steps - collected)
            path = rollout(self.env, self.policy, mpl)
            plen = len(path["actions"])
            if plen != max_path_length and not path["terminals"][-1] and discard_incomplete_paths:
                break
            collected += plen
            paths.append(path)
        self.num_paths_total += len(paths)
        self.num_steps_total += collected
        self.epoch_paths.extend(paths)
        return paths

    def end_epoch(self, epoch):
        self.epoch_paths = []

    def get_diagnostics(self):
        return OrderedDict([("num steps total", self.num_steps_total), ("num paths total", self.num_paths_total)])


def _stats(name, x):
    x = np.asarray(x, dtype=np.float64).ravel()
    return OrderedDict([(name + " Mean", float(np.mean(x))), (name + " Std", float(np.std(x))),
                        (name + " Max", float(np.max(x))), (name + " Min", float(np.min(x)))])


def path_information(paths, prefix, expl_len=None):
    d = OrderedDict()
    d.update(_stats(prefix + "path length", [len(p["actions"]) for p in paths]))
    d.update(_stats(prefix + "Rewards", np.vstack([p["rewards"] for p in paths])))
    d.update(_stats(prefix + "Returns", [np.sum(p["rewards"]) for p in paths]))
    if expl_len is not None:      # evaluation/ExplReturns: return truncated at the exploration horizon
        d.update(_stats(prefix + "ExplReturns", [np.sum(p["rewards"][:expl_len]) for p in paths]))
    d.update(_stats(prefix + "Actions", np.vstack([p["actions"] for p in paths])))
    d[prefix + "Num Paths"] = len(paths)
    d[prefix + "Average Returns"] = float(np.mean([np.sum(p["rewards"]) for p in paths]))
    return d


def _rs_pack(rs):
    st = rs.get_state()
    return dict(key=[int(x) for x in st[1]], pos=int(st[2]), has_gauss=int(st[3]), cached=float(st[4]))


def _rs_unpack(rs, d):
    rs.set_state(("MT19937", np.asarray(d["key"], np.uint32), d["pos"], d["has_gauss"], d["cached"]))


def experiment(variant, log_dir=None, seed=1, obs_dim=None, action_dim=None, num_epochs=None, device=0,
               fused_loop=True, quiet=False, checkpoint_dir=None, resume=False):
    """variant.json -> training run.  Returns the list of progress rows (also written to
    <log_dir>/progress.csv when log_dir is given).

    checkpoint_dir: after every epoch the full run state (networks, Adam moments, entropy coefficient, step
    counters, replay buffer, sampling stream, host generators) is saved there (`time/saving (s)`); with
    resume=True a run picks up after the last saved epoch and continues bit for bit -- the reference's own
    snapshots (rlkit_custom.py:68-82) hold the networks only and cannot resume."""
    validate(variant)
    np.random.seed(seed)                                          # scripts/train.py:112 (args.seed, not variant seed)
    O, A = env_dims(variant["expl_environment_kwargs"], obs_dim, action_dim)
    ak, tk = variant["algorithm_kwargs"], variant["trainer_kwargs"]
    expl_env = SyntheticEnv(O, A, variant["expl_environment_kwargs"].get("horizon", 500), seed)
    eval_env = SyntheticEnv(O, A, variant["eval_environment_kwargs"].get("horizon", 500), seed + 1)
    # This driver's runs are a function of `seed` alone: initial weights come from np.random (seeded above), every noise
    # stream from `seed` -- whether or not torch happens to be loaded in the process (the reference's own scripts seed
    # torch as well, train.py:113, and the holders then follow torch's generator by themselves: networks.process_stream).
    qf1, qf2, tqf1, tqf2 = (FlattenMlp(input_size=O + A, output_size=1, rs=np.random, **variant["qf_kwargs"]) for _ in range(4))
    if variant.get("algorithm", "SAC") == "TD3":                  # rlkit_utils.py:107-135
        policy = TanhMlpPolicy(input_size=O, output_size=A, rs=np.random, **variant["policy_kwargs"])
        target_policy = TanhMlpPolicy(input_size=O, output_size=A, rs=np.random, **variant["policy_kwargs"])
        eval_policy = policy
        expl_policy = PolicyWrappedWithExplorationStrategy(
            exploration_strategy=GaussianStrategy(max_sigma=0.1, min_sigma=0.1, seed=seed), policy=policy)
        trainer = TD3Trainer(policy=policy, qf1=qf1, qf2=qf2, target_qf1=tqf1, target_qf2=tqf2, target_policy=target_policy,
                             batch_size=ak["batch_size"], noise_seed=seed, device=device, **tk)
        policy._noise = expl_policy.es._rs                        # (the generator a checkpoint saves as policy_noise)
    else:
        policy = TanhGaussianPolicy(obs_dim=O, action_dim=A, rs=np.random, noise=np.random.RandomState(seed),
                                    **variant["policy_kwargs"])
        eval_policy, expl_policy = MakeDeterministic(policy), policy
        trainer = SACTrainer(env=eval_env, policy=policy, qf1=qf1, qf2=qf2, target_qf1=tqf1, target_qf2=tqf2,
                             batch_size=ak["batch_size"], noise_seed=seed, device=device, **tk)
    buf = EnvReplayBuffer(variant["replay_buffer_size"], obs_dim=O, action_dim=A, device=device)
    expl, evalc = PathCollector(expl_env, expl_policy), PathCollector(eval_env, eval_policy)
    rows, t_start = [], time.time()
    writer, fh = None, None
    first_epoch = 0
    host_rngs = dict(policy_noise=policy._noise, expl_env=expl_env._rs, eval_env=eval_env._rs)
    if resume and checkpoint_dir and checkpoint_exists(checkpoint_dir):
        extra = load_checkpoint(checkpoint_dir, trainer, buf)
        first_epoch = int(extra["epoch"]) + 1
        _rs_unpack(np.random, extra["np_random"])
        for k, rs in host_rngs.items():
            _rs_unpack(rs, extra[k])
        expl.num_steps_total, expl.num_paths_total = extra["expl_totals"]
        evalc.num_steps_total, evalc.num_paths_total = extra["eval_totals"]
        trainer.end_epoch(first_epoch - 1)
    elif ak.get("min_num_steps_before_training", 0) > 0:
        buf.add_paths(expl.collect_new_paths(ak["expl_max_path_length"], ak["min_num_steps_before_training"], False))
        expl.end_epoch(-1)
    for epoch in range(first_epoch, num_epochs if num_epochs is not None else ak["num_epochs"]):
        t0 = time.time()
        evalc.collect_new_paths(ak["eval_max_path_length"], ak["num_eval_steps_per_epoch"], True)
        t1 = time.time()
        new_paths = expl.collect_new_paths(ak["expl_max_path_length"], ak["num_expl_steps_per_train_loop"], False)
        t2 = time.time()
        buf.add_paths(new_paths)
        t3 = time.time()
        # (the buffer samples np.random itself -- bound to its state words: nothing to hand over before or after the block)
        n_train = ak["num_trains_per_train_loop"]
        if fused_loop:
            trainer.train_loop(buf, n_train, batch_size=ak["batch_size"])
        else:
            for _ in range(n_train):
                trainer.train(buf.random_batch(ak["batch_size"]))
        t4 = time.time()
        row = OrderedDict()
        row.update(("replay_buffer/" + k, v) for k, v in buf.get_diagnostics().items())
        row.update(("trainer/" + k, v) for k, v in trainer.get_diagnostics().items())
        row.update(("exploration/" + k, v) for k, v in expl.get_diagnostics().items())
        row.update(path_information(expl.epoch_paths, "exploration/"))
        row.update(("evaluation/" + k, v) for k, v in evalc.get_diagnostics().items())
        row.update(path_information(evalc.epoch_paths, "evaluation/", ak["expl_max_path_length"]))
        trainer.end_epoch(epoch); buf.end_epoch(epoch); expl.end_epoch(epoch); evalc.end_epoch(epoch)
        t5 = time.time()
        if checkpoint_dir:
            extra = dict(epoch=epoch, seed=seed, np_random=_rs_pack(np.random),
                         expl_totals=[expl.num_steps_total, expl.num_paths_total],
                         eval_totals=[evalc.num_steps_total, evalc.num_paths_total])
            extra.update({k: _rs_pack(rs) for k, rs in host_rngs.items()})
            save_checkpoint(checkpoint_dir, trainer, buf, extra)
        t6 = time.time()
        row["time/data storing (s)"] = t3 - t2
        row["time/evaluation sampling (s)"] = t1 - t0
        row["time/exploration sampling (s)"] = t2 - t1
        row["time/logging (s)"] = t5 - t4
        row["time/saving (s)"] = t6 - t5
        row["time/training (s)"] = t4 - t3
        row["time/epoch (s)"] = t6 - t0
        row["time/total (s)"] = t6 - t_start
        row["Epoch"] = epoch
        rows.append(row)
        if log_dir is not None:
            if writer is None:
                os.makedirs(log_dir, exist_ok=True)
                appending = first_epoch > 0 and os.path.exists(os.path.join(log_dir, "progress.csv"))
                fh = open(os.path.join(log_dir, "progress.csv"), "a" if appending else "w", newline="")
                writer = csv.DictWriter(fh, fieldnames=list(row.keys()))
                if not appending:
                    writer.writeheader()
            writer.writerow(row)
            fh.flush()
        if not quiet:
            print(f"epoch {epoch}: buffer {row['replay_buffer/size']}  QF1 {row['trainer/QF1 Loss']:.4f}  "
                  f"policy loss {row['trainer/Policy Loss']:.4f}  training {row['time/training (s)']:.3f}s "
                  f"({n_train / max(row['time/training (s)'], 1e-9):.0f} steps/s)", flush=True)
    if fh:
        fh.close()
    return rows
