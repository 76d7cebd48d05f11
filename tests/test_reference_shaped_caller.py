"""The import-only drop-in, driven the way the reference drives it.

This file restates, call for call, what the reference does to the duck types once `util/rlkit_utils.py`'s rlkit imports
are replaced by `robosuite_benchmark_amd` -- nothing of the reference is imported or read:

  * assembly             /root/reference/util/rlkit_utils.py:64-106,139-164  (`FlattenMlp(input_size=, output_size=,
                         **qf_kwargs)` x4, `TanhGaussianPolicy(obs_dim=, action_dim=, **policy_kwargs)`,
                         `MakeDeterministic`, `SACTrainer(env=, policy=, qf1=, ..., **trainer_kwargs)` -- no batch_size,
                         no noise_seed --, `EnvReplayBuffer(variant['replay_buffer_size'], expl_env)`, `algorithm.to()`)
  * seeding              /root/reference/scripts/train.py:112-113  (`np.random.seed(args.seed)`, `torch.manual_seed(...)`)
  * the epoch loop       /root/reference/util/rlkit_custom.py:199-242  (prefill, collect, add_paths, training_mode,
                         num_trains_per_train_loop x {random_batch; train})
  * the epoch end        /root/reference/util/rlkit_custom.py:54-82  (`_get_snapshot()` -> `logger.save_itr_params` =
                         torch.save of the trainer's networks and both collectors' policies; get_diagnostics; end_epoch)
  * mode switches        /root/reference/util/rlkit_custom.py:304-312  (`net.to(device)`, `net.train(mode)`)
  * the snapshot's user  /root/reference/util/rlkit_utils.py:173-174,241-250  (`torch.load` -> `data['evaluation/policy']`,
                         `policy.stochastic_policy.cuda()`, `policy.get_action(obs)`)
"""
import io
import json
import os
import pickle

import numpy as np
import pytest
import torch

from robosuite_benchmark_amd import (EnvReplayBuffer, FlattenMlp, GaussianStrategy, MakeDeterministic,
                                      PolicyWrappedWithExplorationStrategy, SACTrainer, TanhGaussianPolicy,
                                      TanhMlpPolicy, TD3Trainer)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "tests", "golden", "Lift-Panda-OSC-POSE-SEED17.variant.json")


class _Box:
    def __init__(self, n):
        self.low, self.high = -np.ones(n), np.ones(n)


class BoxEnv:
    """What the shim sees of NormalizedBoxEnv(GymWrapper(robosuite env)): two Box spaces, reset/step; like robosuite it
    draws its resets from the process-wide np.random stream (a host consumer between training blocks)."""

    def __init__(self, obs_dim, action_dim, mirror=None):
        self.observation_space, self.action_space = _Box(obs_dim), _Box(action_dim)
        self._mirror = mirror                          # a RandomState the test advances in lockstep with np.random

    def reset(self):
        o = np.random.normal(0, 0.5, self.observation_space.low.size)
        if self._mirror is not None:
            assert np.array_equal(o, self._mirror.normal(0, 0.5, self.observation_space.low.size))
        self._o = o
        return o

    def step(self, a):
        self._o = 0.9 * self._o + 0.1 * np.tanh(np.resize(a, self._o.size))
        return self._o, float(np.exp(-np.sum(np.square(a)))), False, {}


class PathCollector:
    """rlkit MdpPathCollector as the loop uses it: collect_new_paths / get_snapshot -> dict(env=, policy=) / end_epoch."""

    def __init__(self, env, policy):
        self._env, self._policy = env, policy

    def collect_new_paths(self, max_path_length, num_steps, discard_incomplete_paths):
        paths, n = [], 0
        while n < num_steps:
            T = min(max_path_length, num_steps - n)
            o = self._env.reset()
            self._policy.reset()
            obs, act, rew, nobs, term = [], [], [], [], []
            for _ in range(T):
                a, _info = self._policy.get_action(o)
                no, r, d, _ = self._env.step(a)
                obs.append(o); act.append(a); rew.append(r); nobs.append(no); term.append(d)
                o = no
            paths.append(dict(observations=np.array(obs), actions=np.array(act), rewards=np.array(rew).reshape(-1, 1),
                              next_observations=np.array(nobs), terminals=np.array(term).reshape(-1, 1),
                              agent_infos=[{}] * T, env_infos=[{}] * T))
            n += T
        return paths

    def get_snapshot(self):
        return dict(env=self._env, policy=self._policy)

    def end_epoch(self, epoch):
        pass


def assemble(variant, obs_dim, action_dim, mirror=None, agent="SAC"):
    """rlkit_utils.py:56-164 with the shim's classes behind rlkit's names."""
    expl_env, eval_env = BoxEnv(obs_dim, action_dim, mirror), BoxEnv(obs_dim, action_dim, mirror)
    obs_dim = expl_env.observation_space.low.size
    action_dim = eval_env.action_space.low.size
    qf1 = FlattenMlp(input_size=obs_dim + action_dim, output_size=1, **variant["qf_kwargs"])
    qf2 = FlattenMlp(input_size=obs_dim + action_dim, output_size=1, **variant["qf_kwargs"])
    target_qf1 = FlattenMlp(input_size=obs_dim + action_dim, output_size=1, **variant["qf_kwargs"])
    target_qf2 = FlattenMlp(input_size=obs_dim + action_dim, output_size=1, **variant["qf_kwargs"])
    if agent == "SAC":
        expl_policy = TanhGaussianPolicy(obs_dim=obs_dim, action_dim=action_dim, **variant["policy_kwargs"])
        eval_policy = MakeDeterministic(expl_policy)
        trainer = SACTrainer(env=eval_env, policy=expl_policy, qf1=qf1, qf2=qf2, target_qf1=target_qf1,
                             target_qf2=target_qf2, **variant["trainer_kwargs"])
    else:
        eval_policy = TanhMlpPolicy(input_size=obs_dim, output_size=action_dim, **variant["policy_kwargs"])
        target_policy = TanhMlpPolicy(input_size=obs_dim, output_size=action_dim, **variant["policy_kwargs"])
        es = GaussianStrategy(action_space=expl_env.action_space, max_sigma=0.1, min_sigma=0.1)
        expl_policy = PolicyWrappedWithExplorationStrategy(exploration_strategy=es, policy=eval_policy)
        trainer = TD3Trainer(policy=eval_policy, qf1=qf1, qf2=qf2, target_qf1=target_qf1, target_qf2=target_qf2,
                             target_policy=target_policy, **variant["trainer_kwargs"])
    return dict(expl_env=expl_env, eval_env=eval_env, trainer=trainer, expl_policy=expl_policy, eval_policy=eval_policy,
                expl=PathCollector(expl_env, expl_policy), evalc=PathCollector(eval_env, eval_policy))


def get_snapshot(trainer, expl, evalc, replay_buffer=None):
    """rlkit_custom.py:68-82."""
    snapshot = {}
    for k, v in trainer.get_snapshot().items():
        snapshot["trainer/" + k] = v
    for k, v in expl.get_snapshot().items():
        if k == "env":
            continue
        snapshot["exploration/" + k] = v
    for k, v in evalc.get_snapshot().items():
        if k == "env":
            continue
        snapshot["evaluation/" + k] = v
    if replay_buffer is not None:
        for k, v in replay_buffer.get_snapshot().items():
            snapshot["replay_buffer/" + k] = v
    return snapshot


def load_variant():
    return json.load(open(VARIANT))


TD3_KWARGS = dict(discount=0.99, reward_scale=1.0, policy_learning_rate=1e-3, qf_learning_rate=1e-3,
                  policy_and_target_update_period=2, tau=0.005, target_policy_noise=0.2,
                  target_policy_noise_clip=0.5)          # scripts/train.py:38-47


# ---------------------------------------------------------------------------------------------------------------
# CPU: assembly, mode switches, the epoch end's pickling and the snapshot's consumer (no device state yet)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("agent", ["SAC", "TD3"])
def test_epoch_end_snapshot_pickles_and_the_loaded_policy_acts(tmp_path, agent):
    v = load_variant()
    if agent == "TD3":
        v = dict(v, trainer_kwargs=TD3_KWARGS)
    np.random.seed(17); torch.manual_seed(17)
    a = assemble(v, 42, 7, agent=agent)
    tr = a["trainer"]
    for net in tr.networks:                                # algorithm.to(ptu.device); training_mode(True / False)
        assert net.to("cuda:0") is net
        net.train(True); net.train(False)
    snap = get_snapshot(tr, a["expl"], a["evalc"])
    assert {"trainer/policy", "trainer/qf1", "trainer/qf2", "trainer/target_qf1", "trainer/target_qf2",
            "exploration/policy", "evaluation/policy"} <= set(snap)
    pickle.dumps(snap)                                     # (what broke in round 2: a holder bound to a CDLL handle)
    path = tmp_path / "params.pkl"
    torch.save(snap, path)                                 # logger.save_itr_params(epoch, snapshot)
    data = torch.load(path, weights_only=False)            # rlkit_utils.py:173 (a file this test wrote itself)
    policy = data["evaluation/policy"]
    # rlkit_utils.py:250:  policy.cuda() if not isinstance(policy, MakeDeterministic) else policy.stochastic_policy.cuda()
    policy.cuda() if not isinstance(policy, MakeDeterministic) else policy.stochastic_policy.cuda()
    obs = np.random.RandomState(0).normal(0, 0.5, 42)
    act, info = policy.get_action(obs)
    want, _ = a["eval_policy"].get_action(obs)
    assert act.shape == (7,) and info == {} and np.array_equal(act, want)
    # the loaded holders are self-contained host networks: same weights, no binding to a trainer
    for k in ("policy", "qf1", "target_qf2"):
        live, loaded = getattr(tr, k), data["trainer/" + k]
        assert loaded._trainer is None and np.array_equal(loaded.flat(), live.flat())
    expl_loaded = data["exploration/policy"]
    a1, _ = expl_loaded.get_action(obs)
    assert a1.shape == (7,) and np.all(np.abs(a1) <= 1)


def test_a_trainer_itself_pickles_without_device_state():
    v = load_variant()
    tr = assemble(v, 10, 3)["trainer"]
    tr2 = pickle.loads(pickle.dumps(tr))
    assert tr2._h is None and tr2.policy._trainer is tr2 and np.array_equal(tr2.policy.flat(), tr.policy.flat())
    assert tr2.noise_seed == tr.noise_seed and tr2.reward_scale == tr.reward_scale


def test_seeds_reach_the_initial_weights_and_both_noise_streams():
    """train.py:112-113 seed np.random and torch; rlkit draws initial weights, rsample noise and exploration noise from
    torch's generator.  Same seed -> same run; another seed -> other weights, other device-noise key, other exploration."""
    v = load_variant()
    obs = np.zeros(42)

    def run(seed):
        np.random.seed(seed); torch.manual_seed(seed)
        a = assemble(v, 42, 7)
        state_before = np.random.get_state()[1].copy()
        draws = np.stack([a["expl_policy"].get_action(obs)[0] for _ in range(3)])
        # like rlkit, neither the assembly nor acting touches np.random: random_batch alone consumes it
        assert np.array_equal(np.random.get_state()[1], state_before)
        assert np.random.get_state()[2] == np.random.RandomState(seed).get_state()[2]
        return a["trainer"].policy.flat(), a["trainer"].noise_seed, draws

    w1, n1, d1 = run(17)
    w1b, n1b, d1b = run(17)
    w2, n2, d2 = run(59)
    assert np.array_equal(w1, w1b) and n1 == n1b and np.array_equal(d1, d1b)
    assert not np.array_equal(w1, w2) and n1 != n2 and not np.array_equal(d1, d2)
    assert n1 == 17 and n2 == 59                          # torch.initial_seed()
    assert not np.array_equal(d1[0], d1[1])               # (a stream, not a constant)


# ---------------------------------------------------------------------------------------------------------------
# GPU: the whole loop on the device path, two seeds
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("seed", [17, 59])
def test_reference_shaped_run_two_epochs(tmp_path, seed):
    v = load_variant()
    ak = dict(v["algorithm_kwargs"], num_trains_per_train_loop=40, num_expl_steps_per_train_loop=300,
              min_num_steps_before_training=700, num_eval_steps_per_epoch=200, expl_max_path_length=100,
              eval_max_path_length=100)
    np.random.seed(seed); torch.manual_seed(seed)          # train.py:112-113
    ref = np.random.RandomState(seed)                      # what np.random must look like to every host consumer
    a = assemble(v, 42, 7, mirror=ref)
    tr, expl, evalc = a["trainer"], a["expl"], a["evalc"]
    replay_buffer = EnvReplayBuffer(v["replay_buffer_size"], a["expl_env"])           # rlkit_utils.py:139-142
    for net in tr.networks:
        net.to("cuda:0")
    B = ak["batch_size"]
    replay_buffer.add_paths(expl.collect_new_paths(ak["expl_max_path_length"], ak["min_num_steps_before_training"], False))
    expl.end_epoch(-1)
    alphas, snapshots = [], []
    for epoch in range(2):
        evalc.collect_new_paths(ak["eval_max_path_length"], ak["num_eval_steps_per_epoch"], True)
        new_paths = expl.collect_new_paths(ak["expl_max_path_length"], ak["num_expl_steps_per_train_loop"], False)
        replay_buffer.add_paths(new_paths)
        for net in tr.networks:
            net.train(True)
        size = replay_buffer.num_steps_can_sample()
        for i in range(ak["num_trains_per_train_loop"]):
            train_data = replay_buffer.random_batch(B)
            want_idx = ref.randint(0, size, B)             # rlkit: np.random.randint(0, self._size, batch_size)
            if i % 13 == 0:                                # (reading the indices does not disturb the device batch)
                assert np.array_equal(train_data.indices(), want_idx)
            tr.train(train_data)
        for net in tr.networks:
            net.train(False)
        # np.random itself is where rlkit would have left it -- without any hand-over call
        assert np.array_equal(np.random.get_state()[1], ref.get_state()[1]) and np.random.get_state()[2] == ref.get_state()[2]
        # _end_epoch (rlkit_custom.py:54-66)
        snapshot = get_snapshot(tr, expl, evalc, replay_buffer)
        path = tmp_path / f"itr_{epoch}.pkl"
        torch.save(snapshot, path)
        snapshots.append(path)
        assert replay_buffer.get_diagnostics()["size"] == 700 + 300 * (epoch + 1)
        d = tr.get_diagnostics()
        assert "QF1 Loss" in d and "Alpha" in d and np.isfinite(d["Policy Loss"])
        alphas.append(d["Alpha"])
        assert tr.reward_scale == v["trainer_kwargs"]["reward_scale"]          # rlkit_custom.py:291
        expl.end_epoch(epoch); evalc.end_epoch(epoch); replay_buffer.end_epoch(epoch); tr.end_epoch(epoch)
    assert tr._h is not None and tr._num_train_steps == 80
    # KA1: the very first step's alpha; one epoch later alpha has moved
    assert alphas[0] == pytest.approx(float(np.float32(np.exp(-v["trainer_kwargs"]["policy_lr"]))), abs=1e-7)
    assert alphas[1] < alphas[0]
    # the snapshot's consumer (rlkit_utils.py:173-174,241-250)
    data = torch.load(snapshots[-1], weights_only=False)
    policy = data["evaluation/policy"]
    policy.cuda() if not isinstance(policy, MakeDeterministic) else policy.stochastic_policy.cuda()
    obs = np.random.RandomState(1).normal(0, 0.5, (5, 42))
    for o in obs:
        got, _ = policy.get_action(o)                      # host network loaded from the file
        want, _ = a["eval_policy"].get_action(o)           # the live policy (sac_policy_act on the trained weights)
        assert np.allclose(got, want, atol=2e-6)
    # the two saved epochs differ (training went on), and the file holds the TRAINED weights, not the initial ones
    first = torch.load(snapshots[0], weights_only=False)
    assert not np.array_equal(first["trainer/policy"].flat(), data["trainer/policy"].flat())
    assert np.array_equal(data["trainer/qf1"].flat(), tr._get_params("qf1"))


@pytest.mark.gpu
def test_two_seeds_draw_different_indices_and_noise():
    """Round 2's hole: constructed the reference's way every --seed drew the same index stream and the same noise."""
    v = load_variant()
    out = []
    for seed in (17, 59):
        np.random.seed(seed); torch.manual_seed(seed)
        a = assemble(v, 42, 7)
        buf = EnvReplayBuffer(5000, a["expl_env"])
        rs = np.random.RandomState(0)
        buf.add_block(rs.normal(size=(5000, 42)), rs.uniform(-1, 1, (5000, 7)), rs.uniform(size=(5000, 1)),
                      rs.normal(size=(5000, 42)), np.zeros((5000, 1), np.uint8))
        b = buf.random_batch(128)
        idx = b.indices()
        assert np.array_equal(idx, np.random.RandomState(seed).randint(0, 5000, 128))
        a["trainer"].train(b)
        out.append((idx, a["trainer"].debug_fetch("a_new", 128 * 7)))
    assert not np.array_equal(out[0][0], out[1][0]) and not np.array_equal(out[0][1], out[1][1])


@pytest.mark.gpu
def test_td3_reference_shaped_epoch(tmp_path):
    v = dict(load_variant(), trainer_kwargs=TD3_KWARGS)
    np.random.seed(3); torch.manual_seed(3)
    a = assemble(v, 42, 7, agent="TD3")
    tr, expl, evalc = a["trainer"], a["expl"], a["evalc"]
    buf = EnvReplayBuffer(10_000, a["expl_env"])
    buf.add_paths(expl.collect_new_paths(100, 500, False))
    for _ in range(10):
        tr.train(buf.random_batch(128))
    snap = get_snapshot(tr, expl, evalc, buf)
    assert "trainer/target_policy" in snap
    torch.save(snap, tmp_path / "params.pkl")
    data = torch.load(tmp_path / "params.pkl", weights_only=False)
    o = np.random.RandomState(1).normal(0, 0.5, 42)
    got, _ = data["evaluation/policy"].get_action(o)
    want, _ = a["eval_policy"].get_action(o)
    assert np.allclose(got, want, atol=2e-6)
    e1, _ = data["exploration/policy"].get_action(o)       # PolicyWrappedWithExplorationStrategy survives the pickle
    assert e1.shape == (7,) and np.all(np.abs(e1) <= 1)
