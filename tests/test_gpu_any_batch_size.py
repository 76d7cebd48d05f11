"""GPU: batch sizes that are not a multiple of the 16-row block (rlkit takes any batch_size; the reference's CLI
default is 128, /root/reference/util/arguments.py).  The slot is padded to whole row-blocks; pad rows carry zero
weight in every mean, so the step must match the oracle on exactly the rows given."""
import numpy as np
import pytest

from robosuite_benchmark_amd import EnvReplayBuffer
from robosuite_benchmark_amd._lib import DIAG_NAMES, TD3_DIAG_NAMES
from tests.helpers import make_pair, make_td3_pair, rel_err, synth_transitions

pytestmark = pytest.mark.gpu


def _batch(B, O, A, seed):
    obs, act, rew, term, nobs = synth_transitions(B, O, A, seed=seed, term_frac=0.1)
    rs = np.random.RandomState(seed + 1)
    eps = (rs.standard_normal((B, A)).astype(np.float32), rs.standard_normal((B, A)).astype(np.float32))
    return dict(observations=obs, actions=act, rewards=rew, terminals=term.astype(np.float32), next_observations=nobs), eps


@pytest.mark.parametrize("O,A,B", [(42, 7, 100), (42, 7, 1), (11, 3, 17), (46, 7, 250), (89, 14, 300), (10, 2, 530)])
def test_sac_step_any_batch_size(O, A, B):
    oracle, hip = make_pair(O, A, B, seed=11)
    nb, eps = _batch(B, O, A, 21)
    want = oracle.step(nb["observations"], nb["actions"], nb["rewards"], nb["terminals"], nb["next_observations"], *eps)
    diag = hip.train(nb, eps=eps)
    for i, name in enumerate(DIAG_NAMES):
        if name in want:
            assert abs(float(diag[i]) - want[name]) <= 1e-5 * max(1.0, abs(want[name])), (name, float(diag[i]), want[name])
    L = oracle.last
    assert hip.debug_fetch("q_target", B).shape == (B,)
    for name, ref in (("q1", L["q1"]), ("q_target", L["y"]), ("q1_new", L["q1_new"]), ("log_pi", L["log_pi"])):
        assert rel_err(hip.debug_fetch(name, B), ref.detach().numpy().ravel()) < 2e-5, name
    for g in ("g_qf1", "g_qf2", "g_policy"):
        ws, bs = L[g][:len(L[g]) // 2], L[g][len(L[g]) // 2:]          # (oracle: all weight grads, then all bias grads)
        ref = np.concatenate([np.concatenate([w.ravel(), b.ravel()]) for w, b in zip(ws, bs)])
        got = hip.debug_fetch(g, ref.size)
        assert np.max(np.abs(got - ref)) <= 5e-5 * max(1e-30, np.max(np.abs(ref))), g


def test_td3_step_and_loops_with_an_odd_batch_size():
    O, A, B, n = 42, 7, 50, 4000
    oracle, hip = make_td3_pair(O, A, B, seed=3)
    nb, eps = _batch(B, O, A, 5)
    want = oracle.step(nb["observations"], nb["actions"], nb["rewards"], nb["terminals"], nb["next_observations"], eps[1])
    diag = hip.train(nb, eps=eps[1])
    for i, name in enumerate(TD3_DIAG_NAMES):
        if name in want:
            assert abs(float(diag[i]) - want[name]) <= 1e-5 * max(1.0, abs(want[name])), name
    # fused loop == stepwise (device batches) == NumPy's index stream, at a batch size of 50
    _, fused = make_td3_pair(O, A, B, seed=4, noise_seed=7)
    _, stepw = make_td3_pair(O, A, B, seed=4, noise_seed=7)
    obs, act, rew, term, nobs = synth_transitions(n, O, A, seed=8)
    bufs = []
    for _ in range(2):
        b = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
        b.add_block(obs, act, rew, nobs, term)
        b.seed(17)
        bufs.append(b)
    fused.train_loop(bufs[0], 30, batch_size=B)
    rs = np.random.RandomState(17)
    for _ in range(30):
        batch = bufs[1].random_batch(B)
        assert np.array_equal(batch.indices(), rs.randint(0, n, B))
        stepw.train(batch)
    sa, sb = fused.state_dict(), stepw.state_dict()
    for k in sa["params"]:
        assert np.array_equal(sa["params"][k], sb["params"][k]), k


def test_replay_buffer_api_with_odd_batch_sizes():
    O, A, n = 13, 4, 3000
    obs, act, rew, term, nobs = synth_transitions(n, O, A, seed=2, term_frac=0.2)
    buf = EnvReplayBuffer(n, obs_dim=O, action_dim=A, lazy_batches=False)
    buf.add_block(obs, act, rew, nobs, term)
    buf.seed(9)
    rs = np.random.RandomState(9)
    for B in (1, 7, 100, 33):
        batch, idx = buf.random_batch(B, return_indices=True)
        want = rs.randint(0, n, B)
        assert np.array_equal(idx, want)
        assert batch["observations"].shape == (B, O) and np.array_equal(batch["observations"], obs[want])
        assert np.array_equal(batch["rewards"], rew[want]) and np.array_equal(batch["terminals"].ravel() != 0, term[want].ravel() != 0)
    assert np.array_equal(buf.sample_indices(5, 3), rs.randint(0, n, 15).reshape(3, 5))
    g = buf.gather(np.array([5, 0, 2999], np.int64))
    assert np.array_equal(g["next_observations"], nobs[[5, 0, 2999]])
