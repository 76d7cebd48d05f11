"""GPU: size-independent properties of the hot path at BASELINE.json's full sizes -- configs[1] Lift obs 42 / act 7 batch 256
(the fused step), configs[2] Door 46 / 7 batch 1024 (the four-launch step, column split 1), configs[3] TwoArmLift 89 / 14
batch 256 -- each on a FULL 1e6-slot replay buffer, where the torch oracle would take minutes.  Each property pins a different part of
`random_batch -> train` (reference loop: /root/reference/util/rlkit_custom.py:233-240) against something computable on
the host from the synthetic transitions alone:

* discount 0      => the Bellman target of a step is reward_scale * r of exactly the rows NumPy's stream picks
                     (index stream, gather, target arithmetic, statistics);
* learning rates 0 => a loop leaves the trained networks bit-identical (Adam with a zero step, no stray writer) and the
                     targets on the closed form of their Polyak recursion;
* tau 1, period 1  => after every step the target critics ARE the critics, bit for bit (Polyak in the Adam launch);
* one transition repeated in every slot => every row of every row-block computes the same Q(s, a): max == min, std == 0
                     (row-block / lane / column-part maps of the MFMA tiles);
* a permuted batch (noise permuted with it) => the same statistics (order-free ones bit for bit)."""
import numpy as np
import pytest

from robosuite_benchmark_amd._lib import DIAG_NAMES
from tests.helpers import make_pair, synth_transitions

pytestmark = pytest.mark.gpu

O, A, B, N = 42, 7, 256, 1_000_000
D = {n: i for i, n in enumerate(DIAG_NAMES)}
SHAPES = {"Lift-256": (42, 7, 256), "Door-1024": (46, 7, 1024), "TwoArmLift-256": (89, 14, 256)}
_cache = {}


@pytest.fixture(scope="module")
def transitions():
    return synth_transitions(N, O, A, seed=77, term_frac=0.02)


@pytest.fixture(params=list(SHAPES))
def shaped(request):
    """(obs_dim, act_dim, batch, transitions) of a BASELINE configuration; one set of transitions alive at a time."""
    o, a, b = SHAPES[request.param]
    if _cache.get("key") != (o, a):
        _cache.clear()
        _cache.update(key=(o, a), data=synth_transitions(N, o, a, seed=77, term_frac=0.02))
    return o, a, b, _cache["data"]


def _full_buffer(transitions):
    from robosuite_benchmark_amd import EnvReplayBuffer
    obs, act, rew, term, nobs = transitions
    buf = EnvReplayBuffer(N, obs_dim=obs.shape[1], action_dim=act.shape[1])
    buf.add_block(obs, act, rew, nobs, term)
    assert buf.num_steps_can_sample() == N
    return buf


def test_discount_zero_targets_are_the_sampled_rewards(shaped):
    O, A, B, transitions = shaped
    rew = transitions[2].reshape(-1).astype(np.float64)
    scale, steps = 3.0, 40
    _, hip = make_pair(O, A, B, seed=5, noise_seed=2, discount=0.0, reward_scale=scale)
    buf = _full_buffer(transitions)
    buf.seed(1234)
    hip.train_loop(buf, steps, batch_size=B)
    assert hip.is_fused() == (B <= 256)
    trace = hip.debug_fetch("diag_trace", steps * 32).reshape(steps, 32)
    rs = np.random.RandomState(1234)                       # the generator the reference seeds (scripts/train.py:112)
    for k in range(steps):
        idx = rs.randint(0, N, B)
        y = np.float32(scale) * rew[idx].astype(np.float32)            # y = reward_scale * r + (1 - d) * 0 * (...)
        got = trace[k, D["Q Targets Mean"]:D["Q Targets Mean"] + 4]
        assert got[2] == y.max() and got[3] == y.min(), (k, got, y.max(), y.min())
        assert abs(got[0] - y.astype(np.float64).mean()) <= 1e-6 * scale
        assert abs(got[1] - y.astype(np.float64).std()) <= 2e-6 * scale


def test_zero_learning_rates_leave_the_networks_bit_identical(transitions):
    _, hip = make_pair(O, A, B, seed=6, noise_seed=3, policy_lr=0.0, qf_lr=0.0)
    buf = _full_buffer(transitions)
    buf.seed(9)
    before = hip.state_dict()
    first, last = hip.train_loop(buf, 300, batch_size=B)
    after = hip.state_dict()
    for k in ("policy", "qf1", "qf2"):
        assert np.array_equal(before["params"][k], after["params"][k]), k
    assert last[D["Alpha"]] == 1.0 and np.isfinite(last).all()
    # the targets follow the closed form of 60 Polyak updates (steps 0, 5, ..., 295) towards critics that never moved
    keep = (1.0 - 0.005) ** 60
    for k in ("qf1", "qf2"):
        want = before["params"]["target_" + k].astype(np.float64) * keep + before["params"][k].astype(np.float64) * (1.0 - keep)
        assert np.max(np.abs(after["params"]["target_" + k] - want)) <= 2e-6
    # and the step still did its forward work on fresh batches: the losses of two different batches differ
    assert first[D["QF1 Loss"]] != last[D["QF1 Loss"]]


def test_tau_one_makes_the_targets_the_critics_after_every_step(shaped):
    O, A, B, transitions = shaped
    _, hip = make_pair(O, A, B, seed=7, noise_seed=4, soft_target_tau=1.0, target_update_period=1)
    buf = _full_buffer(transitions)
    buf.seed(10)
    for n in (1, 7, 64):
        hip.train_loop(buf, n, batch_size=B)
        st = hip.state_dict()
        assert np.array_equal(st["params"]["qf1"], st["params"]["target_qf1"])
        assert np.array_equal(st["params"]["qf2"], st["params"]["target_qf2"])


def test_identical_rows_give_identical_predictions_in_every_tile():
    from robosuite_benchmark_amd import EnvReplayBuffer
    obs, act, rew, term, nobs = synth_transitions(1, O, A, seed=3)
    buf = EnvReplayBuffer(N, obs_dim=O, action_dim=A)
    rep = 50_000
    for _ in range(N // rep):                              # one transition in all 1e6 slots
        buf.add_block(np.repeat(obs, rep, 0), np.repeat(act, rep, 0), np.repeat(rew, rep, 0), np.repeat(nobs, rep, 0),
                      np.repeat(term, rep, 0))
    buf.seed(5)
    _, hip = make_pair(O, A, B, seed=8, noise_seed=5)
    steps = 12
    hip.train_loop(buf, steps, batch_size=B)
    trace = hip.debug_fetch("diag_trace", steps * 32).reshape(steps, 32)
    for k in range(steps):
        for q in ("Q1 Predictions", "Q2 Predictions"):
            mx, mn, sd = trace[k, D[q + " Max"]], trace[k, D[q + " Min"]], trace[k, D[q + " Std"]]
            assert mx == mn, (k, q, mx, mn)
            assert sd <= 1e-3 * max(1.0, abs(mx))          # (sqrt of a difference of two float64 means: zero up to rounding)
    # the predictions move from step to step (the networks are training on that one transition)
    assert trace[0, D["Q1 Predictions Max"]] != trace[steps - 1, D["Q1 Predictions Max"]]


def test_a_permuted_batch_gives_the_same_statistics(transitions):
    obs, act, rew, term, nobs = (x[:B] for x in transitions)
    rs = np.random.RandomState(11)
    eps = (rs.normal(size=(B, A)).astype(np.float32), rs.normal(size=(B, A)).astype(np.float32))
    perm = rs.permutation(B)
    out = []
    for p in (np.arange(B), perm):
        _, hip = make_pair(O, A, B, seed=9)
        batch = dict(observations=obs[p], actions=act[p], rewards=rew[p], terminals=term[p], next_observations=nobs[p])
        out.append(hip.train(batch, eps=(eps[0][p], eps[1][p])))
    a, b = out
    for name in DIAG_NAMES:
        i = D[name]
        if name.endswith(" Max") or name.endswith(" Min") or name in ("Alpha",):
            assert a[i] == b[i], (name, a[i], b[i])        # order-free: bit for bit
        else:
            assert abs(a[i] - b[i]) <= 2e-6 * max(1.0, abs(a[i])), (name, a[i], b[i])
