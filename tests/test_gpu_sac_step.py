"""GPU parity: one SAC gradient step (libsac_hip.so through the C ABI) against the CPU oracle
(oracle/sac_step_torch.py, "rlkit-equivalent restatement") from identical state, identical
batch and identical N(0,1) draws.

Tolerance (BASELINE.json north_star): losses / logged scalars within 1e-5, read as
|a-b| <= 1e-5 * max(1, |b|) (SURVEY.md section 7 "1e-5 on fp32 losses").  fp32 MFMA accumulates
in a different order than the CPU GEMM, so per-element tensors are compared at 2e-5 of the
tensor's scale."""
import numpy as np
import pytest

from robosuite_benchmark_amd._lib import DIAG_NAMES
from tests.helpers import TASK_DIMS, flat_of, make_pair, rel_err, synth_transitions

pytestmark = pytest.mark.gpu
TOL = 1e-5


def batch_and_noise(B, O, A, seed, term_frac=0.0):
    obs, act, rew, term, nobs = synth_transitions(B, O, A, seed=seed, term_frac=term_frac)
    rs = np.random.RandomState(seed + 1)
    e1 = rs.normal(0, 1, (B, A)).astype(np.float32)
    e2 = rs.normal(0, 1, (B, A)).astype(np.float32)
    np_batch = dict(observations=obs, actions=act, rewards=rew, terminals=term.astype(np.float32),
                    next_observations=nobs)
    return np_batch, (e1, e2)


def check_diag(diag, want, tol=TOL):
    for i, name in enumerate(DIAG_NAMES):
        assert abs(diag[i] - want[name]) <= tol * max(1.0, abs(want[name])), (name, diag[i], want[name])


def scale_err(got, want):
    want = np.asarray(want, np.float64)
    return float(np.max(np.abs(np.asarray(got, np.float64) - want)) / max(1e-30, np.max(np.abs(want))))


# batch sizes cover every column split of the hidden layers: SP = 4 (B <= 256), 2 (B <= 512 and odd
# row-block counts above), 1 (B >= 528 with an even row-block count)
CASES = [("Lift", 128, 0.0), ("Lift", 256, 0.0), ("Door", 1024, 0.0), ("TwoArmLift", 256, 0.0),
         ("Lift", 256, 0.05), ("Wipe", 128, 0.0), ("Lift", 16, 0.0), ("Lift", 512, 0.0), ("Lift", 560, 0.0),
         ("Stack", 48, 0.1),
         # every (O, A) of the 8-task sweep (BASELINE.json configs[4], parallel.SWEEP) at the sweep's batch sizes, and
         # the fork's LiftModded-Jaco 64/4 (/root/reference/training_configs/**/variant.json), Lift-Jaco 50/4
         ("TwoArmPegInHole", 256, 0.0), ("TwoArmHandoff", 256, 0.0), ("Stack", 256, 0.0), ("Wipe", 256, 0.0),
         ("PickPlaceCan", 128, 0.0), ("NutAssemblyRound", 128, 0.0), ("LiftModded", 128, 0.0), ("LiftJaco", 128, 0.05),
         ("Door", 256, 0.0), ("WipeJV", 128, 0.0)]


@pytest.mark.parametrize("task,B,term_frac", CASES)
def test_single_step_from_identical_state(task, B, term_frac):
    O, A = TASK_DIMS[task]
    oracle, hip = make_pair(O, A, B, seed=11)
    np_batch, eps = batch_and_noise(B, O, A, seed=21, term_frac=term_frac)
    want = oracle.step(np_batch["observations"], np_batch["actions"], np_batch["rewards"], np_batch["terminals"],
                       np_batch["next_observations"], *eps)
    diag = hip.train(np_batch, eps=eps)
    check_diag(diag, want)
    L = oracle.last
    # forward intermediates
    for name, ref in (("a_new", L["a_new"]), ("mu", L["mu"]), ("log_std", L["log_std"]), ("a_next", L["a2"])):
        assert scale_err(hip.debug_fetch(name, B * A), ref.detach().numpy().ravel()) < 2e-5, name
    for name, ref in (("log_pi", L["log_pi"]), ("log_pi_next", L["log_pi2"]), ("q1", L["q1"]), ("q2", L["q2"]),
                      ("q_target", L["y"]), ("q1_new", L["q1_new"]), ("q2_new", L["q2_new"])):
        assert rel_err(hip.debug_fetch(name, B), ref.detach().numpy().ravel()) < 2e-5, name
    # gradients (flat nn.Linear layout), relative to the gradient's own scale
    for name, key in (("g_policy", "g_policy"), ("g_qf1", "g_qf1"), ("g_qf2", "g_qf2")):
        ws, bs = L[key][:len(L[key]) // 2], L[key][len(L[key]) // 2:]
        ref = np.concatenate([np.concatenate([w.ravel(), b.ravel()]) for w, b in zip(ws, bs)])
        got = hip.debug_fetch(name, ref.size)
        assert scale_err(got, ref) < 5e-5, name
    # parameters after Adam + Polyak.  Adam's first step moves each weight by ~lr*sign(g): where the
    # true gradient is ~0 the sign is numerically undetermined, so compare at a fraction of lr.
    st = hip.state_dict()
    after = oracle.export_nets()
    for name in ("policy", "qf1", "qf2", "target_qf1", "target_qf2"):
        ref = flat_of(after[name])
        d = np.abs(st["params"][name] - ref)
        lr = 1e-3 if name == "policy" else 5e-4
        assert np.max(d) <= 2.0 * lr + 1e-7, name
        assert np.mean(d > 0.02 * lr) < 2e-3, (name, float(np.mean(d > 0.02 * lr)))
    sc = st["scalars"]
    assert sc[3] == 1 and sc[4] == 1
    assert abs(sc[5] - want["Alpha"]) < 1e-7


def test_ten_steps_track_the_oracle():
    """Both sides step their own state for 10 steps on the same batches/noise; logged scalars stay
    within 1e-4 (trajectories diverge chaotically later -- SURVEY.md section 7)."""
    O, A, B = 42, 7, 256
    oracle, hip = make_pair(O, A, B, seed=5)
    for s in range(10):
        np_batch, eps = batch_and_noise(B, O, A, seed=100 + s)
        want = oracle.step(np_batch["observations"], np_batch["actions"], np_batch["rewards"],
                           np_batch["terminals"], np_batch["next_observations"], *eps)
        diag = hip.train(np_batch, eps=eps)
        check_diag(diag, want, tol=1e-4 if s else TOL)
    sc = hip.state_dict()["scalars"]
    assert sc[3] == 10 and sc[4] == 10


def _flat_state(torch_opt, net, key):
    """torch.optim.Adam state of one oracle net as a flat nn.Linear vector (W, b per layer)."""
    return np.concatenate([np.concatenate([torch_opt.state[w][key].numpy().ravel(), torch_opt.state[b][key].numpy().ravel()])
                           for w, b in zip(net.ws, net.bs)])


def _adam_f32(p, m, v, g, lr, t):
    """torch.optim.Adam (fp32 tensors, double bias corrections) restated in NumPy float32."""
    f = np.float32
    m = m + f(1.0 - 0.9) * (g - m)                                  # exp_avg.lerp_(grad, 1 - beta1)
    v = v * f(0.999) + f(1.0 - 0.999) * g * g                       # mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    bc1, bc2s = 1.0 - 0.9 ** t, np.sqrt(1.0 - 0.999 ** t)
    denom = np.sqrt(v) / f(bc2s) + f(1e-8)
    p = p + (f(-(lr / bc1)) * m) / denom                            # addcdiv_(exp_avg, denom, value=-step_size)
    return p.astype(f), m.astype(f), v.astype(f)


@pytest.mark.parametrize("task,B", [("Lift", 256), ("TwoArmHandoff", 64)])
def test_adam_moments_and_update_arithmetic(task, B):
    """(a) exp_avg / exp_avg_sq of the three trained nets and of log_alpha against torch.optim.Adam's own state after
    steps 1, 2 and 3, at 5e-5 of each tensor's scale (moments are linear / quadratic in g: no sign ambiguity);
    (b) the update arithmetic -- bias corrections, beta accumulation at step >= 2, eps placement -- bit-level: the
    parameters after every step equal a float32 NumPy restatement of torch's Adam applied to the HIP path's OWN
    gradients (so gradient noise, which decides the sign of m/sqrt(v) where g ~ 0, does not enter)."""
    O, A = TASK_DIMS[task]
    oracle, hip = make_pair(O, A, B, seed=13)
    nets = {"policy": (oracle.policy, oracle.policy_opt, 1e-3), "qf1": (oracle.qf1, oracle.qf1_opt, 5e-4),
            "qf2": (oracle.qf2, oracle.qf2_opt, 5e-4)}
    st = hip.state_dict()
    ref = {k: (st["params"][k].copy(), np.zeros_like(st["params"][k]), np.zeros_like(st["params"][k])) for k in nets}
    for step in (1, 2, 3):
        np_batch, eps = batch_and_noise(B, O, A, seed=300 + step)
        oracle.step(np_batch["observations"], np_batch["actions"], np_batch["rewards"], np_batch["terminals"],
                    np_batch["next_observations"], *eps)
        hip.train(np_batch, eps=eps)
        st = hip.state_dict()
        for name, (net, opt, lr) in nets.items():
            m_hip, v_hip = st["opt"][name]
            assert scale_err(m_hip, _flat_state(opt, net, "exp_avg")) < 5e-5, (name, step, "exp_avg")
            assert scale_err(v_hip, _flat_state(opt, net, "exp_avg_sq")) < 5e-5, (name, step, "exp_avg_sq")
            g = hip.debug_fetch("g_" + name, m_hip.size)
            ref[name] = _adam_f32(*ref[name], g, lr, step)
            p_ref, m_ref, v_ref = ref[name]
            assert np.max(np.abs(st["params"][name] - p_ref)) <= 1e-7, (name, step, "params vs restated Adam")
            assert np.max(np.abs(m_hip - m_ref)) <= 1e-9 + 1e-6 * np.max(np.abs(m_ref)), (name, step)
            assert np.max(np.abs(v_hip - v_ref)) <= 1e-12 + 1e-6 * np.max(np.abs(v_ref)), (name, step)
        a_st = oracle.alpha_opt.state[oracle.log_alpha]
        sc = st["scalars"]                          # log_alpha, its exp_avg, exp_avg_sq, adam_t, n_steps, alpha
        assert abs(sc[0] - float(oracle.log_alpha)) <= 1e-6 * max(1.0, abs(float(oracle.log_alpha)))
        assert abs(sc[1] - float(a_st["exp_avg"])) <= 1e-5 * abs(float(a_st["exp_avg"])) + 1e-9
        assert abs(sc[2] - float(a_st["exp_avg_sq"])) <= 1e-5 * abs(float(a_st["exp_avg_sq"])) + 1e-12
        assert sc[3] == step
    # where the gradient is well above its own noise the parameters agree with the oracle's after three steps
    after = oracle.export_nets()
    for name, (net, opt, lr) in nets.items():
        v_ref = _flat_state(opt, net, "exp_avg_sq")
        strong = np.sqrt(v_ref) > 1e-2 * np.sqrt(np.max(v_ref))
        d = np.abs(st["params"][name] - flat_of(after[name]))
        assert strong.sum() > 100 and np.max(d[strong]) <= 0.02 * lr, (name, float(np.max(d[strong])))


def test_known_answers_on_device():
    """KA1 (Alpha after the first step, Alpha Loss = -0.0) and KA3 (logged Policy Loss has no alpha)."""
    O, A, B = 42, 7, 128
    _, hip = make_pair(O, A, B, seed=2)
    np_batch, eps = batch_and_noise(B, O, A, seed=7)
    d = hip.train(np_batch, eps=eps)
    assert d[DIAG_NAMES.index("Alpha")] == np.float32(np.exp(np.float32(-1e-3)))
    assert d[DIAG_NAMES.index("Alpha Loss")] == 0.0 and np.signbit(d[DIAG_NAMES.index("Alpha Loss")])
    lp = hip.debug_fetch("log_pi", B)
    qn = np.minimum(hip.debug_fetch("q1_new", B), hip.debug_fetch("q2_new", B))
    assert abs(d[DIAG_NAMES.index("Policy Loss")] - np.mean(lp - qn)) < 1e-5
    assert d[DIAG_NAMES.index("Policy log std Max")] <= 2.0


def test_fixed_alpha_and_reward_scale_variant():
    O, A, B = 46, 7, 64
    oracle, hip = make_pair(O, A, B, seed=9, use_automatic_entropy_tuning=False, reward_scale=3.0,
                            target_update_period=1, soft_target_tau=0.01, discount=0.9)
    for s in range(3):
        np_batch, eps = batch_and_noise(B, O, A, seed=40 + s, term_frac=0.2)
        want = oracle.step(np_batch["observations"], np_batch["actions"], np_batch["rewards"],
                           np_batch["terminals"], np_batch["next_observations"], *eps)
        diag = hip.train(np_batch, eps=eps)
        check_diag(diag, want, tol=1e-4 if s else TOL)


def test_device_noise_stream_is_standard_normal_and_reproducible():
    O, A, B = 42, 7, 1024
    _, hip = make_pair(O, A, B, seed=1, noise_seed=123)
    _, hip2 = make_pair(O, A, B, seed=1, noise_seed=123)
    np_batch, _ = batch_and_noise(B, O, A, seed=3)
    d1, d2 = hip.train(np_batch), hip2.train(np_batch)
    assert np.array_equal(d1, d2)
    # z = mu + std*eps with mu ~ 0, std ~ 1 at init => z ~ N(0,1)
    out = np.empty(B * A, np.float32)
    z = hip.debug_fetch("a_new", B * A)
    z = np.arctanh(np.clip(z, -0.999999, 0.999999))
    assert abs(np.mean(z)) < 0.05 and abs(np.std(z) - 1.0) < 0.05


def test_errors_are_reported_not_fatal():
    from robosuite_benchmark_amd import FlattenMlp, SACTrainer, TanhGaussianPolicy
    pol = TanhGaussianPolicy([256, 256], 10, 3)
    qs = [FlattenMlp([256, 256], 1, 13) for _ in range(4)]
    with pytest.raises(RuntimeError, match="must be positive"):
        SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], batch_size=0)
    for bad in ([], [64] * 8, [5000, 64]):                        # no hidden layer; more than seven; wider than 4096
        pol2 = TanhGaussianPolicy(bad, 10, 3)
        with pytest.raises(RuntimeError, match="hidden_sizes"):
            SACTrainer(policy=pol2, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], batch_size=64)


def _zero_pad(layers, sizes_in, sizes_out):
    """[(W, b)] of a narrow MLP -> the same function as a 256-wide one (zero rows / columns / biases)."""
    out = []
    for (w, b), (n, k), (N, K) in zip(layers, sizes_in, sizes_out):
        W = np.zeros((N, K), np.float32); W[:n, :k] = w
        Bv = np.zeros(N, np.float32); Bv[:n] = b
        out.append((W, Bv))
    return out


@pytest.mark.parametrize("hidden,hidden_q,task,B", [((128, 128), (128, 128), "Lift", 256), ((64, 192), (256, 96), "Door", 128),
                                                   ((256, 256), (32, 32), "TwoArmHandoff", 64), ((100, 50), (7, 255), "Lift", 48)])
def test_narrower_hidden_layers(hidden, hidden_q, task, B):
    """--policy_hidden_sizes / --qf_hidden_sizes below 256 (/root/reference/util/arguments.py:98,104): the kernels' layers
    are 256 wide, a narrower layer is the same function with zero rows / columns -- whose gradients, Adam updates and
    Polyak averages are identically zero.  (a) parity with the oracle built at the narrow sizes; (b) EXACTNESS of the
    embedding: a 256-wide trainer holding the zero-padded weights produces bit-identical diagnostics step after step."""
    from robosuite_benchmark_amd import FlattenMlp, SACTrainer, TanhGaussianPolicy
    O, A = TASK_DIMS[task]
    oracle, hip = make_pair(O, A, B, seed=11, hidden=hidden, hidden_q=hidden_q)
    assert hip.state_dict()["params"]["policy"].size == hidden[0] * O + hidden[0] + hidden[1] * hidden[0] + hidden[1] + 2 * (A * hidden[1] + A)
    # the same nets embedded in 256-wide ones
    nets = oracle.export_nets()
    ps_in = [(hidden[0], O), (hidden[1], hidden[0]), (A, hidden[1]), (A, hidden[1])]
    ps_out = [(256, O), (256, 256), (A, 256), (A, 256)]
    qs_in = [(hidden_q[0], O + A), (hidden_q[1], hidden_q[0]), (1, hidden_q[1])]
    qs_out = [(256, O + A), (256, 256), (1, 256)]
    pol = TanhGaussianPolicy([256, 256], O, A)
    pol.load_flat(flat_of(_zero_pad(nets["policy"], ps_in, ps_out)))
    qn = [FlattenMlp([256, 256], 1, O + A) for _ in range(4)]
    for q, name in zip(qn, ("qf1", "qf2", "target_qf1", "target_qf2")):
        q.load_flat(flat_of(_zero_pad(nets[name], qs_in, qs_out)))
    wide = SACTrainer(policy=pol, qf1=qn[0], qf2=qn[1], target_qf1=qn[2], target_qf2=qn[3], batch_size=B, policy_lr=1e-3,
                      qf_lr=5e-4, soft_target_tau=0.005, target_update_period=5)
    for s_ in range(6):
        np_batch, eps = batch_and_noise(B, O, A, seed=500 + s_, term_frac=0.05)
        want = oracle.step(np_batch["observations"], np_batch["actions"], np_batch["rewards"], np_batch["terminals"],
                           np_batch["next_observations"], *eps)
        d_narrow = hip.train(np_batch, eps=eps)
        d_wide = wide.train(np_batch, eps=eps)
        check_diag(d_narrow, want, tol=1e-4 if s_ else TOL)
        assert np.array_equal(d_narrow, d_wide), s_
    # the padded units of the wide trainer are still exactly zero, and its live part equals the narrow trainer's
    wp, npar = wide.state_dict()["params"], hip.state_dict()["params"]
    w0 = wp["policy"][:256 * O].reshape(256, O)
    assert np.all(w0[hidden[0]:] == 0) and np.array_equal(w0[:hidden[0]].ravel(), npar["policy"][:hidden[0] * O])
