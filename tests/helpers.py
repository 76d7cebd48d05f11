"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import numpy as np

from oracle.sac_step_torch import RlkitEquivalentSAC, init_sac_params

TASK_DIMS = {"Lift": (42, 7), "Door": (46, 7), "Stack": (55, 7), "TwoArmLift": (89, 14), "Wipe": (379, 6),
             "TwoArmPegInHole": (73, 12), "TwoArmHandoff": (86, 14), "PickPlaceCan": (46, 7),
             "NutAssemblyRound": (46, 7), "LiftModded": (64, 4), "LiftJaco": (50, 4), "WipeJV": (379, 7)}


def synth_transitions(n, O, A, seed=1234, term_frac=0.0, reward_scale=1.0):
    """SURVEY.md section 8d synthetic data: obs ~ N(0, .5^2), act ~ U(-1,1), rew ~ U(0,1)."""
    rs = np.random.RandomState(seed)
    obs = rs.normal(0, 0.5, (n, O)).astype(np.float32)
    nobs = rs.normal(0, 0.5, (n, O)).astype(np.float32)
    act = rs.uniform(-1, 1, (n, A)).astype(np.float32)
    rew = (rs.uniform(0, 1, (n, 1)) * reward_scale).astype(np.float32)
    term = (rs.uniform(0, 1, (n, 1)) < term_frac).astype(np.uint8)
    return obs, act, rew, term, nobs


def flat_of(layers):
    return np.concatenate([np.concatenate([w.ravel(), b.ravel()]) for w, b in layers]).astype(np.float32)


def make_pair(O, A, B, seed=3, device=0, **kw):
    """An oracle and a HIP trainer holding identical parameters."""
    from robosuite_benchmark_amd import FlattenMlp, SACTrainer, TanhGaussianPolicy
    kw.setdefault("policy_lr", 1e-3)
    kw.setdefault("qf_lr", 5e-4)
    kw.setdefault("soft_target_tau", 0.005)
    kw.setdefault("target_update_period", 5)
    hidden, hidden_q = tuple(kw.pop("hidden", (256, 256))), kw.pop("hidden_q", None)
    hidden_q = tuple(hidden_q) if hidden_q else hidden
    nets = init_sac_params(O, A, hidden=hidden, seed=seed, hidden_q=hidden_q)
    noise_seed = kw.pop("noise_seed", 0)
    oracle = RlkitEquivalentSAC(nets, A, **kw)
    pol = TanhGaussianPolicy(list(hidden), O, A)
    qs = [FlattenMlp(list(hidden_q), 1, O + A) for _ in range(4)]
    pol.load_flat(flat_of(nets["policy"]))
    for q, name in zip(qs, ("qf1", "qf2", "target_qf1", "target_qf2")):
        q.load_flat(flat_of(nets[name]))
    hip = SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], batch_size=B,
                     device=device, noise_seed=noise_seed, **kw)
    return oracle, hip


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def layers_from_flat(flat, shapes):
    """flat nn.Linear vector (W, b per layer) -> [(W, b), ...] for the oracle."""
    out, off = [], 0
    for (n, k) in shapes:
        w = np.asarray(flat[off:off + n * k], np.float32).reshape(n, k).copy(); off += n * k
        b = np.asarray(flat[off:off + n], np.float32).copy(); off += n
        out.append((w, b))
    assert off == len(flat)
    return out


def make_pair_from_flat(flats, O, A, B, device=0, **kw):
    """Oracle + HIP trainer from flat parameter vectors {policy, qf1, qf2[, target_qf1, target_qf2]}."""
    from collections import OrderedDict
    from robosuite_benchmark_amd import FlattenMlp, SACTrainer, TanhGaussianPolicy
    kw.setdefault("policy_lr", 1e-3)
    kw.setdefault("qf_lr", 5e-4)
    kw.setdefault("soft_target_tau", 0.005)
    kw.setdefault("target_update_period", 5)
    qs, ps = [(256, O + A), (256, 256), (1, 256)], [(256, O), (256, 256), (A, 256), (A, 256)]
    nets = OrderedDict()
    for name in ("qf1", "qf2", "target_qf1", "target_qf2"):
        nets[name] = layers_from_flat(flats.get(name, flats[name.replace("target_", "")]), qs)
    nets["policy"] = layers_from_flat(flats["policy"], ps)
    noise_seed = kw.pop("noise_seed", 0)
    oracle = RlkitEquivalentSAC(nets, A, **kw)
    pol = TanhGaussianPolicy([256, 256], O, A)
    pol.load_flat(flat_of(nets["policy"]))
    qn = [FlattenMlp([256, 256], 1, O + A) for _ in range(4)]
    for q, name in zip(qn, ("qf1", "qf2", "target_qf1", "target_qf2")):
        q.load_flat(flat_of(nets[name]))
    hip = SACTrainer(policy=pol, qf1=qn[0], qf2=qn[1], target_qf1=qn[2], target_qf2=qn[3], batch_size=B,
                     device=device, noise_seed=noise_seed, **kw)
    return oracle, hip


def make_td3_pair(O, A, B, seed=3, device=0, **kw):
    """A TD3 oracle and a HIP TD3 trainer holding identical parameters."""
    from oracle.td3_step_torch import RlkitEquivalentTD3, init_td3_params
    from robosuite_benchmark_amd import FlattenMlp, TanhMlpPolicy, TD3Trainer
    kw.setdefault("policy_learning_rate", 1e-3)
    kw.setdefault("qf_learning_rate", 5e-4)
    hidden = tuple(kw.pop("hidden", (256, 256)))
    nets = init_td3_params(O, A, hidden=hidden, seed=seed)
    noise_seed = kw.pop("noise_seed", 0)
    oracle = RlkitEquivalentTD3(nets, A, **kw)
    pols = [TanhMlpPolicy(list(hidden), A, O) for _ in range(2)]
    qs = [FlattenMlp(list(hidden), 1, O + A) for _ in range(4)]
    for p, name in zip(pols, ("policy", "target_policy")):
        p.load_flat(flat_of(nets[name]))
    for q, name in zip(qs, ("qf1", "qf2", "target_qf1", "target_qf2")):
        q.load_flat(flat_of(nets[name]))
    hip = TD3Trainer(policy=pols[0], qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], target_policy=pols[1],
                     batch_size=B, device=device, noise_seed=noise_seed, **kw)
    return oracle, hip
