"""Pins oracle/sac_step_torch.py against the known answers the reference's shipped
progress.csv rows hold for the hot path.
The reference has no bit-level vector for a gradient step: parity unpinned beyond these."""
import json
import math
import os

import numpy as np
import pytest

from oracle.sac_step_torch import RlkitEquivalentSAC, init_sac_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KA = json.load(open(os.path.join(ROOT, "tests", "golden", "progress_known_answers.json")))


def synth_batch(rs, B, O, A, term_frac=0.0):
    obs = rs.normal(0, 0.5, (B, O)).astype(np.float32)
    nobs = rs.normal(0, 0.5, (B, O)).astype(np.float32)
    act = rs.uniform(-1, 1, (B, A)).astype(np.float32)
    rew = rs.uniform(0, 1, (B, 1)).astype(np.float32)
    term = (rs.uniform(0, 1, (B, 1)) < term_frac).astype(np.float32)
    e1 = rs.normal(0, 1, (B, A)).astype(np.float32)
    e2 = rs.normal(0, 1, (B, A)).astype(np.float32)
    return obs, act, rew, term, nobs, e1, e2


def make(O, A, **kw):
    kw.setdefault("policy_lr", 1e-3)
    kw.setdefault("qf_lr", 5e-4)
    kw.setdefault("soft_target_tau", 0.005)
    kw.setdefault("target_update_period", 5)
    return RlkitEquivalentSAC(init_sac_params(O, A, seed=3), A, **kw)


def test_ka1_alpha_after_first_step_matches_every_shipped_run():
    sac = make(42, 7)
    d = sac.step(*synth_batch(np.random.RandomState(0), 128, 42, 7))
    want = float(np.exp(np.float32(-1e-3)))
    assert d["Alpha"] == pytest.approx(want, rel=0, abs=1e-7)
    assert d["Alpha Loss"] == 0.0 and math.copysign(1, d["Alpha Loss"]) == -1.0   # "-0.0"
    for run, rec in KA.items():
        if run.startswith("_"):
            continue
        row0 = rec["rows"][0]
        assert row0["trainer/Alpha"] == pytest.approx(d["Alpha"], abs=1e-7)
        assert row0["trainer/Alpha Loss"] == 0.0


def test_ka2_alpha_loss_uses_prestep_log_alpha_and_logs_poststep_alpha():
    sac = make(42, 7)
    rs = np.random.RandomState(1)
    prev_log_alpha = 0.0
    for _ in range(4):
        d = sac.step(*synth_batch(rs, 128, 42, 7))
        lp = d["Log Pis Mean"]
        # Alpha Loss = -log_alpha_pre * mean(log_pi + H), H = -A
        assert d["Alpha Loss"] == pytest.approx(-prev_log_alpha * (lp - 7.0), rel=1e-4, abs=1e-7)
        # constant-sign gradient => Adam moves log_alpha by ~lr per step
        assert math.log(d["Alpha"]) == pytest.approx(prev_log_alpha - 1e-3, abs=2e-5)
        prev_log_alpha = math.log(d["Alpha"])
    # same relation in the shipped rows (epochs 1..3 of RUN17): the logged pair is 1000 steps apart,
    # so only the sign/scale relation  -AlphaLoss/(LogPi - A) = log_alpha_pre  ~ log(Alpha) + lr  holds
    rec = KA["Lift-Panda-OSC-POSE-SEED17"]
    for row in rec["rows"][1:4]:
        pre = -row["trainer/Alpha Loss"] / (row["trainer/Log Pis Mean"] - rec["act_dim"])
        assert pre == pytest.approx(math.log(row["trainer/Alpha"]) + 1e-3, abs=5e-5)


def test_ka3_logged_policy_loss_has_no_alpha():
    sac = make(42, 7)
    rs = np.random.RandomState(2)
    for _ in range(300):   # drive alpha away from 1
        d = sac.step(*synth_batch(rs, 64, 42, 7))
    L = sac.last
    lp, qn = L["log_pi"].detach().numpy(), np.minimum(L["q1_new"].detach().numpy(), L["q2_new"].detach().numpy())
    assert d["Alpha"] < 0.8
    assert d["Policy Loss"] == pytest.approx(float(np.mean(lp - qn)), rel=1e-6)
    assert d["Actor Loss"] == pytest.approx(float(np.mean(d["Alpha"] * lp - qn)), rel=1e-5)
    assert abs(d["Policy Loss"] - d["Actor Loss"]) > 1e-2


def test_ka4_log_std_clamped_at_two():
    sac = make(42, 7)
    with np.errstate(all="ignore"):
        import torch
        with torch.no_grad():
            sac.policy.bs[-1].fill_(5.0)     # force raw log_std ~ 5
    d = sac.step(*synth_batch(np.random.RandomState(4), 64, 42, 7))
    assert d["Policy log std Max"] == 2.0
    assert KA["_scan"]["policy_log_std_max"] == 2.0
    # the clamp blocks the gradient of the saturated head: weight (parameter 3) and bias (parameter 7) of last_fc_log_std
    assert sac.last["log_std"].detach().numpy().min() == 2.0
    assert np.all(sac.last["g_policy"][3] == 0.0) and np.all(sac.last["g_policy"][7] == 0.0)
    assert np.any(sac.last["g_policy"][2] != 0.0)           # (the mean head next to it does learn)


@pytest.mark.parametrize("run", [k for k in KA if not k.startswith("_")])
def test_ka6_initial_log_pi_level(run):
    A = KA[run]["act_dim"]
    O = {7: 42, 14: 89, 6: 379}[A]
    sac = make(O, A)
    d = sac.step(*synth_batch(np.random.RandomState(5), 128, O, A))
    shipped = KA[run]["rows"][0]["trainer/Log Pis Mean"]
    # zero-mean unit-std heads: E[log_pi] ~ -0.676*A ; shipped epoch-0 value within sampling noise
    assert d["Log Pis Mean"] == pytest.approx(-0.676 * A, abs=0.25)
    assert shipped == pytest.approx(d["Log Pis Mean"], abs=0.35)


def test_ka7_qf_loss_is_plain_mean_squared_error():
    sac = make(42, 7)
    d = sac.step(*synth_batch(np.random.RandomState(6), 128, 42, 7))
    L = sac.last
    diff = (L["q1"] - L["y"]).detach().numpy().astype(np.float64)
    assert d["QF1 Loss"] == pytest.approx(float(np.mean(diff ** 2)), rel=1e-6)
    # identity visible in the shipped epoch-0 row: loss = (mean diff)^2 + var(diff), Q pred ~ const
    r = KA["Lift-Panda-OSC-POSE-SEED17"]["rows"][0]
    approx = (r["trainer/Q Targets Mean"] - r["trainer/Q1 Predictions Mean"]) ** 2 + r["trainer/Q Targets Std"] ** 2
    assert r["trainer/QF1 Loss"] == pytest.approx(approx, rel=2e-3)


def test_q_target_uses_alpha_and_reward_scale_and_terminals():
    sac = make(42, 7, reward_scale=2.0)
    b = synth_batch(np.random.RandomState(7), 32, 42, 7, term_frac=0.3)
    d = sac.step(*b)
    L = sac.last
    import torch
    # recompute y from pieces with pre-step target nets is not possible post-step (period=5,
    # step 0 updates targets) => check the structural identity on terminal rows instead
    term = b[3].ravel() > 0
    assert term.any() and (~term).any()
    y = L["y"].numpy().ravel()
    assert np.allclose(y[term], 2.0 * b[2].ravel()[term], atol=1e-6)
    assert d["Q Targets Mean"] == pytest.approx(float(np.mean(y)), rel=1e-6)


def test_target_update_cadence():
    sac = make(42, 7, target_update_period=5)
    rs = np.random.RandomState(8)
    import torch
    snaps = []
    for i in range(7):
        before = sac.target_qf1.ws[1].detach().clone()
        sac.step(*synth_batch(rs, 16, 42, 7))
        snaps.append(not torch.equal(before, sac.target_qf1.ws[1].detach()))
    assert snaps == [True, False, False, False, False, True, False]


def test_ka5_buffer_size_progression_in_shipped_rows():
    rows = KA["Lift-Panda-OSC-POSE-SEED17"]["rows"]
    assert [r["replay_buffer/size"] for r in rows[:3]] == [5800, 8300, 10800]
    assert KA["_scan"]["replay_size_max"] == 1_000_000


RUN17 = "/root/reference/runs/Lift-Panda-OSC-POSE-SEED17"


@pytest.mark.skipif(not os.path.isdir(RUN17), reason="reference tree not mounted (GPU box)")
def test_embedded_policy_source_holds_the_statements_the_oracle_restates():
    """The legacy-format params.pkl of a shipped run embeds the SOURCE TEXT of TanhGaussianPolicy / FlattenMlp (torch
    stores it for its container check).  Walk the pickle with pickletools.genops -- nothing is unpickled, nothing is
    stored -- and assert the forward() statements oracle.PolicyNet / QNet restate (SURVEY.md 8a a4/a7, App. A lines
    1-3): the log-std clamp, exp, the reparameterised TanhNormal sample with its pre-tanh value, the summed log-prob,
    and the concatenating FlattenMlp.  Build-container only."""
    import glob
    import io
    import pickletools
    path = glob.glob(os.path.join(RUN17, "*", "params.pkl"))[0]
    data = open(path, "rb").read()
    texts, pos = [], 0
    for _ in range(4):                       # magic, protocol, sys info, the object
        f = io.BytesIO(data[pos:])
        end = None
        for op, arg, off in pickletools.genops(f):
            if op.name in ("BINUNICODE", "SHORT_BINUNICODE", "UNICODE", "BINUNICODE8") and isinstance(arg, str) and "def forward" in arg:
                texts.append(arg)
            if op.name == "STOP":
                end = off + 1
                break
        pos += end
    policy_src = next(t for t in texts if "class TanhGaussianPolicy" in t)
    mlp_src = next(t for t in texts if "class FlattenMlp" in t)
    norm = lambda t: " ".join(t.split())     # noqa: E731
    p = norm(policy_src)
    assert "log_std = torch.clamp(log_std, LOG_SIG_MIN, LOG_SIG_MAX)" in p
    assert "std = torch.exp(log_std)" in p
    assert "tanh_normal = TanhNormal(mean, std)" in p
    assert "tanh_normal.rsample( return_pretanh_value=True )" in p
    assert "log_prob = tanh_normal.log_prob( action, pre_tanh_value=pre_tanh_value )" in p
    assert "log_prob = log_prob.sum(dim=1, keepdim=True)" in p
    assert "action = torch.tanh(mean)" in p                      # the deterministic branch MakeDeterministic takes
    assert "h = self.hidden_activation(fc(h))" in p and "mean = self.last_fc(h)" in p
    assert "log_std = self.last_fc_log_std(h)" in p
    m = norm(mlp_src)
    assert "flat_inputs = torch.cat(inputs, dim=1)" in m and "return super().forward(flat_inputs, **kwargs)" in m
