"""TD3 oracle (rlkit-equivalent restatement, parity unpinned -- the reference holds no TD3 number): internal
consistency with the algorithm it restates (/root/reference/util/rlkit_utils.py:107-135, scripts/train.py:38-47)."""
import numpy as np
import torch

from oracle.td3_step_torch import RlkitEquivalentTD3, init_td3_params


def _batch(B, O, A, seed):
    rs = np.random.RandomState(seed)
    return (rs.normal(0, 0.5, (B, O)).astype(np.float32), rs.uniform(-1, 1, (B, A)).astype(np.float32),
            rs.uniform(0, 1, (B, 1)).astype(np.float32), (rs.uniform(size=(B, 1)) < 0.1).astype(np.float32),
            rs.normal(0, 0.5, (B, O)).astype(np.float32), rs.standard_normal((B, A)).astype(np.float32))


def test_delayed_policy_and_target_updates():
    O, A, B = 11, 3, 32
    td3 = RlkitEquivalentTD3(init_td3_params(O, A, seed=1), A, policy_and_target_update_period=2, tau=0.005)
    before = td3.export_nets()
    td3.step(*_batch(B, O, A, 0))                     # step 0: critics + policy + all three targets
    mid = td3.export_nets()
    for net in ("qf1", "qf2", "policy", "target_policy", "target_qf1", "target_qf2"):
        assert any(not np.array_equal(a[0], b[0]) for a, b in zip(before[net], mid[net])), net
    # Polyak: target = (1 - tau) target_old + tau source_new
    w_new, w_old, w_src = mid["target_qf1"][1][0], before["target_qf1"][1][0], mid["qf1"][1][0]
    np.testing.assert_allclose(w_new, 0.995 * w_old + 0.005 * w_src, rtol=0, atol=1e-7)
    td3.step(*_batch(B, O, A, 1))                     # step 1: critics only
    after = td3.export_nets()
    for net in ("policy", "target_policy", "target_qf1", "target_qf2"):
        assert all(np.array_equal(a[0], b[0]) for a, b in zip(mid[net], after[net])), net
    assert not np.array_equal(mid["qf1"][0][0], after["qf1"][0][0])
    assert td3.last["g_policy"] is None and td3.last["policy_step"] is False


def test_target_smoothing_noise_is_scaled_then_clipped_and_not_reclipped():
    O, A, B = 5, 2, 16
    td3 = RlkitEquivalentTD3(init_td3_params(O, A, seed=2), A, target_policy_noise=0.2, target_policy_noise_clip=0.5)
    b = list(_batch(B, O, A, 3))
    b[5] = np.full((B, A), 10.0, np.float32)          # 10 * 0.2 = 2 -> clipped to 0.5
    td3.step(*b)
    diff = (td3.last["noisy"] - td3.last["a2"]).numpy()
    np.testing.assert_allclose(diff, 0.5, atol=1e-6)
    assert float(td3.last["noisy"].abs().max()) <= 1.5 and float(td3.last["a2"].abs().max()) <= 1.0


def test_policy_loss_goes_through_the_updated_qf1_and_stats_are_complete():
    O, A, B = 7, 2, 64
    td3 = RlkitEquivalentTD3(init_td3_params(O, A, seed=4), A)
    batch = _batch(B, O, A, 5)
    d = td3.step(*batch)
    obs = torch.from_numpy(batch[0])
    with torch.no_grad():                              # qf1 has been stepped; the logged loss used exactly these weights
        # (the policy itself moved afterwards, so recompute with the logged actions)
        want = -td3.qf1(obs, td3.last["pa"].detach()).mean()
    assert abs(d["Policy Loss"] - float(want)) < 1e-6
    for k in ("QF1 Loss", "QF2 Loss", "Policy Loss", "Q1 Predictions Mean", "Q2 Predictions Std", "Q Targets Max",
              "Bellman Errors 1 Mean", "Bellman Errors 2 Min", "Policy Action Mean"):
        assert k in d and np.isfinite(d[k])
    assert abs(d["QF1 Loss"] - d["Bellman Errors 1 Mean"]) < 1e-6
