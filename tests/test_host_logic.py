"""CPU: host-side mirror of the reference interface (network holders, variant schema, env dims,
path collection, multi-GPU task assignment)."""
import json
import os

import numpy as np
import pytest

from robosuite_benchmark_amd import networks, parallel, variant
from robosuite_benchmark_amd.driver import PathCollector, SyntheticEnv, path_information

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_network_holders_follow_the_shipped_layer_names_and_flat_layout():
    pol = networks.TanhGaussianPolicy([256, 256], obs_dim=42, action_dim=7)
    q = networks.FlattenMlp([256, 256], output_size=1, input_size=49)
    assert list(pol.layers) == ["fc0", "fc1", "last_fc", "last_fc_log_std"]     # params.pkl layer names
    assert list(q.layers) == ["fc0", "fc1", "last_fc"]
    assert pol.flat().size == 80398 and q.flat().size == 78849
    assert pol.layers["fc0"][0].shape == (256, 42) and q.layers["last_fc"][0].shape == (1, 256)
    assert np.abs(pol.layers["last_fc"][0]).max() <= 1e-3 and np.abs(q.layers["last_fc"][0]).max() <= 3e-3
    v = pol.flat()
    pol2 = networks.TanhGaussianPolicy([256, 256], obs_dim=42, action_dim=7)
    pol2.load_flat(v)
    assert np.array_equal(pol2.flat(), v)
    assert set(pol.state_dict()) == {f"{n}.{k}" for n in pol.layers for k in ("weight", "bias")}
    assert pol.to("cuda:0") is pol and pol.train(True) is pol                   # rlkit_custom.py:306-312 no-ops


def test_get_action_contract():
    pol = networks.TanhGaussianPolicy([256, 256], obs_dim=10, action_dim=3)
    a, info = pol.get_action(np.zeros(10))
    assert a.shape == (3,) and info == {} and np.all(np.abs(a) < 1)
    det = networks.MakeDeterministic(pol)
    a1, _ = det.get_action(np.ones(10))
    a2, _ = det.get_action(np.ones(10))
    assert np.array_equal(a1, a2)
    mean, _ = pol._trunk(np.ones((1, 10), np.float32))
    assert np.allclose(a1, np.tanh(mean[0]))
    assert det.stochastic_policy is pol                                        # rlkit_utils.py:250 relies on this


@pytest.mark.parametrize("name,O,A,B", [("Lift-Panda-OSC-POSE-SEED17", 42, 7, 128),
                                        ("TwoArmLift-PandaPanda-OSC-POSE-SEED17", 89, 14, 128),
                                        ("LiftModded-Jaco-OSC-POSITION-SEED251", 64, 4, 128),
                                        ("LiftModded-Jaco-diverse-rewards-2-2-1-0-0", 64, 4, 128)])
def test_shipped_variant_json_parses_unchanged(name, O, A, B):
    v = variant.validate(variant.load_variant(os.path.join(GOLD, name + ".variant.json")))
    assert variant.env_dims(v["expl_environment_kwargs"]) == (O, A)
    assert v["algorithm_kwargs"]["batch_size"] == B and v["replay_buffer_size"] == 1_000_000
    tk = v["trainer_kwargs"]
    assert (tk["target_update_period"], tk["soft_target_tau"], tk["policy_lr"], tk["qf_lr"]) == (5, 0.005, 1e-3, 5e-4)


def test_the_forks_extra_env_kwargs_pass_through():
    """training_configs/**/variant.json carry `weights` (LiftModded's reward weights): an env-constructor kwarg the
    dims lookup and the driver must leave alone."""
    v = variant.validate(variant.load_variant(os.path.join(GOLD, "LiftModded-Jaco-diverse-rewards-2-2-1-0-0.variant.json")))
    assert len(v["expl_environment_kwargs"]["weights"]) == 5
    assert variant.env_dims(v["eval_environment_kwargs"]) == (64, 4)


def test_pinned_dims_table_equals_the_fixture_read_off_the_shipped_snapshots():
    fx = json.load(open(os.path.join(GOLD, "env_dims.json")))["dims"]
    want = {(k.split("|")[0], tuple(k.split("|")[1].split("+")), k.split("|")[2]): tuple(v) for k, v in fx.items()}
    assert want == variant.PINNED_DIMS and len(want) == 31
    # the Panda / Sawyer rule reproduces every pinned entry it covers (so unseen combinations follow the same rule)
    for (env, robots, ctrl), dims in want.items():
        if all(r in variant.RULE_ROBOTS for r in robots) and (env, len(robots)) in variant.OBS_DIMS:
            per_arm = variant.ACT_PER_ARM[ctrl] - (1 if env in variant.NO_GRIPPER else 0)
            assert (variant.OBS_DIMS[(env, len(robots))], per_arm * len(robots)) == dims, (env, robots, ctrl)
    with pytest.raises(KeyError, match="no pinned dims"):
        variant.env_dims(dict(env_name="Stack", robots=["Jaco"], controller="OSC_POSE"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/training_configs"), reason="reference tree not mounted (GPU box)")
def test_every_shipped_variant_json_parses_unchanged():
    """All 459 files under /root/reference/{runs,log/runs,training_configs}: schema check + dims for both envs."""
    import glob
    files = [f for d in ("runs", "log/runs", "training_configs")
             for f in glob.glob(f"/root/reference/{d}/**/variant.json", recursive=True)]
    assert len(files) == 459
    seen = set()
    for f in files:
        v = variant.validate(variant.load_variant(f))
        d = variant.env_dims(v["expl_environment_kwargs"])
        assert d == variant.env_dims(v["eval_environment_kwargs"])
        assert v["policy_kwargs"]["hidden_sizes"] == [256, 256] == v["qf_kwargs"]["hidden_sizes"]
        seen.add(d)
    assert (64, 4) in seen and (50, 4) in seen and (379, 7) in seen and (73, 12) in seen


def test_default_variant_matches_argparse_defaults_and_dims_table():
    v = variant.default_variant()
    assert v["algorithm_kwargs"]["batch_size"] == 256 and v["trainer_kwargs"]["target_update_period"] == 1
    assert variant.env_dims(dict(env_name="Wipe", robots=["Panda"], controller="OSC_POSE")) == (379, 6)
    assert variant.env_dims(dict(env_name="Door", robots="Panda", controller="JOINT_VELOCITY")) == (46, 8)
    assert variant.env_dims(dict(env_name="Foo", robots=["Panda"]), obs_dim=5, action_dim=2) == (5, 2)
    with pytest.raises(KeyError):
        variant.env_dims(dict(env_name="Foo", robots=["Panda"]))
    with pytest.raises(ValueError):
        variant.validate(dict(v, algorithm="DDPG"))
    td3 = variant.validate(variant.default_variant(agent="TD3"))     # scripts/train.py:38-47, arguments.py:141-156
    assert td3["algorithm"] == "TD3" and td3["trainer_kwargs"] == dict(
        target_policy_noise=0.2, discount=0.99, reward_scale=1.0, policy_learning_rate=1e-3, qf_learning_rate=5e-4,
        policy_and_target_update_period=2, tau=0.005)


def test_path_collection_matches_the_shipped_epoch0_counts():
    """progress.csv epoch 0: 3300 + 2500 exploration steps in 12 paths (6x500 + 300 + 5x500)."""
    env = SyntheticEnv(6, 2, seed=0)
    pol = networks.TanhGaussianPolicy([256, 256], obs_dim=6, action_dim=2)
    pc = PathCollector(env, pol)
    p0 = pc.collect_new_paths(500, 3300, discard_incomplete_paths=False)
    assert [len(p["actions"]) for p in p0] == [500] * 6 + [300]
    pc.end_epoch(-1)
    p1 = pc.collect_new_paths(500, 2500, discard_incomplete_paths=False)
    assert len(p1) == 5 and pc.get_diagnostics() == {"num steps total": 5800, "num paths total": 12}
    ev = PathCollector(env, networks.MakeDeterministic(pol))
    ev.collect_new_paths(500, 2500, discard_incomplete_paths=True)
    info = path_information(ev.epoch_paths, "evaluation/", expl_len=500)
    for col in ("evaluation/Returns Mean", "evaluation/ExplReturns Mean", "evaluation/Actions Std",
                "evaluation/Num Paths", "evaluation/Average Returns", "evaluation/path length Max"):
        assert col in info
    assert p1[0]["observations"].shape == (500, 6) and p1[0]["rewards"].shape == (500, 1)


def test_progress_columns_cover_the_reference_header():
    from robosuite_benchmark_amd._lib import DIAG_NAMES
    ka = json.load(open(os.path.join(GOLD, "progress_known_answers.json")))
    shipped = [k for k in ka["Lift-Panda-OSC-POSE-SEED17"]["rows"][0] if k.startswith("trainer/")]
    ours = {"trainer/" + n for n in DIAG_NAMES if n != "Actor Loss"}
    assert set(shipped) == ours


def test_rank_to_task_assignment():
    assert parallel.task_for_rank(0) == ("Lift", 42, 7, 17)
    assert parallel.task_for_rank(5)[:3] == ("Lift", 42, 7) and parallel.task_for_rank(5)[3] == 22
    tasks = [parallel.task_for_rank(r, sweep=True)[0] for r in range(8)]
    assert tasks == ["Lift", "Door", "Stack", "Wipe", "PickPlaceCan", "NutAssemblyRound", "TwoArmPegInHole",
                     "TwoArmHandoff"]
    assert parallel.aggregate_steps_per_second(8, 1000, 0.5) == 16000


def test_hidden_sizes_any_depth_are_accepted_by_the_sac_holder_and_bounded():
    """variant['policy_kwargs'|'qf_kwargs']['hidden_sizes'] (arguments.py:98,104): SACTrainer takes 1..7 hidden layers of
    1..4096 units (two of <= 256: the fused kernels, anything else: the general step); so does TD3Trainer."""
    import pytest
    from robosuite_benchmark_amd import FlattenMlp, SACTrainer, TanhGaussianPolicy, TanhMlpPolicy, TD3Trainer

    def sac(hp, hq):
        pol = TanhGaussianPolicy(hp, 10, 3, rs=np.random.RandomState(0))
        qs = [FlattenMlp(hq, 1, 13, rs=np.random.RandomState(1)) for _ in range(4)]
        return SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3])     # (no handle yet: CPU)

    t = sac([512, 512, 64], [300])
    assert t._hidden("policy") == [512, 512, 64] and t._hidden("qf1") == [300]
    assert t.policy.flat().size == 512 * 10 + 512 + 512 * 512 + 512 + 64 * 512 + 64 + 2 * (3 * 64 + 3)
    for bad in ([], [64] * 8, [5000]):
        with pytest.raises(RuntimeError, match="hidden_sizes"):
            sac(bad, [256, 256])._hidden("policy")
    pols = [TanhMlpPolicy([256, 256, 256], 3, 10, rs=np.random.RandomState(2)) for _ in range(2)]
    qs = [FlattenMlp([256, 256], 1, 13, rs=np.random.RandomState(3)) for _ in range(4)]
    td3 = TD3Trainer(policy=pols[0], qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], target_policy=pols[1])
    assert td3._hidden("policy") == [256, 256, 256] == td3._hidden("target_policy") and td3._hidden("qf1") == [256, 256]
    td3.target_policy = TanhMlpPolicy([256, 256], 3, 10, rs=np.random.RandomState(2))
    with pytest.raises(RuntimeError, match="share their hidden_sizes"):
        td3._create(32)
