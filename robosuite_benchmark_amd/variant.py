"""variant.json handling (mirror of the schema scripts/train.py builds and consumes).

Reference: /root/reference/scripts/train.py:53-96 (variant dict / --variant JSON),
/root/reference/util/arguments.py:87-202 (defaults), /root/reference/util/rlkit_utils.py:39-62
(env construction -> obs_dim / action_dim).  robosuite is not installed where this library runs, so
the (obs_dim, action_dim) the reference reads off the env come from a table pinned by the fc0 /
last_fc shapes inside the shipped params.pkl files (dims table)."""
from __future__ import annotations

import json

# Every (env_name, robots, controller) the reference ships a run or a training config for -> (obs_dim, action_dim),
# read off the tensor shapes inside its params.pkl snapshots (policy fc0 / last_fc, qf fc0) by
# tests/golden/make_dims_fixture.py -> tests/golden/env_dims.json (pickletools walk, nothing unpickled).
# LiftModded (the fork's env, /root/reference/training_configs/*) adds object / goal features: 64 with the Jaco arm.
PINNED_DIMS = {
    ("Door", ('Panda',), "JOINT_VELOCITY"): (46, 8),
    ("Door", ('Panda',), "OSC_POSE"): (46, 7),
    ("Door", ('Sawyer',), "JOINT_VELOCITY"): (46, 8),
    ("Door", ('Sawyer',), "OSC_POSE"): (46, 7),
    ("LiftModded", ('Jaco',), "OSC_POSITION"): (64, 4),
    ("Lift", ('Jaco',), "OSC_POSITION"): (50, 4),
    ("Lift", ('Panda',), "JOINT_VELOCITY"): (42, 8),
    ("Lift", ('Panda',), "OSC_POSE"): (42, 7),
    ("Lift", ('Sawyer',), "JOINT_VELOCITY"): (42, 8),
    ("Lift", ('Sawyer',), "OSC_POSE"): (42, 7),
    ("NutAssemblyRound", ('Panda',), "OSC_POSE"): (46, 7),
    ("NutAssemblyRound", ('Sawyer',), "OSC_POSE"): (46, 7),
    ("PickPlaceCan", ('Panda',), "OSC_POSE"): (46, 7),
    ("PickPlaceCan", ('Sawyer',), "OSC_POSE"): (46, 7),
    ("PickPlaceMilk", ('Panda',), "OSC_POSE"): (46, 7),
    ("PickPlaceMilk", ('Sawyer',), "OSC_POSE"): (46, 7),
    ("Stack", ('Panda',), "JOINT_VELOCITY"): (55, 8),
    ("Stack", ('Panda',), "OSC_POSE"): (55, 7),
    ("Stack", ('Sawyer',), "JOINT_VELOCITY"): (55, 8),
    ("Stack", ('Sawyer',), "OSC_POSE"): (55, 7),
    ("TwoArmHandoff", ('Panda', 'Panda'), "OSC_POSE"): (86, 14),
    ("TwoArmHandoff", ('Sawyer', 'Sawyer'), "OSC_POSE"): (86, 14),
    ("TwoArmLift", ('Panda', 'Panda'), "OSC_POSE"): (89, 14),
    ("TwoArmLift", ('Sawyer', 'Sawyer'), "OSC_POSE"): (89, 14),
    ("TwoArmPegInHole", ('Panda', 'Panda'), "OSC_POSE"): (73, 12),
    ("TwoArmPegInHole", ('Panda', 'Sawyer'), "OSC_POSE"): (73, 12),
    ("TwoArmPegInHole", ('Sawyer', 'Sawyer'), "OSC_POSE"): (73, 12),
    ("Wipe", ('Panda',), "JOINT_VELOCITY"): (379, 7),
    ("Wipe", ('Panda',), "OSC_POSE"): (379, 6),
    ("Wipe", ('Sawyer',), "JOINT_VELOCITY"): (379, 7),
    ("Wipe", ('Sawyer',), "OSC_POSE"): (379, 6),
}
# the rule behind the table, for combinations of known tasks nobody shipped a run for (Panda / Sawyer arms observe
# alike; the Jaco's three-finger gripper adds proprioception, so it is only served from the pinned table):
# (env_name, n_robots) -> obs_dim for runs with use_object_obs=True, no cameras
OBS_DIMS = {
    ("Lift", 1): 42, ("Door", 1): 46, ("PickPlaceCan", 1): 46, ("PickPlaceMilk", 1): 46,
    ("NutAssemblyRound", 1): 46, ("Stack", 1): 55, ("Wipe", 1): 379,
    ("TwoArmPegInHole", 2): 73, ("TwoArmHandoff", 2): 86, ("TwoArmLift", 2): 89,
}
# controller -> action dims per arm (gripper included); Wipe and TwoArmPegInHole run without grippers (one fewer)
NO_GRIPPER = ("Wipe", "TwoArmPegInHole")
ACT_PER_ARM = {"OSC_POSE": 7, "OSC_POSITION": 4, "JOINT_VELOCITY": 8, "JOINT_TORQUE": 8, "JOINT_POSITION": 8,
               "IK_POSE": 7}
RULE_ROBOTS = ("Panda", "Sawyer")


def load_variant(path):
    with open(path) as f:
        return json.load(f)


def default_variant(env="Lift", robots=("Panda",), controller="OSC_POSE", seed=1, batch_size=256,
                    target_update_period=1, agent="SAC"):
    """The dict scripts/train.py:53-77 builds from the argparse defaults (arguments.py); agent "TD3": the trainer
    kwargs of scripts/train.py:38-47 with the defaults of arguments.py:141-156."""
    envkw = dict(env_name=env, robots=list(robots), horizon=500, control_freq=20, controller=controller,
                 reward_scale=1.0, hard_reset=False, ignore_done=True)
    if agent == "TD3":
        v = default_variant(env, robots, controller, seed, batch_size, target_update_period)
        v["algorithm"] = "TD3"
        v["trainer_kwargs"] = dict(target_policy_noise=0.2, discount=0.99, reward_scale=1.0, policy_learning_rate=1e-3,
                                   qf_learning_rate=5e-4, policy_and_target_update_period=2, tau=0.005)
        return v
    return dict(
        algorithm="SAC", seed=seed, version="normal", replay_buffer_size=int(1e6),
        qf_kwargs=dict(hidden_sizes=[256, 256]), policy_kwargs=dict(hidden_sizes=[256, 256]),
        algorithm_kwargs=dict(num_epochs=2000, num_eval_steps_per_epoch=2500, num_trains_per_train_loop=1000,
                              num_expl_steps_per_train_loop=2500, min_num_steps_before_training=3300,
                              expl_max_path_length=500, eval_max_path_length=500, batch_size=batch_size),
        trainer_kwargs=dict(discount=0.99, soft_target_tau=5e-3, target_update_period=target_update_period,
                            policy_lr=1e-3, qf_lr=5e-4, reward_scale=1.0, use_automatic_entropy_tuning=True),
        expl_environment_kwargs=dict(envkw), eval_environment_kwargs=dict(envkw))


def env_dims(env_kwargs, obs_dim=None, action_dim=None):
    """(obs_dim, action_dim) the reference would read from the robosuite env (rlkit_utils.py:61-62)."""
    if obs_dim is not None and action_dim is not None:
        return int(obs_dim), int(action_dim)
    # (keys the table does not use -- horizon, control_freq, reward_scale, hard_reset, ignore_done, the fork's
    #  `weights` of LiftModded ... -- belong to the env constructor and pass through untouched)
    name, robots = env_kwargs["env_name"], env_kwargs["robots"]
    robots = [robots] if isinstance(robots, str) else list(robots)
    ctrl = env_kwargs.get("controller", "OSC_POSE")
    pinned = PINNED_DIMS.get((name, tuple(robots), ctrl))
    if pinned is not None:
        return pinned
    key = (name, len(robots))
    if key not in OBS_DIMS or ctrl not in ACT_PER_ARM or any(r not in RULE_ROBOTS for r in robots):
        raise KeyError(f"no pinned dims for env {name!r} x {robots!r} / controller {ctrl!r}: pass obs_dim/action_dim")
    per_arm = ACT_PER_ARM[ctrl] - (1 if name in NO_GRIPPER else 0)
    return OBS_DIMS[key], per_arm * len(robots)


def validate(variant):
    """The checks the reference does implicitly (KeyError / assert at rlkit_utils.py:34)."""
    if variant.get("algorithm", "SAC") not in ("SAC", "TD3"):
        raise ValueError(f"agent {variant.get('algorithm')!r}: the reference knows SAC and TD3 (rlkit_utils.py:91-137)")
    for k in ("replay_buffer_size", "qf_kwargs", "policy_kwargs", "algorithm_kwargs", "trainer_kwargs",
              "expl_environment_kwargs", "eval_environment_kwargs"):
        if k not in variant:
            raise KeyError(f"variant is missing {k!r}")
    return variant
