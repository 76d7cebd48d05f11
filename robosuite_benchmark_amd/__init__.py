"""MI355X-native SAC training inner loop: drop-in for the hot path of
jeremy29tien/robosuite-benchmark (replay_buffer.random_batch -> SACTrainer.train).

Host-side mirror of the rlkit duck types the reference consumes
(/root/reference/util/rlkit_utils.py:64-161, /root/reference/util/rlkit_custom.py:199-312),
over the C ABI of include/sac_hip.h.  All compute is in libsac_hip.so (HIP, gfx950);
there is no CPU fallback."""
from .networks import (FlattenMlp, GaussianStrategy, MakeDeterministic,  # noqa: F401
                       PolicyWrappedWithExplorationStrategy, TanhGaussianPolicy, TanhMlpPolicy)
from .replay_buffer import DeviceBatch, EnvReplayBuffer  # noqa: F401
from .sac import SACTrainer  # noqa: F401
from .td3 import TD3Trainer  # noqa: F401

__all__ = ["EnvReplayBuffer", "DeviceBatch", "FlattenMlp", "TanhGaussianPolicy", "MakeDeterministic", "SACTrainer",
           "TD3Trainer", "TanhMlpPolicy", "GaussianStrategy", "PolicyWrappedWithExplorationStrategy"]
