"""Randomised sweep of the fused step's give-up path: a random fused launch loses a producer (SAC_FUSED_TEST_STALL) inside a
random script of loop calls, stepwise device batches and host batches; the trajectory, the counters and the generator must
end where an undisturbed four-launch run ends, bit for bit.   usage: python scratch/fuzz_giveup.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd import EnvReplayBuffer
from tests.helpers import make_pair, make_td3_pair, synth_transitions

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def buf(n, O, A, seed):
    obs, act, rew, term, nobs = synth_transitions(n, O, A, seed=seed, term_frac=0.05)
    b = EnvReplayBuffer(n, obs_dim=O, action_dim=A)
    b.add_block(obs, act, rew, nobs, term)
    b.seed(seed + 1)
    return b


def same(sa, sb):
    return (all(np.array_equal(sa["params"][k], sb["params"][k]) for k in sa["params"]) and
            all(np.array_equal(sa["opt"][k][j], sb["opt"][k][j]) for k in sa["opt"] for j in range(2)) and
            np.array_equal(sa["scalars"], sb["scalars"]))


bad = 0
for c in range(cases):
    O, A, B = int(rs.choice([rs.randint(1, 113), rs.randint(113, 400)])), int(rs.randint(1, 17)), int(rs.randint(1, 257))
    td3 = c % 3 == 2
    mk = (lambda: make_td3_pair(O, A, B, seed=c, noise_seed=c + 1)[1]) if td3 else (lambda: make_pair(O, A, B, seed=c, noise_seed=c + 1)[1])
    script = [(str(rs.choice(["loop", "step", "host"], p=[0.5, 0.35, 0.15])), int(rs.choice([1, rs.randint(1, 8), rs.randint(8, 60)])))
              for _ in range(int(rs.randint(1, 6)))]
    total = sum(n for _, n in script)
    stall = int(rs.randint(1, total + 1))
    os.environ.pop("SAC_FUSED", None)
    os.environ["SAC_FUSED_TEST_STALL"] = str(stall)
    a = mk()
    os.environ.pop("SAC_FUSED_TEST_STALL")
    os.environ["SAC_FUSED"] = "0"
    b = mk()
    os.environ.pop("SAC_FUSED")
    ba, bb = buf(3000, O, A, c), buf(3000, O, A, c)
    for kind, n in script:
        for tr, bf in ((a, ba), (b, bb)):
            if kind == "loop":
                tr.train_loop(bf, n, batch_size=B)
            else:
                for _ in range(n):
                    batch = bf.random_batch(B)
                    if kind == "host":
                        batch = {k: np.array(batch[k]) for k in ("observations", "actions", "rewards", "terminals", "next_observations")}
                    tr.train(batch)
    a._lib.sac_sync(a._h); b._lib.sac_sync(b._h)
    ok = same(a.state_dict(), b.state_dict()) and not a.is_fused()
    ok = ok and np.array_equal(ba.rng_state()[0], bb.rng_state()[0]) and ba.rng_state()[1] == bb.rng_state()[1]
    print(f"case {c}: {'TD3' if td3 else 'SAC'} obs {O} act {A} batch {B} script {script} stall at fused launch {stall}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
