"""Randomised sweep: the same trajectory through different launch paths must be the same bits -- fused step vs four launches
(batch <= 256; SAC and TD3), sac_train_loop vs the stepwise interface (any batch), random call lengths.
usage: python scratch/fuzz_paths.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd import EnvReplayBuffer
from tests.helpers import make_pair, make_td3_pair, synth_transitions

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
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
    O = int(rs.choice([rs.randint(1, 113), rs.randint(113, 400)]))
    A = int(rs.randint(1, 17))
    B = int(rs.choice([rs.randint(1, 257), rs.randint(1, 257), rs.randint(257, 1200)]))
    td3 = c % 3 == 2
    mk = (lambda **kw: make_td3_pair(O, A, B, seed=c, noise_seed=c + 1, **kw)[1]) if td3 else \
         (lambda **kw: make_pair(O, A, B, seed=c, noise_seed=c + 1, **kw)[1])
    os.environ.pop("SAC_FUSED", None)
    a = mk()
    os.environ["SAC_FUSED"] = "0"
    os.environ["SAC_CHAIN"] = "0"
    b = mk()
    os.environ.pop("SAC_FUSED", None); os.environ.pop("SAC_CHAIN", None)
    s = mk()                                      # stepwise interface, default path
    n_rows = int(rs.randint(max(2, B // 8), 4000))
    ba, bb, bs = buf(n_rows, O, A, c), buf(n_rows, O, A, c), buf(n_rows, O, A, c)
    total = 0
    for _ in range(int(rs.randint(1, 5))):
        n = int(rs.choice([1, rs.randint(1, 6), rs.randint(6, 40)]))
        a.train_loop(ba, n, batch_size=B); b.train_loop(bb, n, batch_size=B)
        for _ in range(n):
            s.train(bs.random_batch(B))
        total += n
    s._lib.sac_sync(s._h)
    sa, sb, ss = a.state_dict(), b.state_dict(), s.state_dict()
    exact_ab = a.fused_mode() != 2                # (k_chain sums its head in another order than the four launches: tolerance-level)
    ok = same(sa, ss) and (same(sa, sb) if exact_ab else True)
    ok = ok and np.array_equal(ba.rng_state()[0], bs.rng_state()[0]) and ba.rng_state()[1] == bs.rng_state()[1]
    print(f"case {c}: {'TD3' if td3 else 'SAC'} obs {O} act {A} batch {B} rows {n_rows} steps {total} kinds {a.fused_mode()}/{b.fused_mode()}: "
          f"{'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
