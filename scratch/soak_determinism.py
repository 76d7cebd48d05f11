"""Soak: two identically seeded runs of N fused-loop steps must end bitwise equal (catches rare races)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
for agent in ("SAC", "TD3"):
    finals = []
    for rep in range(2):
        if agent == "SAC":
            tr, buf = bench.build_replica("Lift", 42, 7, 256, 200_000, 17, 0)
        else:
            tr, buf = bench.build_td3_replica(42, 7, 256, 200_000, 17)
        t0 = time.time()
        first, last = tr.train_loop(buf, N, batch_size=256)
        st = tr.state_dict()
        finals.append((st, buf.rng_state(), last.copy()))
        print(agent, "run", rep, "%.1f s" % (time.time() - t0), "QF1 loss", float(last[0]), "finite", bool(np.all(np.isfinite(last[:28]))), flush=True)
    a, b = finals
    if agent == "SAC":
        finals_sac = finals
    same = all(np.array_equal(a[0]["params"][k], b[0]["params"][k]) for k in a[0]["params"]) and \
        all(np.array_equal(a[0]["opt"][k][0], b[0]["opt"][k][0]) and np.array_equal(a[0]["opt"][k][1], b[0]["opt"][k][1]) for k in a[0]["opt"]) and \
        np.array_equal(a[0]["scalars"], b[0]["scalars"]) and np.array_equal(a[1][0], b[1][0]) and a[1][1] == b[1][1] and np.array_equal(a[2], b[2])
    print(agent, "bitwise identical after", N, "steps:", same, "| fused step:", bool(getattr(tr, "is_fused", lambda: False)()), flush=True)
    assert same
# the fused step against the four-launch step over the same long run
os.environ["SAC_FUSED"] = "0"
tr, buf = bench.build_replica("Lift", 42, 7, 256, 200_000, 17, 0)
os.environ.pop("SAC_FUSED")
first, last = tr.train_loop(buf, N, batch_size=256)
st = tr.state_dict()
same = all(np.array_equal(st["params"][k], finals_sac[0][0]["params"][k]) for k in st["params"]) and np.array_equal(last, finals_sac[0][2])
print("four-launch step == fused step after", N, "steps:", same, flush=True)
assert same
