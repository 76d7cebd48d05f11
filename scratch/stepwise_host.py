"""Host-side cost of the stepwise interface's two calls (they return before the device work is done)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tr, buf = bench.build_replica("Lift", 42, 7, 256, 200_000, 17, 0)
B = 256
for _ in range(200):
    tr.train(buf.random_batch(B))
tr._lib.sac_sync(tr._h)
n = 3000
t0 = time.perf_counter()
bs = [buf.random_batch(B) for _ in range(12)]          # (a token stays valid for 16 draws)
t1 = time.perf_counter()
print("random_batch host cost %.2f us/call" % ((t1 - t0) / 12 * 1e6))
tr._lib.sac_sync(tr._h)
t0 = time.perf_counter()
for _ in range(n):
    b = buf.random_batch(B)
    tr.train(b)
t1 = time.perf_counter()
tr._lib.sac_sync(tr._h)
t2 = time.perf_counter()
print("loop issue %.2f us/step, incl. drain %.2f us/step" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    tr.train(buf.random_batch(B))
pr.disable()
tr._lib.sac_sync(tr._h)
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
