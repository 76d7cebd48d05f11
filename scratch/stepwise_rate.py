import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tr, buf = bench.build_replica("Lift", 42, 7, 256, 1_000_000, 17, 0)
for lazy in (True, False):
    buf.lazy_batches = lazy
    for _ in range(100):
        tr.train(buf.random_batch(256))
    n = 3000
    tr._lib.sac_sync(tr._h)
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train(buf.random_batch(256))
    tr._lib.sac_sync(tr._h)
    el = time.perf_counter() - t0
    print("stepwise drop-in interface, lazy=%s: %.0f steps/s (%.1f us/step)" % (lazy, n / el, el / n * 1e6))
