"""stepwise interface: steps/s and the host's own time per step (launch calls without waiting for the device)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tr, buf = bench.build_replica("Lift", 42, 7, 256, 200_000, 17, 0)
tr.train_loop(buf, 300, batch_size=256)
for rep in range(3):
    n = 3000
    tr._lib.sac_sync(tr._h)
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train(buf.random_batch(256))
    t1 = time.perf_counter()
    tr._lib.sac_sync(tr._h)
    t2 = time.perf_counter()
    print(f"stepwise: {n / (t2 - t0):8.0f} steps/s; host loop alone {1e6 * (t1 - t0) / n:6.2f} us/step, total {1e6 * (t2 - t0) / n:6.2f} us/step")
