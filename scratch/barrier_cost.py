"""What the bench's bracket costs on an idle device: torch.cuda.synchronize(), sac_sync, and the Python side of train_loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
tr, buf = bench.build_replica("Lift", 42, 7, 256, 200_000, 17, 0)
tr.train_loop(buf, 50, batch_size=256)
def t(f, n=200):
    v = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); v.append(time.perf_counter() - t0)
    return np.median(v) * 1e6, np.percentile(v, 90) * 1e6
print("torch.cuda.synchronize() idle: median %.1f us p90 %.1f" % t(torch.cuda.synchronize))
print("sac_sync idle: median %.1f us p90 %.1f" % t(lambda: tr._lib.sac_sync(tr._h)))
import ctypes as C
def dev_sync():
    torch.cuda.current_stream().synchronize()
print("torch current_stream().synchronize() idle: median %.1f us p90 %.1f" % t(dev_sync))
for n in (1, 20):
    v = []
    for _ in range(30):
        torch.cuda.synchronize(); time.sleep(0.001)
        t0 = time.perf_counter(); tr.train_loop(buf, n, batch_size=256); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tr._lib.sac_sync(tr._h); t3 = time.perf_counter()
        v.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
    v = np.median(np.array(v), axis=0)
    print(f"train_loop({n}): call {v[0]:.1f} us, torch.cuda.synchronize() behind it {v[1]:.1f} us, sac_sync behind that {v[2]:.1f} us")
