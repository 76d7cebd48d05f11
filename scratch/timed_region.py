"""Split the bench's timed region of a 20-step call into its host parts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from robosuite_benchmark_amd import _lib
tr, buf = bench.build_replica("Lift", 42, 7, 256, 200_000, 17, 0)
bench.measure_peaks(0)
tr.train_loop(buf, 5, batch_size=256)
def sync():
    torch.cuda.synchronize(); tr._lib.sac_sync(tr._h)
for rep in range(6):
    sync()
    t0 = time.perf_counter()
    first, last = np.empty(32, np.float32), np.empty(32, np.float32)
    t1 = time.perf_counter()
    _lib.check(tr._lib.sac_train_loop(tr._h, buf._h, 20, _lib.ptr(first), _lib.ptr(last)), "x")
    t2 = time.perf_counter()
    tr._need_to_update_eval_statistics = True; tr._record(first)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    tr._lib.sac_sync(tr._h)
    t5 = time.perf_counter()
    d = tr.loop_timing_ms()
    print(f"alloc {1e6*(t1-t0):5.1f}  C call {1e6*(t2-t1):6.1f} (device span {1e3*d['steps']:6.1f})  record {1e6*(t3-t2):5.1f}  torch.sync {1e6*(t4-t3):5.1f}  sac_sync {1e6*(t5-t4):5.1f}  total {1e6*(t5-t0):6.1f}")
