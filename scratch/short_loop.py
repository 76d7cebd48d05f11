"""Where does a short sac_train_loop call spend its time?  usage: python scratch/short_loop.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tr, buf = bench.build_replica("Lift", 42, 7, 256, 200_000, 17, 0)
tr.train_loop(buf, 300, batch_size=256)
res = {}
for steps in (n, n, n, n, n, 1, 1, 520):
    tr._lib.sac_sync(tr._h)
    time.sleep(0.002)
    t0 = time.perf_counter()
    tr.train_loop(buf, steps, batch_size=256)
    t1 = time.perf_counter()
    d = tr.loop_timing_ms()
    res.setdefault(steps, []).append((1e6*(t1-t0), 1e3*d['steps'], 1e3*d['sample'], 1e3*d['gather']))
for steps, v in res.items():
    v = np.median(np.array(v), axis=0)
    print(f"plan {os.environ.get('SAC_CHUNK_PLAN','default'):>14s} steps {steps:4d}: wall {v[0]:8.1f} us  device span {v[1]:8.1f} us  per step {v[1]/steps:6.2f}  "
          f"sample {v[2]:6.1f} gather {v[3]:6.1f}  wall-device {v[0]-v[1]:6.1f}  -> {steps/v[0]*1e6:8.0f} steps/s")
