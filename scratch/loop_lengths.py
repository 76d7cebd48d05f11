"""Per-step device time against loop length and against time (is the slowdown of long loops positional or thermal?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tr, buf = bench.build_replica("Lift", 42, 7, 256, 200_000, 17, 0)
tr.train_loop(buf, 300, batch_size=256)
for rep in range(2):
    for steps in (256, 512, 768, 1024, 2048, 4096, 256, 256):
        tr._lib.sac_sync(tr._h)
        tr.train_loop(buf, steps, batch_size=256)
        d = tr.loop_timing_ms()
        print(f"steps {steps:5d}: per step {1e3*d['steps']/steps:6.2f} us", flush=True)
    time.sleep(0.5)
