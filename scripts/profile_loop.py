#!/usr/bin/env python3
"""Minimal driver for rocprofv3 passes: the BASELINE config-2 workload (Lift 42/7, batch 256,
1e6-slot buffer) run for a few hundred steps through sac_train_loop, nothing else in the process.
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 scripts/profile_loop.py --steps 500
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 scripts/profile_loop.py"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--buffer", type=int, default=1_000_000)
    ap.add_argument("--task", type=str, default="Lift")
    args = ap.parse_args()
    task, O, A = next(t for t in bench.parallel.SWEEP + bench.EXTRA_TASKS if t[0] == args.task)
    trainer, buf = bench.build_replica(task, O, A, args.batch, args.buffer, seed=17, device=0)
    trainer.train_loop(buf, 50, batch_size=args.batch)
    first, last = trainer.train_loop(buf, args.steps, batch_size=args.batch)
    print("done", args.steps, "steps; QF1 loss", float(last[0]), "device ms", trainer.loop_timing_ms())
