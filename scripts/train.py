#!/usr/bin/env python3
"""Counterpart of the reference's scripts/train.py for the MI355X path:
    python scripts/train.py --variant <variant.json> --seed S --log_dir DIR [--epochs N]
Runs the variant unchanged (batch size, lrs, tau, period, buffer size ... from the JSON) on the
HIP library with a synthetic environment of the task's dimensions (robosuite is not installed)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosuite_benchmark_amd.driver import experiment  # noqa: E402
from robosuite_benchmark_amd.variant import default_variant, load_variant  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", type=str, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--log_dir", type=str, default=None)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--env", type=str, default="Lift")
    ap.add_argument("--batch_size", type=int, default=256)
    ap.add_argument("--agent", type=str, default="SAC", choices=["SAC", "TD3"])
    ap.add_argument("--resume", type=str, default=None,
                    help="an existing run directory (…_0000--s-0): continue it from its checkpoint/ after the last saved epoch")
    ap.add_argument("--no_checkpoint", action="store_true", help="do not save <run_dir>/checkpoint after every epoch")
    args = ap.parse_args()
    variant = load_variant(args.variant) if args.variant else default_variant(env=args.env, seed=args.seed,
                                                                              batch_size=args.batch_size, agent=args.agent)
    run_dir = None
    if args.resume:
        import json
        run_dir = args.resume
        variant = json.load(open(os.path.join(run_dir, "variant.json")))
    elif args.log_dir:
        # the reference's run-directory layout (rlkit setup_logger, observed under runs/):
        #   <log_dir>/<Prefix-with-dashes>/<prefix>_<timestamp>_0000--s-0/{variant.json, progress.csv}
        import datetime
        import json
        ek = variant["expl_environment_kwargs"]
        prefix = "{}_{}_{}_SEED{}".format(ek["env_name"], "".join(ek["robots"]), ek["controller"], args.seed)
        stamp = datetime.datetime.now().strftime("%Y_%m_%d_%H_%M_%S")
        run_dir = os.path.join(args.log_dir, prefix.replace("_", "-"), f"{prefix}_{stamp}_0000--s-0")
        os.makedirs(run_dir, exist_ok=True)
        json.dump(variant, open(os.path.join(run_dir, "variant.json"), "w"), indent=2, sort_keys=True)
    ckpt = os.path.join(run_dir, "checkpoint") if (run_dir and not args.no_checkpoint) else None
    experiment(variant, log_dir=run_dir, seed=args.seed, num_epochs=args.epochs, checkpoint_dir=ckpt,
               resume=bool(args.resume))
