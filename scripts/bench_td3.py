#!/usr/bin/env python3
"""Shorthand for `python bench.py --agent TD3` (TD3 grad-steps/s on the headline workload shape, one JSON line)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv.insert(1, "--agent=TD3")
import bench  # noqa: E402

bench.main()
