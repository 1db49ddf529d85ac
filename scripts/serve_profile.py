#!/usr/bin/env python3
"""cProfile of bench.py --serve's host side (where do the scheduler's microseconds per step go?).  Usage: serve_profile.py [requests]"""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402

args = bench.parse_args(["--serve", "--serve-requests", sys.argv[1] if len(sys.argv) > 1 else "512"] + sys.argv[2:])
dev = torch.device("cuda:0")
pr = cProfile.Profile()
pr.enable()
rec = bench.serve_measure(args, dev)
pr.disable()
print({k: rec[k] for k in ("value", "decode_steps", "step_us", "host_us_per_step", "gpu_wait_us_per_step", "batch_occupancy", "prefill_s", "preemptions")})
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
