#!/usr/bin/env python3
"""The decode attention at the shapes the serving loop runs (bench.py --serve: ~100-200 running sequences, contexts = prompt U{4..512}
+ up to 488 generated): what does the default entry (with the harness's mean-length hint) pick, and what would be best?
us per launch (hipGraph of 12 launches over 2 table sets), bytes that exist, fraction of 8 TB/s."""
import dataclasses, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from vllmini_amd import ops
from vllmini_amd.workload import CONFIGS, make_workload
import test_pick_guard_gpu as G

dev = torch.device("cuda:0")
names = ops.variant_names()
rng = np.random.default_rng(0)
for B in (64, 128, 192, 256):
    lens = np.minimum(rng.integers(4, 513, B) + np.minimum(rng.geometric(1 / 122, B), 488) // 2, 1023).astype(np.int32)
    cfg = dataclasses.replace(CONFIGS["cfg3"], name=f"serve_b{B}", batch=B, num_blocks=2 * B * 64 + 8)
    wl = make_workload(cfg, dev, seed=B, table_sets=2)
    wl.seq_lens = torch.from_numpy(lens).to(dev)
    out = torch.empty((B, 12, 64), dtype=torch.float16, device=dev)
    nbytes = int(lens.sum()) * 12 * 64 * 4 + 2 * B * 12 * 64 * 2
    hint = ops.pick_variant(B, 12, 64, int(lens.max()), 16, mean_seq_len=int(lens.mean()))
    cands = [("default (no hint)", 0), (f"hinted -> {names[hint - 1] if hint else '?'}", hint)] + \
        [(n, names.index(n) + 1) for n in ("q_d64_s1q2", "d64_h4_w1_u1_nt1", "d64_h1_w2_u1_nt1", "d64_h1_w4_u1_nt1", "d64_h1_w8_u1_nt1", "d64_h1_w8_u1_nt0", "d64_h1_w4_u2_nt0", "d64_h1_w16_u1_nt0")]
    print(f"B={B} mean len {lens.mean():.0f} max {lens.max()} bytes {nbytes / 1e6:.0f} MB")
    for what, vid in cands:
        try:
            us, label = G._graph_us(wl, out, vid, dev)
        except RuntimeError as e:
            print(f"    {what:40s} refused {str(e)[:60]}")
            continue
        print(f"    {what:40s} {label:22s} {us:8.2f} us  {nbytes / us / 1e6 / 8000:.3f}")
