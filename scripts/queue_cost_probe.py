"""What does an item cost a worker of the balanced kernel beside its pages?  Equal lengths L, forced mode Q (ranked
snake), batch chosen so that every worker runs the same number of items: time / items-per-worker = a + b*L.
`python scripts/queue_cost_probe.py [--kv fp8]`."""
import argparse
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import _lib, ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="cfg3")
ap.add_argument("--iters", type=int, default=100)
args = ap.parse_args()
lib = _lib.use_diag().__enter__()   # the diagnostic build for the whole process (python -m vllmini_amd.build --diag)
names = {lib.vmi_paged_attention_v1_variant_name(i).decode(): i
         for i in range(1, lib.vmi_paged_attention_v1_variant_count() + 1)}
dev = torch.device("cuda:0")


def flags(mode=0, wq=0, nosort=0, team=0):
    return mode | (wq << 2) | (nosort << 11) | (team << 12)


for batch in (256, 512):
    cfg = dataclasses.replace(CONFIGS[args.cfg], batch=batch)
    D = cfg.head_size
    wl = make_workload(cfg, dev, seed=0, ragged=False)
    out = torch.empty((cfg.batch, cfg.num_heads, D), dtype=torch.float16, device=dev)
    variant = names[f"q_d{D}_s1q2"]
    N = batch * cfg.num_heads
    for label, f, W in (("solo", flags(2, 2, 0, 1), 1536), ("solo4", flags(2, 4, 0, 1), 3072), ("team", flags(2, 0, 0, 2), 768)):
        lib.vmi_debug_set_queue_flags(f)
        xs, ys = [], []
        for L in (16, 32, 64, 128, 256, 384, 512, 768, 1024):
            wl.seq_lens.fill_(L)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
            for i in range(args.iters + 5):
                if i >= 5:
                    ev[i - 5][0].record()
                ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale,
                                       wl.tables[i % len(wl.tables)], wl.seq_lens, cfg.block_size, cfg.seq_len, None,
                                       "auto", 1.0, 0, 0, 1, 1, 0, _variant=variant)
                if i >= 5:
                    ev[i - 5][1].record()
            torch.cuda.synchronize()
            t = float(np.median([a.elapsed_time(b) * 1e3 for a, b in ev]))
            per = t / (N / W)
            xs.append(L)
            ys.append(per)
            print(f"batch {batch:5d} {label} L {L:5d}: {t:8.1f} us, {N / W:.0f} items per worker -> {per:6.2f} us per item", flush=True)
        b, a = np.polyfit(xs, ys, 1)
        print(f"   fit: {a:.2f} us + {b * 1000:.2f} ns/token  -> fixed cost = {a / b:.0f} tokens")
lib.vmi_debug_set_queue_flags(0)
