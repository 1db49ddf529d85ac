"""Heavy-tailed batches: the default paged_attention_v1 entry against paged_attention_v2 (512-token partitions + reduce,
caller-owned scratch), cfg3 shapes.  `python scripts/v1_vs_v2_heavy_tail_probe.py [--seq-len 1024]`."""
import argparse
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="cfg3")
ap.add_argument("--iters", type=int, default=200)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CONFIGS[args.cfg]
g = torch.Generator().manual_seed(1)
Lm = cfg.seq_len
cases = {
    "equal lengths": None,
    "U{1..max}": torch.randint(1, Lm + 1, (cfg.batch,), generator=g),
    "1/8 full, rest 1/8": torch.where(torch.rand(cfg.batch, generator=g) < 0.125, Lm, Lm // 8),
    "exponential mean 1/4": torch.clamp((torch.empty(cfg.batch).exponential_(1.0, generator=g) * Lm / 4).long() + 1, max=Lm),
    "exponential mean 1/8": torch.clamp((torch.empty(cfg.batch).exponential_(1.0, generator=g) * Lm / 8).long() + 1, max=Lm),
    "lognormal(5, 1)": torch.clamp(torch.empty(cfg.batch).log_normal_(5.0, 1.0, generator=g).long() + 1, max=Lm),
    "one full, rest 1/16": torch.where(torch.arange(cfg.batch) < 1, Lm, Lm // 16),
}
wl = make_workload(cfg, dev, seed=0, ragged=False)
P = (cfg.seq_len + 511) // 512
es = torch.empty((cfg.batch, cfg.num_heads, P), dtype=torch.float32, device=dev)
ml = torch.empty_like(es)
tmp = torch.empty((cfg.batch, cfg.num_heads, P, cfg.head_size), dtype=torch.float16, device=dev)
out1 = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
out2 = torch.empty_like(out1)


def v1(t):
    ops.paged_attention_v1(out1, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[t], wl.seq_lens,
                           cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0)


def v2(t):
    ops.paged_attention_v2(out2, es, ml, tmp, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[t],
                           wl.seq_lens, cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0)


def timeit(fn):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
    for i in range(args.iters + 10):
        if i >= 10:
            ev[i - 10][0].record()
        fn(i % len(wl.tables))
        if i >= 10:
            ev[i - 10][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return sum(ts) / len(ts)


for name, lens in cases.items():
    if lens is not None:
        wl.seq_lens = lens.to(torch.int32).to(dev)
    t1, t2 = timeit(v1), timeit(v2)
    d = (out1.float() - out2.float()).abs().max().item()
    bytes_us = int(wl.seq_lens.sum().item()) * cfg.kv_heads * cfg.head_size * 4 / 6.5e12 * 1e6
    print(f"{name:24s} bytes at 6.5 TB/s {bytes_us:6.1f} us | v1 default {t1:7.1f} us | v2 {t2:7.1f} us | max|v1-v2| {d:.2e}", flush=True)
