"""Batches of two and more items per resident wave over fp16 pages (head size 64): the balanced kernel with its default two
solo workers per workgroup against four (knob bits 2-4 = 4, everything else automatic — the teams-or-solo rule included).
Diagnostic library; median of 60 HIP-event pairs, launches back to back.  `python scripts/wq4_large_batch_probe.py [B ...]`"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import _lib, ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

lib = _lib.use_diag().__enter__()
dev = torch.device("cuda:0")
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
for B in [int(a) for a in sys.argv[1:]] or [384, 512, 768, 1024]:
    cfg = dataclasses.replace(CONFIGS["cfg3"], name=f"b{B}", batch=B, num_blocks=2 * B * 64)
    wl = make_workload(cfg, dev, seed=0)
    L = cfg.seq_len
    g = torch.Generator().manual_seed(B)
    u = torch.rand(B, generator=g)
    kinds = {"equal": torch.full((B,), L), "U{1..L}": (u * L).long() + 1, "U[1/4..1]": (L / 4 + u * 0.75 * L).long(),
             "3/4 full, rest 1/16": torch.where(u < 0.75, L, L // 16), "2/3 full, rest 1/16": torch.where(u < 0.667, L, L // 16),
             "half full, half 1/16": torch.where(u < 0.5, L, L // 16),
             "exponential mean 1/4": torch.clamp((torch.empty(B).exponential_(1.0, generator=g) * L / 4).long() + 1, max=L),
             "1/8 full, rest 1/8": torch.where(u < 0.125, L, L // 8)}
    out = torch.empty((B, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    for kind, lens in kinds.items():
        wl.seq_lens = lens.clamp(1, L).to(torch.int32).to(dev)
        res = {}
        for rep in range(2):
            for flags in (0, 4 << 2):
                lib.vmi_debug_set_queue_flags(flags)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
                for i in range(75):
                    if i >= 15:
                        ev[i - 15][0].record()
                    ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[i % len(wl.tables)],
                                           wl.seq_lens, cfg.block_size, L, None, "auto", 1.0, 0, 0, 1, 1, 0, _variant=names["q_d64_s1q2"])
                    if i >= 15:
                        ev[i - 15][1].record()
                torch.cuda.synchronize()
                ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
                res.setdefault(flags, []).append(ts[len(ts) // 2])
        lib.vmi_debug_set_queue_flags(0)
        a, b = min(res[0]), min(res[16])
        print(f"batch {B:5d}  {kind:22s} two workers {a:7.1f} us   four workers {b:7.1f} us   {b / a:.3f}", flush=True)
