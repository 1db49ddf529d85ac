"""The default entry (on a full chip: the balanced kernel) against plain several-waves-per-head kernels — which leave the
balancing to the hardware dispatcher by running several times the resident waves — over the length distributions of
scripts/heavy_tail_probe.py.  Product library; median of 60 HIP-event pairs, launches back to back.
`python scripts/default_vs_waves_probe.py [--kv fp8] [B ...]`"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

args = sys.argv[1:]
f8 = "--kv" in args and args[args.index("--kv") + 1] == "fp8"
seq_len = int(args[args.index("--seq-len") + 1]) if "--seq-len" in args else 0
base = args[args.index("--cfg") + 1] if "--cfg" in args else "cfg3"
skip = {args.index(o) + 1 for o in ("--kv", "--seq-len", "--cfg") if o in args}
Bs = [int(a) for i, a in enumerate(args) if a.isdigit() and i not in skip] or [256, 512]
dev = torch.device("cuda:0")
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
Dh = CONFIGS[base].head_size
cands = ([f"fp8_d{Dh}_bs16_h1_w4_u2_nt1", f"fp8_d{Dh}_bs16_h1_w8_u2_nt1"] if f8 else
         [f"d{Dh}_h1_w4_u1_nt1", f"d{Dh}_h1_w8_u1_nt1", f"d{Dh}_h1_w2_u1_nt1"])
for B in Bs:
    cfg = dataclasses.replace(CONFIGS[base], name=f"b{B}", batch=B, seq_len=seq_len or CONFIGS[base].seq_len)
    cfg = dataclasses.replace(cfg, num_blocks=2 * B * cfg.blocks_per_seq)
    wl = make_workload(cfg, dev, seed=0)
    kc, vc, kvd = wl.key_cache, wl.value_cache, "auto"
    if f8:
        g8 = torch.Generator(device=dev).manual_seed(9)
        D = cfg.head_size
        kc = torch.randint(0, 64, (cfg.num_blocks, cfg.kv_heads, D // 16, 16, 16), dtype=torch.uint8, device=dev, generator=g8)
        vc = torch.randint(0, 64, (cfg.num_blocks, cfg.kv_heads, D, 16), dtype=torch.uint8, device=dev, generator=g8)
        kvd = "fp8"
    L = cfg.seq_len
    g = torch.Generator().manual_seed(B)
    u = torch.rand(B, generator=g)
    kinds = {"equal": torch.full((B,), L), "U{1..L}": (u * L).long() + 1, "U[1/4..1]": (L / 4 + u * 0.75 * L).long(),
             "triangular": (torch.minimum(u, torch.rand(B, generator=g)) * L).long() + 1,
             "3/4 full, rest 1/16": torch.where(u < 0.75, L, L // 16), "half full, half 1/16": torch.where(u < 0.5, L, L // 16),
             "1/8 full, rest 1/8": torch.where(u < 0.125, L, L // 8), "1/16 full, rest 1/16": torch.where(u < 1 / 16, L, L // 16),
             "exponential mean 1/4": torch.clamp((torch.empty(B).exponential_(1.0, generator=g) * L / 4).long() + 1, max=L),
             "exponential mean 1/8": torch.clamp((torch.empty(B).exponential_(1.0, generator=g) * L / 8).long() + 1, max=L),
             "lognormal(5, 1)": torch.clamp(torch.empty(B).log_normal_(5.0, 1.0, generator=g).long() + 1, max=L),
             "one full, rest 1/16": torch.where(torch.arange(B) < 1, L, L // 16)}
    out = torch.empty((B, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    for kind, lens in kinds.items():
        wl.seq_lens = lens.clamp(1, L).to(torch.int32).to(dev)
        res = {}
        for rep in range(2):
            for v in ["auto"] + cands:
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
                for i in range(75):
                    if i >= 15:
                        ev[i - 15][0].record()
                    ops.paged_attention_v1(out, wl.query, kc, vc, cfg.kv_heads, wl.scale, wl.tables[i % len(wl.tables)], wl.seq_lens,
                                           cfg.block_size, L, None, kvd, 1.0, 0, 0, 1, 1, 0, _variant=0 if v == "auto" else names[v])
                    if i >= 15:
                        ev[i - 15][1].record()
                torch.cuda.synchronize()
                ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
                res.setdefault(v, []).append(ts[len(ts) // 2])
                if v == "auto":
                    label = ops.last_launch_label()
        print(f"batch {B:4d} {kind:22s} default [{label:18s}] {min(res['auto']):7.1f}   " +
              "   ".join(f"{c.split('_h1_')[1].split('_')[0]} {min(res[c]):7.1f}" for c in cands), flush=True)
