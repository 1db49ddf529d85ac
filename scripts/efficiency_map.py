"""Achieved HBM bandwidth of the DEFAULT paged_attention_v1 entry over a grid of batch sizes x context lengths (equal
lengths, 12 heads x 64, fp16 pages, random page tables, launches back to back, median of 40 HIP-event pairs): a map for
finding steps and dips of the work-decomposition heuristic.  `python scripts/efficiency_map.py [--heads 12] [--head-size 64]
[--kv auto|fp8] [--ragged]` -> stdout + gpurun_out/efficiency_map_*.json"""
import argparse
import dataclasses
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--heads", type=int, default=12)
ap.add_argument("--head-size", type=int, default=64)
ap.add_argument("--kv", default="auto", choices=["auto", "fp8"])
ap.add_argument("--ragged", action="store_true")
ap.add_argument("--batches", default="16,32,48,64,96,128,160,192,204,208,224,256,257,288,320,321,352,384,448,512,640,768,1024")
ap.add_argument("--lens", default="128,256,512,1024,2048,4096")
args = ap.parse_args()
dev = torch.device("cuda:0")
Bs = [int(x) for x in args.batches.split(",")]
Ls = [int(x) for x in args.lens.split(",")]
H, D = args.heads, args.head_size
res = []
print(f"H{H} D{D} {args.kv} {'ragged U{1..L}' if args.ragged else 'equal'}: TB/s of the bytes that exist (kernel us) [kernel]")
for L in Ls:
    row = []
    for B in Bs:
        if B * H * L * D * 4 > 12e9:
            row.append("      -      ")
            continue
        cfg = dataclasses.replace(CONFIGS["cfg3"], name=f"b{B}_l{L}", batch=B, num_heads=H, head_size=D, seq_len=L,
                                  num_blocks=2 * B * (-(-L // 16)) + 8)
        wl = make_workload(cfg, dev, seed=B + L, table_sets=2, ragged=args.ragged)
        if args.kv == "fp8":
            g8 = torch.Generator(device=dev).manual_seed(9)
            wl.key_cache = torch.randint(0, 64, (cfg.num_blocks, H, D // 16, 16, 16), dtype=torch.uint8, device=dev, generator=g8)
            wl.value_cache = torch.randint(0, 64, (cfg.num_blocks, H, D, 16), dtype=torch.uint8, device=dev, generator=g8)
        out = torch.empty((B, H, D), dtype=torch.float16, device=dev)

        def run(t):
            ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, H, wl.scale, wl.tables[t], wl.seq_lens, 16, L,
                                   None, args.kv, 1.0, 0, 0, 1, 1, 0)

        for i in range(20):
            run(i % 2)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for i, (a, b) in enumerate(ev):
            a.record()
            run(i % 2)
            b.record()
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[20]
        nbytes = int(wl.seq_lens.sum().item()) * H * D * 2 * (1 if args.kv == "fp8" else 2)
        tbps = nbytes / us / 1e6
        name = ops.variant_names()[ops.last_variant() - 1]
        res.append({"batch": B, "seq_len": L, "us": us, "TBps": tbps, "kernel": name})
        row.append(f"{tbps:4.2f} ({us:6.1f})")
        del wl, out
        torch.cuda.empty_cache()
    print(f"L{L:5d} | " + " ".join(row), flush=True)
print("batch   | " + " ".join(f"{b:^13d}" for b in Bs))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/efficiency_map_h{H}_d{D}_{args.kv}{'_ragged' if args.ragged else ''}.json", "w"), indent=1)
