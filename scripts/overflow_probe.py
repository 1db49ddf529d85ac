"""Batches with a little more (sequence, head) items than the balanced kernel's grid has waves: the default entry (which
launches the OVF twin for nwaves < items <= nwaves + workgroups: one item per wave as in mode S, the remainder by 4-wave
teams) against the default instantiation forced by name (solo workers, ceil(items / workers) rounds), equal and ragged
lengths, fp16 and fp8 pages.
`python scripts/overflow_probe.py [--kv auto|fp8]` -> stdout + gpurun_out/overflow_probe[_fp8].json"""
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
ap.add_argument("--kv", default="auto", choices=["auto", "fp8"])
ap.add_argument("--iters", type=int, default=100)
args = ap.parse_args()
dev = torch.device("cuda:0")
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
base = CONFIGS["cfg3"]
plain = "fp8_q_d64_s2q4m" if args.kv == "fp8" else "q_d64_s1q2"
res = []
print(f"{'batch':>6} {'lengths':>10} | default entry us (kernel) | forced {plain} us | ideal at 6.7 TB/s")
for B in (256, 257, 264, 272, 288, 304, 320, 321, 352, 512):
    cfg = dataclasses.replace(base, name=f"b{B}", batch=B, num_blocks=2 * B * base.blocks_per_seq)
    for ragged in (False, True, "exp", "tail"):
        wl = make_workload(cfg, dev, seed=B, table_sets=2, ragged=(ragged is True))   # (exp / tail: full tables, lengths overwritten)
        if ragged in ("exp", "tail"):
            gl = torch.Generator().manual_seed(B)
            lens = (torch.clamp((torch.empty(B).exponential_(1.0, generator=gl) * cfg.seq_len / 4).long() + 1, max=cfg.seq_len)
                    if ragged == "exp" else torch.where(torch.rand(B, generator=gl) < 0.125, cfg.seq_len, cfg.seq_len // 8))
            wl.seq_lens = lens.to(torch.int32).to(dev)
        if args.kv == "fp8":
            g8 = torch.Generator(device=dev).manual_seed(9)
            D = cfg.head_size
            wl.key_cache = torch.randint(0, 64, (cfg.num_blocks, cfg.kv_heads, D // 16, 16, 16), dtype=torch.uint8, device=dev, generator=g8)
            wl.value_cache = torch.randint(0, 64, (cfg.num_blocks, cfg.kv_heads, D, 16), dtype=torch.uint8, device=dev, generator=g8)
        out = torch.empty((B, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)

        def run(t, variant):
            ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[t], wl.seq_lens,
                                   cfg.block_size, cfg.seq_len, None, args.kv, 1.0, 0, 0, 1, 1, 0, _variant=variant)

        def timeit(variant):
            for i in range(25):
                run(i % 2, variant)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
            for i, (a, b) in enumerate(ev):
                a.record()
                run(i % 2, variant)
                b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
            return ts[len(ts) // 2]

        auto = timeit(0)
        label = ops.last_launch_label()
        forced = timeit(names[plain])
        ref = out.clone()
        run(0, names[plain]); torch.cuda.synchronize(); ref = out.clone()
        run(0, 0); torch.cuda.synchronize()
        dmax = float((out.float() - ref.float()).abs().max())
        kv_bytes = int(wl.seq_lens.sum().item()) * cfg.kv_heads * cfg.head_size * 2 * (1 if args.kv == "fp8" else 2)
        ideal = kv_bytes / 6.7e12 * 1e6
        res.append({"batch": B, "ragged": ragged, "default_us": auto, "default_kernel": label, "forced_plain_us": forced,
                    "ideal_us": ideal, "max_abs_diff": dmax})
        print(f"{B:6d} {({False: 'equal', True: 'U{1..L}'}.get(ragged, ragged)):>10} | {auto:7.1f} ({label.split(' ')[0]}) | {forced:7.1f} | {ideal:6.1f} | max|d| {dmax:.1e}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/overflow_probe{'_fp8' if args.kv == 'fp8' else ''}.json", "w"), indent=1)
