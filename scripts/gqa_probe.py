"""Diagnostic: paged_attention_v1 with grouped-query attention (num_kv_heads < num_heads): µs and TB/s of the UNIQUE
K/V bytes — how much of the q-heads-per-KV-head re-reading is absorbed by L2.  PYTHONPATH=. python scripts/gqa_probe.py"""
import sys
import torch
from vllmini_amd import ops

dev = torch.device("cuda:0")
B, L, D, BS = 256, 1024, 128, 16
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for (H, Hkv) in ((32, 32), (32, 8), (32, 4), (64, 8), (12, 12)):
    nb = L // BS
    NB = 2 * B * nb
    kc = torch.empty((NB, Hkv, D // 8, BS, 8), dtype=torch.float16, device=dev).uniform_(-1, 1)
    vc = torch.empty((NB, Hkv, D, BS), dtype=torch.float16, device=dev).uniform_(-1, 1)
    q = torch.randn((B, H, D), dtype=torch.float16, device=dev)
    out = torch.empty_like(q)
    tabs = [(torch.randperm(B * nb, device=dev).to(torch.int32) + t * B * nb).view(B, nb) for t in range(2)]
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for i in range(35):
        k = i - 5
        if k >= 0:
            ev[k][0].record()
        ops.paged_attention_v1(out, q, kc, vc, Hkv, D ** -0.5, tabs[i % 2], lens, BS, L, None, "auto", 1.0, _variant=variant)
        if k >= 0:
            ev[k][1].record()
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) for a, b in ev)[15] * 1e3
    uniq = 2 * B * Hkv * L * D * 2
    naive = 2 * B * H * L * D * 2
    name = ops.variant_names()[(variant or ops.pick_variant(B, H, D, L, BS, num_kv_heads=Hkv)) - 1]
    print(f"H={H} Hkv={Hkv}: {us:.1f} us  unique {uniq / 1e6:.0f} MB -> {uniq / us / 1e6:.2f} TB/s of unique bytes "
          f"(per-q-head bytes {naive / 1e6:.0f} MB -> {naive / us / 1e6:.2f} TB/s)  [{name}]", flush=True)
    del kc, vc
    torch.cuda.empty_cache()
