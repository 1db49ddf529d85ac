"""Diagnostic: reshape_and_cache at prefill scale (T tokens with consecutive slots), µs and GB/s."""
import sys
import torch
from vllmini_amd import cache_ops

dev = torch.device("cuda:0")
for (T, H, D) in ((256, 12, 64), (1024, 12, 64), (4096, 12, 64), (4096, 32, 128), (16384, 32, 128)):
    BS = 16
    nb = T // BS + 8
    kc = torch.zeros((nb, H, D // 8, BS, 8), dtype=torch.float16, device=dev)
    vc = torch.zeros((nb, H, D, BS), dtype=torch.float16, device=dev)
    qkv = torch.randn((T, 3 * H * D), dtype=torch.float16, device=dev)
    k = qkv[:, H * D:2 * H * D].view(T, H, D)
    v = qkv[:, 2 * H * D:].view(T, H, D)
    perm = torch.randperm(nb, device=dev)[: T // BS]
    slots = (perm.view(-1, 1) * BS + torch.arange(BS, device=dev).view(1, -1)).reshape(-1).to(torch.int64)
    for _ in range(3):
        cache_ops.reshape_and_cache(k, v, kc, vc, slots, "auto", 1.0)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); cache_ops.reshape_and_cache(k, v, kc, vc, slots, "auto", 1.0); b.record()
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2] * 1e3
    nbytes = 2 * T * H * D * 2 * 2
    print(f"T={T} H={H} D={D}: {us:.1f} us, {nbytes / us / 1e3:.0f} GB/s (read+write {nbytes / 1e6:.1f} MB)", flush=True)
