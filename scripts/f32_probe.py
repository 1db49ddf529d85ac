"""Diagnostic: paged_attention_v1 over float32 tensors (x = 4) on the cfg3 / cfg4 shapes — µs and TB/s of the fp32 KV bytes.
PYTHONPATH=. python scripts/f32_probe.py"""
import torch
from vllmini_amd import ops
from vllmini_amd import _lib
_lib.use_extras().__enter__()   # bfloat16 / float32 / E5M2 / block-sparse live in libvmi_paged_attention_extras.so (build.py --extras)

dev = torch.device("cuda:0")
for name, B, H, D, L in (("cfg3", 256, 12, 64, 1024), ("cfg4", 128, 32, 128, 2048)):
    bs = 16
    nb = L // bs
    NB = 2 * B * nb
    kc = torch.empty((NB, H, D // 4, bs, 4), dtype=torch.float32, device=dev).uniform_(-1, 1)
    vc = torch.empty((NB, H, D, bs), dtype=torch.float32, device=dev).uniform_(-1, 1)
    q = torch.randn((B, H, D), dtype=torch.float32, device=dev)
    out = torch.empty_like(q)
    tabs = [(torch.randperm(B * nb, device=dev).to(torch.int32) + t * B * nb).view(B, nb) for t in range(2)]
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for i in range(24):
        if i >= 4:
            ev[i - 4][0].record()
        ops.paged_attention_v1(out, q, kc, vc, H, D ** -0.5, tabs[i % 2], lens, bs, L, None, "auto", 1.0)
        if i >= 4:
            ev[i - 4][1].record()
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) for a, b in ev)[10] * 1e3
    byts = 2 * B * H * L * D * 4
    print(f"{name} float32: {us:.1f} us, {byts / us / 1e6:.2f} TB/s of {byts / 1e6:.0f} MB")
