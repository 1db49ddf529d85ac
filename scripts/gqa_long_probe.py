"""Diagnostic: long-context, small-batch grouped-query decode — paged_attention_v1 vs paged_attention_v2."""
import torch
from vllmini_amd import ops

dev = torch.device("cuda:0")
B, H, Hkv, D, L, BS = 4, 32, 8, 128, 16384, 16
nb = L // BS
NB = 2 * B * nb
kc = torch.empty((NB, Hkv, D // 8, BS, 8), dtype=torch.float16, device=dev).uniform_(-1, 1)
vc = torch.empty((NB, Hkv, D, BS), dtype=torch.float16, device=dev).uniform_(-1, 1)
q = torch.randn((B, H, D), dtype=torch.float16, device=dev)
out = torch.empty_like(q)
tabs = [(torch.randperm(B * nb, device=dev).to(torch.int32) + t * B * nb).view(B, nb) for t in range(2)]
lens = torch.full((B,), L, dtype=torch.int32, device=dev)
P = L // 512
es = torch.empty((B, H, P), dtype=torch.float32, device=dev)
ml = torch.empty_like(es)
tmp = torch.empty((B, H, P, D), dtype=torch.float16, device=dev)
uniq = 2 * B * Hkv * L * D * 2


def timed(fn):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for i in range(24):
        k = i - 4
        if k >= 0:
            ev[k][0].record()
        fn(i % 2)
        if k >= 0:
            ev[k][1].record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[10] * 1e3


us1 = timed(lambda t: ops.paged_attention_v1(out, q, kc, vc, Hkv, D ** -0.5, tabs[t], lens, BS, L, None, "auto", 1.0))
o1 = out.clone()
us2 = timed(lambda t: ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, Hkv, D ** -0.5, tabs[t], lens, BS, L, None, "auto", 1.0))
print(f"B{B} H{H}/Hkv{Hkv} D{D} L{L}: unique KV {uniq / 1e6:.0f} MB | v1 {us1:.1f} us ({uniq / us1 / 1e6:.2f} TB/s) | "
      f"v2 {us2:.1f} us ({uniq / us2 / 1e6:.2f} TB/s) | max |v1-v2| {float((o1.float() - out.float()).abs().max()):.2e}")
