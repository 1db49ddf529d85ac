"""Diagnostic: paged_attention_v1 with grouped-query attention (num_kv_heads < num_heads): µs and TB/s of the UNIQUE
K/V bytes.  PYTHONPATH=. python scripts/gqa_probe.py [H Hkv D [variant-name-substring]]"""
import sys
import torch
from vllmini_amd import ops

dev = torch.device("cuda:0")
B, L, BS = 256, 1024, 16
H, Hkv, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 8, 128)
sub = sys.argv[4] if len(sys.argv) > 4 else None
names = ops.variant_names()
qpk = H // Hkv
vids = [0] + [i + 1 for i, n in enumerate(names)
              if n.startswith(f"d{D}_") and "_bs" not in n and "LOADSONLY" not in n and (sub is None or sub in n) and
              ("_gq" not in n or qpk % int(n.split("_gq")[1].split("_")[0]) == 0)]
nb = L // BS
NB = 2 * B * nb
kc = torch.empty((NB, Hkv, D // 8, BS, 8), dtype=torch.float16, device=dev).uniform_(-1, 1)
vc = torch.empty((NB, Hkv, D, BS), dtype=torch.float16, device=dev).uniform_(-1, 1)
q = torch.randn((B, H, D), dtype=torch.float16, device=dev)
out = torch.empty_like(q)
tabs = [(torch.randperm(B * nb, device=dev).to(torch.int32) + t * B * nb).view(B, nb) for t in range(2)]
lens = torch.full((B,), L, dtype=torch.int32, device=dev)
uniq = 2 * B * Hkv * L * D * 2
res = []
for vid in vids:
    try:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for i in range(24):
            k = i - 4
            if k >= 0:
                ev[k][0].record()
            ops.paged_attention_v1(out, q, kc, vc, Hkv, D ** -0.5, tabs[i % 2], lens, BS, L, None, "auto", 1.0, _variant=vid)
            if k >= 0:
                ev[k][1].record()
        torch.cuda.synchronize()
    except RuntimeError:
        continue
    us = sorted(a.elapsed_time(b) for a, b in ev)[10] * 1e3
    name = names[vid - 1] if vid else "auto=" + names[ops.pick_variant(B, H, D, L, BS, num_kv_heads=Hkv) - 1]
    res.append((us, name))
for us, name in sorted(res)[:12]:
    print(f"H={H} Hkv={Hkv} D={D}: {us:7.1f} us  {uniq / us / 1e6:.2f} TB/s of unique KV bytes  [{name}]")
