#!/usr/bin/env python3
"""Which variants can the library's heuristics RETURN?  Sweeps every pick entry over a wide grid of launches (host only, no GPU)
and prints the names reached / not reached.  The product menu is cut to the reached set + neighbours (round 6)."""
import itertools
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import ops  # noqa: E402

names = ops.variant_names()
hit = {}
Bs = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 56, 64, 96, 128, 160, 192, 200, 224, 240, 256, 320, 384, 512, 768, 1024, 2048, 4096]
Ls = [1, 16, 64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 3400, 4096, 8192, 16384, 32768, 65536]
Hs = [1, 2, 4, 5, 8, 12, 16, 20, 24, 25, 28, 32, 40, 48, 64, 96]
for D, bs in itertools.product((64, 80, 96, 112, 128, 192, 256), (8, 16, 32)):
    for B, L, H in itertools.product(Bs, Ls, Hs):
        for fp8 in (False, True):
            if fp8 and bs == 8:
                continue
            for mean in (0, max(L // 2, 1), max(L // 8, 1)):
                v = ops.pick_variant(B, H, D, L, bs, mean_seq_len=mean, fp8=fp8)
                if v:
                    hit.setdefault(names[v - 1], (B, H, D, L, bs, mean, fp8))
            if not fp8:
                v = ops.pick_variant(B, H, D, L, bs, workspace=True)
                if v:
                    hit.setdefault(names[v - 1], (B, H, D, L, bs, "ws"))
            for q in (2, 3, 4, 7, 8, 16):
                if H % q == 0:
                    v = ops.pick_variant(B, H, D, L, bs, fp8=fp8, num_kv_heads=H // q)
                    if v:
                        hit.setdefault(names[v - 1], (B, H, D, L, bs, f"gqa{q}", fp8))
                    if not fp8:
                        v = ops.pick_variant(B, H, D, L, bs, num_kv_heads=H // q, workspace=True)
                        if v:
                            hit.setdefault(names[v - 1], (B, H, D, L, bs, f"gqa{q} ws"))
print(len(names), "variants,", len(hit), "reachable")
print("REACHED", json.dumps(sorted(hit)))
print("NOT", json.dumps([n for n in names if n not in hit]))
