#!/usr/bin/env python3
"""The prefill shape of paged_attention_v1 (every prompt position a "sequence" of length t + 1 over its prompt's table): which
work decomposition serves it best?  64 prompts U{4..512}, 12 x 64 heads; us per launch by HIP events."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import ops

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
H, D, bs = 12, 64, 16
for n in (8, 64):
    lens = rng.integers(4, 513, n)
    nblk = (lens + bs - 1) // bs
    NB = int(nblk.sum()) + 8
    g = torch.Generator(device=dev).manual_seed(1)
    kc = torch.empty((NB, H, D // 8, bs, 8), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
    vc = torch.empty((NB, H, D, bs), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
    mbe = int(nblk.max())
    tab = np.full((n, mbe), -1, dtype=np.int32)
    perm, at = rng.permutation(NB).astype(np.int32), 0
    for s in range(n):
        tab[s, : nblk[s]] = perm[at: at + nblk[s]]
        at += nblk[s]
    seq_of_tok = np.repeat(np.arange(n), lens)
    pos = np.concatenate([np.arange(T) for T in lens])
    T = len(pos)
    tables = torch.from_numpy(tab[seq_of_tok]).to(dev)
    lt = torch.from_numpy((pos + 1).astype(np.int32)).to(dev)
    qkv = torch.empty((T, 3 * H * D), dtype=torch.float16, device=dev).normal_(0, 1, generator=g)
    q = qkv[:, : H * D].view(T, H, D)
    out = torch.empty((T, H, D), dtype=torch.float16, device=dev)
    names = ops.variant_names()
    ref = None
    cands = [0] + [names.index(x) + 1 for x in ("q_d64_s1q2", "d64_h4_w1_u1_nt1", "d64_h4_w1_u2_nt1", "d64_h4_w1_u4_nt0", "d64_h4_w1_u1a4_nt1",
                                                "d64_h1_w2_u1_nt1", "d64_h1_w4_u1_nt1", "d64_h1_w4_u2_nt0", "d64_h1_w8_u1_nt0")]
    print(f"n={n} rows={T} token-reads={int((pos + 1).sum())} ({int((pos + 1).sum()) * H * D * 4 / 1e9:.2f} GB if nothing is reused)")
    for vid in cands:
        def run():
            ops.paged_attention_v1(out, q, kc, vc, H, D ** -0.5, tables, lt, bs, mbe * bs, None, "auto", 1.0, 0, 0, 1, 1, 0, _variant=vid)
        try:
            for _ in range(3):
                run()
        except RuntimeError as e:
            print("   ", names[vid - 1], "refused:", str(e)[:80])
            continue
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); run(); b.record()
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) for a, b in ev)[5] * 1e3
        name = names[ops.last_variant() - 1]
        if ref is None:
            ref = out.clone()
        d = float((out.float() - ref.float()).abs().max())
        print(f"    {('default -> ' if vid == 0 else '') + name:34s} {us:9.1f} us   max|d| vs default {d:.1e}")
