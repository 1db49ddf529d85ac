"""Diagnostic: the opt-in "_pvm" grouped-query kernels (P.V on the matrix cores) — distance from the kernel model and
from an fp64 attention, next to the default gq kernels.  PYTHONPATH=. python tests/soak/pvm_probe.py"""
import numpy as np
import torch
import oracle
from vllmini_amd import ops

dev = torch.device("cuda:0")
names = ops.variant_names()
rng = np.random.default_rng(5)
for H, hkv, D in ((16, 4, 128), (32, 4, 128), (16, 4, 64)):
    lens = np.array([1, 16, 17, 100, 333, 1024, 47, 2, 600, 31], np.int32)
    S, bs = len(lens), 16
    nb = 64
    NB = S * nb + 3
    kc = rng.uniform(-1, 1, (NB, hkv, D // 8, bs, 8)).astype(np.float16)
    vc = rng.uniform(-1, 1, (NB, hkv, D, bs)).astype(np.float16)
    q = rng.standard_normal((S, H, D)).astype(np.float16)
    tables = rng.permutation(NB)[:S * nb].astype(np.int32).reshape(S, nb)
    scale = D ** -0.5
    ref = oracle.paged_attention_v1(q, kc, vc, hkv, scale, tables, lens, bs, threads=8)
    tq, tk, tv = (torch.from_numpy(a).to(dev) for a in (q, kc, vc))
    tt, tl = torch.from_numpy(tables).to(dev), torch.from_numpy(lens).to(dev)
    # fp64 attention
    kk = kc.astype(np.float64).transpose(0, 1, 3, 2, 4).reshape(NB, hkv, bs, D)      # [NB,h,tok,D]
    vv = vc.astype(np.float64).transpose(0, 1, 3, 2)                                    # [NB,h,tok,D]
    exact = np.zeros((S, H, D))
    for s in range(S):
        L = lens[s]
        K = kk[tables[s]].transpose(1, 0, 2, 3).reshape(hkv, -1, D)[:, :L]
        V = vv[tables[s]].transpose(1, 0, 2, 3).reshape(hkv, -1, D)[:, :L]
        for h in range(H):
            lg = (K[h // (H // hkv)] @ q[s, h].astype(np.float64)) * scale
            pr = np.exp(lg - lg.max()); pr /= pr.sum()
            exact[s, h] = pr @ V[h // (H // hkv)]
    for vid, name in enumerate(names, start=1):
        if not name.startswith(f"d{D}_gq") or (H // hkv) % int(name.split("_gq")[1].split("_")[0]):
            continue
        if "_pvm" not in name and "_u2_" not in name and "_w4_" not in name:
            continue
        out = torch.empty_like(tq)
        try:
            ops.paged_attention_v1(out, tq, tk, tv, hkv, scale, tt, tl, bs, 1024, None, "auto", 1.0, _variant=vid)
        except RuntimeError as e:
            continue
        got = out.cpu().numpy().astype(np.float64)
        print(f"H{H}/{hkv} D{D} {name:34s} max|hip-model| {np.abs(got - ref).max():.2e}   "
              f"max|hip-fp64| {np.abs(got - exact).max():.2e}   (model-fp64 {np.abs(ref - exact).max():.2e})", flush=True)
