"""Soak of the harness's layer library (not collected by pytest): N random (M, N, K, LayerNorm, epilogue, bias) cases of
gpt2_layer.linear against the module chain with its rounding points, plain and packed weights; random qkv-cache cases against
linear + reshape_and_cache; random argmax / top-k rows against torch.  `python tests/soak/soak_layer.py [cases] [seed]`"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpt2_layer import _chain, _close  # noqa: E402

from vllmini_amd import cache_ops, gpt2_layer as gl  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
rng = np.random.default_rng(seed)
g = torch.Generator(device="cpu").manual_seed(seed)
kernels = {}
for case in range(cases):
    M = int(rng.integers(1, 601))
    N = 16 * int(rng.integers(1, 257))
    ln = bool(rng.integers(0, 2))
    K = 32 * int(rng.integers(1, (2048 if ln else 4608) // 32 + 1))
    epi = ["bias", "gelu", "res"][int(rng.integers(0, 3))]
    x = (torch.randn(M, K, generator=g) * float(rng.uniform(0.3, 3)) + float(rng.uniform(-1, 1))).half().to(dev)
    w = (torch.randn(N, K, generator=g) * (0.6 / K ** 0.5)).half().to(dev)
    b = (torch.randn(N, generator=g) * 0.1).half().to(dev) if rng.integers(0, 4) else None
    lnp = ((1 + 0.2 * torch.randn(K, generator=g)).half().to(dev), (0.1 * torch.randn(K, generator=g)).half().to(dev), 1e-5) if ln else None
    res = torch.randn(M, N, generator=g).half().to(dev) if epi == "res" else None
    name = gl.kernel_name(M, N, K, ln)
    kernels[name] = kernels.get(name, 0) + 1
    got = gl.linear(x, w, b, ln=lnp, gelu=epi == "gelu", residual=res)
    _close(got, _chain(x, w, b, lnp, epi == "gelu", res), (case, M, N, K, ln, epi, name))
    assert torch.equal(gl.linear(x, gl.pack_weight(w), b, ln=lnp, gelu=epi == "gelu", residual=res), got), (case, M, N, K)
    if case % 100 == 99:
        print(f"linear: {case + 1} cases ok", flush=True)
print("kernels:", kernels)
for case in range(max(cases // 10, 10)):
    H, D, BS = [(12, 64, 16), (4, 128, 8), (2, 64, 32), (7, 80, 16)][int(rng.integers(0, 4))]
    M = int(rng.integers(1, 300))
    E = H * D
    if E % 32:
        continue
    NB = (M + BS - 1) // BS * 2 + 2
    x = torch.randn(M, E, generator=g).half().to(dev)
    w = (torch.randn(3 * E, E, generator=g) * (0.6 / E ** 0.5)).half().to(dev)
    b = (torch.randn(3 * E, generator=g) * 0.1).half().to(dev)
    kc0 = torch.randn(NB, H, D // 8, BS, 8, generator=g).half().to(dev)
    vc0 = torch.randn(NB, H, D, BS, generator=g).half().to(dev)
    slots = torch.randperm(NB * BS, generator=g)[:M].to(torch.int64)
    slots[torch.rand(M, generator=g) < 0.05] = -1
    slots = slots.to(dev)
    ref = gl.linear(x, w, b)
    kr, vr = kc0.clone(), vc0.clone()
    cache_ops.reshape_and_cache(ref[:, E:2 * E].view(M, H, D), ref[:, 2 * E:].view(M, H, D), kr, vr, slots, "auto", 1.0)
    kg, vg = kc0.clone(), vc0.clone()
    got = gl.linear_qkv_cache(x, w, b, kg, vg, slots, H)
    assert torch.equal(got, ref) and torch.equal(kg, kr) and torch.equal(vg, vr), (case, M, H, D, BS)
print("qkv cache cases ok")
for case in range(max(cases // 10, 10)):
    B, V = int(rng.integers(1, 300)), int(rng.integers(1, 65537))
    logits = (torch.randn(B, V, generator=g) * float(rng.uniform(0.1, 5))).half().to(dev)
    if rng.integers(0, 3) == 0:
        logits = (logits * 4).round() / 4          # many ties
    assert torch.equal(gl.argmax(logits), logits.argmax(-1)), (case, B, V)
    k = int(rng.integers(1, 65))
    u = torch.rand(B, generator=g).to(dev)
    got = gl.sample_top_k(logits, k, 1.0, uniform=u)
    kk = min(k, V)
    vals, idx = torch.topk(logits.float(), kk, dim=-1)
    picked = logits.gather(-1, got[:, None]).squeeze(-1).float()
    assert (picked >= vals[:, -1]).all(), (case, B, V, k)      # every draw is one of the k largest values
    zero = gl.sample_top_k(logits, k, 1.0, uniform=torch.zeros(B, device=dev))
    assert torch.equal(logits.gather(-1, zero[:, None]).squeeze(-1), logits.max(-1).values), (case, B, V, k)
print("argmax / top-k cases ok")
