#!/usr/bin/env python3
"""Preemption move of ONE sequence (GPT-2 small, 12 layers, ~1000 tokens: 12 x 63 blocks of each cache = 37 MB): the reference's
form — one async memcpy per block (cache_ops.swap_blocks, cache_kernels.cu:56-62), K then V — against ONE launch that moves both
caches, the pinned host pool addressed by the GPU (cache_ops.swap_blocks_batched).  Host time to issue, and time to completion."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import cache_ops

dev = torch.device("cuda:0")
NB, H, D, BS = 8192, 12, 64, 16
kc = torch.randn((NB, H, D // 8, BS, 8), device=dev).half()
vc = torch.randn((NB, H, D, BS), device=dev).half()
for nblocks in (12 * 8, 12 * 32, 12 * 63):
    hk = torch.empty((nblocks,) + tuple(kc.shape[1:]), dtype=torch.float16).pin_memory()
    hv = torch.empty((nblocks,) + tuple(vc.shape[1:]), dtype=torch.float16).pin_memory()
    rng = np.random.default_rng(nblocks)
    src = rng.permutation(NB)[:nblocks]
    out_map = torch.from_numpy(np.stack([src, np.arange(nblocks)], 1).astype(np.int64))
    in_map = torch.from_numpy(np.stack([np.arange(nblocks), rng.permutation(NB)[:nblocks]], 1).astype(np.int64))
    mb = 2 * nblocks * kc[0].numel() * 2 / 1e6
    res = {}
    for name, fn_out, fn_in in (
            ("per block (swap_blocks)", lambda: (cache_ops.swap_blocks(kc, hk, out_map), cache_ops.swap_blocks(vc, hv, out_map)),
             lambda: (cache_ops.swap_blocks(hk, kc, in_map), cache_ops.swap_blocks(hv, vc, in_map))),
            ("one launch (swap_blocks_batched)", lambda: cache_ops.swap_blocks_batched(kc, vc, hk, hv, out_map),
             lambda: cache_ops.swap_blocks_batched(hk, hv, kc, vc, in_map))):
        for direction, fn in (("out", fn_out), ("in", fn_in)):
            ts_issue, ts_done = [], []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
                ts_issue.append(t1 - t0); ts_done.append(t2 - t0)
            res[(name, direction)] = (sorted(ts_issue)[2] * 1e3, sorted(ts_done)[2] * 1e3)
    print(f"{nblocks} blocks per cache, {mb:.1f} MB:")
    for (name, direction), (ti, td) in res.items():
        print(f"    {name:34s} {direction:3s}  host issue {ti:7.3f} ms   done {td:7.3f} ms   {mb / td:6.1f} GB/s")
