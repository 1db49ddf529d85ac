"""Does WHERE a layer's pages lie in the pool change the headline kernel's time?  The reference keeps ONE pool for all layers
(vllmini/kv_cache.py:13-14), so in a decode step the 12 layers' attention calls walk 12 disjoint page sets of one big tensor,
while `bench.py`'s headline call reads the same 16 384 pages of a 32 768-block pool over and over.  `e2e_step` reports the
layer calls 5 - 20 % slower than the headline; this probe separates the candidates: the pool's size (the span of addresses a
call's pages are scattered over = TLB reach), the cycling over 12 page sets, and the order of the pages inside a table.

Cases, all BASELINE configs[2] shaped calls (batch 256 x 1024 tokens, 12 heads x 64, block 16 -> 16 384 pages = 806 MB per call):
  pool        blocks in the pool (32 768 = the headline's; 196 672 = a 12-layer pool with every page in use)
  placement   "scattered": a call's pages are a random sample of the WHOLE pool (a shuffled free list);
              "arena": the pages of layer l lie in [l * 16 384, (l + 1) * 16 384), shuffled inside it;
              "sequential": arena, and every sequence's pages ascending and adjacent
  sets        1 = the same table every call; 12 = twelve disjoint tables (layers) in turn
`python scripts/pool_locality_probe.py [out.json]`"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, D, L, BS = 256, 12, 64, 1024, 16
PER = L // BS
NPG = B * PER            # pages per call
LAYERS = 12
scale = D ** -0.5
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((B, H, D), dtype=torch.float16, device=dev, generator=g)
out = torch.empty_like(q)
lens = torch.full((B,), L, dtype=torch.int32, device=dev)
res = []


def tables_for(pool_blocks, placement, sets, rng):
    if placement == "scattered":
        perm = rng.permutation(pool_blocks)[:sets * NPG]
        return [perm[i * NPG:(i + 1) * NPG].reshape(B, PER) for i in range(sets)]
    tabs = []
    for i in range(sets):
        base = i * NPG
        ids = np.arange(base, base + NPG)
        if placement == "arena":
            ids = rng.permutation(ids)
        tabs.append(ids.reshape(B, PER))
    return tabs


def run(pool_blocks, placement, sets):
    kc = torch.empty((pool_blocks, H, D // 8, BS, 8), dtype=torch.float16, device=dev)
    vc = torch.empty((pool_blocks, H, D, BS), dtype=torch.float16, device=dev)
    kc.uniform_(-1, 1, generator=g)
    vc.uniform_(-1, 1, generator=g)
    rng = np.random.default_rng(1)
    tabs = [torch.from_numpy(t.astype(np.int32)).to(dev) for t in tables_for(pool_blocks, placement, sets, rng)]

    def call(i):
        ops.paged_attention_v1(out, q, kc, vc, H, scale, tabs[i % sets], lens, BS, L, None, "auto", 1.0, 0, 0, 1, 1, 0)

    for i in range(2 * sets + 2):
        call(i)
    torch.cuda.synchronize()
    label = ops.last_launch_label() if hasattr(ops, "last_launch_label") else ""
    reps = 48
    samples = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
            call(i)
        b.record()
        torch.cuda.synchronize()
        samples.append(a.elapsed_time(b) / reps * 1e3)
    rec = {"pool_blocks": pool_blocks, "pool_GB": round(2 * pool_blocks * H * D * BS * 2 / 1e9, 2), "placement": placement,
           "sets": sets, "us_per_call_median": round(float(np.median(samples)), 2), "us_per_call_min": round(min(samples), 2),
           "kernel": label}
    res.append(rec)
    print(json.dumps(rec), flush=True)
    del kc, vc, tabs
    torch.cuda.empty_cache()


BIG = LAYERS * NPG + 64
for pool_blocks, placement, sets in ((32768, "scattered", 1), (32768, "arena", 1), (32768, "sequential", 1),
                                     (BIG, "scattered", 1), (BIG, "arena", 1),
                                     (BIG, "scattered", 12), (BIG, "arena", 12), (BIG, "sequential", 12),
                                     (4 * BIG, "scattered", 12), (4 * BIG, "arena", 12),
                                     (32768, "scattered", 1)):
    run(pool_blocks, placement, sets)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
