#!/usr/bin/env python3
"""prefill_batch: the causal attention through paged_attention_v1 (paged_prefill) against the eager masked attention.
GPT-2 small, prompts U{4..512}; time per call and tokens/s, and the two paths' logits side by side."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd.gpt2_decode import GPT2Dims, GPT2PagedDecoder, random_state_dict
from vllmini_amd.kv_pool import PagedKVPool
from vllmini_amd import ops

dev = torch.device("cuda:0")
dims = GPT2Dims()
sd = random_state_dict(dims, dev)
rng = np.random.default_rng(0)
for n in (1, 8, 32, 64):
    prompts = [rng.integers(0, dims.vocab_size, int(rng.integers(4, 513))).tolist() for _ in range(n)]
    res = {}
    for paged in (False, True):
        pool = PagedKVPool(12 * 40 * n + 64, 12, 64, 16, 65, 12, device=dev, max_seqs=n)
        dec = GPT2PagedDecoder(dims, sd, pool, paged_prefill=paged)
        ts = []
        for rep in range(4):
            for s in range(n):
                pool.free(s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lg = dec.prefill_batch(list(range(n)), prompts)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[paged] = (min(ts[1:]), lg.float().cpu())
        label = ops.variant_names()[ops.last_variant() - 1] if paged else "-"
        del dec, pool
    tok = sum(len(p) for p in prompts)
    d = (res[True][1] - res[False][1]).abs().max().item()
    same = (res[True][1].argmax(-1) == res[False][1].argmax(-1)).float().mean().item()
    print(f"n={n:3d} tokens={tok:6d} eager {res[False][0]*1e3:8.2f} ms ({tok/res[False][0]/1e3:7.1f} k tok/s)  paged {res[True][0]*1e3:8.2f} ms "
          f"({tok/res[True][0]/1e3:7.1f} k tok/s)  kernel {label}  max|dlogit| {d:.3e} (|logit| max {res[False][1].abs().max():.2f}) argmax agree {same:.3f}", flush=True)
