"""Host microseconds per operator call (GPU idle, tiny launches) and what they are made of.
`python scripts/host_overhead_probe.py` -> gpurun_out/host_overhead.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import _lib, cache_ops, ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
res = {}


def per_call(fn, n=2000, sync_every=200):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn()
        if (i + 1) % sync_every == 0:
            torch.cuda.synchronize()     # keep the launch queue from filling up: this measures the host, not the GPU
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name in ("b1", "cfg2", "cfg3"):
    cfg = CONFIGS[name]
    wl = make_workload(cfg, dev, seed=0, table_sets=1)
    out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    pa = lambda: ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale,  # noqa: E731
                                        wl.tables[0], wl.seq_lens, cfg.block_size, cfg.seq_len, None, "auto", 1.0,
                                        0, 0, 1, 1, 0)
    rc = lambda: cache_ops.reshape_and_cache(wl.key, wl.value, wl.key_cache, wl.value_cache, wl.slots[0], "auto", 1.0)  # noqa: E731
    args = ops._pa_common(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[0], wl.seq_lens,
                          cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0)
    raw = lambda: lib.vmi_paged_attention_v1_f16(*args)  # noqa: E731
    common = lambda: ops._pa_common(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[0],  # noqa: E731
                                    wl.seq_lens, cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0)
    n = 2000 if name != "cfg3" else 400
    r = {"paged_attention_v1_call_us": per_call(pa, n, 50 if name == "cfg3" else 200),
         "reshape_and_cache_call_us": per_call(rc, n),
         "c_abi_call_only_us": per_call(raw, n, 50 if name == "cfg3" else 200)}
    t0 = time.perf_counter()
    for _ in range(5000):
        common()
    r["validation_only_us"] = (time.perf_counter() - t0) / 5000 * 1e6
    res[name] = r
    print(name, json.dumps(r), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/host_overhead.json", "w"), indent=1)
