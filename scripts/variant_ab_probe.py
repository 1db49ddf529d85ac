"""Same-box comparison of named kernel variants of the DIAGNOSTIC library on cfg3 (equal lengths): the attention launch
between HIP event pairs, launches back to back, variants interleaved round by round.
`python scripts/variant_ab_probe.py name1 name2 ...` ("auto" = the default entry)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import _lib, ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

lib = _lib.use_diag().__enter__()
dev = torch.device("cuda:0")
cfg = CONFIGS["cfg3"]
wl = make_workload(cfg, dev, seed=0)
out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
res = {n: [] for n in sys.argv[1:]}
for rnd in range(3):
    for n in sys.argv[1:]:
        vid = 0 if n == "auto" else names[n]
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for i in range(230):
            if i >= 30:
                ev[i - 30][0].record()
            ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[i % 2], wl.seq_lens,
                                   cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0, _variant=vid)
            if i >= 30:
                ev[i - 30][1].record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        res[n].append(ts[len(ts) // 2])
for n, v in res.items():
    print(f"{n:28s} median us per round: " + "  ".join(f"{x:6.1f}" for x in v), flush=True)
