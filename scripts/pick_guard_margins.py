import sys, dataclasses
sys.path.insert(0, "tests")
import torch
import test_pick_guard_gpu as G
from vllmini_amd import ops
from vllmini_amd.workload import CONFIGS, make_workload
dev = torch.device("cuda:0")
for cell in G.CELLS:
    name, batch, heads, head_size, seq_len, ragged, candidates = cell
    per = -(-seq_len // 16)
    heads, kv_heads = heads if isinstance(heads, tuple) else (heads, 0)
    cfg = dataclasses.replace(CONFIGS["cfg3"], name=name, batch=batch, num_heads=heads, head_size=head_size, seq_len=seq_len,
                              num_blocks=2 * batch * per + 8, num_kv_heads=kv_heads)
    wl = make_workload(cfg, dev, seed=21, table_sets=2, ragged=ragged)
    out = torch.empty((batch, heads, head_size), dtype=torch.float16, device=dev)
    ids = {n: i + 1 for i, n in enumerate(ops.variant_names())}
    a, label = G._graph_us(wl, out, 0, dev)
    times = {c: min(G._graph_us(wl, out, ids[c], dev)[0] for _ in range(2)) for c in candidates}
    b, _ = G._graph_us(wl, out, 0, dev)
    d = min(a, b)
    best = min(times, key=times.get)
    print(f"{name:26s} default {label:24s} {d:8.2f}  best {best:22s} {times[best]:8.2f}  ratio {d / times[best]:.3f}", flush=True)
    del wl, out
    torch.cuda.empty_cache()
