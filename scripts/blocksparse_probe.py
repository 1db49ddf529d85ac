"""Diagnostic: block-sparse paged_attention_v1 (blocksparse_vert_stride > 1) on the cfg3 / cfg4 shapes — µs next to the
dense operator and the fraction of cache blocks the pattern attends.  PYTHONPATH=. python scripts/blocksparse_probe.py"""
import torch
from vllmini_amd import ops
from vllmini_amd import _lib
_lib.use_extras().__enter__()   # bfloat16 / float32 / E5M2 / block-sparse live in libvmi_paged_attention_extras.so (build.py --extras)
from vllmini_amd.workload import CONFIGS, make_workload

dev = torch.device("cuda:0")


def timed(fn, n=30):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n + 4):
        if i >= 4:
            ev[i - 4][0].record()
        fn(i)
        if i >= 4:
            ev[i - 4][1].record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2] * 1e3


for name in ("cfg3", "cfg4"):
    cfg = CONFIGS[name]
    wl = make_workload(cfg, dev, seed=0, table_sets=2)
    out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    nblk = cfg.seq_len // cfg.block_size

    def run(i, sp=(0, 1, 1, 0)):
        ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.num_heads, wl.scale, wl.tables[i % 2],
                               wl.seq_lens, cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, *sp)

    dense = timed(run)
    print(f"{name}: dense {dense:.1f} us")
    for sp in ((2, 4, 64, 1), (1, 8, 64, 1), (4, 2, 128, 0), (16, 16, 16, 1)):
        loc, vert, bsz, step = sp
        att = 0
        for h in range(cfg.num_heads):
            off = h * step + 1
            qb = (cfg.seq_len - 1) // bsz
            for b in range(nblk):
                kb = b * cfg.block_size // bsz
                att += ((kb + off) % vert == 0) or (kb > qb - loc)
        frac = att / (nblk * cfg.num_heads)
        us = timed(lambda i: run(i, sp))
        print(f"{name}: local {loc} vert {vert} sparse-block {bsz} step {step}: {frac:.3f} of the blocks attended, "
              f"{us:.1f} us = {us / dense:.3f} of dense")
