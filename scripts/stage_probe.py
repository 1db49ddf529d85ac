"""LDS-staged experiment kernels (pa_stage.hip) against the direct-to-register kernels: results and HIP-event time,
back-to-back launches.  `python scripts/stage_probe.py` -> gpurun_out/stage_probe.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import _lib, ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

lib = _lib.use_diag().__enter__()   # the diagnostic build for the whole process (python -m vllmini_amd.build --diag)
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
dev = torch.device("cuda:0")
res = {}


def timeit(fn, iters=40):
    for i in range(5):
        fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for i in range(iters):
        ev[i][0].record()
        fn(i)
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return sum(ts) / len(ts)


for cname, kv in (("cfg3", "auto"), ("cfg3", "fp8"), ("cfg4", "auto"), ("cfg4", "fp8")):
    cfg = CONFIGS[cname]
    wl = make_workload(cfg, dev, seed=0)
    D = cfg.head_size
    if kv == "fp8":
        g = torch.Generator(device=dev).manual_seed(9)
        kshape = (cfg.num_blocks, cfg.kv_heads, D // 16, 16, 16)
        wl.key_cache = (torch.randint(0, 64, kshape, dtype=torch.uint8, device=dev, generator=g)
                        | (torch.randint(0, 2, kshape, dtype=torch.uint8, device=dev, generator=g) << 7))
        vshape = (cfg.num_blocks, cfg.kv_heads, D, 16)
        wl.value_cache = (torch.randint(0, 64, vshape, dtype=torch.uint8, device=dev, generator=g)
                          | (torch.randint(0, 2, vshape, dtype=torch.uint8, device=dev, generator=g) << 7))
    out = torch.empty((cfg.batch, cfg.num_heads, D), dtype=torch.float16, device=dev)

    def run(i, variant, o=None):
        ops.paged_attention_v1(out if o is None else o, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale,
                               wl.tables[i % len(wl.tables)], wl.seq_lens, 16, cfg.seq_len, None, kv, 1.0, 0, 0, 1, 1, 0,
                               _variant=variant)

    pre = "fp8_" if kv == "fp8" else ""
    direct = [n for n in names if n in (f"d{D}_h4_w1_u1_nt1", f"fp8_d{D}_bs16_h4_w1_u1_nt1", f"fp8_d{D}_bs16_h4_w1_u2_nt1",
                                        f"d{D}_mh4_h4_u1_nt1_lock")]
    direct = [n for n in direct if n.startswith("fp8_") == (kv == "fp8")]
    ref = torch.empty_like(out)
    run(0, 0, ref)
    torch.cuda.synchronize()
    rows = {"default_entry": timeit(lambda i: run(i, 0))}
    for n in direct:
        rows[n] = timeit(lambda i, n=n: run(i, names[n]))
    for n in [x for x in names if x.startswith(f"stage_{pre}d{D}_")]:
        o = torch.full_like(out, float("nan"))
        run(0, names[n], o)
        torch.cuda.synchronize()
        maxd = float((o.float() - ref.float()).abs().nan_to_num(nan=1e9).max())
        rows[n] = timeit(lambda i, n=n: run(i, names[n]))
        rows[n + ":max|d| vs default"] = maxd
    res[f"{cname}_{kv}"] = rows
    print(cname, kv, json.dumps({k: (round(v, 1) if v > 1 else v) for k, v in rows.items()}), flush=True)
    del wl
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/stage_probe.json", "w"), indent=1)
