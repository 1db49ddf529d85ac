"""Balanced (work-queue) kernel probe: bit-equality with the one-wave-per-head kernel and timing of every
mode, on cfg3 / cfg4 with equal and ragged lengths.  `python scripts/queue_probe.py [--cfg cfg3] [--iters 60]`.
Writes gpurun_out/queue_probe_<cfg>.json."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import _lib, ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", default="cfg3")
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--kv", default="auto", choices=["auto", "fp8"])
args = ap.parse_args()

lib = _lib.use_diag().__enter__()   # the diagnostic build for the whole process (python -m vllmini_amd.build --diag)
names = {lib.vmi_paged_attention_v1_variant_name(i).decode(): i
         for i in range(1, lib.vmi_paged_attention_v1_variant_count() + 1)}
dev = torch.device("cuda:0")
cfg = CONFIGS[args.cfg]
if args.batch:
    import dataclasses
    cfg = dataclasses.replace(cfg, batch=args.batch)
D = cfg.head_size


def flags(mode=0, wq=0, nosort=0, team=0):
    return mode | (wq << 2) | (nosort << 11) | (team << 12)


def to_fp8(wl):
    c = wl.cfg
    g = torch.Generator(device=dev).manual_seed(9)
    kshape = (c.num_blocks, c.kv_heads, c.head_size // 16, 16, 16)
    vshape = (c.num_blocks, c.kv_heads, c.head_size, 16)
    wl.key_cache = (torch.randint(0, 64, kshape, dtype=torch.uint8, device=dev, generator=g)
                    | (torch.randint(0, 2, kshape, dtype=torch.uint8, device=dev, generator=g) << 7))
    wl.value_cache = (torch.randint(0, 64, vshape, dtype=torch.uint8, device=dev, generator=g)
                      | (torch.randint(0, 2, vshape, dtype=torch.uint8, device=dev, generator=g) << 7))
    return wl


def run(wl, out, t, variant):
    c = wl.cfg
    ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, c.kv_heads, wl.scale, wl.tables[t],
                           wl.seq_lens, c.block_size, c.seq_len, None, args.kv, 1.0, 0, 0, 1, 1, 0, _variant=variant)


def timeit(wl, out, variant, iters):
    for i in range(5):
        run(wl, out, i % len(wl.tables), variant)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    torch.cuda.synchronize()
    for i in range(iters):
        ev[i][0].record()
        run(wl, out, i % len(wl.tables), variant)
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return {"mean": sum(ts) / len(ts), "median": ts[len(ts) // 2], "min": ts[0]}


res = {"cfg": cfg.name, "batch": cfg.batch}
PRE = "fp8_" if args.kv == "fp8" else ""
qnames = [n for n in names if n.startswith(f"{PRE}q_d{D}_")]
ref_name = f"fp8_d{D}_bs16_h4_w1_u1_nt1" if args.kv == "fp8" else f"d{D}_h4_w1_u1_nt1"
auto_name = None
for ragged in (False, True, "sorted"):
    wl = make_workload(cfg, dev, seed=0, ragged=ragged)
    if args.kv == "fp8":
        to_fp8(wl)
    tag = "rag-sort" if ragged == "sorted" else ("ragged" if ragged else "uniform")
    kv_bytes = int(wl.seq_lens.sum().item()) * cfg.kv_heads * D * 2 * (1 if args.kv == "fp8" else 2)
    out_ref = torch.empty((cfg.batch, cfg.num_heads, D), dtype=torch.float16, device=dev)
    run(wl, out_ref, 0, names[ref_name])
    torch.cuda.synchronize()
    rows = {}
    rows[ref_name] = timeit(wl, out_ref, names[ref_name], args.iters)
    lib.vmi_debug_set_queue_flags(0)
    rows["default_entry"] = timeit(wl, out_ref, 0, args.iters)
    if args.kv == "fp8":
        hint = lib.vmi_paged_attention_v1_pick_variant_fp8(cfg.batch, cfg.num_heads, D, 16, cfg.seq_len,
                                                           int(wl.seq_lens.float().mean().item()))
    else:
        hint = lib.vmi_paged_attention_v1_pick_variant_hint(cfg.batch, cfg.num_heads, D, 16, cfg.seq_len,
                                                            int(wl.seq_lens.float().mean().item()), 0)
    rows["hint:" + lib.vmi_paged_attention_v1_variant_name(hint).decode()] = timeit(wl, out_ref, hint, args.iters)
    run(wl, out_ref, 0, names[ref_name])
    torch.cuda.synchronize()
    for qn in qnames:
        for label, f in [("auto", flags()), ("S", flags(1)), ("S_earlysort", flags(1) | (1 << 15)), ("Q_solo", flags(2, 2, 0, 1)),
                         ("Q_solo_earlysort", flags(2, 2, 0, 1) | (1 << 15)), ("Q_solo_nosort", flags(2, 2, 1, 1)),
                         ("Q_solo_w4", flags(2, 4, 0, 1)), ("Q_solo_w4_nosort", flags(2, 4, 1, 1)),
                         ("Q_team", flags(2, 0, 0, 2)), ("Q_team_nosort", flags(2, 0, 1, 2))]:
            lib.vmi_debug_set_queue_flags(f)
            out = torch.full_like(out_ref, float("nan"))
            run(wl, out, 0, names[qn])
            torch.cuda.synchronize()
            maxd = float((out.float() - out_ref.float()).abs().nan_to_num(nan=1e9).max().item())
            same = bool(torch.equal(out.view(torch.int16), out_ref.view(torch.int16)))
            if "team" in label:   # other fp32 summation order: not bit-identical to the solo kernels by design
                same = maxd <= 2e-3
            # a second launch straight after: the ticket slot must have been left clean
            out2 = torch.full_like(out_ref, float("nan"))
            run(wl, out2, 0, names[qn])
            torch.cuda.synchronize()
            same2 = bool(torch.equal(out2.view(torch.int16), out.view(torch.int16)))
            r = timeit(wl, out, names[qn], args.iters)
            r.update(bit_identical=same and same2, max_abs_diff=maxd)
            rows[f"{qn}:{label}"] = r
        lib.vmi_debug_set_queue_flags(0)
    for k, r in rows.items():
        r["TBps"] = kv_bytes / (r["mean"] * 1e-6) / 1e12
        print(f"{tag:8s} {k:34s} mean {r['mean']:8.1f} us  median {r['median']:8.1f}  min {r['min']:8.1f}  "
              f"{r['TBps']:.2f} TB/s  {'' if 'bit_identical' not in r else ('BIT-IDENTICAL' if r['bit_identical'] else 'DIFFERS max|d|=%g' % r['max_abs_diff'])}",
              flush=True)
    res[tag] = {"kv_bytes": kv_bytes, "rows": rows}
if args.kv == "fp8":
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open(f"gpurun_out/queue_probe_{cfg.name}_b{cfg.batch}_fp8.json", "w"), indent=1)
    sys.exit(0)
# ---- where does mode Q start to pay?  other length distributions, full tables (lengths overwritten) ----
wl = make_workload(cfg, dev, seed=0, ragged=False)
g = torch.Generator().manual_seed(1)
Lm = cfg.seq_len
dists = {
    "U[1/4..1]": torch.randint(Lm // 4, Lm + 1, (cfg.batch,), generator=g),
    "U[1/2..1]": torch.randint(Lm // 2, Lm + 1, (cfg.batch,), generator=g),
    "U[3/4..1]": torch.randint(3 * Lm // 4, Lm + 1, (cfg.batch,), generator=g),
    "U[7/8..1]": torch.randint(7 * Lm // 8, Lm + 1, (cfg.batch,), generator=g),
    "half full, half 1/16": torch.where(torch.rand(cfg.batch, generator=g) < 0.5, Lm, Lm // 16),
    "1/8 full, rest 1/8": torch.where(torch.rand(cfg.batch, generator=g) < 0.125, Lm, Lm // 8),
    "exponential mean 1/4": torch.clamp((torch.empty(cfg.batch).exponential_(1.0, generator=g) * Lm / 4).long() + 1, max=Lm),
    "exponential mean 1/8": torch.clamp((torch.empty(cfg.batch).exponential_(1.0, generator=g) * Lm / 8).long() + 1, max=Lm),
    "lognormal(5, 1)": torch.clamp(torch.empty(cfg.batch).log_normal_(5.0, 1.0, generator=g).long() + 1, max=Lm),
    "one full, rest 1/16": torch.where(torch.arange(cfg.batch) < 1, Lm, Lm // 16),
}
res["distributions"] = {}
for dname, lens in dists.items():
    wl.seq_lens = lens.to(torch.int32).to(dev)
    kv_bytes = int(wl.seq_lens.sum().item()) * cfg.kv_heads * D * 2 * 2
    out = torch.empty((cfg.batch, cfg.num_heads, D), dtype=torch.float16, device=dev)
    rows = {}
    lib.vmi_debug_set_queue_flags(0)
    rows["default_entry"] = timeit(wl, out, 0, args.iters)
    hint = lib.vmi_paged_attention_v1_pick_variant_hint(cfg.batch, cfg.num_heads, D, 16, cfg.seq_len,
                                                        int(wl.seq_lens.float().mean().item()), 0)
    rows["hint:" + lib.vmi_paged_attention_v1_variant_name(hint).decode()] = timeit(wl, out, hint, args.iters)
    qn = f"q_d{D}_s1q2"
    for label, f in [("auto", flags()), ("S", flags(1)), ("Q_solo", flags(2, 2, 0, 1)),
                     ("Q_solo_earlysort", flags(2, 2, 0, 1) | (1 << 15)), ("Q_team", flags(2, 0, 0, 2))]:
        lib.vmi_debug_set_queue_flags(f)
        rows[f"{qn}:{label}"] = timeit(wl, out, names[qn], args.iters)
    lib.vmi_debug_set_queue_flags(0)
    ideal = kv_bytes / 6.5e12 * 1e6
    print(f"{dname:24s} mean/max {float(wl.seq_lens.float().mean()) / Lm:.2f}  bytes/6.5TBps {ideal:6.1f} us | " +
          "  ".join(f"{k} {r['mean']:.1f}" for k, r in rows.items()), flush=True)
    res["distributions"][dname] = {"kv_bytes": kv_bytes, "rows": rows}
os.makedirs("gpurun_out", exist_ok=True)
with open(f"gpurun_out/queue_probe_{cfg.name}_b{cfg.batch}.json", "w") as f:
    json.dump(res, f, indent=1)
