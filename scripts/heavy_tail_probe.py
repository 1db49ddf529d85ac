"""Heavy-tailed batches through the balanced kernel: the kernel's own choice (round 3: teams for the long items + solo
quads for the short ones) against round 2's teams-for-everything (QF_NOHYBRID) and forced solo workers, on the length
distributions of scripts/queue_probe.py.  HIP events, launches back to back, table sets alternating.
`python scripts/heavy_tail_probe.py [--cfg cfg3] [--kv auto|fp8] [--iters 100]` -> stdout + gpurun_out/heavy_tail_<cfg>[_fp8].json"""
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
ap.add_argument("--kv", default="auto", choices=["auto", "fp8"])
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--kernel", default="", help="balanced kernel by name (default: the config's default)")
args = ap.parse_args()
lib = _lib.use_diag().__enter__()   # the mode knob lives in the diagnostic build
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
dev = torch.device("cuda:0")
cfg = CONFIGS[args.cfg]
if args.batch:
    import dataclasses
    cfg = dataclasses.replace(cfg, batch=args.batch, num_blocks=2 * args.batch * cfg.blocks_per_seq)
D, Lm = cfg.head_size, cfg.seq_len
wl = make_workload(cfg, dev, seed=0, ragged=False)
if args.kv == "fp8":
    g8 = torch.Generator(device=dev).manual_seed(9)
    for name, shape in (("key_cache", (cfg.num_blocks, cfg.kv_heads, D // 16, 16, 16)), ("value_cache", (cfg.num_blocks, cfg.kv_heads, D, 16))):
        setattr(wl, name, torch.randint(0, 64, shape, dtype=torch.uint8, device=dev, generator=g8)
                | (torch.randint(0, 2, shape, dtype=torch.uint8, device=dev, generator=g8) << 7))
qn = {("auto", 64): "q_d64_s1q2", ("auto", 128): "q_d128_s1q1", ("fp8", 64): "fp8_q_d64_s2q4m", ("fp8", 128): "fp8_q_d128_s1q2m"}[(args.kv, D)]
qn = args.kernel or qn


def flags(mode=0, wq=0, nosort=0, team=0, nohybrid=0):
    return mode | (wq << 2) | (nohybrid << 10) | (nosort << 11) | (team << 12)


def run(out, t):
    ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[t], wl.seq_lens,
                           cfg.block_size, Lm, None, args.kv, 1.0, 0, 0, 1, 1, 0, _variant=names[qn])


def timeit(out):
    for i in range(25):      # (the first launches after a pause run on a lower clock: 5 warm-ups read 3 us high)
        run(out, i % len(wl.tables))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
    torch.cuda.synchronize()
    for i in range(args.iters):
        ev[i][0].record()
        run(out, i % len(wl.tables))
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]


g = torch.Generator().manual_seed(1)
B = cfg.batch
dists = {
    "equal": torch.full((B,), Lm),
    "U{1..L}": torch.randint(1, Lm + 1, (B,), generator=g),
    "U[1/4..1]": torch.randint(Lm // 4, Lm + 1, (B,), generator=g),
    "U[1/8..1]": torch.randint(Lm // 8, Lm + 1, (B,), generator=g),
    "triangular (min of 2 U)": torch.minimum(torch.randint(1, Lm + 1, (B,), generator=g), torch.randint(1, Lm + 1, (B,), generator=g)),
    "3/4 full, rest 1/16": torch.where(torch.rand(B, generator=g) < 0.75, Lm, Lm // 16),
    "2/3 full, rest 1/16": torch.where(torch.rand(B, generator=g) < 0.667, Lm, Lm // 16),
    "half full, half 1/16": torch.where(torch.rand(B, generator=g) < 0.5, Lm, Lm // 16),
    "half U[1/2..1], half 1/8": torch.where(torch.rand(B, generator=g) < 0.5, torch.randint(Lm // 2, Lm + 1, (B,), generator=g), torch.full((B,), Lm // 8)),
    "1/8 full, rest 1/8": torch.where(torch.rand(B, generator=g) < 0.125, Lm, Lm // 8),
    "1/16 full, rest 1/16": torch.where(torch.rand(B, generator=g) < 0.0625, Lm, Lm // 16),
    "exponential mean 1/4": torch.clamp((torch.empty(B).exponential_(1.0, generator=g) * Lm / 4).long() + 1, max=Lm),
    "exponential mean 1/8": torch.clamp((torch.empty(B).exponential_(1.0, generator=g) * Lm / 8).long() + 1, max=Lm),
    "lognormal(5, 1)": torch.clamp(torch.empty(B).log_normal_(5.0, 1.0, generator=g).long() + 1, max=Lm),
    "one full, rest 1/16": torch.where(torch.arange(B) < 1, Lm, Lm // 16),
    "4 full, rest 1/32": torch.where(torch.arange(B) < 4, Lm, Lm // 32),
}
MODES = [("auto", flags()), ("teams only (r02)", flags(2, 0, 0, 2, 1)), ("hybrid forced", flags(2, 0, 0, 2)), ("solo forced", flags(2, 2, 0, 1)),
         ("solo 4 workers", flags(2, 4, 0, 1))]
res = {}
print(f"{cfg.name} {args.kv} {qn}: median kernel us by HIP events ({args.iters} launches)")
for dname, lens in dists.items():
    wl.seq_lens = lens.to(torch.int32).to(dev)
    kv_bytes = int(wl.seq_lens.sum().item()) * cfg.kv_heads * D * 2 * (1 if args.kv == "fp8" else 2)
    out = torch.empty((B, cfg.num_heads, D), dtype=torch.float16, device=dev)
    row, outs = {}, {}
    for label, f in MODES:
        lib.vmi_debug_set_queue_flags(f)
        o = torch.full_like(out, float("nan"))
        run(o, 0)
        torch.cuda.synchronize()
        outs[label] = o
        row[label] = timeit(out)
    lib.vmi_debug_set_queue_flags(0)
    ref = outs["solo forced"].float()
    dmax = max(float((outs[k].float() - ref).abs().nan_to_num(nan=1e9).max()) for k in outs)
    ideal = kv_bytes / 6.7e12 * 1e6
    res[dname] = {"kv_bytes": kv_bytes, "ideal_us_at_6.7TBps": ideal, "us": row, "max_abs_diff_between_modes": dmax}
    print(f"{dname:24s} mean/max {float(lens.float().mean()) / Lm:.2f} bytes/6.7TB/s {ideal:6.1f} | " +
          "  ".join(f"{k} {v:6.1f}" for k, v in row.items()) + f" | max|d| between modes {dmax:.1e}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"cfg": cfg.name, "kv": args.kv, "kernel": qn, "rows": res},
          open(f"gpurun_out/heavy_tail_{cfg.name}{'_b%d' % args.batch if args.batch else ''}{'_fp8' if args.kv == 'fp8' else ''}.json", "w"), indent=1)
