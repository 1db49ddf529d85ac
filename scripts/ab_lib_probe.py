"""Same-box A/B of two builds of the library: `python scripts/ab_lib_probe.py <path/to/lib.so> [--cfg cfg3]` times the
default entry on equal and on ragged lengths (HIP events, launches back to back).  Run it alternately with each build."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import build as _build  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("lib")
ap.add_argument("--cfg", default="cfg3")
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--flags", type=lambda x: int(x, 0), default=0, help="queue flags; needs a -DVMI_DIAG build")
args = ap.parse_args()
_build.LIB_PATH = os.path.abspath(args.lib)
from vllmini_amd import _lib, ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

if args.flags or "diag" in os.path.basename(args.lib):   # the mode knob exists in a diagnostic build only: the path must name such a library
    _build.DIAG_LIB_PATH = _build.LIB_PATH
    _build.LIB_PATH = os.path.join(_build.OUT_DIR, _build.LIB_NAME)
    lib = _lib.use_diag().__enter__()
    lib.vmi_debug_set_queue_flags(args.flags)
else:
    lib = _lib.load()     # the product build at the given path
dev = torch.device("cuda:0")
cfg = CONFIGS[args.cfg]
g = torch.Generator().manual_seed(1)
Lm = cfg.seq_len
cases = [("uniform", False, None), ("ragged", True, None),
         ("1/8 full, rest 1/8", False, torch.where(torch.rand(cfg.batch, generator=g) < 0.125, Lm, Lm // 8)),
         ("exponential mean 1/4", False,
          torch.clamp((torch.empty(cfg.batch).exponential_(1.0, generator=g) * Lm / 4).long() + 1, max=Lm)),
         ("exponential mean 1/8", False,
          torch.clamp((torch.empty(cfg.batch).exponential_(1.0, generator=g) * Lm / 8).long() + 1, max=Lm)),
         ("lognormal(5, 1)", False, torch.clamp(torch.empty(cfg.batch).log_normal_(5.0, 1.0, generator=g).long() + 1, max=Lm)),
         ("lognormal(5.5, 0.7)", False, torch.clamp(torch.empty(cfg.batch).log_normal_(5.5, 0.7, generator=g).long() + 1, max=Lm)),
         ("one full, rest 1/16", False, torch.where(torch.arange(cfg.batch) < 1, Lm, Lm // 16))]
for tag, ragged, lens in cases:
    wl = make_workload(cfg, dev, seed=0, ragged=ragged)
    if lens is not None:
        wl.seq_lens = lens.to(torch.int32).to(dev)
    out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
    for i in range(args.iters + 30):
        if i >= 30:
            ev[i - 30][0].record()
        ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale,
                               wl.tables[i % len(wl.tables)], wl.seq_lens, cfg.block_size, cfg.seq_len, None, "auto", 1.0,
                               0, 0, 1, 1, 0)
        if i >= 30:
            ev[i - 30][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    print(f"{os.path.basename(args.lib):32s} flags {args.flags:#x} {tag:22s} mean {sum(ts) / len(ts):7.1f} us  "
          f"median {ts[len(ts) // 2]:7.1f}  min {ts[0]:7.1f}", flush=True)
