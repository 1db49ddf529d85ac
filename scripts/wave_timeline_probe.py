"""When do the waves of a balanced-kernel launch start and finish?  Diagnostic library only: its q_* kernels write one
record per wave — start and end in ticks of the constant 100 MHz clock, HW_ID, XCC_ID (vmi_diag_set_wave_timeline).
For cfg3 (batch 256 x 12 heads, 1024 tokens) with equal, ragged and heavy-tailed lengths, fp16 and fp8 pages: the span of
the launch, how far the waves' start times are spread (dispatch), the distribution of their end times (what the slowest
wave adds to the median one), and the same per XCD and per CU — is the chip served evenly?
`python scripts/wave_timeline_probe.py [out.json]`"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import _lib, ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

lib = _lib.use_diag().__enter__()
dev = torch.device("cuda:0")
cfg = CONFIGS["cfg3"]
TICK_US = 0.01
rec = torch.zeros((8192, 4), dtype=torch.int64, device=dev)
g = torch.Generator().manual_seed(1)
L = cfg.seq_len
cases = [("equal", False, None, "auto"), ("U{1..1024}", True, None, "auto"),
         ("1/8 full, rest 1/8", False, torch.where(torch.rand(cfg.batch, generator=g) < 0.125, L, L // 8), "auto"),
         ("equal, fp8 pages", False, None, "fp8")]
results = {}
for tag, ragged, lens, kvd in cases:
    wl = make_workload(cfg, dev, seed=0, ragged=ragged)
    if lens is not None:
        wl.seq_lens = lens.to(torch.int32).to(dev)
    kc, vc = wl.key_cache, wl.value_cache
    if kvd == "fp8":
        gg = torch.Generator(device=dev).manual_seed(5)
        code = lambda shape: (torch.randint(0, 64, shape, dtype=torch.uint8, device=dev, generator=gg)        # noqa: E731
                              | (torch.randint(0, 2, shape, dtype=torch.uint8, device=dev, generator=gg) << 7))
        kc = code((cfg.num_blocks, cfg.kv_heads, cfg.head_size // 16, cfg.block_size, 16))
        vc = code((cfg.num_blocks, cfg.kv_heads, cfg.head_size, cfg.block_size))
    out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)

    def launch(i):
        ops.paged_attention_v1(out, wl.query, kc, vc, cfg.kv_heads, wl.scale, wl.tables[i % len(wl.tables)], wl.seq_lens,
                               cfg.block_size, cfg.seq_len, None, kvd, 1.0, 0, 0, 1, 1, 0)

    for i in range(30):
        launch(i)
    torch.cuda.synchronize()
    spans, p50s, p99s, skews, per_xcc, per_cu_spread, durs, med_xcc, idle = [], [], [], [], [], [], [], [], []
    for rep in range(8):
        rec.zero_()
        torch.cuda.synchronize()
        assert lib.vmi_diag_set_wave_timeline(rec.data_ptr(), 0) == 0
        launch(rep)
        launch(rep + 1)            # two launches back to back: the second one's records stay (a launch behind a launch)
        torch.cuda.synchronize()
        assert lib.vmi_diag_set_wave_timeline(None, 0) == 0
        r = rec.cpu().numpy().astype(np.int64)
        r = r[r[:, 1] > 0]
        t0 = r[:, 0].min()
        start, end = (r[:, 0] - t0) * TICK_US, (r[:, 1] - t0) * TICK_US
        busy = (end - start) > 1.0                     # (waves that retire at once in mode Q are not workers)
        spans.append(float(end.max()))
        p50s.append(float(np.median(end[busy])))
        p99s.append(float(np.percentile(end[busy], 99)))
        skews.append(float(start.max()))
        durs.append(float(np.median((end - start)[busy])))
        xcc = r[:, 3] & 0xF
        per_xcc.append([float(end[busy & (xcc == x)].max()) if (busy & (xcc == x)).any() else 0.0 for x in range(8)])
        med_xcc.append([float(np.median(end[busy & (xcc == x)])) if (busy & (xcc == x)).any() else 0.0 for x in range(8)])
        cu_key = (xcc << 16) | ((r[:, 2] >> 8) & 0xFF)            # XCC, SE / SH / CU bits of HW_ID
        last_by_cu = np.array([end[busy & (cu_key == k)].max() for k in np.unique(cu_key[busy])])
        per_cu_spread.append([float(last_by_cu.min()), float(np.median(last_by_cu)), float(last_by_cu.max()), int(len(last_by_cu))])
        idle.append(float(1.0 - last_by_cu.mean() / last_by_cu.max()))   # share of CU-time between a CU's last wave and the launch's
    label = ops.last_launch_label()
    m = lambda a: float(np.median(np.array(a), axis=0)) if np.ndim(a) == 1 else np.median(np.array(a), axis=0).round(2).tolist()   # noqa: E731
    results[tag] = {"kernel": label, "waves_recorded": int(len(r)), "workers": int(busy.sum()),
                    "span_us": m(spans), "last_wave_start_us": m(skews), "median_worker_busy_us": m(durs),
                    "end_p50_us": m(p50s), "end_p99_us": m(p99s),
                    "median_end_per_xcc_us": m(med_xcc), "last_end_per_xcc_us": m(per_xcc),
                    "last_end_per_cu_us_min_median_max_n": m(per_cu_spread), "cu_time_idle_at_the_end": m(idle)}
    print(tag, json.dumps(results[tag]), flush=True)
if len(sys.argv) > 1:
    json.dump(results, open(sys.argv[1], "w"), indent=1)
