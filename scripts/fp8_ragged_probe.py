"""fp8 pages on a ragged full chip (cfg3, seq_lens ~ U{1..1024}: the lowest fraction on file) — where does the time go, and
what do the kernel's other schedules do with the same batch?  Diagnostic library: the balanced fp8 kernel with its mode knob
(auto / solo with 2 or 4 workers per workgroup / teams), the several-waves-per-head kernels the dispatcher balances, each by the
median of 60 HIP-event pairs in the call pair; and the per-wave timeline of the default (when does each wave finish, how long
was it busy, how many tokens did it own).  `python scripts/fp8_ragged_probe.py [out.json]`"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import _lib, cache_ops, ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

lib = _lib.use_diag().__enter__()
dev = torch.device("cuda:0")
cfg = CONFIGS["cfg3"]
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
res = {}


def flags(mode=0, wq=0, team=0):
    return mode | (wq << 2) | (team << 12)


for tag, ragged in (("U{1..1024}", True), ("equal", False)):
    wl = make_workload(cfg, dev, seed=4321, table_sets=2, ragged=ragged)
    g = torch.Generator(device=dev).manual_seed(5)
    code = lambda shape: (torch.randint(0, 64, shape, dtype=torch.uint8, device=dev, generator=g)          # noqa: E731
                          | (torch.randint(0, 2, shape, dtype=torch.uint8, device=dev, generator=g) << 7))
    kc = code((cfg.num_blocks, cfg.kv_heads, cfg.head_size // 16, 16, 16))
    vc = code((cfg.num_blocks, cfg.kv_heads, cfg.head_size, 16))
    out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    tok = int(wl.seq_lens.sum().item())
    nbytes = 2 * tok * cfg.num_heads * cfg.head_size + 2 * cfg.batch * cfg.num_heads * cfg.head_size * 2

    def pair(i, vid, ev=None):
        t = i % 2
        cache_ops.reshape_and_cache(wl.key, wl.value, kc, vc, wl.slots[t], "fp8", 1.0)
        if ev:
            ev[0].record()
        ops.paged_attention_v1(out, wl.query, kc, vc, cfg.kv_heads, wl.scale, wl.tables[t], wl.seq_lens, 16, cfg.seq_len, None,
                               "fp8", 1.0, 0, 0, 1, 1, 0, _variant=vid)
        if ev:
            ev[1].record()

    def timed(vid, fl):
        lib.vmi_debug_set_queue_flags(fl)
        for i in range(15):
            pair(i, vid)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
        for i, ev in enumerate(evs):
            pair(i, vid, ev)
        torch.cuda.synchronize()
        lib.vmi_debug_set_queue_flags(0)
        return sorted(a.elapsed_time(b) * 1e3 for a, b in evs)[30]

    rows = {}
    q = names["fp8_q_d64_s2q4m"]
    for label, vid, fl in (("default entry", 0, 0), ("balanced, auto", q, 0), ("balanced, solo x4", q, flags(2, 4, 1)),
                           ("balanced, solo x2", q, flags(2, 2, 1)), ("balanced, solo x3", q, flags(2, 3, 1)),
                           ("balanced, teams", q, flags(2, 0, 2)), ("balanced, mode S", q, flags(1)),
                           ("fp8_q_d64_s1q2 auto", names["fp8_q_d64_s1q2"], 0),
                           ("4 waves per head", names["fp8_d64_bs16_h1_w4_u2_nt1"] if "fp8_d64_bs16_h1_w4_u2_nt1" in names else 0, 0),
                           ("2 waves per head", names.get("fp8_d64_bs16_h1_w2_u2_nt1", 0), 0),
                           ("8 waves per head", names.get("fp8_d64_bs16_h1_w8_u2_nt1", 0), 0)):
        if label != "default entry" and not vid:
            continue
        us = timed(vid, fl)
        rows[label] = {"us": round(us, 1), "TBps": round(nbytes / us / 1e6, 2), "kernel": ops.last_launch_label()}
        print(tag, label, rows[label], flush=True)
    # per-wave timeline of the balanced kernel, automatic mode
    rec = torch.zeros((8192, 4), dtype=torch.int64, device=dev)
    spans, busy_med, busy_max, ends = [], [], [], []
    for rep in range(6):
        rec.zero_()
        torch.cuda.synchronize()
        assert lib.vmi_diag_set_wave_timeline(rec.data_ptr(), 0) == 0
        pair(rep, q)
        pair(rep + 1, q)
        torch.cuda.synchronize()
        assert lib.vmi_diag_set_wave_timeline(None, 0) == 0
        r = rec.cpu().numpy().astype(np.int64)
        r = r[r[:, 1] > 0]
        t0 = r[:, 0].min()
        st, en = (r[:, 0] - t0) * 0.01, (r[:, 1] - t0) * 0.01
        w = (en - st) > 1.0
        spans.append(float(en.max()))
        busy_med.append(float(np.median((en - st)[w])))
        busy_max.append(float((en - st)[w].max()))
        ends.append(np.percentile(en[w], [10, 50, 90, 99]).round(1).tolist())
    rows["timeline (balanced, auto)"] = {"span_us": float(np.median(spans)), "median_worker_busy_us": float(np.median(busy_med)),
                                         "longest_worker_busy_us": float(np.median(busy_max)), "workers": int(w.sum()),
                                         "end_p10_p50_p90_p99_us": np.median(np.array(ends), axis=0).round(1).tolist()}
    print(tag, "timeline", rows["timeline (balanced, auto)"], flush=True)
    rows["bytes"] = nbytes
    res[tag] = rows
    del wl, kc, vc
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
