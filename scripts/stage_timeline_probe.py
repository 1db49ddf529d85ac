"""The latency chain of an UNDER-FILLED chip, term by term (BASELINE configs[1] = batch 32 x 512 tokens, and batch 1 / 8 / 16 /
32 at 1024 tokens): where do the microseconds of a paged_attention_v1 launch go when the launch is a chain of memory round
trips rather than a stream?

  python scripts/stage_timeline_probe.py --stamps out.json     diagnostic library: every wave of pa_v1_kernel writes ten
                                                               stage stamps (vmi_diag_set_stage_stamps, 100 MHz clock)
  python scripts/stage_timeline_probe.py --plain               product library, no stamps: every case = 10 + 50 call pairs
                                                               (reshape_and_cache, paged_attention_v1); run it under
                                                               `rocprofv3 --kernel-trace` and hand the trace to
  python scripts/stage_timeline_probe.py --summarize DIR out.json    per case: the attention kernel's duration as rocprofv3
                                                               sees it (dispatch begin -> end, incl. what lies before the first
                                                               wave's first instruction and behind the last wave's last)
Cases: `name:batch:seq_len:variant` ("auto" = the default entry, "auto_nows" = the default entry without a workspace);
CASES below or --cases a,b,c."""
import csv
import glob
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [
    "cfg2:32:512:auto", "cfg2:32:512:d64_h1_w8_u1_nt1", "cfg2:32:512:d64_h1_w8_u1_nt0", "cfg2:32:512:d64_h1_w8_u1a2_nt0",
    "cfg2:32:512:d64_h1_w4_u2_nt0", "cfg2:32:512:d64_h1_w4_u2a4_nt0", "cfg2:32:512:d64_h1_w8_u2_nt0",
    "b1:1:1024:auto", "b8:8:1024:auto", "b16:16:1024:auto", "b32:32:1024:auto", "b32:32:1024:d64_h1_w8_u1a2_nt0",
    "b24:24:1024:auto", "b24:24:1024:d64_h1_w8_u1a2_nt0", "b1_l16:1:16:auto", "b1_l256:1:256:auto",
]
STAGES = ["entry", "seq_len known", "first pages requested", "first K group consumed", "K pass done", "maxima exchanged",
          "probabilities written", "V pass done", "partial outputs exchanged", "out stored"]
WARM, TIMED = 10, (50 if "--few" not in sys.argv else 12)


def arg_cases():
    for i, a in enumerate(sys.argv):
        if a == "--cases":
            return sys.argv[i + 1].split(",")
    return CASES


def setup(case, dev):
    import dataclasses
    import torch
    from vllmini_amd.workload import CONFIGS, make_workload
    name, b, L, var = case.split(":")[:4]
    kv = (case.split(":") + ["auto"])[4]            # optional fifth field: "fp8" = E4M3 pages, kv_scale 1
    hd = (case.split(":") + ["auto", "12x64"])[5] if len(case.split(":")) > 5 else "12x64"   # optional sixth field: heads x head size [x KV heads]
    b, L = int(b), int(L)
    per = -(-L // 16)
    cfg = dataclasses.replace(CONFIGS["cfg2"], name=name, batch=b, seq_len=L, num_blocks=max(4096, 2 * b * per),
                              num_heads=int(hd.split("x")[0]), head_size=int(hd.split("x")[1]),
                              num_kv_heads=int((hd.split("x") + ["0"])[2]))     # "32x128x8": 8 KV heads (grouped-query)
    wl = make_workload(cfg, dev, seed=7, table_sets=2)
    wl.kv = kv
    if kv == "fp8":
        g8 = torch.Generator(device=dev).manual_seed(9)
        H, D = cfg.kv_heads, cfg.head_size
        wl.key_cache = torch.randint(0, 64, (cfg.num_blocks, H, D // 16, 16, 16), dtype=torch.uint8, device=dev, generator=g8)
        wl.value_cache = torch.randint(0, 64, (cfg.num_blocks, H, D, 16), dtype=torch.uint8, device=dev, generator=g8)
    out = torch.empty((b, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    return cfg, wl, out, var


def pair(ops, cache_ops, cfg, wl, out, i, vid, scatter=True):
    t = i % len(wl.tables)
    if scatter:
        cache_ops.reshape_and_cache(wl.key, wl.value, wl.key_cache, wl.value_cache, wl.slots[t], wl.kv, 1.0)
    ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[t], wl.seq_lens,
                           cfg.block_size, cfg.seq_len, None, wl.kv, 1.0, 0, 0, 1, 1, 0, _variant=vid)


def run_plain():
    import torch
    from vllmini_amd import cache_ops, ops
    dev = torch.device("cuda:0")
    names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
    order = []
    for case in arg_cases():
        cfg, wl, out, var = setup(case, dev)
        ops.set_workspace_enabled(var != "auto_nows")      # "auto_nows": the default entry without a workspace (round 4's picks)
        vid = 0 if var.startswith("auto") else names[var]
        try:
            pair(ops, cache_ops, cfg, wl, out, 0, vid, scatter=False)      # (a refused launch — a split kernel whose workgroups
        except RuntimeError as e:                                         #  would not all be resident — leaves no kernel behind)
            print(f"skipped {case}: {str(e)[:120]}", file=sys.stderr)
            ops.set_workspace_enabled(True)
            continue
        torch.cuda.synchronize()
        for i in range(WARM + TIMED):
            pair(ops, cache_ops, cfg, wl, out, i, vid)
        torch.cuda.synchronize()
        ops.set_workspace_enabled(True)
        order.append({"case": case, "kernel": ops.variant_names()[(vid or ops.last_variant()) - 1], "launches": WARM + TIMED + 1})
        del wl, out
    print(json.dumps(order))


def summarize(trace_dir, out_path):
    order = json.loads(open(os.path.join(trace_dir, "order.json")).read().strip().splitlines()[-1])
    f = sorted(glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True))[0]
    every = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in every if "pa_v1_kernel" in r["Kernel_Name"] or "pa_q_kernel" in r["Kernel_Name"] or
            "pa_split" in r["Kernel_Name"]]
    scat = [r for r in every if "reshape_and_cache" in r["Kernel_Name"]]      # the other half of the call pair
    res, k, ks = [], 0, 0
    us = lambda chunk: np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in chunk])   # noqa: E731
    have_scat = len(scat) == sum(o["launches"] - 1 for o in order)      # (a case = 1 probing attention launch + launches - 1 call pairs)
    for o in order:
        chunk = rows[k: k + o["launches"]][WARM + 1:]
        d = us(chunk)
        rec = {**o, "rocprofv3_us_mean": round(float(d.mean()), 2), "rocprofv3_us_median": round(float(np.median(d)), 2),
               "rocprofv3_us_min": round(float(d.min()), 2), "grid": chunk[0].get("Grid_Size_X", "") + "x" + chunk[0].get("Grid_Size_Y", ""),
               "workgroup": chunk[0].get("Workgroup_Size_X", "")}
        if have_scat:
            ds = us(scat[ks: ks + o["launches"] - 1][WARM:])
            rec["scatter_us_median"] = round(float(np.median(ds)), 2)
            rec["scatter_grid"] = scat[ks + WARM].get("Grid_Size_X", "") + "x" + scat[ks + WARM].get("Grid_Size_Y", "")
        k += o["launches"]
        ks += o["launches"] - 1
        res.append(rec)
        print(json.dumps(res[-1]), flush=True)
    assert k == len(rows), (k, len(rows))
    json.dump(res, open(out_path, "w"), indent=1)


def run_stamps(out_path):
    import torch
    from vllmini_amd import _lib, cache_ops, ops
    lib = _lib.use_diag().__enter__()
    dev = torch.device("cuda:0")
    names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
    rec = torch.zeros((65536, 12), dtype=torch.int64, device=dev)
    results, raw = {}, {}
    for case in arg_cases():
        cfg, wl, out, var = setup(case, dev)
        vid = 0 if var == "auto" else names[var]
        for i in range(20):
            pair(ops, cache_ops, cfg, wl, out, i, vid)
        torch.cuda.synchronize()
        kern = ops.variant_names()[(vid or ops.last_variant()) - 1]
        if kern.startswith("q_"):
            print(case, "-> balanced kernel, no stage stamps", flush=True)
            continue
        reps = []
        for rep in range(12):
            rec.zero_()
            torch.cuda.synchronize()
            assert lib.vmi_diag_set_stage_stamps(rec.data_ptr(), 0) == 0
            pair(ops, cache_ops, cfg, wl, out, rep, vid)           # in the call pair, behind a launch of the same kind:
            pair(ops, cache_ops, cfg, wl, out, rep + 1, vid)       # the second launch's records stay
            torch.cuda.synchronize()
            assert lib.vmi_diag_set_stage_stamps(None, 0) == 0
            r = rec.cpu().numpy().astype(np.int64)
            r = r[r[:, 0] > 0]
            reps.append(r)
        # per repetition: every stamp relative to the launch's first wave entry, in us
        med_of_waves, last_of_waves, first_of_waves, spans, dur = [], [], [], [], []
        for r in reps:
            t0 = r[:, 0].min()
            ts = (r[:, :10] - t0) * 0.01
            ts = np.where(r[:, :10] > 0, ts, np.nan)
            med_of_waves.append(np.nanmedian(ts, axis=0))
            last_of_waves.append(np.nanmax(ts, axis=0))
            first_of_waves.append(np.nanmin(ts, axis=0))
            spans.append(float(np.nanmax(ts[:, 9])))
            dur.append(np.nanmedian(np.diff(ts, axis=1), axis=0))
        m = lambda a: np.nanmedian(np.array(a), axis=0).round(2).tolist()   # noqa: E731
        r = reps[-1]
        results[case] = {"kernel": kern, "waves": int(len(r)), "blocks_per_wave_max": int((r[:, 11] >> 8).max()),
                         "xcds_used": int(len(np.unique(r[:, 11] & 0xF))),
                         "cus_used": int(len(np.unique(((r[:, 11] & 0xF) << 16) | ((r[:, 10] >> 8) & 0xFF)))),
                         "span_first_entry_to_last_store_us": float(np.median(spans)),
                         "stages": STAGES,
                         "stamp_us_first_wave": m(first_of_waves), "stamp_us_median_wave": m(med_of_waves),
                         "stamp_us_last_wave": m(last_of_waves), "stage_duration_us_median_wave": m(dur)}
        print(case, json.dumps(results[case]), flush=True)
        raw[case.replace(":", "_")] = reps[-1]              # the last repetition, every wave: for per-XCD / per-CU questions
        del wl, out
    json.dump(results, open(out_path, "w"), indent=1)
    np.savez_compressed(out_path.replace(".json", "_raw.npz"), **raw)


if __name__ == "__main__":
    if "--plain" in sys.argv:
        run_plain()
    elif "--summarize" in sys.argv:
        i = sys.argv.index("--summarize")
        summarize(sys.argv[i + 1], sys.argv[i + 2])
    else:
        i = sys.argv.index("--stamps")
        run_stamps(sys.argv[i + 1])
