#!/usr/bin/env python3
"""Diagnostic companions of bench.py (not part of the bench contract; run on the GPU box via gpurun):

    python scripts/bench_diag.py --sweep  [bench flags]   every kernel variant of the config's head size, kernel us by
                                                          HIP events -> gpurun_out/sweep_<cfg>.json
    python scripts/bench_diag.py --matrix [bench flags]   every (head size, block size, element type) the operators are
                                                          built for -> gpurun_out/matrix.json
    python scripts/bench_diag.py --diag   [bench flags]   plain and gather read bandwidth of this box with the math
                                                          removed (needs the DIAGNOSTIC library: vmi_diag_* are not in
                                                          the product .so; `python -m vllmini_amd.build --diag`)

The workload flags (--config, --batch, --seq-len, --kv, --steps, --warmup) are bench.py's.
"""
from __future__ import annotations

import dataclasses
import json
import os
import statistics
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import bench  # noqa: E402
from vllmini_amd import ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402


def out_path(name):
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    return os.path.join(REPO, "gpurun_out", name)


def run_matrix(args, base, dev):
    res = []
    for kind in ("float16", "bfloat16", "fp8_kv", "fp8_e5m2_kv", "float32"):
        dt = {"bfloat16": torch.bfloat16, "float32": torch.float32}.get(kind, torch.float16)
        for D in (64, 80, 96, 112, 128, 192, 256):
            for bs in (8, 16, 32):
                if kind.startswith("fp8") and bs == 8:
                    continue
                per = -(-base.seq_len // bs)
                c = dataclasses.replace(base, name=f"m_d{D}_bs{bs}", head_size=D, block_size=bs,
                                        num_blocks=2 * base.batch * per + 8)
                wl = make_workload(c, dev, seed=5, table_sets=2)
                bench.KV_DTYPE = "auto"
                if kind == "bfloat16":
                    wl.key_cache, wl.value_cache, wl.qkv = (wl.key_cache.to(dt), wl.value_cache.to(dt), wl.qkv.to(dt))
                elif kind == "float32":          # x = 4 layout: same bytes per chunk, half the elements
                    wl.key_cache = torch.empty((c.num_blocks, c.num_heads, D // 4, bs, 4), dtype=dt, device=dev).uniform_(-1, 1)
                    wl.value_cache = torch.empty((c.num_blocks, c.num_heads, D, bs), dtype=dt, device=dev).uniform_(-1, 1)
                    wl.qkv = wl.qkv.to(dt)
                elif kind.startswith("fp8"):
                    bench.KV_DTYPE = "fp8" if kind == "fp8_kv" else "fp8_e5m2"
                    gk = torch.Generator(device=dev).manual_seed(3)
                    wl.key_cache = torch.randint(0, 64, (c.num_blocks, c.num_heads, D // 16, bs, 16), dtype=torch.uint8,
                                                 device=dev, generator=gk)
                    wl.value_cache = torch.randint(0, 64, (c.num_blocks, c.num_heads, D, bs), dtype=torch.uint8,
                                                   device=dev, generator=gk)
                out = torch.empty((c.batch, c.num_heads, D), dtype=dt, device=dev)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                for i in range(args.warmup + args.steps):
                    k = i - args.warmup
                    if k >= 0:
                        ev[k][0].record()
                    bench.attend(wl, out, i % len(wl.tables), 0)
                    if k >= 0:
                        ev[k][1].record()
                torch.cuda.synchronize(dev)
                us = statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3
                vid = ops.last_variant()      # what the launches really ran (0 for the float32 kernels)
                nbytes = bench.alg_bytes(c)
                if kind == "float32":            # K and V bytes double
                    nbytes += 2 * c.batch * c.kv_heads * c.seq_len * c.head_size * 2
                row = {"dtype": kind, "head_size": D, "block_size": bs, "us_median": us,
                       "gbps": nbytes / (us * 1e-6) / 1e9,
                       "variant": "pa_v1_f32_kernel" if kind == "float32" or not vid else ops.variant_names()[vid - 1]}
                res.append(row)
                print(json.dumps(row), file=sys.stderr, flush=True)
                del wl, out
                torch.cuda.empty_cache()
    bench.KV_DTYPE = "auto"
    with open(out_path("matrix.json"), "w") as f:
        json.dump({"batch": base.batch, "num_heads": base.num_heads, "seq_len": base.seq_len, "rows": res}, f, indent=1)


def run_diag(wl, dev):
    from vllmini_amd import _lib

    lib = _lib.load_diag()          # raises when only the product library exists
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    src = wl.key_cache
    nbytes = src.numel() * 2
    stream = torch.cuda.current_stream(dev).cuda_stream
    res = []
    for nt in (0, 1):
        for blocks in (1024, 2048, 4096, 8192, 16384):
            for _ in range(3):
                lib.vmi_diag_stream_read(src.data_ptr(), nbytes, sink.data_ptr(), blocks, nt, dev.index, stream)
            evs = []
            for i in range(20):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_ = wl.value_cache if i % 2 else wl.key_cache   # alternate pools: defeat the 256 MiB MALL
                a.record()
                lib.vmi_diag_stream_read(s_.data_ptr(), nbytes, sink.data_ptr(), blocks, nt, dev.index, stream)
                b.record()
                evs.append((a, b))
            torch.cuda.synchronize(dev)
            ms = statistics.median(a.elapsed_time(b) for a, b in evs)
            res.append({"nt": nt, "blocks": blocks, "bytes": nbytes, "us": ms * 1e3, "gbps": nbytes / (ms * 1e-3) / 1e9})
            print(json.dumps(res[-1]), file=sys.stderr, flush=True)
    # gather reads: contiguous chunk size x KiB in flight per wave x waves (768 blocks of 256 = 3072 waves = cfg3)
    for blocks in (768, 384, 192):
        for kb in (2, 4, 8, 16):
            for infl in (1, 2, 4, 8, 16):
                evs = []
                for i in range(16):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s_ = wl.value_cache if i % 2 else wl.key_cache
                    a.record()
                    rc = lib.vmi_diag_gather_read(s_.data_ptr(), nbytes, sink.data_ptr(), kb, infl, blocks, 1, dev.index, stream)
                    b.record()
                    assert rc == 0
                    evs.append((a, b))
                torch.cuda.synchronize(dev)
                ms = statistics.median(a.elapsed_time(b) for a, b in evs[4:])
                res.append({"kind": "gather", "chunk_kb": kb, "inflight_kb_per_wave": infl, "waves": blocks * 4,
                            "inflight_kb_per_cu": infl * blocks * 4 / 256, "bytes": nbytes,
                            "us": ms * 1e3, "gbps": nbytes / (ms * 1e-3) / 1e9})
                print(json.dumps(res[-1]), file=sys.stderr, flush=True)
    with open(out_path("diag.json"), "w") as f:
        json.dump(res, f, indent=1)


def run_sweep(args, cfg, wl, out, dev):
    res = []
    for vid, name in enumerate(ops.variant_names(), start=1):
        if not name.startswith(f"{bench.KV_PREFIX[args.kv]}d{cfg.head_size}_") or "_gq" in name:
            continue            # (gq kernels need num_heads / num_kv_heads > 1: scripts/gqa_probe.py)
        try:
            kern_ms = bench.kernel_pass(wl, out, args.steps, vid, dev, warm=args.warmup)
        except RuntimeError as e:
            res.append({"variant": vid, "name": name, "error": str(e)})
            continue
        us = statistics.mean(kern_ms) * 1e3
        res.append({"variant": vid, "name": name, "us_mean": us, "us_median": statistics.median(kern_ms) * 1e3,
                    "us_min": min(kern_ms) * 1e3, "gbps": bench.alg_bytes(cfg) / (us * 1e-6) / 1e9})
        print(json.dumps(res[-1]), file=sys.stderr, flush=True)
    with open(out_path(f"sweep_{cfg.name}.json"), "w") as f:
        json.dump({"config": cfg.name, "kv": args.kv,
                   "picked": ops.pick_variant(cfg.batch, cfg.num_heads, cfg.head_size, cfg.seq_len,
                                              fp8=bench.FP8_ARG[args.kv]), "results": res}, f, indent=1)


def main():
    ap = bench.build_parser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--diag", action="store_true")
    ap.add_argument("--matrix", action="store_true")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = CONFIGS[args.config]
    if args.batch or args.seq_len:
        b_, l_ = args.batch or cfg.batch, args.seq_len or cfg.seq_len
        cfg = dataclasses.replace(cfg, name=f"{cfg.name}_b{b_}_l{l_}", batch=b_, seq_len=l_,
                                  num_blocks=max(2 * b_ * (-(-l_ // cfg.block_size)), 64))
    if args.matrix:
        return run_matrix(args, cfg, dev)
    wl = make_workload(cfg, dev, seed=1234, table_sets=2, ragged=args.ragged)
    out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    if args.kv.startswith("fp8"):
        bench.KV_DTYPE = args.kv
        gk = torch.Generator(device=dev).manual_seed(99)
        wl.key_cache = bench.random_fp8_codes((cfg.num_blocks, cfg.kv_heads, cfg.head_size // 16, cfg.block_size, 16), dev, gk)
        wl.value_cache = bench.random_fp8_codes((cfg.num_blocks, cfg.kv_heads, cfg.head_size, cfg.block_size), dev, gk)
    if args.diag:
        return run_diag(wl, dev)
    if args.sweep:
        return run_sweep(args, cfg, wl, out, dev)
    raise SystemExit("one of --sweep / --diag / --matrix")


if __name__ == "__main__":
    main()
