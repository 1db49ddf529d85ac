#!/usr/bin/env python3
"""bench.py — decode-step throughput of the MI355X paged-attention hot path.

    python bench.py --gpus N --steps K --warmup W            (N=1: plain python;
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   for N>1)

One "step" = one decode step of ONE transformer layer's attention over one batch of synthetic
input, i.e. the reference's per-layer call pair (vllmini/model/gpt2.py:44 then :62):
    cache_ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, "auto", 1.0)
    paged_attention_v1(out, query, key_cache, value_cache, ...)
through the drop-in Python surface -> C-ABI -> HIP kernels.  Inputs are resident in HBM before
the timed region.  Default workload = BASELINE.json configs[2] ("cfg3": GPT-2 small heads,
batch 256, seq 1024, block 16, num_blocks 32768), the configuration the metric is quoted on.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      achieved = algorithmic bytes of paged_attention_v1 / mean kernel
                duration from HIP events recorded around the attention launch on every n-th step of the
                timed region (--event-stride, default: every 4th, at least 5 samples), on the launch
                stream: an event pair costs the stream 6.5 us (profiles/r02k_event_cost.md), so probing
                every step would put 5 % of instrumentation into `value`; peak = 8000 GB/s (HBM3E spec)
  cpu_baseline  the reference's CPU fallback — PyTorch eager attention over the gathered pages
                (oracle/eager.py, restating vllmini/model/gpt2.py:71-78) — timed on this box's host
                cores on the same synthetic workload (rank 0, N=1 only)

Beside `value` (never as it): `fused_step` (the pair as one launch), `fp8_kv_step` (fp8 E4M3 pages), `ragged_step`
(seq_lens ~ U{1..seq_len} through the same default entry: the shape a continuous-batching server produces),
`graph_step` (the call pair replayed from one hipGraph: what is left when the host is out of the way).

Multi-GPU: sequences are sharded over ranks as independent KV pools, no collective on the data
path.  --scaling weak (default): 256 sequences per GPU, pool of 65536 blocks (BASELINE configs[4]); --scaling strong:
2048 sequences in all, 2048/N per GPU, pool = max(65536, what the batch needs).  For N > 1 every timed step ends
with the one exchange a decode loop has — the all_gather of the step's sampled token ids (8 B per sequence, RCCL
over xGMI; vllmini_amd/shard.py:gather_token_ids) — and its share of the step is reported as `token_exchange_us`.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from vllmini_amd import cache_ops, ops, shard  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="override the config's batch (diagnostic sweeps)")
    ap.add_argument("--pv-mfma", action="store_true",
                    help="with --kv-heads: opt into the grouped-query kernels that run P.V on the matrix cores too "
                         "(ops.set_pv_mfma; north-star 1e-3 instead of 1-2 ulp)")
    ap.add_argument("--kv-heads", type=int, default=0,
                    help="grouped-query attention: num_kv_heads < num_heads (diagnostic; BASELINE configs are multi-head)")
    ap.add_argument("--seq-len", type=int, default=0, help="override the config's seq_len (diagnostic sweeps)")
    ap.add_argument("--op", default="v1", choices=["v1", "v2", "fused"],
                    help="attention operator in the step: paged_attention_v1 (headline) or the split-KV paged_attention_v2")
    ap.add_argument("--variant", type=int, default=0, help="force a kernel work decomposition (0 = heuristic)")
    ap.add_argument("--variant-name", default="", help="the same by kernel name (ops.variant_names())")
    ap.add_argument("--sweep", action="store_true", help="time every kernel variant, write gpurun_out/sweep.json")
    ap.add_argument("--diag", action="store_true",
                    help="report the plain 16-B/lane read bandwidth of this box over the K pool (stderr + gpurun_out/diag.json)")
    ap.add_argument("--e2e", action="store_true",
                    help="end-to-end GPT-2 small decode (12 layers, random weights) on the batched harness: "
                         "extra JSON line on stderr + gpurun_out/e2e.json")
    ap.add_argument("--e2e-fused", action="store_true", help="e2e with one fused append+attention launch per layer")
    ap.add_argument("--e2e-context", type=int, default=1008, help="context length the e2e sequences start at")
    ap.add_argument("--e2e-ragged", action="store_true", help="e2e with contexts ~ U{16..e2e-context} instead of equal ones")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="skip the extra fused-step measurement")
    ap.add_argument("--no-fp8", action="store_true", help="skip the extra fp8-KV-cache measurement")
    ap.add_argument("--noop-instead-of-reshape", action="store_true",
                    help="diagnostic: a 4-byte fill kernel takes reshape_and_cache's place in the step")
    ap.add_argument("--reshape-other-set", action="store_true",
                    help="diagnostic: reshape_and_cache writes the OTHER table set's blocks (no freshly written lines are read)")
    ap.add_argument("--skip-reshape", action="store_true",
                    help="DIAGNOSTIC (invalid as a bench line): attention launches back to back, no reshape_and_cache")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--sequential-tables", action="store_true",
                    help="physically sequential pages instead of a random permutation (diagnostic)")
    ap.add_argument("--ragged", action="store_true", help="seq_lens ~ U{1..L} (diagnostic)")
    ap.add_argument("--ragged-sorted", action="store_true", help="same lengths, longest sequence first (diagnostic)")
    ap.add_argument("--kv", default="auto", choices=["auto", "fp8", "fp8_e5m2"],
                    help="KV cache element type: auto = fp16 (the BASELINE metric); fp8 = E4M3 bytes (diagnostic line, "
                         "SURVEY row f-4)")
    ap.add_argument("--matrix", action="store_true",
                    help="attention kernel time for every (head size, block size) of the reference's dispatch set at "
                         "this config's batch/heads/seq_len, fp16 and bf16 -> gpurun_out/matrix.json (diagnostic)")
    ap.add_argument("--hint-mean", action="store_true",
                    help="pass the batch's mean length to the heuristic (what a host-side scheduler can do)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = 256 sequences per GPU (default), strong = 2048 sequences in all")
    ap.add_argument("--no-ragged", action="store_true", help="skip the extra ragged-batch measurement")
    ap.add_argument("--no-graph", action="store_true", help="skip the extra hipGraph-replay measurement")
    ap.add_argument("--e2e-eager", action="store_true", help="e2e: plain launches instead of hipGraph replay")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="record the HIP event pair around the attention launch on every n-th timed step (1 = every "
                         "step; the records are measurement and cost the stream 6.5 us a pair); lowered so that at "
                         "least 5 launches are probed")
    return ap.parse_args()


def init_dist(n_gpus: int):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if n_gpus > 1 or world > 1 or os.environ.get("VMI_FORCE_DIST") == "1":   # VMI_FORCE_DIST: 1-GPU smoke of the RCCL path
        import torch.distributed as dist

        if world != n_gpus:
            raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                             f"--nproc-per-node {n_gpus}")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        # RCCL prints a banner on STDOUT — when the communicator is created and again, with the lazily created
        # communicator of the first all_gather, later on.  stdout must carry exactly ONE JSON line, so fd 1 points at
        # stderr for the whole run and the line goes to the saved descriptor (emit_line).
        global _REAL_STDOUT
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", rank=rank, world_size=world,  # "nccl" IS RCCL on ROCm
                                device_id=torch.device("cuda", local_rank))
        warm = torch.zeros(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        return dist, rank, world, local_rank
    torch.cuda.set_device(0)
    return None, 0, 1, 0


_REAL_STDOUT = None


def emit_line(line: dict) -> None:
    text = json.dumps(line) + "\n"
    if _REAL_STDOUT is None:
        sys.stdout.write(text)
        sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, text.encode())


_V2_SCRATCH = {}
SKIP_RESHAPE = False
KV_DTYPE = "auto"      # "fp8" / "fp8_e5m2": byte caches in the x = 16 layout (--kv)
KV_PREFIX = {"auto": "", "fp8": "fp8_", "fp8_e5m2": "fp8e5m2_"}          # variant-name prefixes
FP8_ARG = {"auto": False, "fp8": True, "fp8_e5m2": "e5m2"}               # ops.pick_variant(fp8=...)
KV_LABEL = {"auto": "fp16", "fp8": "fp8 E4M3", "fp8_e5m2": "fp8 E5M2"}


def alg_bytes(cfg):
    """Algorithmic bytes per attention launch; an fp8 cache halves the K/V term."""
    b = cfg.algorithmic_bytes()
    if KV_DTYPE.startswith("fp8"):
        b -= 2 * cfg.batch * cfg.kv_heads * cfg.seq_len * cfg.head_size
    return b


def attend(wl, out, t, variant, op="v1"):
    c = wl.cfg
    if op == "fused":   # reshape_and_cache + paged_attention_v1 in one launch (extension, include/vmi_paged_attention.h)
        ops.paged_attention_v1_append(out, wl.query, wl.key, wl.value, wl.key_cache, wl.value_cache, c.kv_heads,
                                      wl.scale, wl.tables[t], wl.seq_lens, c.block_size, c.seq_len, _variant=variant)
        return
    if op == "v1":
        ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, c.kv_heads, wl.scale,
                               wl.tables[t], wl.seq_lens, c.block_size, c.seq_len, None, KV_DTYPE, 1.0,
                               0, 0, 1, 1, 0, _variant=variant)
        return
    key = (c.name, out.device)
    if key not in _V2_SCRATCH:      # caller-owned scratch, as in the reference's v2 signature
        P = (c.seq_len + 511) // 512
        _V2_SCRATCH[key] = (torch.empty((c.batch, c.num_heads, P), dtype=torch.float32, device=out.device),
                            torch.empty((c.batch, c.num_heads, P), dtype=torch.float32, device=out.device),
                            torch.empty((c.batch, c.num_heads, P, c.head_size), dtype=torch.float16, device=out.device))
    es, ml, tmp = _V2_SCRATCH[key]
    ops.paged_attention_v2(out, es, ml, tmp, wl.query, wl.key_cache, wl.value_cache, c.kv_heads, wl.scale,
                           wl.tables[t], wl.seq_lens, c.block_size, c.seq_len, None, "auto", 1.0,
                           0, 0, 1, 1, 0, _variant=variant)


RESHAPE_OTHER = False
NOOP_RESHAPE = None


def one_step(wl, out, i, variant, op="v1"):
    """The reference's per-layer decode call pair, in its call order (gpt2.py:44, :62)."""
    t = i % len(wl.tables)
    if NOOP_RESHAPE is not None:
        NOOP_RESHAPE.fill_(1.0)
    elif op != "fused" and not SKIP_RESHAPE:
        cache_ops.reshape_and_cache(wl.key, wl.value, wl.key_cache, wl.value_cache,
                                    wl.slots[(t + 1) % len(wl.tables) if RESHAPE_OTHER else t], KV_DTYPE, 1.0)
    attend(wl, out, t, variant, op)


_EXCHANGE = {}     # per-run state of the N > 1 token exchange: ids tensor, global batch, per-step event pairs


def exchange_tokens(dist, i=None):
    """The decode loop's only collective: every rank hands the ids it sampled for its sequences to
    all ranks.  Synthetic ids here; the all_gather is the real one."""
    if dist is None:
        return
    ev = _EXCHANGE.get("events")
    if ev is not None and i is not None:
        ev[i][0].record()
    _EXCHANGE["gathered"] = shard.gather_token_ids(_EXCHANGE["ids"], _EXCHANGE["global_batch"], dist,
                                                   out=_EXCHANGE["out"])
    if ev is not None and i is not None:
        ev[i][1].record()


EVENT_STRIDE = 1


def time_steps(wl, out, steps, warmup, variant, dist, dev, op="v1"):
    """W untimed steps, then EXACTLY K timed steps between barrier+synchronize pairs
    (vllmini_amd/shard.py:timed_steps — the same code the 2-rank gloo test exercises)."""
    c = wl.cfg
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if dist is not None:
        world, rank = dist.get_world_size(), dist.get_rank()
        _EXCHANGE["global_batch"] = c.batch * world
        _EXCHANGE["ids"] = torch.arange(rank * c.batch, (rank + 1) * c.batch, dtype=torch.int64, device=dev)
        _EXCHANGE["out"] = torch.empty(c.batch * world, dtype=torch.int64, device=dev)
        _EXCHANGE["events"] = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                               for _ in range(steps)]

    def step(i):
        one_step(wl, out, i, variant, op)
        exchange_tokens(dist)

    def timed(i):
        t = i % len(wl.tables)
        if NOOP_RESHAPE is not None:
            NOOP_RESHAPE.fill_(1.0)
        elif op != "fused" and not SKIP_RESHAPE:
            cache_ops.reshape_and_cache(wl.key, wl.value, wl.key_cache, wl.value_cache,
                                        wl.slots[(t + 1) % len(wl.tables) if RESHAPE_OTHER else t], KV_DTYPE, 1.0)
        probe = i % EVENT_STRIDE == 0
        if probe:
            ev[i][0].record()  # HIP events on the launch stream (torch's current stream)
        attend(wl, out, t, variant, op)
        if probe:
            ev[i][1].record()
        exchange_tokens(dist, i)

    elapsed = shard.timed_steps(step, steps, warmup, dist,
                                sync=lambda: torch.cuda.synchronize(dev), timed_step=timed)
    kern_ms = [a.elapsed_time(b) for i, (a, b) in enumerate(ev) if i % EVENT_STRIDE == 0]
    if dist is not None:
        _EXCHANGE["us"] = statistics.mean(a.elapsed_time(b) for a, b in _EXCHANGE["events"]) * 1e3
        g = _EXCHANGE["gathered"]
        assert g.numel() == _EXCHANGE["global_batch"] and int(g[0]) == 0 and int(g[-1]) == g.numel() - 1
        _EXCHANGE["events"] = None
    return elapsed, kern_ms


def graph_steps(wl, out, steps, variant, dev, per_graph=1):
    """The reference's call pair captured ONCE into a hipGraph (one graph per table set) and replayed: the launch
    work the host does per step is one graph launch, so this is the step the GPU can do when the host is out of
    the way.  Returns seconds for `steps` replays."""
    graphs = []
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for t in range(len(wl.tables)):
            one_step(wl, out, t, variant)
    torch.cuda.current_stream(dev).wait_stream(side)
    if per_graph <= 1:
        for t in range(len(wl.tables)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one_step(wl, out, t, variant)
            graphs.append(g)
        per_graph = 1
    else:   # `per_graph` consecutive steps (table sets in the loop's order) in ONE graph
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(per_graph):
                one_step(wl, out, t, variant)
        graphs.append(g)
    replays = -(-steps // per_graph)
    for i in range(max(2, 10 // per_graph)):
        graphs[i % len(graphs)].replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(replays):
        graphs[i % len(graphs)].replay()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * steps / (replays * per_graph)


def pmc_traffic(cfg_name: str, kernel_variant: str):
    """HBM bytes per launch of the attention kernel from the rocprofv3 PMC passes committed under
    profiles/ (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command; FETCH_SIZE x1024 x2 per the
    gfx950 correction in MI355X_MICROARCH.md §HBM).  bench.py cannot run a profiler around itself, so
    the figure comes from the latest recorded pass for this workload and kernel variant, else None."""
    cfg_name = cfg_name.replace("cfg5", "cfg3")   # cfg5 = the cfg3 launch over a larger pool (N > 1 runs)
    path = os.path.join(REPO, "profiles", f"pmc_{cfg_name}_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel_variant") and d["kernel_variant"] != kernel_variant:
            return None, None
        return d["traffic_bytes_corrected"]["total"], os.path.relpath(path, REPO)
    except (OSError, KeyError, ValueError):
        return None, None


def cpu_baseline(wl, steps):
    """Reference CPU fallback: eager attention (gather by block_tables -> matmul/softmax/matmul)."""
    from oracle.eager import torch_eager_decode  # checker/baseline only; never on the product path

    c = wl.cfg
    torch.set_num_threads(os.cpu_count() or 1)
    kc, vc = wl.key_cache.cpu(), wl.value_cache.cpu()
    q = wl.query.cpu().contiguous()
    tab = wl.tables[0].cpu()
    # bound the sample to ~10-30 s of CPU work: time one step, then decide how many to run
    t0 = time.perf_counter()
    torch_eager_decode(q, kc, vc, wl.scale, tab, c.seq_len, dtype=torch.float32)
    first = time.perf_counter() - t0
    n = max(1, min(steps, int(20.0 / max(first, 1e-3))))
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        torch_eager_decode(q, kc, vc, wl.scale, tab, c.seq_len, dtype=torch.float32)
        ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    return {
        "value": c.batch / med, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
        "ms_per_step": med * 1e3,
        "sample": f"full {c.name} batch ({c.batch} seqs x {c.seq_len} tokens, H{c.num_heads} D{c.head_size}), "
                  f"fp32 torch eager incl. page gather, median of {n} steps after 1 warm-up, "
                  f"{torch.get_num_threads()} threads",
    }


def run_e2e(args, dist, rank, world, local_rank, dev):
    """GPT-2 small, `batch` sequences per GPU at ~seq_len context, one token per sequence per step,
    through vllmini_amd.gpt2_decode (hipGraph replay of the whole step).  KV is synthetic: pages are
    filled with random fp16 and sequences are registered at the target context length."""
    import numpy as np

    from vllmini_amd.gpt2_decode import GPT2Dims, GPT2PagedDecoder, random_state_dict
    from vllmini_amd.kv_pool import PagedKVPool

    cfg = CONFIGS[args.config]
    # GPT-2 small with the position table extended past 1024 so contexts can cross seq_len 1024
    dims = GPT2Dims(n_positions=2048)
    assert (cfg.num_heads, cfg.head_size) == (dims.n_head, dims.head_size)
    ctx0 = args.e2e_context
    total_steps = args.warmup + args.steps + 2
    mb = -(-(ctx0 + total_steps) // cfg.block_size) + 1
    blocks_needed = cfg.batch * dims.n_layer * (mb - 1)
    pool = PagedKVPool(blocks_needed + 64, dims.n_head, dims.head_size, cfg.block_size, mb, dims.n_layer, device=dev,
                       max_seqs=cfg.batch, multi_block_prefill=True, kv_cache_dtype=args.kv)
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    if args.kv.startswith("fp8"):      # random E4M3 / E5M2 codes of magnitude < 2 (codes 0..63 in either format)
        for c in (pool.key_cache, pool.value_cache):
            c.copy_(torch.randint(0, 64, c.shape, dtype=torch.uint8, device=dev, generator=g)
                    | (torch.randint(0, 2, c.shape, dtype=torch.uint8, device=dev, generator=g) << 7))
    else:
        pool.key_cache.uniform_(-1, 1, generator=g)
        pool.value_cache.uniform_(-1, 1, generator=g)
    # shuffle the free list so pages are scattered like a long-running pool's
    perm = np.random.default_rng(rank).permutation(pool.num_blocks)
    pool.free_blocks = perm.tolist()
    ctxs = np.random.default_rng(100 + rank).integers(16, ctx0 + 1, cfg.batch) if args.e2e_ragged else [ctx0] * cfg.batch
    for s in range(cfg.batch):
        pool.allocate_for_prefill(s, int(ctxs[s]))   # bookkeeping only: the pages already hold synthetic KV
    dec = GPT2PagedDecoder(dims, random_state_dict(dims, dev, seed=rank), pool, fused_append=args.e2e_fused)
    ids = list(range(cfg.batch))
    tok = torch.randint(0, dims.vocab_size, (cfg.batch,), device=dev, generator=g)

    def step(i):
        nonlocal tok
        logits = dec.decode(ids, tok, use_graph=not args.e2e_eager)
        tok = logits.argmax(-1)                       # greedy: stays on the device, no host sync

    elapsed = shard.timed_steps(step, args.steps, args.warmup, dist, sync=lambda: torch.cuda.synchronize(dev))
    elapsed = shard.max_over_ranks(elapsed, dist, dev)
    res = {"metric": "gpt2_small_decode_tokens_per_sec_end_to_end", "value": cfg.batch * world * args.steps / elapsed,
           "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3,
           "context": f"U{{16..{ctx0}}} (mean {float(np.mean(ctxs)):.0f})" if args.e2e_ragged else ctx0, "batch_per_gpu": cfg.batch,
           "data": "synthetic KV + random-init GPT-2 small weights", "dtype": "f16",
           "note": ("12 x (c_attn, paged_attention_v1_append [fused], c_proj, MLP) + lm_head, hipGraph replay, greedy"
                    if args.e2e_fused else
                    "12 x (c_attn, reshape_and_cache, paged_attention_v1, c_proj, MLP) + lm_head, hipGraph replay, greedy")
                   .replace("hipGraph replay", "plain launches" if args.e2e_eager else "hipGraph replay")}
    if rank == 0:
        print(json.dumps(res), file=sys.stderr, flush=True)
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        res["kv_cache_dtype"] = args.kv
        with open(os.path.join(REPO, "gpurun_out", "e2e_ragged.json" if args.e2e_ragged else "e2e_fused.json" if args.e2e_fused else
                               ("e2e_fp8.json" if args.kv == "fp8" else ("e2e_fp8_e5m2.json" if args.kv == "fp8_e5m2" else "e2e.json"))), "w") as f:
            json.dump(res, f, indent=1)


def run_matrix(args, base, dev):
    """Every (head size, block size, element type / cache type) the operators are built for, at base's
    batch/heads/seq_len: fp16, bf16, float32, and fp16 query over fp8 E4M3 / E5M2 caches."""
    import dataclasses

    global KV_DTYPE
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
                KV_DTYPE = "auto"
                if kind == "bfloat16":
                    wl.key_cache, wl.value_cache, wl.qkv = (wl.key_cache.to(dt), wl.value_cache.to(dt), wl.qkv.to(dt))
                elif kind == "float32":          # x = 4 layout: same bytes per chunk, half the elements
                    wl.key_cache = torch.empty((c.num_blocks, c.num_heads, D // 4, bs, 4), dtype=dt, device=dev).uniform_(-1, 1)
                    wl.value_cache = torch.empty((c.num_blocks, c.num_heads, D, bs), dtype=dt, device=dev).uniform_(-1, 1)
                    wl.qkv = wl.qkv.to(dt)
                elif kind.startswith("fp8"):
                    KV_DTYPE = "fp8" if kind == "fp8_kv" else "fp8_e5m2"
                    gk = torch.Generator(device=dev).manual_seed(3)
                    ks = (c.num_blocks, c.num_heads, D // 16, bs, 16)
                    vs = (c.num_blocks, c.num_heads, D, bs)
                    wl.key_cache = torch.randint(0, 64, ks, dtype=torch.uint8, device=dev, generator=gk)
                    wl.value_cache = torch.randint(0, 64, vs, dtype=torch.uint8, device=dev, generator=gk)
                out = torch.empty((c.batch, c.num_heads, D), dtype=dt, device=dev)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                for i in range(args.warmup + args.steps):
                    t = i % len(wl.tables)
                    k = i - args.warmup
                    if k >= 0:
                        ev[k][0].record()
                    attend(wl, out, t, 0)
                    if k >= 0:
                        ev[k][1].record()
                torch.cuda.synchronize(dev)
                us = statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3
                vid = ops.last_variant()      # what the launches really ran (0 for the float32 kernels)
                nbytes = alg_bytes(c)
                if kind == "float32":            # K and V bytes double
                    nbytes += 2 * c.batch * c.kv_heads * c.seq_len * c.head_size * 2
                row = {"dtype": kind, "head_size": D, "block_size": bs, "us_median": us,
                       "gbps": nbytes / (us * 1e-6) / 1e9,
                       "variant": "pa_v1_f32_kernel" if kind == "float32" or not vid else ops.variant_names()[vid - 1]}
                res.append(row)
                print(json.dumps(row), file=sys.stderr, flush=True)
                del wl, out
                torch.cuda.empty_cache()
    KV_DTYPE = "auto"
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "matrix.json"), "w") as f:
        json.dump({"batch": base.batch, "num_heads": base.num_heads, "seq_len": base.seq_len, "rows": res}, f, indent=1)


def main():
    global KV_DTYPE, SKIP_RESHAPE, RESHAPE_OTHER, NOOP_RESHAPE, EVENT_STRIDE
    args = parse_args()
    RESHAPE_OTHER = args.reshape_other_set
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path exists for the product kernels)")
    dist, rank, world, local_rank = init_dist(args.gpus)
    dev = torch.device("cuda", local_rank)
    if args.noop_instead_of_reshape:
        NOOP_RESHAPE = torch.zeros(1, device=dev)
    cfg = CONFIGS[args.config]
    if (world > 1 or dist is not None) and args.config == "cfg3":
        # BASELINE.json configs[4]: batch 2048 over 8 GPUs = 256 sequences per GPU (the cfg3 shape) with a
        # per-GPU KV pool of 65536 blocks.  Same kernel work per GPU as N=1; only the pool is larger.
        cfg = CONFIGS["cfg5"]
        if args.scaling == "strong":
            # total batch 2048 whatever N; N = 1 needs 2 x 131072 blocks for two disjoint table sets,
            # more than the stated 65536 -> the pool is max(65536, needed)
            import dataclasses
            total = 2048
            if total % world:
                raise SystemExit(f"--scaling strong: {total} sequences do not divide over {world} GPUs")
            b_ = total // world
            cfg = dataclasses.replace(cfg, name="cfg5_strong", batch=b_,
                                      num_blocks=max(cfg.num_blocks, 2 * b_ * cfg.blocks_per_seq))
    if args.variant_name:
        args.variant = ops.variant_names().index(args.variant_name) + 1
    if args.pv_mfma:
        ops.set_pv_mfma(True)
    if args.kv_heads:
        import dataclasses
        args.no_cpu_baseline = True      # the eager CPU baseline is written for the multi-head BASELINE configs
        cfg = dataclasses.replace(cfg, name=f"{cfg.name}_kv{args.kv_heads}", num_kv_heads=args.kv_heads)
    if args.batch or args.seq_len:
        import dataclasses
        b_ = args.batch or cfg.batch
        l_ = args.seq_len or cfg.seq_len
        per = -(-l_ // cfg.block_size)
        cfg = dataclasses.replace(cfg, name=f"{cfg.name}_b{b_}_l{l_}", batch=b_, seq_len=l_,
                                  num_blocks=max(2 * b_ * per, 64))
    if args.matrix:
        run_matrix(args, cfg, dev)
        return
    if args.e2e:
        run_e2e(args, dist, rank, world, local_rank, dev)
        if dist is not None:
            dist.destroy_process_group()
        return
    wl = make_workload(cfg, dev, seed=1234 + rank, table_sets=2, ragged=("sorted" if args.ragged_sorted else args.ragged))
    if args.sequential_tables:
        for t, tab in enumerate(wl.tables):
            per = cfg.num_blocks // len(wl.tables)
            seq = torch.arange(cfg.batch * cfg.blocks_per_seq, dtype=torch.int32, device=dev) + t * per
            tab[:, : cfg.blocks_per_seq] = seq.view(cfg.batch, cfg.blocks_per_seq)
    out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    if args.kv.startswith("fp8"):
        KV_DTYPE = args.kv
        if args.op != "v1":
            raise SystemExit("--kv fp8 is built for --op v1")
        args.no_fused = True
        args.no_cpu_baseline = True
        gk = torch.Generator(device=dev).manual_seed(99 + rank)
        kshape = (cfg.num_blocks, cfg.kv_heads, cfg.head_size // 16, cfg.block_size, 16)
        vshape = (cfg.num_blocks, cfg.kv_heads, cfg.head_size, cfg.block_size)
        del wl.key_cache, wl.value_cache
        torch.cuda.empty_cache()
        # random codes 0..63 + sign: magnitude < 2 in E4M3 (exponent field <= 7) and in E5M2 (<= 15), no NaN codes
        wl.key_cache = (torch.randint(0, 64, kshape, dtype=torch.uint8, device=dev, generator=gk)
                        | (torch.randint(0, 2, kshape, dtype=torch.uint8, device=dev, generator=gk) << 7))
        wl.value_cache = (torch.randint(0, 64, vshape, dtype=torch.uint8, device=dev, generator=gk)
                          | (torch.randint(0, 2, vshape, dtype=torch.uint8, device=dev, generator=gk) << 7))

    if args.diag:
        from vllmini_amd import _lib
        lib = _lib.load()
        sink = torch.zeros(1, dtype=torch.int32, device=dev)
        src = wl.key_cache
        nbytes = src.numel() * 2
        stream = torch.cuda.current_stream(dev).cuda_stream
        res = []
        for nt in (0, 1):
            for blocks in (1024, 2048, 4096, 8192, 16384):
                for _ in range(3):
                    lib.vmi_diag_stream_read(src.data_ptr(), nbytes, sink.data_ptr(), blocks, nt, local_rank, stream)
                evs = []
                for i in range(20):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s_ = wl.value_cache if i % 2 else wl.key_cache   # alternate pools: defeat the 256 MiB MALL
                    a.record()
                    lib.vmi_diag_stream_read(s_.data_ptr(), nbytes, sink.data_ptr(), blocks, nt, local_rank, stream)
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
                        rc = lib.vmi_diag_gather_read(s_.data_ptr(), nbytes, sink.data_ptr(), kb, infl, blocks, 1, local_rank, stream)
                        b.record()
                        assert rc == 0, _lib.last_error()
                        evs.append((a, b))
                    torch.cuda.synchronize(dev)
                    ms = statistics.median(a.elapsed_time(b) for a, b in evs[4:])
                    res.append({"kind": "gather", "chunk_kb": kb, "inflight_kb_per_wave": infl, "waves": blocks * 4,
                                "inflight_kb_per_cu": infl * blocks * 4 / 256, "bytes": nbytes,
                                "us": ms * 1e3, "gbps": nbytes / (ms * 1e-3) / 1e9})
                    print(json.dumps(res[-1]), file=sys.stderr, flush=True)
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "diag.json"), "w") as f:
            json.dump(res, f, indent=1)
        return

    if args.sweep:
        names = ops.variant_names()
        res = []
        for vid, name in enumerate(names, start=1):
            if not name.startswith(f"{KV_PREFIX[args.kv]}d{cfg.head_size}_") or "_gq" in name:
                continue            # (gq kernels need num_heads / num_kv_heads > 1: scripts/gqa_probe.py)
            try:
                _, kern_ms = time_steps(wl, out, args.steps, args.warmup, vid, dist, dev)
            except RuntimeError as e:
                res.append({"variant": vid, "name": name, "error": str(e)})
                continue
            us = statistics.mean(kern_ms) * 1e3
            res.append({"variant": vid, "name": name, "us_mean": us, "us_median": statistics.median(kern_ms) * 1e3,
                        "us_min": min(kern_ms) * 1e3, "gbps": alg_bytes(cfg) / (us * 1e-6) / 1e9})
            if rank == 0:
                print(json.dumps(res[-1]), file=sys.stderr, flush=True)
        if rank == 0:
            os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
            with open(os.path.join(REPO, "gpurun_out", f"sweep_{cfg.name}.json"), "w") as f:
                json.dump({"config": cfg.name, "kv": args.kv,
                           "picked": ops.pick_variant(cfg.batch, cfg.num_heads, cfg.head_size, cfg.seq_len,
                                                      fp8=FP8_ARG[args.kv]), "results": res}, f, indent=1)
        return

    SKIP_RESHAPE = args.skip_reshape
    EVENT_STRIDE = max(1, min(args.event_stride, args.steps // 5))
    if args.hint_mean and not args.variant and args.op in ("v1", "fused"):
        lens_h = wl.seq_lens.cpu()
        args.variant = ops.pick_variant(cfg.batch, cfg.num_heads, cfg.head_size, int(lens_h.max()), cfg.block_size,
                                        mean_seq_len=int(lens_h.float().mean()), fp8=FP8_ARG[args.kv])
    elapsed, kern_ms = time_steps(wl, out, args.steps, args.warmup, args.variant, dist, dev, op=args.op)
    elapsed = shard.max_over_ranks(elapsed, dist, dev)
    kern_mean_ms = shard.max_over_ranks(statistics.mean(kern_ms), dist, dev)

    tokens = cfg.batch * world * args.steps          # one new token per sequence per step
    ms_per_step = elapsed / args.steps * 1e3
    achieved = alg_bytes(cfg) / (kern_mean_ms * 1e-3) / 1e9
    # the variant the library actually launched (it knows the launch's kv_scale, the pick queries do not)
    vid = args.variant or (ops.last_variant() if args.op in ("v1", "fused") else 0)
    vname = ops.variant_names()[vid - 1] if vid else f"paged_attention_v2 variant {args.variant or 'auto'}"
    traffic, traffic_src = (pmc_traffic(cfg.name + {"auto": "", "fp8": "_fp8", "fp8_e5m2": "_fp8_e5m2"}[args.kv], vname) if args.op == "v1" else
                            pmc_traffic(cfg.name + "_fused", vname) if args.op == "fused" else (None, None))
    line = {
        "metric": "decode_tokens_per_sec_paged_attention_v1_per_layer",
        "value": tokens / elapsed,
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling if dist is not None else "weak",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {
            "workload": f"{cfg.name}: paged_attention_v1+reshape_and_cache decode step, batch {cfg.batch}/GPU, "
                        f"seq_len {cfg.seq_len}, {cfg.num_heads} heads"
                        f"{'' if cfg.kv_heads == cfg.num_heads else ' (' + str(cfg.kv_heads) + ' KV heads)'} x {cfg.head_size}, "
                        f"block_size {cfg.block_size}, "
                        f"num_blocks {cfg.num_blocks}/GPU, {KV_LABEL[args.kv]} KV, "
                        f"random-permutation block tables"
                        + (" (SEQUENTIAL tables)" if args.sequential_tables else "")
                        + (" (ragged lens)" if args.ragged else "")
                        + (" (DIAGNOSTIC: reshape_and_cache skipped)" if args.skip_reshape else ""),
            "global_batch": cfg.batch * world,
            "seq_len": cfg.seq_len,
            "parallelism": f"dp{world} (independent KV pools, no data-path collective"
                           + ("; per-step all_gather of the sampled token ids over RCCL)" if dist is not None else ")"),
            "kernel_variant": vname, "op": args.op,
        },
        "paged_attention_v1_us_per_step": kern_mean_ms * 1e3,
        "paged_attention_v1_us_median": statistics.median(kern_ms) * 1e3,
        "paged_attention_v1_us_min": min(kern_ms) * 1e3,
        "kernel_event_stride": EVENT_STRIDE, "kernel_event_samples": len(kern_ms),
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel": "pa_q_kernel" if vname.startswith(("q_d", "bf16_q_d")) else "pa_v1_kernel",
            "algorithmic_bytes_per_launch": alg_bytes(cfg),
        },
    }
    if args.op == "v1" and not args.no_fused:
        # the same step as ONE launch (vmi_paged_attention_v1_append_f16: bit-identical caches and out,
        # tests/test_parity_gpu.py); reported beside `value`, which stays the reference's two-op call pair
        f_elapsed, f_kern = time_steps(wl, out, args.steps, args.warmup, args.variant, dist, dev, op="fused")
        f_elapsed = shard.max_over_ranks(f_elapsed, dist, dev)
        line["fused_step"] = {"op": "paged_attention_v1_append (reshape_and_cache + paged_attention_v1, one launch)",
                              "value": tokens / f_elapsed, "unit": "tokens/s",
                              "ms_per_step": f_elapsed / args.steps * 1e3,
                              "kernel_us_mean": statistics.mean(f_kern) * 1e3}
    if args.op == "v1" and args.kv == "auto" and not args.no_fp8 and not args.ragged and not args.ragged_sorted:
        # the same step over an fp8 E4M3 KV cache (kv_cache_dtype "fp8"): half the K/V bytes.
        # A different data format, so it is reported beside `value`, never as it.
        k16, v16 = wl.key_cache, wl.value_cache
        gk = torch.Generator(device=dev).manual_seed(99 + rank)
        kshape = (cfg.num_blocks, cfg.kv_heads, cfg.head_size // 16, cfg.block_size, 16)
        wl.key_cache = (torch.randint(0, 64, kshape, dtype=torch.uint8, device=dev, generator=gk)
                        | (torch.randint(0, 2, kshape, dtype=torch.uint8, device=dev, generator=gk) << 7))
        wl.value_cache = (torch.randint(0, 64, v16.shape, dtype=torch.uint8, device=dev, generator=gk)
                          | (torch.randint(0, 2, v16.shape, dtype=torch.uint8, device=dev, generator=gk) << 7))
        KV_DTYPE = "fp8"
        try:
            q_elapsed, q_kern = time_steps(wl, out, args.steps, args.warmup, 0, dist, dev, op="v1")
        finally:
            KV_DTYPE = "auto"
            wl.key_cache, wl.value_cache = k16, v16
        q_elapsed = shard.max_over_ranks(q_elapsed, dist, dev)
        q_us = statistics.mean(q_kern) * 1e3
        KV_DTYPE = "fp8"
        q_bytes = alg_bytes(cfg)
        KV_DTYPE = "auto"
        line["fp8_kv_step"] = {"op": "reshape_and_cache + paged_attention_v1, kv_cache_dtype='fp8' (E4M3), kv_scale 1.0",
                               "value": tokens / q_elapsed, "unit": "tokens/s",
                               "ms_per_step": q_elapsed / args.steps * 1e3, "kernel_us_mean": q_us,
                               "achieved_GBps": q_bytes / (q_us * 1e-6) / 1e9,
                               "frac_of_hbm_peak": q_bytes / (q_us * 1e-6) / 1e9 / HBM_PEAK_GBPS}
    if dist is not None:
        line["token_exchange_us"] = shard.max_over_ranks(_EXCHANGE["us"], dist, dev)
        line["token_exchange"] = f"all_gather of {cfg.batch} int64 ids per rank ({cfg.batch * world * 8} B in all), backend nccl (RCCL)"
    if args.op == "v1" and args.kv == "auto" and not args.no_graph and not args.ragged and not args.ragged_sorted and \
            dist is None and not args.skip_reshape:
        g_elapsed = graph_steps(wl, out, args.steps, args.variant, dev)
        line["graph_step"] = {"op": "reshape_and_cache + paged_attention_v1 replayed from one hipGraph per table set",
                              "value": cfg.batch * args.steps / g_elapsed, "unit": "tokens/s",
                              "ms_per_step": g_elapsed / args.steps * 1e3}
        n_sets = len(wl.tables)
        g_elapsed = graph_steps(wl, out, args.steps, args.variant, dev, per_graph=n_sets)
        line["graph_step"]["steps_per_graph"] = {"steps": n_sets, "ms_per_step": g_elapsed / args.steps * 1e3,
                                                 "value": cfg.batch * args.steps / g_elapsed}
    if args.op == "v1" and args.kv == "auto" and not args.no_ragged and not args.ragged and not args.ragged_sorted and \
            not args.variant and not args.sequential_tables:
        # the same call pair, same default entry (no hint, no variant), on a RAGGED batch: seq_lens ~ U{1..seq_len}
        pools = (wl.key_cache, wl.value_cache)
        rwl = make_workload(cfg, dev, seed=4321 + rank, table_sets=2, ragged=True)
        del rwl.key_cache, rwl.value_cache
        rwl.key_cache, rwl.value_cache = pools          # same pools: only tables, lengths and slots differ
        r_elapsed, r_kern = time_steps(rwl, out, args.steps, args.warmup, 0, dist, dev, op="v1")
        r_elapsed = shard.max_over_ranks(r_elapsed, dist, dev)
        r_us = shard.max_over_ranks(statistics.mean(r_kern), dist, dev) * 1e3
        tok = int(rwl.seq_lens.sum().item())
        r_bytes = 2 * tok * cfg.kv_heads * cfg.head_size * 2 + 2 * cfg.batch * cfg.num_heads * cfg.head_size * 2 + \
            int(((rwl.seq_lens + cfg.block_size - 1) // cfg.block_size).sum().item()) * 4 + cfg.batch * 4
        line["ragged_step"] = {"op": "reshape_and_cache + paged_attention_v1, default entry, seq_lens ~ U{1..%d}" % cfg.seq_len,
                               "value": tokens / r_elapsed, "unit": "tokens/s",
                               "ms_per_step": r_elapsed / args.steps * 1e3, "kernel_us_mean": r_us,
                               "algorithmic_bytes_per_launch": r_bytes,
                               "achieved_GBps": r_bytes / (r_us * 1e-6) / 1e9,
                               "frac_of_hbm_peak": r_bytes / (r_us * 1e-6) / 1e9 / HBM_PEAK_GBPS}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(wl, args.cpu_steps)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        emit_line(line)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
