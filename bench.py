#!/usr/bin/env python3
"""bench.py — decode-step throughput of the MI355X paged-attention hot path.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 with no WORLD_SIZE in the environment LAUNCHES ITSELF: this process starts N
ranks of this same file (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_ADDR=127.0.0.1 set), waits for them and
exits with their status; rank 0 prints the line.  Under `python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N ...` the ranks already exist and each simply runs.

One "step" = one decode step of ONE transformer layer's attention over one batch of synthetic
input, i.e. the reference's per-layer call pair (vllmini/model/gpt2.py:44 then :62):
    cache_ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, "auto", 1.0)
    paged_attention_v1(out, query, key_cache, value_cache, ...)
through the drop-in Python surface -> C-ABI -> HIP kernels.  Inputs are resident in HBM before
the timed region.  Default workload = BASELINE.json configs[2] ("cfg3": GPT-2 small heads,
batch 256, seq 1024, block 16, num_blocks 32768), the configuration the metric is quoted on.

Prints ONE JSON line on rank 0 (contract in the task statement).  `value` comes from a timed region that holds
nothing but the K steps (no event records).  Two extra objects:
  roofline      achieved = algorithmic bytes of paged_attention_v1 / MEDIAN kernel duration from a
                separate pass of >= 50 call pairs with a HIP event pair around every attention launch, on the launch
                stream (--kernel-samples; mean and min are on the line too); for N > 1 the slowest rank's median;
                peak = 8000 GB/s (HBM3E spec)
  cpu_baseline  the reference's CPU fallback — PyTorch eager attention over the gathered pages
                (oracle/eager.py, restating vllmini/model/gpt2.py:71-78) — timed on this box's host
                cores on the same synthetic workload (rank 0, after the GPU work), at several thread counts;
                the best one is `value`, all of them are listed

Beside `value` (never as it), each a sub-record of the same line:
  fused_step    the call pair as one launch (extension)
  fp8_kv_step   the call pair over fp8 E4M3 pages
  ragged_step   seq_lens ~ U{1..seq_len} through the same default entry: what a continuous-batching server produces
  graph_step    the call pair replayed from one hipGraph (N = 1); steps_per_graph: 12 pairs (a token's 12 layers) per graph
  cfg4_step     the call pair on BASELINE configs[3] (32 heads x 128, batch 128, seq 2048): ms/step, kernel us, frac
  e2e_step      GPT-2 small end to end (12 layers + lm_head, random weights, batch 256/GPU, context ~1008) through
                vllmini_amd.gpt2_decode + kv_pool: decode tokens/s — BASELINE's first metric — and the share of the
                step spent in the two operators

  deferred_scatter_step   a TOKEN of a 12-layer model on the cfg3 shape (12 disjoint table sets): 12 call pairs against 12 append-read
                attention launches + ONE reshape_and_cache for the 12 layers' rows, per layer step (round 6, N = 1)
  serve_step    the continuous-batching scheduler itself (vllmini_amd/scheduler.py over GPT2PagedDecoder): a seeded closed-loop
                trace of ragged requests, batched admission, refills, preemption by swap — generated tokens/s, per-token latency,
                occupancy, host us per step beside the GPU wait, swap traffic (`--serve` alone runs the 2048-request trace)

The headline's timed region holds EXACTLY K steps; a region shorter than --min-timed-ms (50) is repeated and the line reports the
MEDIAN region with every region listed (`ms_per_step_regions`).  The e2e records report the median of three regions.

Multi-GPU: sequences are sharded over ranks as independent KV pools, no collective on the data
path.  --scaling weak (default): 256 sequences per GPU, pool of 65536 blocks (BASELINE configs[4]); --scaling strong:
2048 sequences in all, 2048/N per GPU, pool = max(65536, what the batch needs).  For N > 1 the timed region also holds
the one exchange a decode loop has — the all_gather of the sampled token ids (8 B per sequence, RCCL over xGMI;
vllmini_amd/shard.py:gather_token_ids) — once per TOKEN, i.e. on every 12th layer step (GPT-2 small has 12 layers and a
step here is one layer's call pair); its duration alone is reported as `token_exchange_us`.

Diagnostic modes (--sweep, --diag, --matrix) live in scripts/bench_diag.py.
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from vllmini_amd import shard  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_parser():
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
    ap.add_argument("--op", default="v1", choices=["v1", "v2", "fused", "newest"],
                    help="attention operator in the step: paged_attention_v1 (headline) or the split-KV paged_attention_v2")
    ap.add_argument("--variant", type=int, default=0, help="force a kernel work decomposition (0 = heuristic)")
    ap.add_argument("--variant-name", default="", help="the same by kernel name (ops.variant_names())")
    ap.add_argument("--e2e", action="store_true",
                    help="ONLY the end-to-end GPT-2 small decode (12 layers, random weights) on the batched harness: "
                         "JSON on stderr + gpurun_out/e2e*.json")
    ap.add_argument("--e2e-fused", action="store_true", help="e2e with one fused append+attention launch per layer")
    ap.add_argument("--e2e-deferred", action="store_true", help="e2e with append-read attention and ONE scatter per token")
    ap.add_argument("--e2e-context", type=int, default=1008, help="context length the e2e sequences start at")
    ap.add_argument("--e2e-ragged", action="store_true", help="e2e with contexts ~ U{16..e2e-context} instead of equal ones")
    ap.add_argument("--e2e-eager", action="store_true", help="e2e: plain launches instead of hipGraph replay")
    ap.add_argument("--e2e-scatter-in-c-attn", action="store_true",
                    help="e2e: the q/k/v projection writes k and v into the paged cache itself (no reshape_and_cache launch)")
    ap.add_argument("--e2e-sampler", default="greedy", choices=("greedy", "top_k", "top_k_torch"),
                    help="e2e: how the next token is chosen (top_k = scheduler.py:144-153 in one launch; top_k_torch = the torch chain)")
    ap.add_argument("--e2e-torch-layers", action="store_true",
                    help="e2e: the block's linear layers as torch modules instead of csrc/gpt2_layer.hip")
    ap.add_argument("--serve", action="store_true",
                    help="the serving loop alone: BatchScheduler over GPT2PagedDecoder with a seeded request trace (ragged "
                         "contexts, refills, preemption by swap); prints its record to stderr, gpurun_out/serve.json")
    ap.add_argument("--serve-requests", type=int, default=2048)
    ap.add_argument("--serve-max-batch", type=int, default=256)
    ap.add_argument("--serve-pool-blocks", type=int, default=0, help="KV pool size in blocks (0: sized so that preemption happens)")
    ap.add_argument("--serve-max-prompt", type=int, default=512)
    ap.add_argument("--serve-mean-new", type=int, default=128, help="mean of the geometric output lengths")
    ap.add_argument("--serve-sampler", default="top_k", choices=("top_k", "greedy"))
    ap.add_argument("--serve-admit-every", type=int, default=16, help="decode steps between admissions of queued requests (one prefill call each)")
    ap.add_argument("--serve-prefill-tokens", type=int, default=16384, help="prompt tokens per prefill call, at most")
    ap.add_argument("--serve-headroom", type=int, default=0,
                    help="blocks kept free per running sequence at admission (0: a prompt is admitted whenever it fits; growth preempts)")
    ap.add_argument("--serve-pool-frac", type=float, default=0.7, help="pool = this share of what max_batch mid-life sequences hold")
    ap.add_argument("--serve-rate", type=float, default=0.0,
                    help="open loop: Poisson arrivals at this many requests/s (0 = closed loop, every request queued at t = 0)")
    ap.add_argument("--serve-preempt", default="swap", choices=("swap", "drop"))
    ap.add_argument("--serve-kv", default="auto", choices=("auto", "fp8"), help="KV pages of the serving run (fp8 = E4M3: the same pool bytes hold twice the tokens)")
    ap.add_argument("--serve-no-deferred-scatter", action="store_true", help="the reference's call pair per layer instead")
    ap.add_argument("--serve-eager", action="store_true", help="plain launches instead of hipGraph replay")
    ap.add_argument("--no-serve", action="store_true", help="skip the extra serving-loop measurement")
    ap.add_argument("--min-timed-ms", type=float, default=50.0,
                    help="repeat the K-step timed region until the regions hold this many ms in all; the line reports the MEDIAN region")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline call pair: no sub-records, no CPU baseline (what the rocprofv3 passes run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="skip the extra fused-step measurement")
    ap.add_argument("--no-fp8", action="store_true", help="skip the extra fp8-KV-cache measurement")
    ap.add_argument("--no-ragged", action="store_true", help="skip the extra ragged-batch measurement")
    ap.add_argument("--no-graph", action="store_true", help="skip the extra hipGraph-replay measurement")
    ap.add_argument("--no-cfg4", action="store_true", help="skip the extra BASELINE configs[3] measurement")
    ap.add_argument("--no-cfg2", action="store_true", help="skip the extra BASELINE configs[1] measurement")
    ap.add_argument("--no-deferred", action="store_true", help="skip the extra 12-layer token measurement (call pairs against deferred scatter)")
    ap.add_argument("--no-strong", action="store_true", help="skip the N = 1 anchor of the strong-scaling curve (batch 2048 on one GPU)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the extra end-to-end GPT-2 measurement")
    ap.add_argument("--no-long", action="store_true", help="skip the extra few-sequences-x-long-context measurement (workspace on / off)")
    ap.add_argument("--noop-instead-of-reshape", action="store_true",
                    help="diagnostic: a 4-byte fill kernel takes reshape_and_cache's place in the step")
    ap.add_argument("--reshape-other-set", action="store_true",
                    help="diagnostic: reshape_and_cache writes the OTHER table set's blocks (no freshly written lines are read)")
    ap.add_argument("--skip-reshape", action="store_true",
                    help="DIAGNOSTIC (invalid as a bench line): attention launches back to back, no reshape_and_cache")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU baseline's thread-count sweep")
    ap.add_argument("--sequential-tables", action="store_true",
                    help="physically sequential pages instead of a random permutation (diagnostic)")
    ap.add_argument("--ragged", action="store_true", help="seq_lens ~ U{1..L} (diagnostic)")
    ap.add_argument("--ragged-sorted", action="store_true", help="same lengths, longest sequence first (diagnostic)")
    ap.add_argument("--kv", default="auto", choices=["auto", "fp8", "fp8_e5m2"],
                    help="KV cache element type: auto = fp16 (the BASELINE metric); fp8 = E4M3 bytes (diagnostic line, "
                         "SURVEY row f-4)")
    ap.add_argument("--hint-mean", action="store_true",
                    help="pass the batch's mean length to the heuristic (what a host-side scheduler can do)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = 256 sequences per GPU (default), strong = 2048 sequences in all")
    ap.add_argument("--kernel-samples", type=int, default=60,
                    help="launches in the separate kernel-timing pass (HIP event pair around each attention launch); "
                         "at least 50")
    ap.add_argument("--standin-cpu", action="store_true",
                    help="TEST ONLY: run the launch / rendezvous / timing / JSON plumbing on CPU over gloo with a "
                         "stand-in step (no product kernel runs; the line says so and is not a measurement)")
    return ap


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    if args.headline_only:
        args.no_cpu_baseline = args.no_fused = args.no_fp8 = args.no_ragged = args.no_graph = True
        args.no_cfg4 = args.no_e2e = args.no_cfg2 = args.no_strong = args.no_long = args.no_deferred = args.no_serve = True
    args.kernel_samples = max(50, args.kernel_samples)
    return args


# ---- launching N ranks from a plain `python bench.py --gpus N` ------------------------------------------------------

def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def needs_self_launch(args, env=os.environ) -> bool:
    """`--gpus N > 1` started without a launcher: no rank environment exists, so this process becomes the launcher."""
    return args.gpus > 1 and "WORLD_SIZE" not in env and "RANK" not in env


def self_launch(args, argv) -> int:
    """Start args.gpus ranks of this file, one per GPU index, on 127.0.0.1; rank 0 inherits stdout (the ONE JSON
    line), the other ranks' stdout goes to stderr.  Returns the first non-zero exit status (0 if all succeed)."""
    n = args.gpus
    if not args.standin_cpu:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f"bench.py: --gpus {n} but this node shows {have} HIP device(s)", file=sys.stderr)
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VMI_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else sys.stderr.fileno()))
    status = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0 and status == 0:
                status = rc
                for q in alive:            # one rank failed: the others would wait in a collective forever
                    q.terminate()
    return status


def init_dist(args):
    """-> (dist or None, rank, world, local_rank, device)."""
    n_gpus = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cpu = args.standin_cpu
    if n_gpus > 1 or world > 1 or os.environ.get("VMI_FORCE_DIST") == "1":   # VMI_FORCE_DIST: 1-GPU smoke of the RCCL path
        import torch.distributed as dist

        if world != n_gpus:
            raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}")
        # one rank = one GPU = the host cores next to it (before RCCL / gloo start their threads: they inherit the mask)
        global _PLACEMENT
        _PLACEMENT = shard.place_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), None if cpu else local_rank,
                                      bind=os.environ.get("VMI_BENCH_NO_BIND") != "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL prints a banner on STDOUT — when the communicator is created and again, with the lazily created
        # communicator of the first all_gather, later on.  stdout must carry exactly ONE JSON line, so fd 1 points at
        # stderr for the whole run and the line goes to the saved descriptor (emit_line).
        global _REAL_STDOUT
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
        if cpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            return dist, rank, world, local_rank, torch.device("cpu")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,  # "nccl" IS RCCL on ROCm
                                device_id=torch.device("cuda", local_rank))
        warm = torch.zeros(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        return dist, rank, world, local_rank, torch.device("cuda", local_rank)
    if cpu:
        return None, 0, 1, 0, torch.device("cpu")
    torch.cuda.set_device(0)
    return None, 0, 1, 0, torch.device("cuda", 0)


_REAL_STDOUT = None
_PLACEMENT = None      # shard.place_rank() of this rank (N > 1)


def emit_line(line: dict) -> None:
    text = json.dumps(line) + "\n"
    if _REAL_STDOUT is None:
        sys.stdout.write(text)
        sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, text.encode())


# ---- the step ---------------------------------------------------------------------------------------------------------

_V2_SCRATCH = {}
SKIP_RESHAPE = False
KV_DTYPE = "auto"      # "fp8" / "fp8_e5m2": byte caches in the x = 16 layout (--kv)
KV_PREFIX = {"auto": "", "fp8": "fp8_", "fp8_e5m2": "fp8e5m2_"}          # variant-name prefixes
FP8_ARG = {"auto": False, "fp8": True, "fp8_e5m2": "e5m2"}               # ops.pick_variant(fp8=...)
KV_LABEL = {"auto": "fp16", "fp8": "fp8 E4M3", "fp8_e5m2": "fp8 E5M2"}
RESHAPE_OTHER = False
NOOP_RESHAPE = None


def alg_bytes(cfg, kv=None):
    """Algorithmic bytes per attention launch; an fp8 cache halves the K/V term."""
    b = cfg.algorithmic_bytes()
    if (kv or KV_DTYPE).startswith("fp8"):
        b -= 2 * cfg.batch * cfg.kv_heads * cfg.seq_len * cfg.head_size
    return b


def attend(wl, out, t, variant, op="v1"):
    from vllmini_amd import ops

    c = wl.cfg
    if op in ("fused", "newest"):   # reshape_and_cache + paged_attention_v1 in one launch (extension, include/vmi_paged_attention.h)
        # ("newest": the same attention — the newest token read from this step's rows — WITHOUT the cache write)
        ops.paged_attention_v1_append(out, wl.query, wl.key, wl.value, wl.key_cache, wl.value_cache, c.kv_heads,
                                      wl.scale, wl.tables[t], wl.seq_lens, c.block_size, c.seq_len, _variant=variant,
                                      write_cache=op == "fused")
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


def scatter(wl, t, op="v1"):
    """reshape_and_cache of the step (gpt2.py:44) — or its diagnostic stand-ins."""
    from vllmini_amd import cache_ops

    if NOOP_RESHAPE is not None:
        NOOP_RESHAPE.fill_(1.0)
    elif op not in ("fused", "newest") and not SKIP_RESHAPE:
        cache_ops.reshape_and_cache(wl.key, wl.value, wl.key_cache, wl.value_cache,
                                    wl.slots[(t + 1) % len(wl.tables) if RESHAPE_OTHER else t], KV_DTYPE, 1.0)


def one_step(wl, out, i, variant, op="v1"):
    """The reference's per-layer decode call pair, in its call order (gpt2.py:44, :62)."""
    t = i % len(wl.tables)
    scatter(wl, t, op)
    attend(wl, out, t, variant, op)


_EXCHANGE = {}     # per-run state of the N > 1 token exchange: ids tensor, global batch, per-step event pairs


EXCHANGE_EVERY = 12    # layer steps per token: GPT-2 small's n_layer (the model BASELINE's configs name)


def exchange_tokens(dist, i=None):
    """The decode loop's only collective: every rank hands the ids it sampled for its sequences to
    all ranks.  Synthetic ids here; the all_gather is the real one.  A decode loop samples once per TOKEN, i.e. once
    per n_layer of this bench's steps (a step is ONE layer's call pair), so inside the timed region the exchange runs
    on every EXCHANGE_EVERY-th step, starting with the first (i = None: unconditionally)."""
    if dist is None or (i is not None and i % EXCHANGE_EVERY):
        return
    if i is None:      # measured alone (exchange_pass): the blocking form, the compute stream waits for the collective
        _EXCHANGE["gathered"] = shard.gather_token_ids(_EXCHANGE["ids"], _EXCHANGE["global_batch"], dist,
                                                       out=_EXCHANGE["out"][0])
        return
    # In the decode loop every rank samples its own sequences' ids, so the next token's layers do not depend on the
    # gathered vector: the all_gather is issued asynchronously (the process group's stream, behind the kernels already
    # enqueued) and runs BESIDE the next token's call pairs; the previous token's gather is waited for (stream-side) before
    # its buffer's turn comes again — two buffers alternate.
    k = (i // EXCHANGE_EVERY) & 1
    prev = _EXCHANGE["work"][k]
    if prev is not None:
        prev.wait()
    _EXCHANGE["gathered"], _EXCHANGE["work"][k] = shard.gather_token_ids_async(
        _EXCHANGE["ids"], _EXCHANGE["global_batch"], dist, _EXCHANGE["out"][k])


def setup_exchange(batch, dist, dev):
    if dist is None:
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    _EXCHANGE["global_batch"] = batch * world
    _EXCHANGE["ids"] = torch.arange(rank * batch, (rank + 1) * batch, dtype=torch.int64, device=dev)
    _EXCHANGE["out"] = [torch.empty(batch * world, dtype=torch.int64, device=dev) for _ in range(2)]
    _EXCHANGE["work"] = [None, None]


def device_sync(dev):
    return (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else None


def time_steps(wl, out, steps, warmup, variant, dist, dev, op="v1", legacy=False):
    """W untimed steps, then EXACTLY K timed steps between barrier+synchronize pairs
    (vllmini_amd/shard.py:timed_steps — the same code the 2-rank gloo test exercises).  The timed region holds the
    steps and, for N > 1, the per-step token exchange: no event records, no host reads."""
    setup_exchange(wl.cfg.batch, dist, dev)

    def step(i):
        one_step(wl, out, i, variant, op)
        if legacy:      # rounds 1-2: a BLOCKING exchange on every step (the compute stream waits for the collective)
            exchange_tokens(dist, None)
        else:
            exchange_tokens(dist, i)

    elapsed = shard.timed_steps(step, steps, warmup, dist, sync=device_sync(dev), clock_behind_barrier=legacy)   # (its closing device synchronise
    if dist is not None:                                                             #  covers the process group's stream)
        for w in _EXCHANGE["work"]:
            if w is not None:
                w.wait()
        _EXCHANGE["work"] = [None, None]
        g = _EXCHANGE["gathered"]
        assert g.numel() == _EXCHANGE["global_batch"] and int(g[0]) == 0 and int(g[-1]) == g.numel() - 1
    return elapsed


def kernel_pass(wl, out, n, variant, dev, op="v1", warm=5):
    """Separate measurement pass: n call pairs with a HIP event pair around every attention launch, on the launch
    stream (torch's current stream is the one ops.py launches on).  -> list of n durations in ms."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(warm):
        one_step(wl, out, i, variant, op)
    for i in range(n):
        t = i % len(wl.tables)
        scatter(wl, t, op)
        ev[i][0].record()
        attend(wl, out, t, variant, op)
        ev[i][1].record()
    torch.cuda.synchronize(dev)
    return [a.elapsed_time(b) for a, b in ev]


def exchange_pass(n, dist, dev):
    """Median duration of the token all_gather alone (event pair around each of n exchanges), us."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        exchange_tokens(dist)
        b.record()
    torch.cuda.synchronize(dev)
    return statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3


def kernel_stats(kern_ms, dist, dev):
    """mean / median / min of one rank's samples; for N > 1 the slowest rank's figure of each, and every rank's median."""
    ks = {k: shard.max_over_ranks(f(kern_ms), dist, dev) * 1e3
          for k, f in (("mean", statistics.mean), ("median", statistics.median), ("min", min))}
    ks["median_per_rank"] = [m * 1e3 for m in shard.all_ranks(statistics.median(kern_ms), dist, dev)]
    return ks


def graph_steps(wl, out, steps, variant, dev, per_graph=1, attend_only=False):
    """The reference's call pair captured ONCE into a hipGraph (one graph per table set) and replayed: the launch
    work the host does per step is one graph launch, so this is the step the GPU can do when the host is out of
    the way.  Returns seconds for `steps` replays."""
    graphs = []
    if attend_only:      # (the attention launch alone, table sets alternating: the device's own time per launch)
        def one_step(wl, out, i, variant):           # noqa: F811 — shadows the call pair for this measurement
            attend(wl, out, i % len(wl.tables), variant)
    else:
        one_step = globals()["one_step"]
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for t in range(len(wl.tables)):
            one_step(wl, out, t, variant)
    torch.cuda.current_stream(dev).wait_stream(side)
    if per_graph <= 1:
        for t in range(len(wl.tables)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):      # (the stream the warm-up ran on: its workspace exists, ops.workspace_for)
                one_step(wl, out, t, variant)
            graphs.append(g)
        per_graph = 1
    else:   # `per_graph` consecutive steps (table sets in the loop's order) in ONE graph
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for t in range(per_graph):
                one_step(wl, out, t % len(wl.tables), variant)
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


_LIB_SHA = None


def library_sha16() -> str:
    """First 16 hex digits of the SHA-256 of the product library this process loads: ties a bench line to its binary, and a
    committed rocprofv3 pass (profiles/pmc_*_latest.json carry the same field) to the binary it profiled."""
    global _LIB_SHA
    if _LIB_SHA is None:
        import hashlib
        from vllmini_amd import build as _b
        h = hashlib.sha256()
        with open(_b.LIB_PATH, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        _LIB_SHA = h.hexdigest()[:16]
    return _LIB_SHA


def _committed_pass(cfg_name: str, kernel_variant: str):
    """profiles/pmc_<cfg>_latest.json if it describes THIS kernel variant of THIS binary, else (None, why)."""
    path = os.path.join(REPO, "profiles", f"pmc_{cfg_name}_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, "no committed pass"
    if d.get("kernel_variant") and d["kernel_variant"] != kernel_variant:
        return None, f"committed pass is of kernel {d.get('kernel_variant')}"
    if d.get("library_sha16") and d["library_sha16"] != library_sha16():
        return None, f"committed pass is of library {d['library_sha16']}, this run loads {library_sha16()}"
    d["_path"] = os.path.relpath(path, REPO)
    return d, None


def pmc_traffic(cfg_name: str, kernel_variant: str):
    """HBM bytes per launch of the attention kernel from the rocprofv3 PMC passes committed under
    profiles/ (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command; FETCH_SIZE x1024 x2 per the
    gfx950 correction in MI355X_MICROARCH.md §HBM).  bench.py cannot run a profiler around itself, so
    the figure comes from the latest recorded pass for this workload and kernel variant, else None."""
    if cfg_name == "cfg5":
        cfg_name = "cfg3"   # cfg5 = the cfg3 launch over a larger pool (N > 1 runs)
    d, why = _committed_pass(cfg_name, kernel_variant)
    try:
        return (d["traffic_bytes_corrected"]["total"], d["_path"]) if d else (None, why)
    except (KeyError, TypeError):
        return None, "committed pass holds no traffic figure"


# ---- the CPU baseline -------------------------------------------------------------------------------------------------

def physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:  # noqa: BLE001
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_thread_candidates(logical: int, physical: int) -> list:
    """Thread counts the baseline tries: 8 (what BASELINE.md §3 was measured on), a quarter / half / all of the
    physical cores, and every logical CPU (round 2's only setting)."""
    cand = {8, 16, physical // 4, physical // 2, physical, logical}
    return sorted(c for c in cand if 1 <= c <= logical)


def cpu_baseline(wl, budget_s: float):
    """Reference CPU fallback: eager attention (gather by block_tables -> matmul/softmax/matmul), fp32, on the host
    cores.  One warm-up + timed steps at each thread count; `value` is the best count's median."""
    from oracle.eager import torch_eager_decode  # checker/baseline only; never on the product path

    c = wl.cfg
    kc, vc = wl.key_cache.cpu(), wl.value_cache.cpu()
    q = wl.query.cpu().contiguous()
    tab = wl.tables[0].cpu()
    logical, physical = os.cpu_count() or 1, physical_cores()

    def run():
        t0 = time.perf_counter()
        torch_eager_decode(q, kc, vc, wl.scale, tab, c.seq_len, dtype=torch.float32)
        return time.perf_counter() - t0

    tried = {}
    cands = cpu_thread_candidates(logical, physical)
    per = budget_s * 0.6 / len(cands)
    for n in cands:
        torch.set_num_threads(n)
        first = run()                                           # warm-up (page faults, thread pool start)
        reps = max(1, min(3, int((per - first) / max(first, 1e-3))))
        tried[n] = statistics.median(run() for _ in range(reps))
    best = min(tried, key=tried.get)
    torch.set_num_threads(best)
    reps = max(3, min(7, int(budget_s * 0.4 / max(tried[best], 1e-3))))
    ts = [run() for _ in range(reps)]
    med = min(statistics.median(ts), tried[best])
    return {
        "value": c.batch / med, "unit": "tokens/s", "cores": best, "kind": "port",
        "ms_per_step": med * 1e3,
        "threads_tried": {str(n): round(t * 1e3, 1) for n, t in tried.items()},
        "host": {"logical_cpus": logical, "physical_cores": physical},
        "ms_per_step_all_logical_cpus": tried.get(logical, float("nan")) * 1e3,
        "sample": f"full {c.name} batch ({c.batch} seqs x {c.seq_len} tokens, H{c.num_heads} D{c.head_size}), "
                  f"fp32 torch eager incl. page gather, median of {reps} steps at the best of {len(cands)} thread "
                  f"counts ({best} threads); every count: 1 warm-up + up to 3 steps",
    }


# ---- end to end (GPT-2 small over the batched harness) ----------------------------------------------------------------

def e2e_measure(args, cfg, dist, rank, world, dev, kv="auto", fused=False, ragged=False, eager=False,
                ctx0=1008, operator_share=True, native_layers=True, scatter_in_c_attn=False, sampler="greedy", deferred=False):
    """GPT-2 small, `batch` sequences per GPU at ~seq_len context, one token per sequence per step,
    through vllmini_amd.gpt2_decode (hipGraph replay of the whole step).  KV is synthetic: pages are
    filled with random fp16 and sequences are registered at the target context length."""
    import numpy as np

    from vllmini_amd import cache_ops, gpt2_layer, ops
    from vllmini_amd.gpt2_decode import GPT2Dims, GPT2PagedDecoder, random_state_dict
    from vllmini_amd.kv_pool import PagedKVPool

    # GPT-2 small with the position table extended past 1024 so contexts can cross seq_len 1024
    dims = GPT2Dims(n_positions=2048)
    assert (cfg.num_heads, cfg.head_size) == (dims.n_head, dims.head_size)
    total_steps = args.warmup + 3 * args.steps + 2      # (three timed regions, below)
    mb = -(-(ctx0 + total_steps) // cfg.block_size) + 1
    blocks_needed = cfg.batch * dims.n_layer * (mb - 1)
    pool = PagedKVPool(blocks_needed + 64, dims.n_head, dims.head_size, cfg.block_size, mb, dims.n_layer, device=dev,
                       max_seqs=cfg.batch, multi_block_prefill=True, kv_cache_dtype=kv)
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    if kv.startswith("fp8"):      # random E4M3 / E5M2 codes of magnitude < 2 (codes 0..63 in either format)
        for c in (pool.key_cache, pool.value_cache):
            c.copy_(torch.randint(0, 64, c.shape, dtype=torch.uint8, device=dev, generator=g)
                    | (torch.randint(0, 2, c.shape, dtype=torch.uint8, device=dev, generator=g) << 7))
    else:
        pool.key_cache.uniform_(-1, 1, generator=g)
        pool.value_cache.uniform_(-1, 1, generator=g)
    # shuffle the free list so pages are scattered like a long-running pool's
    perm = np.random.default_rng(rank).permutation(pool.num_blocks)
    pool.free_blocks = perm.tolist()
    ctxs = np.random.default_rng(100 + rank).integers(16, ctx0 + 1, cfg.batch) if ragged else [ctx0] * cfg.batch
    for s in range(cfg.batch):
        pool.allocate_for_prefill(s, int(ctxs[s]))   # bookkeeping only: the pages already hold synthetic KV
    dec = GPT2PagedDecoder(dims, random_state_dict(dims, dev, seed=rank), pool, fused_append=fused, native_layers=native_layers,
                           scatter_in_c_attn=scatter_in_c_attn, deferred_scatter=deferred)
    ids = list(range(cfg.batch))
    tok = torch.randint(0, dims.vocab_size, (cfg.batch,), device=dev, generator=g)

    # N > 1: the sampled ids go back to every rank once per token — asynchronously, beside the next token's layers
    # (shard.gather_token_ids_async; a rank's next step needs only its own ids), two buffers alternating
    gathered = [torch.empty(cfg.batch * world, dtype=torch.int64, device=dev) for _ in range(2)] if dist is not None else None
    works = [None, None]

    def step(i):
        nonlocal tok
        logits = dec.decode(ids, tok, use_graph=not eager)
        if sampler == "greedy":
            tok = dec.greedy(logits)                  # stays on the device, no host sync
        elif sampler == "top_k":                      # the reference's sampling (scheduler.py:144-153), one launch of this build
            tok = gpt2_layer.sample_top_k(logits, 50, 1.0, generator=g)
        else:                                         # ... as the torch chain the reference spells out
            vals, idx = torch.topk(logits.float(), 50, dim=-1)
            tok = idx.gather(-1, torch.multinomial(torch.softmax(vals, -1), 1, generator=g)).squeeze(-1)
        if dist is not None:
            k = i & 1
            if works[k] is not None:
                works[k].wait()
            _, works[k] = shard.gather_token_ids_async(tok, cfg.batch * world, dist, gathered[k])

    # THREE consecutive timed regions of `steps` (each in the full bracket): these sub-records run seconds after multi-GB pools were
    # allocated and released, and one region in ten or so caught a ~100 ms stall of the runtime (a one-sequence token read 4 637 us
    # instead of 527 in one of three otherwise identical runs).  Each region's contexts are `steps` tokens longer than the last.
    regions = []
    for c in range(3):
        regions.append(shard.max_over_ranks(shard.timed_steps(step, args.steps, args.warmup if c == 0 else 0, dist,
                                                              sync=device_sync(dev)), dist, dev) / args.steps)
    for w in works:
        if w is not None:
            w.wait()
    # the MEDIAN of three consecutive regions (round 5 reported the faster of two: a selection); all three are on
    # the record, and `stall_suspected` says when they disagree by more than a quarter
    elapsed = statistics.median(regions) * args.steps
    note = ("12 x (c_attn, paged_attention_v1_append [fused], c_proj, MLP) + lm_head, hipGraph replay, greedy" if fused else
            "12 x (c_attn, reshape_and_cache, paged_attention_v1, c_proj, MLP) + lm_head, hipGraph replay, greedy")
    if deferred:
        note = note.replace("c_attn, reshape_and_cache, paged_attention_v1", "c_attn, paged_attention_v1 over cache + this step's rows "
                            "[append-read]") + "; ONE reshape_and_cache per token for the 12 layers' rows"
    if scatter_in_c_attn:
        note = note.replace("c_attn, reshape_and_cache, paged_attention_v1", "c_attn [writes k, v into the cache itself], paged_attention_v1")
    note += ("; the block's linear layers on this build's kernels (ln_1 + c_attn, c_proj + residual, ln_2 + c_fc + GELU, "
             "mlp.c_proj + residual: four launches, csrc/gpt2_layer.hip)" if native_layers else
             "; the block's linear layers as torch modules (layer_norm, F.linear -> hipBLASLt, gelu, add: eleven launches)")
    res = {"metric": "gpt2_small_decode_tokens_per_sec_end_to_end", "value": cfg.batch * world * args.steps / elapsed,
           "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3,
           "context": f"U{{16..{ctx0}}} (mean {float(np.mean(ctxs)):.0f})" if ragged else ctx0, "batch_per_gpu": cfg.batch,
           "data": "synthetic KV + random-init GPT-2 small weights", "dtype": "f16", "kv_cache_dtype": kv,
           "layers": "native" if native_layers else "torch_modules",
           "timed_regions_ms_per_step": [r * 1e3 for r in regions], "stall_suspected": max(regions) > 1.25 * min(regions),
           "value_is": "median of three consecutive timed regions",
           "note": note.replace("hipGraph replay", "plain launches" if eager else "hipGraph replay")}
    if operator_share and not fused:
        # the two operators alone on the decoder's own buffers: the 12 layers' call pairs back to back (the tables,
        # slots and lengths of the last step, a [B, 3E] projection output as q/k/v), one event pair around the chain
        st = dec._static
        E, H, D = dims.n_embd, dims.n_head, dims.head_size
        qkv = torch.randn((cfg.batch, 3 * E), dtype=torch.float16, device=dev, generator=g)
        q, k, v = (qkv[:, i * E:(i + 1) * E].view(cfg.batch, H, D) for i in range(3))
        out = torch.empty((cfg.batch, H, D), dtype=torch.float16, device=dev)
        var = st.get("variant", 0)
        n = 24
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for j in range(n + 2):
            if j >= 2:
                ev[j - 2][0].record()
            for i in range(dims.n_layer):
                cache_ops.reshape_and_cache(k, v, pool.key_cache, pool.value_cache, st["slots"][i], kv, pool.kv_scale)
                ops.paged_attention_v1(out, q, pool.key_cache, pool.value_cache, H, dec.scale, st["tables"][i],
                                       st["seq_lens"], pool.block_size, dec.max_seq_len, None, kv, pool.kv_scale,
                                       0, 0, 1, 1, 0, _variant=var)
            if j >= 2:
                ev[j - 2][1].record()
        torch.cuda.synchronize(dev)
        chain_ms = shard.max_over_ranks(statistics.median(a.elapsed_time(b) for a, b in ev), dist, dev)
        res["operators_ms_per_step"] = chain_ms
        res["operators_share_of_step"] = chain_ms / res["ms_per_step"]
        res["operators_note"] = (f"{dims.n_layer} x (reshape_and_cache + paged_attention_v1) on the decoder's own pool, "
                                 f"tables and lengths, launched back to back; median of {n} chains by HIP events")
    del dec, pool
    torch.cuda.empty_cache()
    return res


def serve_trace(requests, max_prompt, mean_new, max_length, vocab, seed=0):
    """The seeded request trace: prompt lengths U{4..max_prompt}, output lengths geometric with mean `mean_new` (at least 1, at most
    4 x mean_new: without the cap a closed-loop trace ends in hundreds of steps of a few sequences), cut so that prompt + output <= max_length; token ids uniform."""
    import numpy as np

    rng = np.random.default_rng(seed)
    plen = rng.integers(4, max_prompt + 1, requests)
    new = np.minimum(np.minimum(rng.geometric(1.0 / mean_new, requests), 4 * mean_new), max_length - plen).astype(np.int64)
    prompts = [rng.integers(0, vocab, int(n)).tolist() for n in plen]
    return prompts, [int(k) for k in new]


def serve_measure(args, dev, rank=0, requests=None, note_extra=""):
    """The continuous-batching scheduler itself (reference vllmini/scheduler.py:55-130): BatchScheduler over
    GPT2PagedDecoder on ONE GPU, every request queued at t = 0 (closed loop), admitted — several prompts per prefill call —
    as batch slots and blocks free up, contexts ragged, the pool too small for the batch so that sequences are preempted by
    swap and resumed.  Nothing is synthetic but the weights and the token ids: K / V come from real prefills."""
    import numpy as np

    from vllmini_amd.gpt2_decode import GPT2Dims, GPT2PagedDecoder, random_state_dict
    from vllmini_amd.kv_pool import PagedKVPool
    from vllmini_amd.scheduler import BatchScheduler, sample_greedy, sample_top_k

    requests = requests or args.serve_requests
    dims = GPT2Dims()
    max_length, bs, L = dims.n_positions, 16, dims.n_layer
    prompts, new = serve_trace(requests, min(args.serve_max_prompt, max_length - 1), args.serve_mean_new, max_length, dims.vocab_size,
                               seed=17 + rank)
    mb = max_length // bs + 1
    # a pool for ~70 % of what max_batch sequences of the trace's mean mid-life length hold: preemption happens, thrashing does not
    mean_mid = float(np.mean([len(p) + k / 2 for p, k in zip(prompts, new)]))
    want = int(args.serve_pool_frac * args.serve_max_batch * L * (mean_mid / bs + 1))
    nblocks = args.serve_pool_blocks or max(want, L * mb + 64)
    pool = PagedKVPool(nblocks, dims.n_head, dims.head_size, bs, mb, L, device=dev, max_seqs=args.serve_max_batch + 8,
                       host_blocks=max(nblocks // 4, L * mb), kv_cache_dtype=args.serve_kv)
    if args.serve_preempt == "swap":
        pool.reserve_host()          # (the pinned host pool of a server exists before its first request)
    deferred = not args.serve_no_deferred_scatter and args.serve_kv == "auto"      # (the append-read kernels are built for fp16 pages)
    dec = GPT2PagedDecoder(dims, random_state_dict(dims, dev, seed=rank), pool, deferred_scatter=deferred,
                           scatter_in_c_attn=None if not deferred else None, pad_batch_to=0 if args.serve_eager else 32)
    g = torch.Generator(device=dev).manual_seed(5 + rank)
    sampler = sample_greedy if args.serve_sampler == "greedy" else sample_top_k
    if args.serve_sampler == "greedy":
        sampler = lambda logits, generator=None: dec.greedy(logits)      # noqa: E731 — the harness's argmax kernel
    sch = BatchScheduler(dec, max_length=max_length, eos_token_id=dims.eos_token_id, max_batch=args.serve_max_batch,
                         sampler=sampler, use_graph=not args.serve_eager, generator=g, preempt=args.serve_preempt,
                         record_latency=True, admit_every=args.serve_admit_every, max_prefill_tokens=args.serve_prefill_tokens,
                         headroom_blocks=args.serve_headroom)
    # warm-up outside the clock: library handles, the graphs of the padded batch sizes the run will see (captured on first use)
    warm = BatchScheduler(dec, max_length=max_length, eos_token_id=dims.eos_token_id, max_batch=args.serve_max_batch,
                          sampler=sampler, use_graph=not args.serve_eager, generator=g)
    # (request i ends after 2 + i // 32 tokens: the batch passes through every padded size on its way down)
    for i, p in enumerate(prompts[: min(args.serve_max_batch, requests)]):
        warm.submit(p[:48], max_new_tokens=2 + i // 32)
    warm.run()
    assert sorted(pool.free_blocks) == list(range(nblocks))
    pool.free_blocks = list(range(nblocks))
    torch.cuda.synchronize(dev)
    t_pre = [0.0]
    real_prefill_batch, real_prefill = dec.prefill_batch, dec.prefill

    def timed(fn):
        def run(*a, **k):
            t = time.perf_counter()
            out = fn(*a, **k)
            t_pre[0] += time.perf_counter() - t
            return out
        return run
    dec.prefill_batch, dec.prefill = timed(real_prefill_batch), timed(real_prefill)
    t0 = time.perf_counter()
    if args.serve_rate > 0:
        # OPEN loop: Poisson arrivals at `serve_rate` requests/s (seeded); the scheduler steps whenever something is pending and
        # sleeps until the next arrival otherwise — latency under a given load instead of the closed loop's saturation throughput
        arrive = np.cumsum(np.random.default_rng(23 + rank).exponential(1.0 / args.serve_rate, requests))
        ids, i, steps = [], 0, 0
        while i < requests or sch.pending():
            now = time.perf_counter() - t0
            while i < requests and arrive[i] <= now:
                ids.append(sch.submit(prompts[i], max_new_tokens=new[i]))
                i += 1
            if sch.pending():
                sch.step()
                steps += 1
            else:
                time.sleep(max(0.0, min(arrive[i] - now, 1e-3)))
        steps = sch.steps
    else:
        ids = [sch.submit(p, max_new_tokens=k) for p, k in zip(prompts, new)]
        steps = sch.run()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    from vllmini_amd import ops
    ops.check_workspaces()
    done = sum(len(sch.sequences[s]) - len(p) for s, p in zip(ids, prompts))
    by_eos = sum(1 for s, p, k in zip(ids, prompts, new) if len(sch.sequences[s]) < len(p) + k and sch.sequences[s][-1] == dims.eos_token_id)
    short = sum(1 for s, p, k in zip(ids, prompts, new) if len(sch.sequences[s]) < len(p) + k)
    # every request ran to its length (or to an EOS it sampled itself) unless the reference's drop policy was asked for
    assert not sch.pending() and (args.serve_preempt == "drop" or (not sch.evicted and short == by_eos)), (done, sum(new), short, by_eos)
    lat = np.concatenate(sch.token_latency_s) if sch.token_latency_s else np.zeros(1)
    ttft = np.asarray(sch.first_token_s) if sch.first_token_s else np.zeros(1)
    st, ss = sch.stats, pool.swap_stats
    rec = {"metric": "gpt2_small_serving_decode_tokens_per_sec", "value": done / wall, "unit": "tokens/s", "n_gpus": 1,
           "requests": requests, "generated_tokens": int(done), "ended_by_eos": by_eos, "prompt_tokens": int(sum(len(p) for p in prompts)),
           "wall_s": wall, "decode_steps": steps, "step_us": wall / max(steps, 1) * 1e6,
           "tokens_per_s_incl_prompt": (done + sum(len(p) for p in prompts)) / wall,
           "batch_occupancy": st["decode_rows"] / max(steps, 1) / args.serve_max_batch, "max_batch": args.serve_max_batch,
           "token_latency_ms": {"mean": float(lat.mean() * 1e3), "p50": float(np.percentile(lat, 50) * 1e3),
                                "p95": float(np.percentile(lat, 95) * 1e3), "p99": float(np.percentile(lat, 99) * 1e3)},
           "first_token_s": {"mean": float(ttft.mean()), "p95": float(np.percentile(ttft, 95))},
           "host_us_per_step": st["host_s"] / max(steps, 1) * 1e6, "gpu_wait_us_per_step": st["wait_s"] / max(steps, 1) * 1e6,
           "admit_s": st["admit_s"], "admit_every_steps": args.serve_admit_every, "headroom_blocks_per_seq": args.serve_headroom, "prefill_s": t_pre[0], "prefill_calls": st["prefill_calls"], "prompts_per_prefill_call": requests / max(st["prefill_calls"], 1),
           "preemptions": st["preemptions"], "resumes": st["resumes"], "dropped": st["dropped"],
           "swap_out_MB": ss["bytes_out"] / 1e6, "swap_in_MB": ss["bytes_in"] / 1e6,
           "pool_blocks": nblocks, "pool_GB": 2 * nblocks * pool.block_bytes / 1e9, "sampler": args.serve_sampler,
           "preempt": args.serve_preempt, "kv_cache_dtype": args.serve_kv, "deferred_scatter": deferred, "graph_replay": not args.serve_eager,
           "arrival_rate_per_s": args.serve_rate or None,
           "trace": (f"open loop, Poisson arrivals at {args.serve_rate:g} requests/s" if args.serve_rate > 0 else
                     f"closed loop, all {requests} requests queued at t = 0") + f"; {requests} requests; prompts U{{4..{args.serve_max_prompt}}} tokens "
                    f"(mean {np.mean([len(p) for p in prompts]):.0f}), outputs geometric (mean {np.mean(new):.0f}), prompt + output "
                    f"<= {max_length}; seed 17",
           "note": "BatchScheduler (vllmini_amd/scheduler.py) over GPT2PagedDecoder, random-init GPT-2 small, real prefills (several "
                   "prompts per call, their causal attention ONE paged_attention_v1 launch per layer) and decode steps through the operators; "
                   "host_us_per_step = the scheduler's and pool's Python per decode step (bookkeeping, staging, the graph launch) apart "
                   "from gpu_wait_us_per_step, the time it then waits for the sampled ids, and from admit_s, the admission of queued "
                   "requests (prefill_s = the prefill calls in it: host-bound torch launches); all of it is inside wall_s" + note_extra}
    del sch, warm, dec, pool
    torch.cuda.empty_cache()
    return rec


def run_serve(args, dev, rank):
    res = serve_measure(args, dev, rank)
    if rank == 0:
        print(json.dumps(res), file=sys.stderr, flush=True)
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "serve.json"), "w") as f:
            json.dump(res, f, indent=1)


def run_e2e(args, cfg, dist, rank, world, dev):
    res = e2e_measure(args, cfg, dist, rank, world, dev, kv=args.kv, fused=args.e2e_fused, ragged=args.e2e_ragged, deferred=args.e2e_deferred,
                      eager=args.e2e_eager, ctx0=args.e2e_context, native_layers=not args.e2e_torch_layers,
                      scatter_in_c_attn=args.e2e_scatter_in_c_attn, sampler=args.e2e_sampler)
    if rank == 0:
        print(json.dumps(res), file=sys.stderr, flush=True)
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        name = "e2e_ragged.json" if args.e2e_ragged else "e2e_fused.json" if args.e2e_fused else \
            {"fp8": "e2e_fp8.json", "fp8_e5m2": "e2e_fp8_e5m2.json"}.get(args.kv, "e2e.json")
        name = name.replace(".json", "_torch_layers.json") if args.e2e_torch_layers else name
        with open(os.path.join(REPO, "gpurun_out", name), "w") as f:
            json.dump(res, f, indent=1)


# ---- the stand-in (tests of the launch path on CPU) -------------------------------------------------------------------

def standin_main(args, dist, rank, world, dev):
    """--standin-cpu: everything around the kernels — self-launch, rendezvous, barrier-bracketed timing, max over ranks,
    token exchange, one JSON line on rank 0's stdout — with a stand-in step on CPU tensors over gloo.  No product
    kernel runs and the line says so: it exists for tests/test_bench_launch.py, never as a measurement."""
    batch = 256 if args.scaling == "weak" else 2048 // world      # the sequences per rank of the real run (BASELINE configs[4])
    x = torch.ones(batch, 64)
    setup_exchange(batch, dist, dev)

    def step(i):
        x.mul_(1.0)
        exchange_tokens(dist, i)

    elapsed = shard.timed_steps(step, args.steps, args.warmup, dist)
    if dist is not None:
        for w in _EXCHANGE["work"]:       # (on CPU there is no device synchronise: the asynchronous gathers end here)
            if w is not None:
                w.wait()
        _EXCHANGE["work"] = [None, None]
    per_rank = shard.all_ranks(elapsed / args.steps * 1e3, dist, dev)     # the N > 1 line's per-rank figures, same code path
    exch_us = None
    if dist is not None:     # the exchange alone, as the real line reports it (wall clock here: no device events on CPU)
        t0 = time.perf_counter()
        for _ in range(20):
            exchange_tokens(dist, None)
        exch_us = shard.max_over_ranks((time.perf_counter() - t0) / 20 * 1e6, dist, dev)
    elapsed = shard.max_over_ranks(elapsed, dist, dev)
    placement = shard.gather_objects(_PLACEMENT, dist) if dist is not None else None
    legacy = None
    if dist is not None:     # the rounds-1-2 method beside it: blocking exchange on every step, clock behind the barrier

        def legacy_step(i):
            x.mul_(1.0)
            exchange_tokens(dist, None)

        l_el = shard.max_over_ranks(shard.timed_steps(legacy_step, args.steps, args.warmup, dist, clock_behind_barrier=True), dist, dev)
        legacy = {"method_version": 2, "ms_per_step": l_el / args.steps * 1e3, "value": batch * world * args.steps / l_el,
                  "unit": "tokens/s", "timing_bracket": shard.TIMING_BRACKET_LEGACY, "token_exchange": "blocking all_gather on every step"}
    if dist is not None:
        g = _EXCHANGE["gathered"]
        assert g.numel() == batch * world and int(g[-1]) == g.numel() - 1
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit_line({"metric": "STAND-IN (launch-path test, not a measurement)", "value": batch * world * args.steps / elapsed,
                   "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
                   "vs_baseline": None, "dtype": "f32", "data": "stand-in",
                   "config": {"workload": "stand-in step on CPU over gloo", "global_batch": batch * world, "batch_per_rank": batch},
                   "roofline": None, "cpu_baseline": None,
                   "method_version": 3, "timing_bracket": shard.TIMING_BRACKET, "ms_per_step_per_rank": per_rank,
                   "rank_placement": placement, "token_exchange_every_steps": EXCHANGE_EVERY,
                   "token_exchange_us": exch_us, "rccl_ranks": world if placement is not None else None,
                   "process_group_backend": "gloo (stand-in)" if placement is not None else None,
                   "legacy_method_step": legacy,
                   "self_launched": os.environ.get("VMI_BENCH_SELF_LAUNCHED") == "1"})


# ---- sub-records ------------------------------------------------------------------------------------------------------

def pair_record(wl, out, args, variant, dist, dev, tokens, nbytes, op="v1"):
    """Timed region + kernel pass for one workload through the default entry -> (record, kernel stats)."""
    elapsed = shard.max_over_ranks(time_steps(wl, out, args.steps, args.warmup, variant, dist, dev, op=op), dist, dev)
    ks = kernel_stats(kernel_pass(wl, out, args.kernel_samples, variant, dev, op=op), dist, dev)
    rec = {"value": tokens / elapsed, "unit": "tokens/s", "ms_per_step": elapsed / args.steps * 1e3,
           "kernel_us_median": ks["median"], "kernel_us_mean": ks["mean"], "kernel_us_min": ks["min"],
           "kernel_event_samples": args.kernel_samples}
    if nbytes:
        gbps = nbytes / (ks["median"] * 1e-6) / 1e9
        rec.update({"algorithmic_bytes_per_launch": nbytes, "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS})
    return rec, ks


def rocprof_kernel_us(cfg_name: str, kernel_variant: str):
    """The attention kernel's average duration in the latest committed rocprofv3 --kernel-trace --stats pass for this workload
    and kernel (profiles/pmc_<cfg>_latest.json), or None — bench.py cannot run a profiler around itself."""
    d, why = _committed_pass(cfg_name, kernel_variant)
    try:
        return (d["pa_v1_dispatches"]["mean_us_after_warmup"], f"{d['_path']} (pass {d.get('profile_tag', '?')})") if d else (None, why)
    except (KeyError, TypeError):
        return None, "committed pass holds no kernel duration"


def cfg2_record(args, dist, rank, world, dev):
    """BASELINE configs[1]: batch 32, seq_len 512, 12 heads x 64, num_blocks 4096 — 50 MB per launch, resident in the 256 MiB
    Infinity Cache, 384 (sequence, head) units on 256 CUs: a LATENCY chain, not a stream, so no fraction of the HBM peak is
    quoted.  The launch is shorter than the host's work per call, so three figures: the plain call pair (host-bound), the
    attention kernel between HIP events (host-bound at this size), and the pair replayed from a hipGraph of 48 pairs
    (the device's own time per pair)."""
    from vllmini_amd import ops
    c2 = CONFIGS["cfg2"]
    wl2 = make_workload(c2, dev, seed=55 + rank, table_sets=2)
    out2 = torch.empty((c2.batch, c2.num_heads, c2.head_size), dtype=torch.float16, device=dev)
    rec, _ = pair_record(wl2, out2, args, 0, dist, dev, c2.batch * world * args.steps, 0)
    v2 = ops.variant_names()[ops.last_variant() - 1]
    n_g = 96
    g = graph_steps(wl2, out2, n_g, 0, dev, per_graph=48)
    ga = graph_steps(wl2, out2, n_g, 0, dev, per_graph=48, attend_only=True)
    us, src = rocprof_kernel_us("cfg2", v2)
    for k in ("kernel_us_median", "kernel_us_mean", "kernel_us_min"):      # not kernel times at this size: see the note
        rec["event_pair_" + k[7:] + "_host_bound"] = rec.pop(k)
    return {"op": "reshape_and_cache + paged_attention_v1, BASELINE configs[1]: batch 32/GPU, seq_len 512, 12 heads x 64, "
                  "block_size 16, num_blocks 4096, fp16", **rec, "kernel_variant": v2,
            "algorithmic_bytes_per_launch": alg_bytes(c2, "auto"),
            "graph_us_per_pair": g / n_g * 1e6,
            "graph_us_per_attention_launch": ga / n_g * 1e6,
            "committed_profile_kernel_us": us, "committed_profile_source": src,
            "regime": "Infinity-Cache-resident (50 MB per launch, two table sets = 100 MB of a 201 MB pool) and under-filled (384 "
                      "units on 256 CUs): a latency chain — launch, table + q, K pages, barrier, softmax, V pages, barrier, store "
                      "(profiles/r04_underfilled_chip.md) — not an HBM stream: no frac quoted",
            "note": "ms_per_step is host-bound (the call pair's Python + launch work exceeds its 15 us of kernels), and so are the "
                    "event pairs around the attention launch (the GPU waits for the host between the two records): NOT kernel "
                    "times at this size.  The device's own figures: graph_us_per_pair = reshape_and_cache + paged_attention_v1 + "
                    "gaps, 48 pairs per hipGraph; graph_us_per_attention_launch = 48 attention launches per graph (kernel + "
                    "~1 us of spacing); committed_profile_kernel_us = the kernel alone in the rocprofv3 pass committed under profiles/ — NOT measured by this run, and null unless that pass profiled this kernel variant of this very binary (library_sha16)"}


def long_context_record(args, dev):
    """Few sequences x long contexts (batch 1 x 16384 tokens and batch 4 x 8192 at 12 heads x 64; batch 4 x 8192 at 32 / 8
    grouped-query heads x 128; batch 48 x 32768 — past the plain kernels' LDS, in rounds): the regime the reference's
    scheduler runs (one sequence at a time, scheduler.py:60) at today's context lengths.  The same default entry with and
    without the wrapper's workspace (round 5: with one, paged_attention_v1 spreads each (sequence, head) over up to 64 waves on
    as many CUs — vmi_paged_attention_v1_f16_ws, pa_split.hpp).  Device time per attention launch, 24 launches per hipGraph."""
    from vllmini_amd import ops
    out = {"op": "paged_attention_v1, default entry, fp16, block_size 16", "unit": "us per attention launch (hipGraph of 24)"}
    for name in ("long_b1", "long_b4", "long_gqa", "long_32k"):
        c = CONFIGS[name]
        wl = make_workload(c, dev, seed=17, table_sets=2)
        o = torch.empty((c.batch, c.num_heads, c.head_size), dtype=torch.float16, device=dev)
        rec = {"batch": c.batch, "seq_len": c.seq_len, "heads": f"{c.num_heads} / {c.kv_heads} x {c.head_size}",
               "algorithmic_bytes_per_launch": alg_bytes(c, "auto")}
        for key, on in (("with_workspace", True), ("without_workspace", False)):
            prev = ops.set_workspace_enabled(on)
            try:
                attend(wl, o, 0, 0)
                torch.cuda.synchronize(dev)
                v = ops.variant_names()[ops.last_variant() - 1]
                n = 96 if c.seq_len * c.batch < (1 << 18) else 24        # (long_32k without a workspace: 3.4 ms per launch)
                us = graph_steps(wl, o, n, 0, dev, per_graph=min(n, 24), attend_only=True) / n * 1e6
            finally:
                ops.set_workspace_enabled(prev)
            rec[key] = {"us": us, "kernel_variant": v, "achieved_GBps": rec["algorithmic_bytes_per_launch"] / us / 1e3}
        rec["speedup"] = rec["without_workspace"]["us"] / rec["with_workspace"]["us"]
        us_p, src = rocprof_kernel_us(name, rec["with_workspace"]["kernel_variant"])
        rec["committed_profile_kernel_us"], rec["committed_profile_source"] = us_p, src
        out[name] = rec
        del wl, o
        torch.cuda.empty_cache()
    return out


def strong_n1_record(args, dev):
    """The N = 1 anchor of the STRONG-scaling curve (BASELINE.md §2 last row): all 2048 sequences of BASELINE
    configs[4] on one GPU, pool = max(65536, what two disjoint table sets need) blocks.  2048 = QSORT_MAX, the most sequences
    the balanced kernel ranks (pa_queue.hpp); the default entry's pick is reported."""
    from vllmini_amd import ops
    c5 = CONFIGS["cfg5"]
    b_ = 2048
    c5 = dataclasses.replace(c5, name="cfg5_strong_n1", batch=b_, num_blocks=max(c5.num_blocks, 2 * b_ * c5.blocks_per_seq))
    wl5 = make_workload(c5, dev, seed=91, table_sets=2)
    out5 = torch.empty((b_, c5.num_heads, c5.head_size), dtype=torch.float16, device=dev)
    rec, _ = pair_record(wl5, out5, args, 0, None, dev, b_ * args.steps, alg_bytes(c5, "auto"))
    v5 = ops.variant_names()[ops.last_variant() - 1]
    del wl5, out5
    torch.cuda.empty_cache()
    t5, t5src = pmc_traffic("cfg5_strong", v5)
    us5, us5src = rocprof_kernel_us("cfg5_strong", v5)
    return {"op": "reshape_and_cache + paged_attention_v1, BASELINE configs[4] on ONE GPU: batch 2048, seq_len 1024, 12 heads x 64, "
                  f"block_size 16, num_blocks {c5.num_blocks}, fp16", **rec, "kernel_variant": v5,
            "traffic": t5, "traffic_source": t5src, "committed_profile_kernel_us": us5, "committed_profile_source": us5src,
            "note": "N = 1 point of `--scaling strong` (2048 sequences in all, 2048/N per GPU); batch 2048 = QSORT_MAX; "
                    "the weak-scaling N = 1 point is the headline itself (256 sequences per GPU)"}


def deferred_scatter_record(args, dev):
    """A TOKEN of a 12-layer model on the cfg3 shape, the two ways a decode loop can run its layers' hot path (N = 1):
      pair      12 x (reshape_and_cache ; paged_attention_v1)                      — the reference's order, gpt2.py:44 then :62
      deferred  12 x paged_attention_v1 over cache + this step's rows (vmi_paged_attention_v1_newest_f16: the append-read
                kernels, bit-identical out) ; ONE reshape_and_cache over the 12 x B rows   — GPT2PagedDecoder(deferred_scatter)
    Twelve DISJOINT table sets (a layer's pages are its own: 9.4 GB of KV), so no launch re-reads what another left in the
    Infinity Cache.  Reported per LAYER STEP (token time / 12), comparable with the headline's ms_per_step."""
    from vllmini_amd import cache_ops, ops
    c = CONFIGS["cfg3"]
    nl = 12
    c12 = dataclasses.replace(c, name="cfg3_12_layers", num_blocks=nl * c.batch * c.blocks_per_seq + 64)
    wl = make_workload(c12, dev, seed=2024, table_sets=nl)
    assert len(wl.tables) == nl
    E = c.num_heads * c.head_size
    g = torch.Generator(device=dev).manual_seed(12)
    qkv = torch.empty((nl, c.batch, 3 * E), dtype=torch.float16, device=dev).normal_(0, 1, generator=g)
    q = [qkv[i, :, :E].view(c.batch, c.num_heads, c.head_size) for i in range(nl)]
    k = [qkv[i, :, E:2 * E].view(c.batch, c.num_heads, c.head_size) for i in range(nl)]
    v = [qkv[i, :, 2 * E:].view(c.batch, c.num_heads, c.head_size) for i in range(nl)]
    rows = qkv.view(nl * c.batch, 3 * E)
    k_all = rows[:, E:2 * E].view(nl * c.batch, c.num_heads, c.head_size)
    v_all = rows[:, 2 * E:].view(nl * c.batch, c.num_heads, c.head_size)
    slots_all = torch.cat(wl.slots)
    out = torch.empty((c.batch, c.num_heads, c.head_size), dtype=torch.float16, device=dev)

    def token_pair():
        for i in range(nl):
            cache_ops.reshape_and_cache(k[i], v[i], wl.key_cache, wl.value_cache, wl.slots[i], "auto", 1.0)
            ops.paged_attention_v1(out, q[i], wl.key_cache, wl.value_cache, c.kv_heads, wl.scale, wl.tables[i], wl.seq_lens,
                                   c.block_size, c.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0)

    def token_deferred():
        for i in range(nl):
            ops.paged_attention_v1_append(out, q[i], k[i], v[i], wl.key_cache, wl.value_cache, c.kv_heads, wl.scale, wl.tables[i],
                                          wl.seq_lens, c.block_size, c.seq_len, write_cache=False)
        cache_ops.reshape_and_cache(k_all, v_all, wl.key_cache, wl.value_cache, slots_all, "auto", 1.0)

    def scatter_only():
        cache_ops.reshape_and_cache(k_all, v_all, wl.key_cache, wl.value_cache, slots_all, "auto", 1.0)

    n = max(8, min(args.steps, 200) // nl)

    def timed(fn, reps):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps

    def graphed(fn):
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            fn()
        return gr

    res = {}
    for name, fn in (("pair", token_pair), ("deferred", token_deferred)):
        t_plain = [timed(fn, n) for _ in range(3)]
        gr = graphed(fn)
        t_graph = [timed(gr.replay, n) for _ in range(3)]
        res[name] = {"us_per_layer_step": statistics.median(t_plain) / nl * 1e6,
                     "graph_us_per_layer_step": statistics.median(t_graph) / nl * 1e6,
                     "us_per_layer_step_regions": [t / nl * 1e6 for t in t_plain],
                     "graph_us_per_layer_step_regions": [t / nl * 1e6 for t in t_graph]}
        del gr
    vname = ops.variant_names()[ops.last_variant() - 1]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record()
        scatter_only()
        b.record()
    torch.cuda.synchronize(dev)
    scat_us = statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3
    nbytes = alg_bytes(c, "auto")
    best = min(res["deferred"]["us_per_layer_step"], res["deferred"]["graph_us_per_layer_step"])
    rec = {"op": "12 layers of one token: 12 x paged_attention_v1 over cache + this step's rows (append-read), then ONE "
                 "reshape_and_cache over the 12 x 256 rows — against 12 x the reference's call pair",
           "layers": nl, "tokens_timed": n, **{f"{k}_{kk}": vv for k, r in res.items() for kk, vv in r.items()},
           "one_scatter_of_12_layers_us": scat_us, "kernel_variant": vname,
           "value": c.batch / (best * 1e-6), "unit": "tokens/s per layer step", "ms_per_step": best * 1e-3,
           "algorithmic_bytes_per_launch": nbytes, "step_frac_of_hbm_peak": nbytes / (best * 1e-6) / 1e9 / HBM_PEAK_GBPS,
           "note": "per layer step = token time / 12; `value` from the faster of plain launches and one hipGraph per token.  The "
                   "attention's output is bit-identical in both forms and the caches end equal (tests/test_serve_gpu.py, "
                   "tests/test_append_balanced_gpu.py); what moves is WHEN the newest token's 24 partial lines per (sequence, "
                   "head) are written: once per token instead of in front of every layer's attention"}
    del wl, qkv
    torch.cuda.empty_cache()
    return rec


def random_fp8_codes(shape, dev, gen):
    """Random E4M3 / E5M2 codes 0..63 + sign: magnitude < 2 in E4M3 (exponent field <= 7) and in E5M2 (<= 15), no NaNs."""
    return (torch.randint(0, 64, shape, dtype=torch.uint8, device=dev, generator=gen)
            | (torch.randint(0, 2, shape, dtype=torch.uint8, device=dev, generator=gen) << 7))


def main(argv=None):
    global KV_DTYPE, SKIP_RESHAPE, RESHAPE_OTHER, NOOP_RESHAPE
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if needs_self_launch(args):
        sys.exit(self_launch(args, argv))
    if args.kv == "fp8_e5m2":           # E5M2 pages are outside the hot path: libvmi_paged_attention_extras.so (a probe option)
        from vllmini_amd import _lib
        _lib.use_extras().__enter__()
    RESHAPE_OTHER = args.reshape_other_set
    if not args.standin_cpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path exists for the product kernels)")
    dist, rank, world, local_rank, dev = init_dist(args)
    if args.standin_cpu:
        standin_main(args, dist, rank, world, dev)
        return
    from vllmini_amd import ops

    if args.noop_instead_of_reshape:
        NOOP_RESHAPE = torch.zeros(1, device=dev)
    cfg = CONFIGS[args.config]
    e2e_cfg = CONFIGS["cfg3"]
    if (world > 1 or dist is not None) and args.config == "cfg3":
        # BASELINE.json configs[4]: batch 2048 over 8 GPUs = 256 sequences per GPU (the cfg3 shape) with a
        # per-GPU KV pool of 65536 blocks.  Same kernel work per GPU as N=1; only the pool is larger.
        cfg = CONFIGS["cfg5"]
        if args.scaling == "strong":
            # total batch 2048 whatever N; N = 1 needs 2 x 131072 blocks for two disjoint table sets,
            # more than the stated 65536 -> the pool is max(65536, needed)
            total = 2048
            if total % world:
                raise SystemExit(f"--scaling strong: {total} sequences do not divide over {world} GPUs")
            b_ = total // world
            cfg = dataclasses.replace(cfg, name="cfg5_strong", batch=b_,
                                      num_blocks=max(cfg.num_blocks, 2 * b_ * cfg.blocks_per_seq))
            e2e_cfg = dataclasses.replace(e2e_cfg, batch=b_)
    if args.variant_name:
        args.variant = ops.variant_names().index(args.variant_name) + 1
    if args.pv_mfma:
        ops.set_pv_mfma(True)
    if args.kv_heads:
        args.no_cpu_baseline = True      # the eager CPU baseline is written for the multi-head BASELINE configs
        cfg = dataclasses.replace(cfg, name=f"{cfg.name}_kv{args.kv_heads}", num_kv_heads=args.kv_heads)
    if args.batch or args.seq_len:
        b_ = args.batch or cfg.batch
        l_ = args.seq_len or cfg.seq_len
        per = -(-l_ // cfg.block_size)
        cfg = dataclasses.replace(cfg, name=f"{cfg.name}_b{b_}_l{l_}", batch=b_, seq_len=l_,
                                  num_blocks=max(2 * b_ * per, 64))
    if args.serve:
        run_serve(args, dev, rank)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.e2e:
        run_e2e(args, cfg, dist, rank, world, dev)
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
        args.no_fused = args.no_cpu_baseline = args.no_cfg4 = args.no_e2e = args.no_cfg2 = args.no_strong = args.no_long = True
        args.no_deferred = args.no_serve = True
        gk = torch.Generator(device=dev).manual_seed(99 + rank)
        kshape = (cfg.num_blocks, cfg.kv_heads, cfg.head_size // 16, cfg.block_size, 16)
        vshape = (cfg.num_blocks, cfg.kv_heads, cfg.head_size, cfg.block_size)
        del wl.key_cache, wl.value_cache
        torch.cuda.empty_cache()
        wl.key_cache = random_fp8_codes(kshape, dev, gk)
        wl.value_cache = random_fp8_codes(vshape, dev, gk)

    SKIP_RESHAPE = args.skip_reshape
    plain = args.op == "v1" and args.kv == "auto" and not args.ragged and not args.ragged_sorted   # the BASELINE line
    if args.hint_mean and not args.variant and args.op in ("v1", "fused"):
        lens_h = wl.seq_lens.cpu()
        args.variant = ops.pick_variant(cfg.batch, cfg.num_heads, cfg.head_size, int(lens_h.max()), cfg.block_size,
                                        mean_seq_len=int(lens_h.float().mean()), fp8=FP8_ARG[args.kv])

    # ---- the headline: K steps, nothing else in the timed region -------------------------------------------------------
    # EXACTLY K steps per timed region (each in the full bracket); a short region (the driver's 20 steps are 2.6 ms) is repeated
    # until the regions hold --min-timed-ms in all, and the line reports the MEDIAN region — every region is on the line
    regions = [shard.max_over_ranks(time_steps(wl, out, args.steps, args.warmup, args.variant, dist, dev, op=args.op), dist, dev)]
    while sum(regions) * 1e3 < args.min_timed_ms and len(regions) < 99:     # (max over ranks: every rank sees the same sums)
        regions.append(shard.max_over_ranks(time_steps(wl, out, args.steps, 0, args.variant, dist, dev, op=args.op), dist, dev))
    elapsed = statistics.median(regions)
    tokens = cfg.batch * world * args.steps          # one new token per sequence per step
    ms_per_step = elapsed / args.steps * 1e3
    # ---- the attention kernel's duration: its own pass ------------------------------------------------------------------
    ks = kernel_stats(kernel_pass(wl, out, args.kernel_samples, args.variant, dev, op=args.op), dist, dev)
    line_bytes = alg_bytes(cfg)
    if args.ragged or args.ragged_sorted:      # the bytes that exist in THIS batch (lengths differ), not the config's
        tok = int(wl.seq_lens.sum().item())
        line_bytes = 2 * tok * cfg.kv_heads * cfg.head_size * (1 if args.kv.startswith("fp8") else 2) + \
            2 * cfg.batch * cfg.num_heads * cfg.head_size * 2 + \
            int(((wl.seq_lens + cfg.block_size - 1) // cfg.block_size).sum().item()) * 4 + cfg.batch * 4
    achieved = line_bytes / (ks["median"] * 1e-6) / 1e9
    # the variant the library actually launched (it knows the launch's kv_scale, the pick queries do not)
    vid = args.variant or (ops.last_variant() if args.op in ("v1", "fused", "newest") else 0)
    vname = ops.variant_names()[vid - 1] if vid else f"paged_attention_v2 variant {args.variant or 'auto'}"
    traffic, traffic_src = (pmc_traffic(cfg.name + {"auto": "", "fp8": "_fp8", "fp8_e5m2": "_fp8_e5m2"}[args.kv], vname) if args.op == "v1" else
                            pmc_traffic(cfg.name + "_" + args.op, vname) if args.op in ("fused", "newest") else (None, None))
    line = {
        "metric": "decode_tokens_per_sec_paged_attention_v1_per_layer",
        "value": tokens / elapsed,
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "timed_region_ms": sum(regions) * 1e3,
        "timed_regions": len(regions),
        "ms_per_step_regions": [r / args.steps * 1e3 for r in regions],
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
            "launch": "self-launched ranks" if os.environ.get("VMI_BENCH_SELF_LAUNCHED") == "1" else
                      ("external launcher" if dist is not None else "single process"),
        },
        "paged_attention_v1_us_per_step": ks["median"],
        "paged_attention_v1_us_median": ks["median"],
        "paged_attention_v1_us_mean": ks["mean"],
        "paged_attention_v1_us_min": ks["min"],
        "paged_attention_v1_us_median_per_rank": ks["median_per_rank"],
        "kernel_event_samples": args.kernel_samples,
        "library_sha16": library_sha16(),
        "committed_profile_kernel_us": rocprof_kernel_us(cfg.name if cfg.name != "cfg5" else "cfg3", vname)[0] if args.op == "v1" and args.kv == "auto" and not args.ragged else None,
        "method_version": 3,
        "method": "v3 (round 3 on): event-free timed region; " + shard.TIMING_BRACKET + "; for N > 1 the token all_gather runs "
                  f"once per token (every {EXCHANGE_EVERY}th layer step), asynchronously on the process group's stream.  v2 (rounds "
                  "1-2): a blocking all_gather on EVERY step, clock stopped behind the closing barrier — for N > 1 the line carries "
                  "that figure too (`legacy_method_step`), so round-over-round comparisons need not mix method and kernel changes",
        "timing_bracket": shard.TIMING_BRACKET,
        "kernel_event_pass": "separate from the timed region: one HIP event pair around every attention launch of "
                             f"{args.kernel_samples} call pairs; roofline.achieved uses the median",
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel": "pa_q_kernel" if vname.startswith(("q_d", "bf16_q_d", "fp8_q_d")) else "pa_v1_kernel",
            "algorithmic_bytes_per_launch": line_bytes,
            **({"mfma": "q.K^T of the K pass", "mfma_note": "fp8 pages: the decoded K tile is the B operand of v_mfma_f32_16x16x32_f16, q "
                "the A operand with 16 equal rows — 1 of 16 result rows is useful work, the matrix pipe is otherwise idle; "
                "p.V stays on the VALU (profiles/r03b_k_pass_on_mfma.md)"} if vname.endswith("m") and "_q_d" in vname else
               {"mfma": 0, "mfma_note": "multi-head decode is M = 1 per KV head: both contractions are GEMVs, the kernel "
                                        "issues no MFMA (SQ_INSTS_MFMA = 0, profiles/pmc_cfg3_latest.json)"}),
        },
    }
    if line.get("committed_profile_kernel_us"):
        # what the HIP event pair around a launch reads above the kernel's own duration in the committed rocprofv3 pass of this
        # binary (dispatch-to-dispatch spacing, event records): the instrumentation share of roofline.achieved's denominator
        line["event_minus_rocprof_us"] = ks["median"] - line["committed_profile_kernel_us"]
    if dist is not None:
        line["token_exchange_us"] = shard.max_over_ranks(exchange_pass(args.kernel_samples, dist, dev), dist, dev)
        line["token_exchange"] = (f"all_gather of {cfg.batch} int64 ids per rank ({cfg.batch * world * 8} B in all), backend nccl "
                                  f"(RCCL); inside the timed region once per token = on every {EXCHANGE_EVERY}th layer step "
                                  f"(steps 0, {EXCHANGE_EVERY}, ...), issued ASYNCHRONOUSLY on the process group's stream — a rank "
                                  "samples its own sequences' ids, the next token's layers do not wait for the gathered vector — "
                                  "and waited for one token later and before the closing synchronise; token_exchange_us = median "
                                  f"of {args.kernel_samples} BLOCKING exchanges alone, by HIP events")
        line["token_exchange_every_steps"] = EXCHANGE_EVERY
        line["rccl_ranks"] = dist.get_world_size()          # ranks in the process group (backend "nccl" = RCCL on ROCm)
        line["process_group_backend"] = dist.get_backend()
        line["rank_placement"] = shard.gather_objects(_PLACEMENT, dist)     # per rank: device uuid / PCI id, NUMA node, bound cores
        # the SAME K steps the way rounds 1-2 measured them: blocking exchange on every step, clock behind the barrier
        l_elapsed = shard.max_over_ranks(time_steps(wl, out, args.steps, args.warmup, args.variant, dist, dev, op=args.op, legacy=True),
                                         dist, dev)
        line["legacy_method_step"] = {"method_version": 2, "ms_per_step": l_elapsed / args.steps * 1e3, "value": tokens / l_elapsed,
                                      "unit": "tokens/s", "timing_bracket": shard.TIMING_BRACKET_LEGACY,
                                      "token_exchange": "blocking all_gather on every step"}
    if args.op == "v1" and not args.no_fused:
        # the same step as ONE launch (vmi_paged_attention_v1_append_f16: bit-identical caches and out,
        # tests/test_parity_gpu.py); reported beside `value`, which stays the reference's two-op call pair
        rec, _ = pair_record(wl, out, args, args.variant, dist, dev, tokens, alg_bytes(cfg), op="fused")
        line["fused_step"] = {"op": "paged_attention_v1_append (reshape_and_cache + paged_attention_v1, one launch)", **rec}
    if plain and not args.no_fp8:
        # the same step over an fp8 E4M3 KV cache (kv_cache_dtype "fp8"): half the K/V bytes.
        # A different data format, so it is reported beside `value`, never as it.
        k16, v16 = wl.key_cache, wl.value_cache
        gk = torch.Generator(device=dev).manual_seed(99 + rank)
        kshape = (cfg.num_blocks, cfg.kv_heads, cfg.head_size // 16, cfg.block_size, 16)
        wl.key_cache = random_fp8_codes(kshape, dev, gk)
        wl.value_cache = random_fp8_codes(v16.shape, dev, gk)
        KV_DTYPE = "fp8"
        try:
            rec, _ = pair_record(wl, out, args, 0, dist, dev, tokens, alg_bytes(cfg, "fp8"))
            rec["kernel_variant"] = ops.variant_names()[ops.last_variant() - 1]
        finally:
            KV_DTYPE = "auto"
            wl.key_cache, wl.value_cache = k16, v16
        line["fp8_kv_step"] = {"op": "reshape_and_cache + paged_attention_v1, kv_cache_dtype='fp8' (E4M3), kv_scale 1.0", **rec}
    if plain and not args.no_graph and dist is None and not args.skip_reshape:
        # what a decode loop captures: the 12 layers' call pairs of one token in ONE hipGraph (gpt2_decode.py replays the whole
        # step that way).  The record is THAT form; one pair per graph is kept only as the measured cost of a graph launch.
        n_g = max(args.steps, 24)
        g12 = graph_steps(wl, out, n_g, args.variant, dev, per_graph=12)
        g1 = graph_steps(wl, out, args.steps, args.variant, dev)
        line["graph_step"] = {"op": "reshape_and_cache + paged_attention_v1, a token's 12 layer call pairs replayed from ONE hipGraph",
                              "value": cfg.batch * n_g / g12, "unit": "tokens/s", "ms_per_step": g12 / n_g * 1e3,
                              "pairs_per_graph": 12,
                              "one_pair_per_graph_ms_per_step": g1 / args.steps * 1e3,
                              "plain_launches_ms_per_step": ms_per_step,
                              "graph_minus_plain_us_per_step": (g12 / n_g * 1e3 - ms_per_step) * 1e3,
                              "note": "graph_minus_plain_us_per_step is what replaying the pair from a hipGraph costs (+) or saves (-) "
                                      "against plain launches: the call pair is NOT launch-bound (its two kernels run 125 us, the host "
                                      "issues them in ~25), so a graph has no host time to hide here, and on this ROCm a graph's kernel "
                                      "nodes are spaced a little wider than stream launches (a few us per pair); one pair per graph adds "
                                      f"the graph launch itself: {(g1 / args.steps - g12 / n_g) * 1e6:.1f} us more per step than 12 pairs per "
                                      "graph.  Where graphs pay is a step whose HOST work exceeds its kernels: the end-to-end decode "
                                      "harness (e2e_step replays the whole token from one graph) and cfg2_step.graph_us_per_pair"}
    if plain and not args.no_ragged and not args.variant and not args.sequential_tables:
        # the same call pair, same default entry (no hint, no variant), on a RAGGED batch: seq_lens ~ U{1..seq_len}
        pools = (wl.key_cache, wl.value_cache)
        rwl = make_workload(cfg, dev, seed=4321 + rank, table_sets=2, ragged=True)
        del rwl.key_cache, rwl.value_cache
        rwl.key_cache, rwl.value_cache = pools          # same pools: only tables, lengths and slots differ
        tok = int(rwl.seq_lens.sum().item())
        r_bytes = 2 * tok * cfg.kv_heads * cfg.head_size * 2 + 2 * cfg.batch * cfg.num_heads * cfg.head_size * 2 + \
            int(((rwl.seq_lens + cfg.block_size - 1) // cfg.block_size).sum().item()) * 4 + cfg.batch * 4
        rec, _ = pair_record(rwl, out, args, 0, dist, dev, tokens, r_bytes)
        rec["kernel_variant"] = ops.variant_names()[ops.last_variant() - 1]
        line["ragged_step"] = {"op": "reshape_and_cache + paged_attention_v1, default entry, seq_lens ~ U{1..%d}" % cfg.seq_len, **rec}
        del rwl
    cpu_wl = None if (args.no_cpu_baseline or rank != 0) else wl
    if cpu_wl is not None:   # host copies now: the device pools are released before the larger sub-records allocate theirs
        cpu_wl = type("HostWorkload", (), {"cfg": cfg, "key_cache": wl.key_cache.cpu(), "value_cache": wl.value_cache.cpu(),
                                           "query": wl.query.cpu(), "tables": [wl.tables[0].cpu()], "scale": wl.scale})()
    del wl
    torch.cuda.empty_cache()
    if plain and not args.no_cfg4 and args.config == "cfg3" and not args.variant:
        # BASELINE configs[3]: 32 heads x 128, batch 128, seq 2048 — the same call pair through the same default entry
        c4 = CONFIGS["cfg4"]
        wl4 = make_workload(c4, dev, seed=77 + rank, table_sets=2)
        out4 = torch.empty((c4.batch, c4.num_heads, c4.head_size), dtype=torch.float16, device=dev)
        rec, _ = pair_record(wl4, out4, args, 0, dist, dev, c4.batch * world * args.steps, alg_bytes(c4, "auto"))
        v4 = ops.variant_names()[ops.last_variant() - 1]
        rec["launch"] = ops.last_launch_label()
        t4, t4src = pmc_traffic("cfg4", v4)
        line["cfg4_step"] = {"op": "reshape_and_cache + paged_attention_v1, BASELINE configs[3]: batch 128/GPU, seq_len 2048, "
                                   "32 heads x 128, block_size 16, num_blocks 32768, fp16", **rec,
                             "kernel_variant": v4, "traffic": t4, "traffic_source": t4src, "mfma": 0,
                             "note": "kernel_us covers the gated double launch (lockstep 4-heads-per-wave kernel + balanced "
                                     "kernel; one of them leaves at once, DESIGN.md §3.7).  BASELINE.json labels this config "
                                     "'MFMA QK^T path': with one query row per KV head the contraction is a GEMV and the "
                                     "kernel issues NO MFMA; the matrix cores are used by the grouped-query kernels only"}
        del wl4, out4
        torch.cuda.empty_cache()
    def guarded(key, fn):
        """A sub-record that fails must not take the headline with it (N = 1; with several ranks an exception on one of them
        would leave the others inside a collective, so there it propagates): the record says what went wrong instead."""
        if dist is not None:
            line[key] = fn()
            return
        try:
            line[key] = fn()
        except Exception as e:       # noqa: BLE001 — reported on the line, never hidden
            line[key] = {"error": f"{type(e).__name__}: {e}"[:500]}
            torch.cuda.empty_cache()

    if plain and not args.no_cfg2 and args.config == "cfg3" and not args.variant:
        guarded("cfg2_step", lambda: cfg2_record(args, dist, rank, world, dev))
    if plain and not args.no_strong and args.config == "cfg3" and not args.variant and dist is None:
        guarded("cfg5_strong_n1", lambda: strong_n1_record(args, dev))
    if plain and not args.no_deferred and args.config == "cfg3" and not args.variant and dist is None:
        guarded("deferred_scatter_step", lambda: deferred_scatter_record(args, dev))
    if plain and not args.no_e2e and args.config == "cfg3" and not args.variant:
        # every step appends a token: a long timed region would measure a longer context than the ~1 k the record is quoted on
        # (the default 200 + 20 steps end at 1 230 tokens: +22 % bytes in the last step) — the sub-record runs at most 24 + 6 steps
        e2e_args = argparse.Namespace(**{**vars(args), "steps": min(args.steps, 24), "warmup": min(args.warmup, 6)})
        args_main, args = args, e2e_args
        res = e2e_measure(args, e2e_cfg, dist, rank, world, dev, ctx0=args.e2e_context)
        line["e2e_step"] = {k: res[k] for k in ("metric", "value", "unit", "ms_per_step", "timed_regions_ms_per_step", "stall_suspected",
                                                "value_is", "context", "batch_per_gpu", "note",
                                                "operators_ms_per_step", "operators_share_of_step", "operators_note")}
        # the same model with the harness's own fused step (one launch instead of the reference's call pair per layer — bit-
        # identical, tests/test_parity_gpu.py): the reference surface stays the pair, the harness may use what is faster
        resf = e2e_measure(args, e2e_cfg, dist, rank, world, dev, ctx0=args.e2e_context, fused=True, operator_share=False)
        line["e2e_step"]["fused_append"] = {k: resf[k] for k in ("value", "unit", "ms_per_step", "note")}
        # ... and with the scatter deferred to one launch per token (round 6: the append-read kernels; same logits, same caches)
        resd = e2e_measure(args, e2e_cfg, dist, rank, world, dev, ctx0=args.e2e_context, deferred=True, operator_share=False)
        line["e2e_step"]["deferred_scatter"] = {k: resd[k] for k in ("value", "unit", "ms_per_step", "timed_regions_ms_per_step", "note")}
        # ... and with the block's linear layers left to the torch modules (rounds 1 - 4's harness), for the comparison
        rest = e2e_measure(args, e2e_cfg, dist, rank, world, dev, ctx0=args.e2e_context, operator_share=False, native_layers=False)
        line["e2e_step"]["torch_module_layers"] = {k: rest[k] for k in ("value", "unit", "ms_per_step", "note")}
        # ... and with the reference's sampling (scheduler.py:144-153: top-k 50, one multinomial draw) in place of the argmax:
        # one launch of this build against the torch chain (topk, softmax, multinomial, gather)
        resk = e2e_measure(args, e2e_cfg, dist, rank, world, dev, ctx0=args.e2e_context, operator_share=False, sampler="top_k")
        reskt = e2e_measure(args, e2e_cfg, dist, rank, world, dev, ctx0=args.e2e_context, operator_share=False, sampler="top_k_torch")
        line["e2e_step"]["top_k_50_sampling"] = {"value": resk["value"], "unit": "tokens/s", "ms_per_step": resk["ms_per_step"],
                                                 "torch_chain_ms_per_step": reskt["ms_per_step"]}
        line["e2e_step"]["steps"], line["e2e_step"]["warmup"] = args.steps, args.warmup
        if dist is None:
            # the regime the reference's scheduler runs (one sequence per step, scheduler.py:60): a token's latency
            b1 = dataclasses.replace(e2e_cfg, batch=1)
            r1 = e2e_measure(args, b1, dist, rank, world, dev, ctx0=args.e2e_context, operator_share=False)   # (the call pair)
            r1s = e2e_measure(args, b1, dist, rank, world, dev, ctx0=args.e2e_context, operator_share=False, scatter_in_c_attn=True)
            r1k = e2e_measure(args, b1, dist, rank, world, dev, ctx0=args.e2e_context, operator_share=False, sampler="top_k")
            r1kt = e2e_measure(args, b1, dist, rank, world, dev, ctx0=args.e2e_context, operator_share=False, sampler="top_k_torch")
            r1t = e2e_measure(args, b1, dist, rank, world, dev, ctx0=args.e2e_context, operator_share=False, native_layers=False)
            line["e2e_step"]["batch_1"] = {"us_per_token": r1["ms_per_step"] * 1e3, "tokens_per_s": r1["value"],
                                           "scatter_in_c_attn_us_per_token": r1s["ms_per_step"] * 1e3,
                                           "torch_module_layers_us_per_token": r1t["ms_per_step"] * 1e3,
                                           "top_k_50_sampling_us_per_token": r1k["ms_per_step"] * 1e3,
                                           "top_k_50_sampling_torch_chain_us_per_token": r1kt["ms_per_step"] * 1e3,
                                           "note": "ONE sequence at the same context, the whole token from one hipGraph; greedy "
                                                   "unless named; top_k_50 = the reference's sampling (scheduler.py:144-153)"}
            # BASELINE configs[1] as a decode step: batch 32 at ~512 tokens (the pool's contexts start 24 tokens short of it)
            c2 = dataclasses.replace(e2e_cfg, batch=32)
            r2 = e2e_measure(args, c2, dist, rank, world, dev, ctx0=480, operator_share=False)
            r2t = e2e_measure(args, c2, dist, rank, world, dev, ctx0=480, operator_share=False, native_layers=False)
            line["e2e_step"]["cfg2_shape"] = {"batch": 32, "context": 480, "tokens_per_s": r2["value"],
                                              "us_per_token": r2["ms_per_step"] * 1e3,
                                              "torch_module_layers_us_per_token": r2t["ms_per_step"] * 1e3}
        args = args_main
    if plain and not args.no_long and args.config == "cfg3" and not args.variant and dist is None:
        guarded("long_context_step", lambda: long_context_record(args, dev))
    if plain and not args.no_serve and args.config == "cfg3" and not args.variant and dist is None:
        # the continuous-batching scheduler itself: a bounded trace inside the default run, 2048 with --serve
        guarded("serve_step", lambda: serve_measure(args, dev, rank, requests=min(args.serve_requests, 1024)))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # after the GPU work and after the process group is gone: the other ranks have exited, the host is this rank's
        line["cpu_baseline"] = None if cpu_wl is None else cpu_baseline(cpu_wl, args.cpu_seconds)
        emit_line(line)


if __name__ == "__main__":
    main()
