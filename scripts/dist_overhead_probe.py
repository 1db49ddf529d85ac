"""Where do the ~5 us per call pair go when a process group exists?  One process, one GPU: the cfg3 call pair timed (event-
free loop of 240 steps, and the median of 60 event pairs around the attention launch) BEFORE torch.distributed is
initialised, AFTER init_process_group("nccl") with no collective issued, after one all_reduce, and after
destroy_process_group.  `python scripts/dist_overhead_probe.py`"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import paged_attention_cuda as ext  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg = CONFIGS["cfg3"]
wl = make_workload(cfg, dev, seed=3, table_sets=2)
out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)


def pair(i, ev=None):
    t = i % 2
    ext.cache_ops.reshape_and_cache(wl.key, wl.value, wl.key_cache, wl.value_cache, wl.slots[t], "auto", 1.0)
    if ev:
        ev[0].record()
    ext.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[t], wl.seq_lens,
                           cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0)
    if ev:
        ev[1].record()


def measure(tag):
    for i in range(30):
        pair(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(240):
        pair(i)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 240 * 1e6
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for i, ev in enumerate(evs):
        pair(i, ev)
    torch.cuda.synchronize()
    k = statistics.median(a.elapsed_time(b) for a, b in evs) * 1e3
    t0 = time.perf_counter()
    for i in range(240):          # host cost of the launches alone: how long the loop takes to ENQUEUE 240 pairs
        pair(i)
    host = (time.perf_counter() - t0) / 240 * 1e6
    torch.cuda.synchronize()
    print(f"{tag:46s} step {step:6.1f} us   attention by events {k:6.1f} us   host enqueue {host:5.1f} us/pair", flush=True)


measure("no process group")
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
measure("after init_process_group(nccl)")
x = torch.ones(4, device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
measure("after one all_reduce")
ids = torch.arange(256, dtype=torch.int64, device=dev)
o = torch.empty(256, dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(o, ids)
torch.cuda.synchronize()
measure("after one all_gather_into_tensor")
dist.barrier()
torch.cuda.synchronize()
measure("after barrier()")
dist.destroy_process_group()
measure("after destroy_process_group")
