"""Sharding of the decode path over the GPUs of one node.

The path partitions into independent units: a sequence touches only its own KV pages, and the
operator's grid is (heads, sequences) with no cross-sequence reduction (attention_kernels.cu:734).
So: one process per GPU, every rank owns a PRIVATE KV pool, free list and block tables, and a
contiguous slice of the global batch.  There is NO collective on the data path.  The only exchange a
decode loop needs is the per-token hand-back of sampled token ids (8 bytes per sequence) — an
all_gather over `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests), issued
asynchronously on the process group's stream so that it runs beside the next token's layers (gather_token_ids_async).
The reference has no distributed layer at all (last row).

Timing helpers implement bench.py's contract: W untimed warm-up steps, then exactly K steps between
a barrier + device synchronise on both sides, and the MAX over ranks.
"""
from __future__ import annotations

import time
from typing import Callable, Optional, Sequence, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced slice [lo, hi) of n_items for `rank` (first n_items % world ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} not in [0, {world})")
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def owner_of(index: int, n_items: int, world: int) -> int:
    """Rank whose shard_range contains `index`."""
    q, r = divmod(n_items, world)
    boundary = r * (q + 1)
    if index < boundary:
        return index // (q + 1)
    return r + (index - boundary) // q if q else world - 1


def barrier_sync(dist, sync: Optional[Callable[[], None]]) -> None:
    if sync is not None:
        sync()
    if dist is not None:
        dist.barrier()
    if sync is not None:
        sync()


TIMING_BRACKET = ("K steps between (device synchronise, barrier, device synchronise) on both sides; a rank's clock stops when ITS "
                  "device has finished the K steps, in front of the closing barrier; the MAX over ranks is reported")
TIMING_BRACKET_LEGACY = "the same bracket with the clock stopped BEHIND the closing barrier (rounds 1-2)"


def timed_steps(step: Callable[[int], None], steps: int, warmup: int, dist=None,
                sync: Optional[Callable[[], None]] = None,
                timed_step: Optional[Callable[[int], None]] = None, clock_behind_barrier: bool = False) -> float:
    """Run `warmup` untimed then exactly `steps` timed calls of step(i); returns local elapsed seconds.
    `sync` = device synchronise (torch.cuda.synchronize on GPU ranks, None on CPU).  `timed_step`, if
    given, replaces `step` inside the timed region (same work plus per-launch event records).

    The K steps are bracketed by (synchronise, barrier, synchronise) on both sides.  The clock starts after the opening
    bracket and stops when THIS rank's device has finished its K steps — after the closing bracket's first synchronise, in
    front of its barrier: the barrier's own latency (an RCCL all-reduce plus a host wake-up, ~90 us: 3.5 % of twenty 130-us
    steps) is not work of the path, and the caller takes the MAX over ranks, which is what a clock stopped behind the barrier
    would read without it."""
    for i in range(warmup):
        step(i)
    barrier_sync(dist, sync)
    t0 = time.perf_counter()
    body = timed_step or step
    for i in range(steps):
        body(i)
    if sync is not None:
        sync()
    elapsed = time.perf_counter() - t0
    barrier_sync(dist, sync)
    if clock_behind_barrier:      # rounds 1-2: the closing barrier's own latency inside the reported time (TIMING_BRACKET_LEGACY)
        elapsed = time.perf_counter() - t0
    return elapsed


def all_ranks(value: float, dist=None, device: torch.device | str = "cpu") -> list:
    """`value` of every rank, in rank order (a [world] all_gather of one float64): per-rank figures beside the max."""
    if dist is None:
        return [float(value)]
    world = dist.get_world_size()
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    got = torch.empty(world, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(got, mine)
    return [float(x) for x in got.cpu()]


def max_over_ranks(value: float, dist=None, device: torch.device | str = "cpu") -> float:
    if dist is None:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_token_ids(local_ids: torch.Tensor, global_batch: int, dist=None,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """all_gather the int64 ids each rank sampled for its slice into the global [B] order (8 bytes per sequence:
    latency-bound, nowhere near a 153 GB/s xGMI link).  When the batch divides evenly over the ranks — the bench's
    and the scheduler's normal case — it is ONE collective into `out` (a preallocated [global_batch] int64 tensor,
    optional); otherwise slices differ by one element and every rank pads to the largest."""
    if dist is None:
        return local_ids
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(global_batch, rank, world)
    assert local_ids.dtype == torch.int64 and local_ids.numel() == hi - lo
    if global_batch % world == 0:
        if out is None:
            out = torch.empty(global_batch, dtype=torch.int64, device=local_ids.device)
        elif out.shape != (global_batch,) or out.dtype != torch.int64 or out.device != local_ids.device or \
                not out.is_contiguous():
            raise ValueError(f"out must be a contiguous int64 [{global_batch}] tensor on {local_ids.device}")
        dist.all_gather_into_tensor(out, local_ids.contiguous())
        return out
    width = -(-global_batch // world)
    padded = torch.zeros(width, dtype=torch.int64, device=local_ids.device)
    padded[: hi - lo] = local_ids
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded)
    parts = []
    for r in range(world):
        a, b = shard_range(global_batch, r, world)
        parts.append(gathered[r][: b - a])
    return torch.cat(parts)


def gather_token_ids_async(local_ids: torch.Tensor, global_batch: int, dist, out: torch.Tensor):
    """The same all_gather issued WITHOUT making the compute stream wait for it: -> (out, work).  Every rank samples the
    ids of its own sequences, so a rank's next decode step does not depend on the gathered vector (the
    gather is the hand-back to the front end) — the collective runs on the process group's own stream, behind the kernels
    already enqueued on the compute stream and BESIDE the next token's layers; `work.wait()` (a stream-side wait, no host
    block on RCCL) belongs in front of whatever reads `out` or reuses it, at the latest one token later.  Even batches only
    (global_batch % world == 0), `out` preallocated: nothing is allocated on the way."""
    world = dist.get_world_size()
    if global_batch % world or out.shape != (global_batch,) or out.dtype != torch.int64 or not out.is_contiguous() or \
            local_ids.dtype != torch.int64 or local_ids.numel() != global_batch // world or not local_ids.is_contiguous():
        raise ValueError(f"gather_token_ids_async: contiguous int64 [{global_batch // max(world, 1)}] ids of an even batch "
                         f"into a contiguous int64 [{global_batch}] tensor")
    return out, dist.all_gather_into_tensor(out, local_ids, async_op=True)


# ---- where a rank runs: host cores near its GPU ------------------------------------------------------------------------
# A decode loop spends ~25 us of host time per 128-us call pair; eight such loops on a 2-socket host whose scheduler is free
# to move them across sockets is where "linear by construction" fails first.  One process per GPU, bound to cores of the NUMA
# node its GPU hangs off (the driver's /sys tree names it); where the node is unknown (containers that hide /sys, the CPU
# stand-in) the process's current core set is dealt evenly to the local ranks.

def _parse_cpulist(text: str) -> list:
    cores = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cores.extend(range(int(a), int(b or a) + 1))
    return cores


def gpu_numa_node(pci_bus_id: str, sysfs: str = "/sys") -> int:
    """NUMA node of the PCI function `dddd:bb:dd.f` (-1: unknown)."""
    try:
        with open(f"{sysfs}/bus/pci/devices/{pci_bus_id.lower()}/numa_node") as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def cores_for_rank(local_rank: int, local_world: int, numa_node: int = -1, allowed=None, sysfs: str = "/sys") -> list:
    """Host cores rank `local_rank` of `local_world` should run on: the allowed cores of its GPU's NUMA node, dealt evenly to
    the ranks whose GPUs share that node as far as this function can know — i.e. sliced by local_rank among local_world —
    or, with the node unknown, an even slice of all allowed cores.  Never empty."""
    import os

    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    pool = allowed
    if numa_node >= 0:
        try:
            with open(f"{sysfs}/devices/system/node/node{numa_node}/cpulist") as f:
                node = set(_parse_cpulist(f.read()))
            near = [c for c in allowed if c in node]
            if near:
                pool = near
        except OSError:
            pass
    lo, hi = shard_range(len(pool), local_rank % max(local_world, 1), max(local_world, 1))
    mine = pool[lo:hi]
    return mine or pool


def place_rank(local_rank: int, local_world: int, device_index=None, bind: bool = True) -> dict:
    """Bind this process to the cores cores_for_rank() gives it and describe where it runs: what the N > 1 bench line
    carries per rank (device UUID, PCI bus id, NUMA node, cores)."""
    import os

    info = {"local_rank": local_rank, "pci_bus_id": None, "uuid": None, "numa_node": -1}
    if device_index is not None and torch.cuda.is_available():
        p = torch.cuda.get_device_properties(device_index)
        bus = getattr(p, "pci_bus_id", None)
        if isinstance(bus, int):        # torch exposes domain / bus / device as integers
            bus = f"{getattr(p, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(p, 'pci_device_id', 0):02x}.0"
        info["pci_bus_id"] = bus
        info["uuid"] = str(getattr(p, "uuid", "")) or None
        if bus:
            info["numa_node"] = gpu_numa_node(bus)
    cores = cores_for_rank(local_rank, local_world, info["numa_node"])
    if bind:
        try:
            os.sched_setaffinity(0, cores)
            info["bound"] = True
        except OSError:
            info["bound"] = False
    else:
        info["bound"] = False
    info["cores"] = f"{cores[0]}-{cores[-1]}" if cores == list(range(cores[0], cores[-1] + 1)) else ",".join(map(str, cores))
    info["n_cores"] = len(cores)
    return info


def gather_objects(obj, dist=None) -> list:
    """`obj` of every rank in rank order (outside any timed region)."""
    if dist is None:
        return [obj]
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, obj)
    return got


def shard_rows(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_range(x.shape[0], rank, world)
    return x[lo:hi]
