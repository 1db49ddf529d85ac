"""Synthetic decode workloads in the reference KV layout (BASELINE.json `configs`).

Pools are filled once with random fp16; every sequence owns DISTINCT physical blocks drawn from
a random permutation, so page gathers are non-sequential, as a live paged allocator produces
them (vllmini/kv_cache.py:16,60,83: FIFO free list with frees appended out of order).

Used by bench.py and the GPU parity tests.  Pure torch; no oracle, no reference.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional

import torch

BLOCK_SIZE = 16
X = 8


@dataclasses.dataclass(frozen=True)
class DecodeConfig:
    name: str
    batch: int          # sequences per GPU
    num_heads: int
    head_size: int
    seq_len: int
    num_blocks: int     # pool size per GPU
    block_size: int = BLOCK_SIZE
    num_kv_heads: int = 0   # 0 = num_heads (multi-head attention); smaller = grouped-query attention

    @property
    def kv_heads(self) -> int:
        return self.num_kv_heads or self.num_heads

    @property
    def blocks_per_seq(self) -> int:
        return (self.seq_len + self.block_size - 1) // self.block_size

    def algorithmic_bytes(self) -> int:
        """2*B*H*L*D*2 (K,V) + 2*B*H*D*2 (q,out) + B*ceil(L/16)*4 (tables) + B*4 (lens)."""
        b, h, d = self.batch, self.num_heads, self.head_size
        return 2 * b * self.kv_heads * self.seq_len * d * 2 + 2 * b * h * d * 2 + b * self.blocks_per_seq * 4 + b * 4

    def flops(self) -> int:
        return 4 * self.batch * self.num_heads * self.seq_len * self.head_size


# BASELINE.json configs[1..4] (configs[0] is the CPU plumbing case, see tests/)
CONFIGS = {
    "cfg1": DecodeConfig("cfg1", 1, 12, 64, 32, 64),
    "cfg2": DecodeConfig("cfg2", 32, 12, 64, 512, 4096),
    "cfg3": DecodeConfig("cfg3", 256, 12, 64, 1024, 32768),
    "cfg4": DecodeConfig("cfg4", 128, 32, 128, 2048, 32768),   # 2 disjoint table sets of 16384 blocks
    "cfg5": DecodeConfig("cfg5", 256, 12, 64, 1024, 65536),
    # not in BASELINE.json: long context, small batch — the shape split-KV (paged_attention_v2) exists for
    "long": DecodeConfig("long", 4, 32, 128, 16384, 8192),
    "b1": DecodeConfig("b1", 1, 12, 64, 1024, 256),
    # BASELINE configs[4] on ONE GPU (the N = 1 point of the strong-scaling curve): all 2048 sequences, two table sets
    "cfg5_strong": DecodeConfig("cfg5_strong", 2048, 12, 64, 1024, 262144),
    # few sequences x long contexts: where a caller-owned workspace lets paged_attention_v1 spread a head over many CUs
    "long_b1": DecodeConfig("long_b1", 1, 12, 64, 16384, 4096),
    "long_b4": DecodeConfig("long_b4", 4, 12, 64, 8192, 8192),
    "long_gqa": DecodeConfig("long_gqa", 4, 32, 128, 8192, 8192, num_kv_heads=8),   # Llama-3-8B-shaped heads (grouped-query)
    # contexts past what several waves' logits fit in one workgroup's LDS (~27 000 tokens), more items than are resident: in rounds
    "long_32k": DecodeConfig("long_32k", 48, 12, 64, 32768, 196608),
}


@dataclasses.dataclass
class DecodeWorkload:
    cfg: DecodeConfig
    key_cache: torch.Tensor      # [NB, H, D/8, bs, 8] fp16
    value_cache: torch.Tensor    # [NB, H, D, bs]      fp16
    qkv: torch.Tensor            # [B, 3*H*D] fp16 — fused projection output, as gpt2.py:35-36 produces
    tables: list                 # list of int32 [B, MB] block tables over disjoint block sets
    seq_lens: torch.Tensor       # int32 [B]
    slots: list                  # list of int64 [B]: slot of the newest token under tables[i]
    scale: float

    @property
    def query(self) -> torch.Tensor:  # strided views, row stride 3*H*D (gpt2.py:35-41)
        c = self.cfg
        return self.qkv[:, : c.num_heads * c.head_size].view(c.batch, c.num_heads, c.head_size)

    @property
    def key(self) -> torch.Tensor:
        c = self.cfg
        hd, kd = c.num_heads * c.head_size, c.kv_heads * c.head_size
        return self.qkv[:, hd: hd + kd].view(c.batch, c.kv_heads, c.head_size)

    @property
    def value(self) -> torch.Tensor:
        c = self.cfg
        hd, kd = c.num_heads * c.head_size, c.kv_heads * c.head_size
        return self.qkv[:, hd + kd: hd + 2 * kd].view(c.batch, c.kv_heads, c.head_size)


def make_workload(cfg: DecodeConfig, device: torch.device | str, seed: int = 0, table_sets: int = 2,
                  ragged: bool = False, max_blocks_per_seq: Optional[int] = None,
                  kv_dist: str = "uniform") -> DecodeWorkload:
    """Allocate and fill pools + metadata on `device`.

    table_sets > 1 builds that many block tables over DISJOINT slices of the pool so that
    consecutive timed steps do not re-read what the previous step left in the 256 MiB
    Infinity Cache (SURVEY.md §7 hard part 3).
    """
    dev = torch.device(device)
    g = torch.Generator(device="cpu").manual_seed(seed)
    c = cfg
    mb = max_blocks_per_seq or c.blocks_per_seq
    need = c.batch * c.blocks_per_seq
    if need * table_sets > c.num_blocks:
        table_sets = max(1, c.num_blocks // need)
    if need > c.num_blocks:
        raise ValueError(f"{c.name}: needs {need} blocks, pool has {c.num_blocks}")

    kshape = (c.num_blocks, c.kv_heads, c.head_size // X, c.block_size, X)
    vshape = (c.num_blocks, c.kv_heads, c.head_size, c.block_size)
    gd = torch.Generator(device=dev).manual_seed(seed) if dev.type == "cuda" else g
    if kv_dist == "uniform":
        key_cache = torch.empty(kshape, dtype=torch.float16, device=dev).uniform_(-1, 1, generator=gd)
        value_cache = torch.empty(vshape, dtype=torch.float16, device=dev).uniform_(-1, 1, generator=gd)
    else:
        key_cache = torch.empty(kshape, dtype=torch.float16, device=dev).normal_(0, 1, generator=gd)
        value_cache = torch.empty(vshape, dtype=torch.float16, device=dev).normal_(0, 1, generator=gd)
    qkv = torch.empty((c.batch, (c.num_heads + 2 * c.kv_heads) * c.head_size), dtype=torch.float16,
                      device=dev).normal_(0, 1, generator=gd)

    if ragged:
        lens = torch.randint(1, c.seq_len + 1, (c.batch,), generator=g, dtype=torch.int32)
        lens[0] = c.seq_len
        if ragged == "sorted":      # longest first: what a length-aware caller could arrange (diagnostic)
            lens = torch.sort(lens, descending=True).values
    else:
        lens = torch.full((c.batch,), c.seq_len, dtype=torch.int32)

    per_set = c.num_blocks // table_sets
    tables, slots = [], []
    for t in range(table_sets):
        perm = torch.randperm(per_set, generator=g)[:need].to(torch.int32) + t * per_set
        tab = torch.full((c.batch, mb), -1, dtype=torch.int32)
        tab[:, : c.blocks_per_seq] = perm.view(c.batch, c.blocks_per_seq)
        last = (lens.to(torch.int64) - 1)
        blk = tab[torch.arange(c.batch), (last // c.block_size)].to(torch.int64)
        slot = blk * c.block_size + last % c.block_size
        # blocks past each sequence's length are never read: poison them with -1 like the reference
        nblk = (lens + c.block_size - 1) // c.block_size
        col = torch.arange(mb).view(1, mb)
        tab = torch.where(col < nblk.view(-1, 1), tab, torch.full_like(tab, -1))
        tables.append(tab.to(dev))
        slots.append(slot.to(dev))
    return DecodeWorkload(cfg=c, key_cache=key_cache, value_cache=value_cache, qkv=qkv, tables=tables,
                          seq_lens=lens.to(dev), slots=slots, scale=1.0 / math.sqrt(c.head_size))
