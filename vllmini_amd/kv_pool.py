"""Paged KV pool bookkeeping — host-side mirror of the reference's layout/metadata producers.

Reference:
    KVCache.__init__              vllmini/kv_cache.py:6-19     cache tensors in the kernel layout + FIFO free list
    KVCache.allocate_for_prefill  vllmini/kv_cache.py:21-37    ONE block per layer, table [[blk,-1,...]], slots arange+blk*bs
    KVCache.append_block          vllmini/kv_cache.py:56-73    pop(0) from the free list into the first -1 entry
    KVCache.free                  vllmini/kv_cache.py:81-86    blocks appended back to the free list
    BlockManager.decode_step      vllmini/block_manager.py:28-63  per layer: slot = last_block*bs + filled; new block when full

The reference keeps one int32 [1, max_blocks] table PER (sequence, layer) on the device and scans
it element by element (a device->host sync per element, block_manager.py:36-39).  Here the same
state lives in host numpy arrays — tables[layer][seq_row] — and one [B, max_blocks] int32 tensor and
one [B] int64 slot tensor per layer are produced per decode step for a whole BATCH of sequences,
which is the shape paged_attention_v1 / reshape_and_cache are batched over.  For any single sequence
the block ids, their order, the slots and the free-list order are exactly the reference's
(tests/test_kv_pool.py replays the reference's own allocator trace from tests/golden/seam_trace.npz).

Reference quirks that are reproduced or flagged, not silently changed:
  * prefill hands out exactly one block per layer, so prompts longer than block_size are rejected
    here (the reference would silently write into the next block, which belongs to another layer:
    kv_cache.py:25-35);
  * decode_step needs a trailing -1 in the table to find the last block (block_manager.py:36-39), so a
    sequence can use at most max_blocks_per_seq-1 blocks per layer; exceeding that raises here
    (the reference dies with UnboundLocalError);
  * exhaustion raises RuntimeError with the reference's messages (kv_cache.py:22-23, 57-58).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

X = 8  # halves per 16-byte K chunk (kv_cache.py:13: head_size // 8, ..., 8)


class PagedKVPool:
    def __init__(self, num_blocks: int, num_heads: int, head_size: int, block_size: int,
                 max_blocks_per_seq: int, num_layers: int, device: torch.device | str = "cuda",
                 allocate_tensors: bool = True):
        self.num_blocks = num_blocks
        self.num_heads = num_heads
        self.head_size = head_size
        self.block_size = block_size
        self.max_blocks_per_seq = max_blocks_per_seq
        self.num_layers = num_layers
        self.device = torch.device(device)
        if allocate_tensors:
            # kv_cache.py:13-14 — ONE pool shared by all layers
            self.key_cache = torch.zeros(num_blocks, num_heads, head_size // X, block_size, X,
                                         dtype=torch.float16, device=self.device)
            self.value_cache = torch.zeros(num_blocks, num_heads, head_size, block_size,
                                           dtype=torch.float16, device=self.device)
        else:  # bookkeeping only (CPU tests)
            self.key_cache = self.value_cache = None
        self.free_blocks: List[int] = list(range(num_blocks))          # kv_cache.py:16, FIFO
        self.allocated_blocks: Dict[int, List[int]] = {}               # kv_cache.py:17
        # per sequence: [layers, max_blocks] int32 table (-1 padded) and per-layer fill of the last block
        self._tables: Dict[int, np.ndarray] = {}
        self._nblocks: Dict[int, np.ndarray] = {}                      # [layers] blocks in use
        self._filled: Dict[int, np.ndarray] = {}                       # [layers] tokens in the last block
        self._seq_len: Dict[int, int] = {}

    # ---- reference-compatible per-sequence operations -------------------------------------------
    def allocate_for_prefill(self, seq_id: int, seq_len: int) -> Tuple[List[int], np.ndarray, np.ndarray]:
        """-> (allocated block ids, slot mappings [layers, seq_len] int64, tables [layers, MB] int32)."""
        if seq_id in self.allocated_blocks:
            raise ValueError(f"sequence {seq_id} already allocated")
        if len(self.free_blocks) < self.num_layers:
            raise RuntimeError("Not enough free blocks for prefill allocation")   # kv_cache.py:22-23
        if seq_len > self.block_size:
            raise RuntimeError(
                f"prefill of {seq_len} tokens does not fit the single block per layer the reference "
                f"allocates (block_size={self.block_size}; kv_cache.py:25-35)")
        allocated = self.free_blocks[: self.num_layers]                # kv_cache.py:25-26
        self.free_blocks = self.free_blocks[self.num_layers:]
        self.allocated_blocks[seq_id] = list(allocated)
        tab = np.full((self.num_layers, self.max_blocks_per_seq), -1, dtype=np.int32)
        tab[:, 0] = allocated                                          # kv_cache.py:31
        self._tables[seq_id] = tab
        self._nblocks[seq_id] = np.ones(self.num_layers, dtype=np.int64)
        self._filled[seq_id] = np.full(self.num_layers, min(seq_len, self.block_size), dtype=np.int64)  # :30
        self._seq_len[seq_id] = seq_len
        slots = (np.arange(seq_len, dtype=np.int64)[None, :] +
                 np.asarray(allocated, dtype=np.int64)[:, None] * self.block_size)   # kv_cache.py:35
        return list(allocated), slots, tab.copy()

    def decode_step(self, seq_id: int, input_len: int = 1) -> Tuple[np.ndarray, np.ndarray]:
        """One new token for one sequence -> (tables [layers, MB] int32, slots [layers] int64).
        Layer by layer, like block_manager.py:34-59 (so a shared free list hands blocks out in the
        reference's order)."""
        tab, nb, filled = self._tables[seq_id], self._nblocks[seq_id], self._filled[seq_id]
        slots = np.empty(self.num_layers, dtype=np.int64)
        for layer in range(self.num_layers):
            if nb[layer] >= self.max_blocks_per_seq:
                # the reference finds the last block by looking for the first -1 (block_manager.py:36-39)
                raise RuntimeError(
                    f"sequence {seq_id} layer {layer}: table of {self.max_blocks_per_seq} entries is full; "
                    "the reference needs a trailing -1 (block_manager.py:36-39)")
            last_block = int(tab[layer, nb[layer] - 1])
            num_filled = int(filled[layer])
            if num_filled == self.block_size:                           # block_manager.py:48-53
                if len(self.free_blocks) == 0:
                    raise RuntimeError("No free blocks available")      # kv_cache.py:57-58
                new_block = self.free_blocks.pop(0)                     # kv_cache.py:60
                self.allocated_blocks[seq_id].append(new_block)
                tab[layer, nb[layer]] = new_block                       # kv_cache.py:64-70
                nb[layer] += 1
                last_block, num_filled = new_block, 0
            slots[layer] = last_block * self.block_size + num_filled    # block_manager.py:55
            filled[layer] = num_filled + input_len                      # block_manager.py:59
        self._seq_len[seq_id] += input_len
        return tab.copy(), slots

    def free(self, seq_id: int) -> None:
        if seq_id in self.allocated_blocks:                             # kv_cache.py:81-86
            self.free_blocks.extend(self.allocated_blocks[seq_id])
            del self.allocated_blocks[seq_id]
            del self._tables[seq_id], self._nblocks[seq_id], self._filled[seq_id], self._seq_len[seq_id]

    def seq_len(self, seq_id: int) -> int:
        return self._seq_len[seq_id]

    # ---- batched view: what the batched operators consume -----------------------------------------
    def decode_step_batch(self, seq_ids: Sequence[int]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """One new token for EVERY sequence in `seq_ids` (in that order) ->
             tables  [layers, B, MB] int32, slots [layers, B] int64, context_lens [B] int32
        context_lens counts the new token: the token written at `slots` attends to itself.  (The
        reference passes the length BEFORE the new token, scheduler.py:96 vs block_manager.py:55 — an
        off-by-one in its caller that makes the newest token invisible; pass context_lens-1 as
        seq_lens to reproduce that.)"""
        B = len(seq_ids)
        tables = np.empty((self.num_layers, B, self.max_blocks_per_seq), dtype=np.int32)
        slots = np.empty((self.num_layers, B), dtype=np.int64)
        lens = np.empty(B, dtype=np.int32)
        for i, sid in enumerate(seq_ids):
            t, s = self.decode_step(sid, 1)
            tables[:, i, :] = t
            slots[:, i] = s
            lens[i] = self._seq_len[sid]
        return tables, slots, lens

    def upload(self, tables: np.ndarray, slots: np.ndarray, lens: np.ndarray):
        """Host metadata -> device tensors in the dtypes the operators require (int32 / int64 / int32)."""
        return (torch.from_numpy(tables).to(self.device, non_blocking=True),
                torch.from_numpy(slots).to(self.device, non_blocking=True),
                torch.from_numpy(lens).to(self.device, non_blocking=True))
