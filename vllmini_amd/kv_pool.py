"""Paged KV pool bookkeeping — host-side mirror of the reference's layout/metadata producers.

Reference:
    KVCache.__init__              vllmini/kv_cache.py:6-19     cache tensors in the kernel layout + FIFO free list
    KVCache.allocate_for_prefill  vllmini/kv_cache.py:21-37    ONE block per layer, table [[blk,-1,...]], slots arange+blk*bs
    KVCache.append_block          vllmini/kv_cache.py:56-73    pop(0) from the free list into the first -1 entry
    KVCache.free                  vllmini/kv_cache.py:81-86    blocks appended back to the free list
    BlockManager.decode_step      vllmini/block_manager.py:28-63  per layer: slot = last_block*bs + filled; new block when full
    BlockManager.swap_to_cpu / swap_from_cpu   vllmini/block_manager.py:70-87  a sequence's blocks to host memory and back

The reference keeps one int32 [1, max_blocks] table PER (sequence, layer) on the device and scans
it element by element (a device->host sync per element, block_manager.py:36-39).  Here the same
state lives in host numpy arrays with one ROW per live sequence —
    tables [layers, rows, max_blocks] int32,  nblocks / filled [layers, rows]
— and a decode step for a whole batch is a handful of vectorised numpy operations that emit
one [layers, B, max_blocks] int32 table tensor and one [layers, B] int64 slot tensor, the shapes
paged_attention_v1 / reshape_and_cache are batched over.  No device round trip is involved.

For any single sequence the block ids, their order, the slots and the free-list order are exactly
the reference's (tests/test_kv_pool.py replays the reference's own allocator trace from
tests/golden/seam_trace.npz); for a batch, new blocks are handed out in the order the reference
would use if it stepped the sequences one after the other (sequence-major, then layer).

Reference quirks that are reproduced or flagged, not silently changed:
  * the reference's prefill hands out exactly one block per layer and silently writes a longer prompt into the next
    block, which belongs to another layer (kv_cache.py:25-35).  Here a prompt gets ceil(len / block_size)
    blocks per layer (`multi_block_prefill=True`, the default since round 6: a serving loop cannot be fed 16-token
    prompts); up to block_size tokens the blocks, tables and slots are the reference's own (the golden allocator traces
    replay under either setting).  `multi_block_prefill=False` pins the reference's limit: longer prompts are refused;
  * decode_step needs a trailing -1 in the table to find the last block (block_manager.py:36-39), so a
    sequence can use at most max_blocks_per_seq-1 blocks per layer before the next step fails; that
    raises here (the reference dies with UnboundLocalError);
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
                 allocate_tensors: bool = True, max_seqs: int = 64, multi_block_prefill: bool = True,
                 kv_cache_dtype: str = "auto", kv_scale: float = 1.0, host_blocks: int = 0):
        self.num_blocks = num_blocks
        self.num_heads = num_heads
        self.head_size = head_size
        self.block_size = block_size
        self.max_blocks_per_seq = max_blocks_per_seq
        self.num_layers = num_layers
        self.device = torch.device(device)
        self.multi_block_prefill = multi_block_prefill
        # "auto": fp16 pages (kv_cache.py:13-14); "fp8" / "fp8_e5m2": E4M3 / E5M2 byte pages in the x = 16 layout, half the bytes
        # (the reference surface's kv_cache_dtype / kv_scale, passed through to both operators)
        self.kv_cache_dtype, self.kv_scale = kv_cache_dtype, float(kv_scale)
        if allocate_tensors and kv_cache_dtype == "fp8_e5m2":
            # E5M2 pages are outside the hot path (include/vmi_paged_attention_extras.h): say so HERE, not at the first launch
            from . import _lib
            _lib.require_extras("PagedKVPool(kv_cache_dtype='fp8_e5m2'): operators over fp8-E5M2 pages")
        if allocate_tensors:
            # kv_cache.py:13-14 — ONE pool shared by all layers
            x, dt = (16, torch.uint8) if kv_cache_dtype in ("fp8", "fp8_e4m3", "fp8_e5m2") else (X, torch.float16)
            self.key_cache = torch.zeros(num_blocks, num_heads, head_size // x, block_size, x,
                                         dtype=dt, device=self.device)
            self.value_cache = torch.zeros(num_blocks, num_heads, head_size, block_size,
                                           dtype=dt, device=self.device)
        else:  # bookkeeping only (CPU tests)
            self.key_cache = self.value_cache = None
        self.free_blocks: List[int] = list(range(num_blocks))          # kv_cache.py:16, FIFO
        self.allocated_blocks: Dict[int, List[int]] = {}               # kv_cache.py:17
        # preemption (block_manager.py:15 cpu_cache): a pinned host pool in the SAME block layout, allocated at the first
        # swap_out; host_blocks = 0 -> as many as the device pool.  swapped[seq_id] = everything needed to put the sequence back
        self.host_blocks = int(host_blocks) or num_blocks
        self.host_key_cache = self.host_value_cache = None
        self._host_free: List[int] = list(range(self.host_blocks))
        self.swapped: Dict[int, dict] = {}
        self.swap_stats = {"swap_outs": 0, "swap_ins": 0, "blocks_out": 0, "blocks_in": 0, "bytes_out": 0, "bytes_in": 0}
        self._row_of: Dict[int, int] = {}
        self._free_rows: List[int] = []
        self._rows = 0
        self._alloc_rows(max_seqs)

    # ---- row storage -----------------------------------------------------------------------------
    def _alloc_rows(self, n: int) -> None:
        L, MB = self.num_layers, self.max_blocks_per_seq
        tables = np.full((L, n, MB), -1, dtype=np.int32)
        nblocks = np.zeros((L, n), dtype=np.int64)
        filled = np.zeros((L, n), dtype=np.int64)
        seq_len = np.zeros(n, dtype=np.int64)
        if self._rows:
            tables[:, : self._rows] = self._tables
            nblocks[:, : self._rows] = self._nblocks
            filled[:, : self._rows] = self._filled
            seq_len[: self._rows] = self._seq_len
        self._free_rows.extend(range(self._rows, n))
        self._tables, self._nblocks, self._filled, self._seq_len = tables, nblocks, filled, seq_len
        self._rows = n

    def _new_row(self, seq_id: int) -> int:
        if not self._free_rows:
            self._alloc_rows(max(2 * self._rows, 1))
        row = self._free_rows.pop(0)
        self._row_of[seq_id] = row
        return row

    def _take(self, n: int) -> List[int]:
        taken = self.free_blocks[:n]                                    # kv_cache.py:25-26 / pop(0) :60
        del self.free_blocks[:n]
        return taken

    # ---- reference-compatible per-sequence operations -------------------------------------------
    def allocate_for_prefill(self, seq_id: int, seq_len: int) -> Tuple[List[int], np.ndarray, np.ndarray]:
        """-> (allocated block ids, slot mappings [layers, seq_len] int64, tables [layers, MB] int32)."""
        if seq_id in self.allocated_blocks:
            raise ValueError(f"sequence {seq_id} already allocated")
        bs, L = self.block_size, self.num_layers
        per_layer = 1
        if seq_len > bs:
            if not self.multi_block_prefill:
                raise RuntimeError(
                    f"prefill of {seq_len} tokens does not fit the single block per layer the reference "
                    f"allocates (block_size={bs}; kv_cache.py:25-35)")
            per_layer = -(-seq_len // bs)
            if per_layer > self.max_blocks_per_seq - 1:
                raise RuntimeError(f"prefill of {seq_len} tokens needs {per_layer} blocks per layer, table "
                                   f"holds {self.max_blocks_per_seq - 1} usable entries")
        if len(self.free_blocks) < L * per_layer:
            raise RuntimeError("Not enough free blocks for prefill allocation")   # kv_cache.py:22-23
        allocated = self._take(L * per_layer)
        self.allocated_blocks[seq_id] = list(allocated)
        row = self._new_row(seq_id)
        # layer l gets blocks [l, L + l, 2L + l, ...]: the first L are the reference's one-per-layer
        blocks = np.ascontiguousarray(np.asarray(allocated, dtype=np.int32).reshape(per_layer, L).T)  # [L, per_layer]
        self._tables[:, row, :] = -1
        self._tables[:, row, :per_layer] = blocks                       # kv_cache.py:31
        self._nblocks[:, row] = per_layer
        self._filled[:, row] = seq_len - (per_layer - 1) * bs           # kv_cache.py:30 (min(seq_len, bs))
        self._seq_len[row] = seq_len
        pos = np.arange(seq_len, dtype=np.int64)
        slots = np.ascontiguousarray(blocks.astype(np.int64)[:, pos // bs] * bs + pos % bs)   # kv_cache.py:35
        return list(allocated), slots, self._tables[:, row, :].copy()

    def decode_step(self, seq_id: int, input_len: int = 1) -> Tuple[np.ndarray, np.ndarray]:
        """One new token for one sequence -> (tables [layers, MB] int32, slots [layers] int64)
        (block_manager.py:28-63)."""
        tables, slots, _ = self._step_rows(np.array([self._row_of[seq_id]]), [seq_id], input_len)
        return tables[:, 0, :], slots[:, 0]

    def free(self, seq_id: int) -> None:
        if seq_id in self.allocated_blocks:                             # kv_cache.py:81-86
            self.free_blocks.extend(self.allocated_blocks[seq_id])
            del self.allocated_blocks[seq_id]
            row = self._row_of.pop(seq_id)
            self._tables[:, row, :] = -1
            self._nblocks[:, row] = 0
            self._free_rows.append(row)

    # ---- preemption: a sequence's blocks to host memory and back (block_manager.py:70-87) -------------------------------
    @property
    def block_bytes(self) -> int:
        """Bytes of one block of ONE cache (K and V blocks have the same size)."""
        esz = 1 if self.kv_cache_dtype in ("fp8", "fp8_e4m3", "fp8_e5m2") else 2
        return self.num_heads * self.head_size * self.block_size * esz

    def reserve_host(self) -> None:
        """Allocate the pinned host pool now (else at the first swap_out: pinning host_blocks x 2 x block_bytes takes a
        fraction of a second per GB — a server does it at start-up, not under its first preemption)."""
        if self.key_cache is not None and self.host_key_cache is None:
            pin = self.device.type == "cuda"    # pinned: the GPU reads / writes these pages directly (one launch per swap)
            self.host_key_cache = torch.empty((self.host_blocks,) + tuple(self.key_cache.shape[1:]),
                                              dtype=self.key_cache.dtype, pin_memory=pin)
            self.host_value_cache = torch.empty((self.host_blocks,) + tuple(self.value_cache.shape[1:]),
                                                dtype=self.value_cache.dtype, pin_memory=pin)

    def _move_blocks(self, to_host: bool, pairs: np.ndarray) -> None:
        if self.key_cache is None:          # bookkeeping only (CPU tests)
            return
        from . import cache_ops
        self.reserve_host()
        dev, host = (self.key_cache, self.value_cache), (self.host_key_cache, self.host_value_cache)
        src, dst = (dev, host) if to_host else (host, dev)
        cache_ops.swap_blocks_batched(src[0], src[1], dst[0], dst[1], torch.from_numpy(np.ascontiguousarray(pairs)))

    def swap_out(self, seq_id: int) -> int:
        """swap_to_cpu (block_manager.py:70-73): copy the sequence's blocks — every layer's — to the host pool, then free
        them.  ONE launch moves them all (cache_ops.swap_blocks_batched; the reference: one .cpu() gather per cache),
        asynchronously on the current stream: the freed blocks may be handed out at once, launches that overwrite them
        queue behind the copy.  Returns the number of blocks moved; RuntimeError when the host pool cannot take them."""
        if seq_id not in self.allocated_blocks:
            raise ValueError(f"No KV cache allocated for sequence {seq_id}")            # kv_cache.py:50-51
        blocks = self.allocated_blocks[seq_id]
        n = len(blocks)
        if len(self._host_free) < n:
            raise RuntimeError(f"Not enough free host blocks to swap out sequence {seq_id} ({n} needed, "
                               f"{len(self._host_free)} of {self.host_blocks} free)")
        host = self._host_free[:n]
        del self._host_free[:n]
        row = self._row_of[seq_id]
        self.swapped[seq_id] = {"host": host, "dev": list(blocks), "tables": self._tables[:, row, :].copy(),
                                "nblocks": self._nblocks[:, row].copy(), "filled": self._filled[:, row].copy(),
                                "seq_len": int(self._seq_len[row])}
        self._move_blocks(True, np.stack([np.asarray(blocks, dtype=np.int64), np.asarray(host, dtype=np.int64)], axis=1))
        self.free(seq_id)
        st = self.swap_stats
        st["swap_outs"] += 1
        st["blocks_out"] += n
        st["bytes_out"] += 2 * n * self.block_bytes
        return n

    def swap_in(self, seq_id: int) -> bool:
        """swap_from_cpu (block_manager.py:75-87): False when the sequence is not swapped out or the pool has not enough
        free blocks (the reference answers its RuntimeError the same way); else the sequence owns as many NEW blocks
        (free-list order), its tables name them in the old positions, and the pages are back — bit for bit."""
        st = self.swapped.get(seq_id)
        if st is None:
            return False                                                    # :76-77
        n = len(st["host"])
        if len(self.free_blocks) < n:
            return False                                                    # :86-87
        new = self._take(n)
        lut = np.full(self.num_blocks, -1, dtype=np.int32)
        lut[np.asarray(st["dev"], dtype=np.int64)] = np.asarray(new, dtype=np.int32)
        row = self._new_row(seq_id)
        t = st["tables"]
        self._tables[:, row, :] = np.where(t >= 0, lut[np.maximum(t, 0)], -1)
        self._nblocks[:, row] = st["nblocks"]
        self._filled[:, row] = st["filled"]
        self._seq_len[row] = st["seq_len"]
        self.allocated_blocks[seq_id] = list(new)
        self._move_blocks(False, np.stack([np.asarray(st["host"], dtype=np.int64), np.asarray(new, dtype=np.int64)], axis=1))
        self._host_free.extend(st["host"])
        del self.swapped[seq_id]
        ss = self.swap_stats
        ss["swap_ins"] += 1
        ss["blocks_in"] += n
        ss["bytes_in"] += 2 * n * self.block_bytes
        return True

    def drop_swapped(self, seq_id: int) -> None:
        """Forget a swapped-out sequence (block_manager.py:65-68: free() also deletes the cpu copy)."""
        st = self.swapped.pop(seq_id, None)
        if st is not None:
            self._host_free.extend(st["host"])

    def blocks_of(self, seq_id: int) -> int:
        """Blocks a sequence holds on the device, or would need to come back."""
        if seq_id in self.allocated_blocks:
            return len(self.allocated_blocks[seq_id])
        return len(self.swapped[seq_id]["host"])

    def seq_len(self, seq_id: int) -> int:
        return int(self._seq_len[self._row_of[seq_id]])

    def table(self, seq_id: int) -> np.ndarray:
        return self._tables[:, self._row_of[seq_id], :].copy()

    # ---- batched step: what the batched operators consume -----------------------------------------
    def _step_rows(self, rows: np.ndarray, seq_ids: Sequence[int], input_len: int):
        bs, MB = self.block_size, self.max_blocks_per_seq
        nb = self._nblocks[:, rows]                                     # [L, B]
        filled = self._filled[:, rows]
        if (nb >= MB).any():
            # the reference finds the last block by looking for the first -1 (block_manager.py:36-39)
            l, b = np.argwhere(nb >= MB)[0]
            raise RuntimeError(
                f"sequence {seq_ids[b]} layer {l}: table of {MB} entries is full; "
                "the reference needs a trailing -1 (block_manager.py:36-39)")
        need = filled == bs                                             # block_manager.py:48
        n_new = int(need.sum())
        if n_new:
            # reference order: sequence by sequence, layer by layer within a sequence
            order = np.argwhere(need.T)                                 # rows of (b, l), b-major
            if n_new > len(self.free_blocks):
                # hand out what the reference would have handed out before failing
                order = order[: len(self.free_blocks)]
            new_blocks = np.asarray(self._take(len(order)), dtype=self._tables.dtype)
            if len(order):
                b_idx, l_idx = order[:, 0], order[:, 1]                     # (sequence, layer) pairs are distinct
                r_idx = rows[b_idx]
                self._tables[l_idx, r_idx, self._nblocks[l_idx, r_idx]] = new_blocks   # kv_cache.py:64-70
                self._nblocks[l_idx, r_idx] += 1
                self._filled[l_idx, r_idx] = 0
                # per-sequence ownership lists, in hand-out order (order is sequence-major)
                cuts = np.flatnonzero(np.diff(b_idx)) + 1
                for seg_b, seg in zip(b_idx[np.r_[0, cuts]], np.split(new_blocks, cuts)):
                    self.allocated_blocks[seq_ids[int(seg_b)]].extend(int(x) for x in seg)
            if len(order) < n_new:
                raise RuntimeError("No free blocks available")          # kv_cache.py:57-58
            nb = self._nblocks[:, rows]
            filled = self._filled[:, rows]
        tables = self._tables[:, rows, :]                               # [L, B, MB] (copy: fancy index)
        last = np.take_along_axis(tables, (nb - 1)[:, :, None], axis=2)[:, :, 0].astype(np.int64)
        slots = last * bs + filled                                      # block_manager.py:55
        self._filled[:, rows] = filled + input_len                      # block_manager.py:59
        self._seq_len[rows] += input_len
        return tables, slots, self._seq_len[rows].astype(np.int32)

    def decode_step_batch(self, seq_ids: Sequence[int]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """One new token for EVERY sequence in `seq_ids` (in that order) ->
             tables  [layers, B, MB] int32, slots [layers, B] int64, context_lens [B] int32
        context_lens counts the new token: the token written at `slots` attends to itself.  (The
        reference passes the length BEFORE the new token, scheduler.py:96 vs block_manager.py:55 — an
        off-by-one in its caller that makes the newest token invisible; pass context_lens-1 as
        seq_lens to reproduce that.)"""
        rows = np.fromiter((self._row_of[s] for s in seq_ids), dtype=np.int64, count=len(seq_ids))
        return self._step_rows(rows, seq_ids, 1)

    def upload(self, tables: np.ndarray, slots: np.ndarray, lens: np.ndarray):
        """Host metadata -> device tensors in the dtypes the operators require (int32 / int64 / int32)."""
        return (torch.from_numpy(np.ascontiguousarray(tables)).to(self.device, non_blocking=True),
                torch.from_numpy(np.ascontiguousarray(slots)).to(self.device, non_blocking=True),
                torch.from_numpy(np.ascontiguousarray(lens)).to(self.device, non_blocking=True))
