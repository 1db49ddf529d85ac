"""Host-side KV bookkeeping mirror vs the allocator trace of the reference's own KVCache/BlockManager
(tests/golden/seam_trace.npz, captured by importing the reference Python — gen_golden.py)."""
from __future__ import annotations

import json
import os

import numpy as np
import pytest

from vllmini_amd.kv_pool import PagedKVPool


def _pool(meta, **kw):
    return PagedKVPool(meta["num_blocks"], meta["num_heads"], meta["head_size"], meta["block_size"],
                       meta["max_blocks_per_seq"], meta["num_layers"], device="cpu", allocate_tensors=False, **kw)


def test_replays_reference_allocator_trace(golden_dir):
    z = np.load(os.path.join(golden_dir, "seam_trace.npz"))
    meta = json.loads(str(z["meta"]))
    pool = _pool(meta)
    allocated, slots, tables = pool.allocate_for_prefill(7, meta["prompt_len"])
    assert np.array_equal(np.array(pool.free_blocks), z["prefill_free_blocks"])
    assert np.array_equal(tables, z["prefill_block_tables"][:, 0, :])
    # the prefill reshape_and_cache calls of the trace carry the reference's slot mappings (kv_cache.py:35)
    for layer in range(meta["num_layers"]):
        assert str(z[f"call{layer:04d}/op"]) == "reshape_and_cache"
        assert np.array_equal(slots[layer], z[f"call{layer:04d}/slot_mapping"])
    call = meta["num_layers"]
    for step in range(meta["num_decode_steps"]):
        tab, slot = pool.decode_step(7, 1)
        assert np.array_equal(tab, z[f"alloc{step:03d}/block_tables"][:, 0, :]), step
        assert np.array_equal(slot, z[f"alloc{step:03d}/slots"]), step
        assert np.array_equal(np.array(pool.free_blocks), z[f"alloc{step:03d}/free_blocks"]), step
        # and these are the very arguments the reference passed to the ops in that step
        for layer in range(meta["num_layers"]):
            assert np.array_equal(z[f"call{call:04d}/slot_mapping"], slot[layer:layer + 1])
            assert np.array_equal(z[f"call{call + 1:04d}/block_tables"][0], tab[layer])
            call += 2
    # the reference passes the length BEFORE the new token (scheduler.py:96): last call saw max_length-1
    assert int(z[f"call{call - 1:04d}/seq_lens"][0]) == meta["max_length"] - 1 == pool.seq_len(7) - 1
    pool.free(7)
    assert pool.free_blocks == meta["final_free_blocks"]


def test_capacity_trace_the_mirror_stops_where_the_reference_stops(golden_dir):
    """capacity_trace.npz: the reference's Scheduler with max_length above the capacity of a block-table row.  The reference
    completes 44 decode steps — the 44th appends the row's last block — and dies in the 45th (block_manager.py:36-39 finds no
    -1 in a full row: UnboundLocalError); the mirror hands out the same tables, slots and free lists step by step and refuses
    the same step with a RuntimeError that says why.  What the kernel saw on the way: seq_len <= 48 under max_seq_len = 64
    (scheduler.py:96-97), i.e. the reference's own callers never reach seq_len > max_seq_len."""
    z = np.load(os.path.join(golden_dir, "capacity_trace.npz"))
    meta = json.loads(str(z["meta"]))
    assert meta["failure"] == {"type": "UnboundLocalError", "message": "local variable 'last_block_info' referenced before assignment"}
    cap = meta["max_blocks_per_seq"] * meta["block_size"]
    assert meta["max_seq_len_passed"] == cap == 64 and meta["largest_seq_len_passed"] == 48 and meta["sequence_length_at_failure"] == 49
    pool = _pool(meta)
    pool.allocate_for_prefill(3, meta["prompt_len"])
    for step in range(meta["num_decode_steps"]):
        tab, slot = pool.decode_step(3, 1)
        assert np.array_equal(tab, z[f"alloc{step:03d}/block_tables"][:, 0, :]), step
        assert np.array_equal(slot, z[f"alloc{step:03d}/slots"]), step
        assert np.array_equal(np.array(pool.free_blocks), z[f"alloc{step:03d}/free_blocks"]), step
    assert pool.seq_len(3) == meta["sequence_length_at_failure"] and (pool.table(3) >= 0).all()      # every row full
    assert np.array_equal(pool.table(3), z["final_table"][:, 0, :])
    with pytest.raises(RuntimeError, match="table of 4 entries is full"):
        pool.decode_step(3, 1)
    assert pool.free_blocks == meta["free_blocks_at_failure"]           # the refused step handed nothing out


def test_batched_decode_equals_per_sequence_reference_order():
    """decode_step_batch == calling the reference-compatible per-sequence step in order; tables are
    [layers, B, MB] int32, slots [layers, B] int64."""
    kw = dict(num_blocks=200, num_heads=12, head_size=64, block_size=16, max_blocks_per_seq=4, num_layers=3)
    a = PagedKVPool(device="cpu", allocate_tensors=False, **kw)
    b = PagedKVPool(device="cpu", allocate_tensors=False, **kw)
    ids = [11, 22, 33, 44]
    for i, sid in enumerate(ids):
        a.allocate_for_prefill(sid, 3 + 4 * i)
        b.allocate_for_prefill(sid, 3 + 4 * i)
    for step in range(30):
        tables, slots, lens = a.decode_step_batch(ids)
        assert tables.dtype == np.int32 and slots.dtype == np.int64 and lens.dtype == np.int32
        assert tables.shape == (3, 4, 4) and slots.shape == (3, 4)
        for i, sid in enumerate(ids):
            t, s = b.decode_step(sid, 1)
            assert np.array_equal(tables[:, i], t) and np.array_equal(slots[:, i], s)
            assert lens[i] == b.seq_len(sid)
        assert a.free_blocks == b.free_blocks
        # slot of the new token is inside the last used block of every layer, at position (len-1) % 16
        last = np.take_along_axis(tables, ((lens - 1) // 16)[None, :, None].repeat(3, 0), axis=2)[..., 0]
        assert np.array_equal(slots, last.astype(np.int64) * 16 + (lens - 1) % 16)
        # no block is owned twice
        owned = [blk for sid in ids for blk in a.allocated_blocks[sid]]
        assert len(owned) == len(set(owned)) and not set(owned) & set(a.free_blocks)


def test_swap_out_and_in_keep_the_sequence_and_move_it_to_new_blocks():
    """BlockManager.swap_to_cpu / swap_from_cpu (block_manager.py:70-87) on the bookkeeping: the blocks return to the free
    list, the host pool holds as many, and the sequence comes back on OTHER blocks with the same shape of table — the next
    decode steps then produce the slots an undisturbed twin produces, block ids renamed."""
    kw = dict(num_blocks=40, num_heads=2, head_size=64, block_size=16, max_blocks_per_seq=6, num_layers=3,
              device="cpu", allocate_tensors=False)
    a, twin = PagedKVPool(host_blocks=16, **kw), PagedKVPool(**kw)
    for p in (a, twin):
        p.allocate_for_prefill(1, 20)          # 2 blocks per layer
        p.allocate_for_prefill(2, 5)
        for _ in range(14):
            p.decode_step_batch([1, 2])
    before = a.table(1)
    n = a.swap_out(1)
    assert n == 3 * 3 and 1 not in a.allocated_blocks and 1 in a.swapped and len(a._host_free) == 16 - n
    assert a.blocks_of(1) == n and not a.swap_in(99)                            # :76-77
    with pytest.raises(ValueError):
        a.swap_out(1)                                                           # kv_cache.py:50-51
    a.allocate_for_prefill(3, 16 * 5)                                           # takes 15 of the free blocks
    a.free_blocks, keep = a.free_blocks[:4], a.free_blocks[4:]
    assert not a.swap_in(1) and 1 in a.swapped                                  # not enough free blocks: False (:86-87)
    a.free_blocks += keep
    assert a.swap_in(1) and 1 not in a.swapped and len(a._host_free) == 16
    after = a.table(1)
    assert np.array_equal(after >= 0, before >= 0)            # the same shape of table, other blocks
    assert a.seq_len(1) == twin.seq_len(1)
    rename = {int(o): int(nw) for o, nw in zip(before[before >= 0], after[after >= 0])}
    for _ in range(20):
        ta, sa, la = a.decode_step_batch([1])
        tt, st, lt = twin.decode_step_batch([1])
        assert np.array_equal(la, lt) and np.array_equal(ta >= 0, tt >= 0)
        # same offsets inside the blocks; blocks known before the swap are the renamed ones
        assert np.array_equal(sa % 16, st % 16)
        for l in range(3):
            if int(st[l, 0] // 16) in rename:
                assert int(sa[l, 0] // 16) == rename[int(st[l, 0] // 16)]
    owned = [b for s in a.allocated_blocks for b in a.allocated_blocks[s]]
    assert len(owned) == len(set(owned)) and not set(owned) & set(a.free_blocks)
    assert a.swap_stats["blocks_out"] == a.swap_stats["blocks_in"] == n


def test_reference_limits_raise():
    pool = PagedKVPool(num_blocks=5, num_heads=2, head_size=64, block_size=16, max_blocks_per_seq=2, num_layers=2,
                       device="cpu", allocate_tensors=False, multi_block_prefill=False)   # (the reference's limit, pinned)
    with pytest.raises(RuntimeError, match="single block per layer"):
        pool.allocate_for_prefill(1, 17)                                   # kv_cache.py:25-35 latent limit
    pool.allocate_for_prefill(1, 16)
    pool.decode_step(1)                                                    # block full -> appends (2 of 2 entries used)
    with pytest.raises(RuntimeError, match="trailing -1"):
        for _ in range(17):
            pool.decode_step(1)                                            # block_manager.py:36-39
    pool2 = PagedKVPool(num_blocks=3, num_heads=2, head_size=64, block_size=16, max_blocks_per_seq=4, num_layers=2,
                        device="cpu", allocate_tensors=False)
    pool2.allocate_for_prefill(1, 16)
    with pytest.raises(RuntimeError, match="Not enough free blocks"):
        pool2.allocate_for_prefill(2, 4)                                   # kv_cache.py:22-23
    with pytest.raises(RuntimeError, match="No free blocks"):
        pool2.decode_step(1)                                               # layer 0 takes the last block, layer 1 fails
