"""N>1 path on CPU: two processes, `gloo` backend, 127.0.0.1 rendezvous.

What runs on the GPU box under RCCL is exactly this code with backend "nccl" and the HIP operator
as the step; here the step is the CPU oracle (checker standing in for the device op) so that the
sharded result can be compared with the unsharded one."""
from __future__ import annotations

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from helpers import BS, make_case
        from vllmini_amd import shard
        from vllmini_amd.kv_pool import PagedKVPool

        B, H, D = 7, 4, 64                       # odd batch: slices of 4 and 3
        rng = np.random.default_rng(123)         # every rank builds the same GLOBAL problem...
        lens = rng.integers(1, 90, B).astype(np.int32)
        case = make_case(rng, B, H, D, lens)
        full = oracle.paged_attention_v1(case["q"], case["kc"], case["vc"], H, case["scale"], case["tables"], lens, BS)

        lo, hi = shard.shard_range(B, rank, world)  # ...and computes only its slice, from its own pool
        assert [shard.owner_of(i, B, world) for i in range(B)] == [0, 0, 0, 0, 1, 1, 1]
        mine = oracle.paged_attention_v1(np.ascontiguousarray(case["q"][lo:hi]), case["kc"], case["vc"], H,
                                         case["scale"], case["tables"][lo:hi], lens[lo:hi], BS)
        assert np.array_equal(mine.view(np.uint16), full[lo:hi].view(np.uint16))

        # per-step token hand-back: each rank "samples" an id per local sequence
        local_ids = torch.arange(lo, hi, dtype=torch.int64) * 10 + rank
        allids = shard.gather_token_ids(local_ids, B, dist)
        want = torch.tensor([i * 10 + shard.owner_of(i, B, world) for i in range(B)], dtype=torch.int64)
        assert torch.equal(allids, want)

        # ... and the even-batch fast path (ONE all_gather_into_tensor into a preallocated buffer: what bench.py and the
        # scheduler run under RCCL), with and without `out`, from a non-contiguous source, and its argument checks
        B8 = 8
        lo8, hi8 = shard.shard_range(B8, rank, world)
        ids8 = torch.arange(lo8, hi8, dtype=torch.int64) * 7 + 1
        want8 = torch.arange(B8, dtype=torch.int64) * 7 + 1
        assert torch.equal(shard.gather_token_ids(ids8, B8, dist), want8)
        buf = torch.full((B8,), -1, dtype=torch.int64)
        got8 = shard.gather_token_ids(ids8, B8, dist, out=buf)
        assert got8 is buf and torch.equal(buf, want8)
        strided = torch.stack([ids8, ids8 + 100], dim=1)[:, 0]          # a view with stride 2
        assert not strided.is_contiguous() and torch.equal(shard.gather_token_ids(strided, B8, dist), want8)
        for bad in (torch.empty(B8 + 1, dtype=torch.int64), torch.empty(B8, dtype=torch.int32)):
            with pytest.raises(ValueError):
                shard.gather_token_ids(ids8, B8, dist, out=bad)

        # the asynchronous form bench.py's timed region uses: issued, then waited for before `out` is read; two buffers
        # in flight at once; argument checks
        bufs = [torch.full((B8,), -1, dtype=torch.int64) for _ in range(2)]
        o0, w0 = shard.gather_token_ids_async(ids8, B8, dist, bufs[0])
        o1, w1 = shard.gather_token_ids_async(ids8 + 1000, B8, dist, bufs[1])
        w0.wait()
        w1.wait()
        assert o0 is bufs[0] and torch.equal(bufs[0], want8) and torch.equal(bufs[1], want8 + 1000)
        for bad_ids, bad_out in ((ids8, torch.empty(B8 + 1, dtype=torch.int64)), (ids8.to(torch.int32), bufs[0]),
                                 (strided, bufs[0])):
            with pytest.raises(ValueError):
                shard.gather_token_ids_async(bad_ids, B8, dist, bad_out)

        # private pools: the same seq ids on different ranks never collide (independent free lists)
        pool = PagedKVPool(64, H, D, 16, 4, 2, device="cpu", allocate_tensors=False)
        for sid in range(lo, hi):
            pool.allocate_for_prefill(sid, 5)
        tables, slots, ctx = pool.decode_step_batch(list(range(lo, hi)))
        assert tables.shape == (2, hi - lo, 4) and (ctx == 6).all()
        assert pool.free_blocks[0] == 2 * (hi - lo)          # each rank consumed from ITS list only

        # timing contract: barrier both sides, max over ranks
        calls = []
        elapsed = shard.timed_steps(lambda i: calls.append(i) or (rank and __import__("time").sleep(0.01)), 5, 2, dist)
        assert calls == [0, 1, 0, 1, 2, 3, 4]
        tmax = shard.max_over_ranks(elapsed, dist)
        gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([elapsed], dtype=torch.float64))
        assert abs(tmax - max(float(g) for g in gathered)) < 1e-12 and tmax >= 0.04
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_decode_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_shard_range_partitions_exactly():
    from vllmini_amd import shard

    for n in (0, 1, 7, 8, 256, 2048, 2049):
        for world in (1, 2, 3, 4, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
            for i in range(n):
                lo, hi = spans[shard.owner_of(i, n, world)]
                assert lo <= i < hi
    with pytest.raises(ValueError):
        shard.shard_range(4, 4, 4)
