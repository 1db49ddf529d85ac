"""The token exchange of the multi-GPU leg on the backend it runs on in production — "nccl" = RCCL — with one rank on
cuda:0 (the GPU box has one GPU; world_size 2 is covered on CPU over gloo, tests/test_shard_gloo.py): the blocking and the
asynchronous form hand back the ids in global order, two asynchronous gathers may be in flight at once, and a gather issued
behind a kernel on the compute stream sees that kernel's result (the process group's stream waits for the compute stream
at issue; `work.wait()` makes the compute stream wait for the gather)."""
from __future__ import annotations

import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_token_exchange_over_rccl_single_rank():
    import torch.distributed as dist

    from vllmini_amd import shard

    if not dist.is_nccl_available():
        pytest.skip("torch.distributed was built without the nccl (RCCL) backend")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        B = 256
        ids = torch.arange(B, dtype=torch.int64, device=dev) * 3 + 1
        assert torch.equal(shard.gather_token_ids(ids, B, dist), ids)
        bufs = [torch.full((B,), -1, dtype=torch.int64, device=dev) for _ in range(2)]
        # produced on the compute stream right in front of the gather: the collective must see the finished values
        big = torch.zeros(1 << 24, dtype=torch.int64, device=dev)
        for it in range(6):
            big.add_(1)                                   # a kernel the gather has to queue behind
            src = (big[:B] * 1000 + ids).contiguous()     # = (it + 1) * 1000 + ids
            k = it & 1
            out, work = shard.gather_token_ids_async(src, B, dist, bufs[k])
            big.add_(0)                                   # compute goes on beside the gather
            work.wait()                                   # stream-side: later kernels on the compute stream see `out`
            assert out is bufs[k]
            assert torch.equal(out, ids + (it + 1) * 1000)
        o0, w0 = shard.gather_token_ids_async(ids, B, dist, bufs[0])
        o1, w1 = shard.gather_token_ids_async(ids + 7, B, dist, bufs[1])
        w0.wait()
        w1.wait()
        torch.cuda.synchronize()
        assert torch.equal(o0, ids) and torch.equal(o1, ids + 7)
        with pytest.raises(ValueError):
            shard.gather_token_ids_async(ids.to(torch.int32), B, dist, bufs[0])
    finally:
        dist.destroy_process_group()
