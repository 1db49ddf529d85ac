"""Host logic of the batching scheduler (counterpart of vllmini/scheduler.py) with a stand-in decoder
on CPU: the real PagedKVPool bookkeeping (no tensors) + deterministic fake logits."""
from __future__ import annotations

import os
import socket
import sys

import numpy as np
import pytest
import torch

from vllmini_amd.kv_pool import PagedKVPool
from vllmini_amd.scheduler import BatchScheduler, deal_requests, sample_greedy, sample_top_k

V, EOS = 97, 96


class FakeDecoder:
    """next-token logits are a pure function of (last token, position): one-hot at (7*tok + pos) % 97."""

    def __init__(self, num_blocks=64, layers=2, max_blocks_per_seq=8, host_blocks=0):
        self.pool = PagedKVPool(num_blocks, 2, 64, 16, max_blocks_per_seq, layers, device="cpu", allocate_tensors=False,
                                host_blocks=host_blocks)
        self.decode_calls = []
        self.prefill_batches = []

    @staticmethod
    def _logits(tok, pos):
        out = torch.zeros(V)
        out[(7 * tok + pos) % V] = 5.0
        return out

    def prefill(self, seq_id, ids):
        self.pool.allocate_for_prefill(seq_id, len(ids))
        return self._logits(ids[-1], len(ids) - 1)

    def prefill_batch(self, seq_ids, prompts):
        self.prefill_batches.append(list(seq_ids))
        return torch.stack([self.prefill(s, p) for s, p in zip(seq_ids, prompts)])

    def decode(self, seq_ids, tokens):
        pos = [self.pool.seq_len(s) for s in seq_ids]
        self.pool.decode_step_batch(seq_ids)
        self.decode_calls.append(list(seq_ids))
        return torch.stack([self._logits(int(t), p) for t, p in zip(tokens, pos)])


def _expected(prompt, max_length):
    seq, tok, pos = list(prompt), prompt[-1], len(prompt) - 1
    while len(seq) < max_length:
        nxt = (7 * tok + pos) % V
        seq.append(nxt)
        if nxt == EOS:
            break
        tok, pos = nxt, len(seq) - 1
    return seq


def test_batched_schedule_equals_one_sequence_at_a_time():
    prompts = [[1, 2, 3], [5], [9, 9, 9, 9, 9, 9], [40, 41]]
    dec = FakeDecoder()
    sch = BatchScheduler(dec, max_length=40, eos_token_id=EOS, sampler=sample_greedy)
    ids = [sch.add_sequence(p) for p in prompts]
    assert sch.run() > 0
    for sid, p in zip(ids, prompts):
        assert sch.sequences[sid] == _expected(p, 40)
    assert not sch.active and not sch.last_logits and not sch.sequence_lengths
    assert sorted(dec.pool.free_blocks) == list(range(dec.pool.num_blocks))      # every block returned (kv_cache.py:81-86)
    assert len(dec.decode_calls[0]) == 4                                          # one call advances the whole batch


def test_oldest_first_and_max_batch():
    dec = FakeDecoder()
    sch = BatchScheduler(dec, max_length=12, eos_token_id=EOS, max_batch=2, sampler=sample_greedy)
    ids = [sch.add_sequence([i + 1]) for i in range(5)]
    stepped = sch.step()
    assert stepped == ids[:2]                                                     # arrival order (PriorityQueue, scheduler.py:16,60)
    sch.run()
    for sid in ids:
        assert sch.sequences[sid] == _expected([sid + 1], 12)


def test_max_length_and_eos_end_sequences():
    dec = FakeDecoder()
    sch = BatchScheduler(dec, max_length=6, eos_token_id=EOS, sampler=sample_greedy)
    a = sch.add_sequence([1, 2, 3, 4, 5, 6])          # already at max_length: ends without sampling (scheduler.py:71-74)
    tok = next(t for t in range(V) for p in [0] if (7 * t + p) % V == EOS)
    b = sch.add_sequence([tok])                       # next token is EOS
    sch.run()
    assert sch.sequences[a] == [1, 2, 3, 4, 5, 6]
    assert sch.sequences[b] == [tok, EOS]
    assert not sch.active


def test_out_of_blocks_preempts_youngest_and_resumes_it_with_the_same_tokens():
    """The pool cannot hold four sequences past 16 tokens: the youngest are swapped out (block_manager.py:70-73), come back
    oldest first when there is room (:75-87), and EVERY sequence ends with the tokens of an unconstrained run."""
    dec = FakeDecoder(num_blocks=10, layers=2)
    sch = BatchScheduler(dec, max_length=40, eos_token_id=EOS, sampler=sample_greedy)
    ids = [sch.add_sequence([3 + i] * 15) for i in range(4)]      # 8 blocks used, 2 free
    sch.run()
    assert not sch.evicted and sch.stats["preemptions"] >= 2 and sch.stats["resumes"] == sch.stats["preemptions"]
    assert all(sch.sequences[s] == _expected([3 + s] * 15, 40) for s in ids)
    assert not sch.active and not sch.swapped and not sch.last_logits and not dec.pool.swapped
    assert sorted(dec.pool.free_blocks) == list(range(10)) and sorted(dec.pool._host_free) == list(range(dec.pool.host_blocks))
    st = dec.pool.swap_stats
    assert st["swap_outs"] == st["swap_ins"] == sch.stats["preemptions"] and st["blocks_out"] == st["blocks_in"] > 0
    # the youngest went first (scheduler.py:117-130) and a preempted sequence never ran while an older one waited on the host
    first_victim_gone = next(i for i, c in enumerate(dec.decode_calls) if ids[-1] not in c)
    assert all(ids[-1] in c for c in dec.decode_calls[:first_victim_gone])
    assert all(ids[0] in c for c in dec.decode_calls[: 40 - 15])  # the oldest never left


def test_full_host_pool_falls_back_to_dropping_like_the_reference():
    dec = FakeDecoder(num_blocks=10, layers=2, host_blocks=2)       # room for ONE preempted 15-token sequence
    sch = BatchScheduler(dec, max_length=40, eos_token_id=EOS, sampler=sample_greedy)
    ids = [sch.add_sequence([3 + i] * 15) for i in range(4)]
    sch.run()
    assert sch.stats["preemptions"] >= 1 and sch.evicted and sch.stats["dropped"] == len(sch.evicted)
    done = [s for s in ids if s not in sch.evicted]
    assert done and all(sch.sequences[s] == _expected([3 + s] * 15, 40) for s in done)
    assert sorted(dec.pool.free_blocks) == list(range(10)) and sorted(dec.pool._host_free) == [0, 1]


def test_submitted_requests_are_admitted_in_groups_and_refill_the_batch():
    """submit() queues; step() prefills as many queued prompts as fit — ONE prefill_batch call — and refills the batch as
    sequences end; per-request max_new_tokens; prompts longer than a block (multi-block prefill is the pool's default)."""
    dec = FakeDecoder(num_blocks=64, layers=2, max_blocks_per_seq=8)
    sch = BatchScheduler(dec, max_length=100, eos_token_id=EOS, max_batch=3, sampler=sample_greedy, record_latency=True)
    prompts = [[5 + i] * (3 + 7 * i) for i in range(7)]             # 3 ... 45 tokens
    new = [4, 9, 2, 6, 1, 5, 3]
    ids = [sch.submit(p, max_new_tokens=k) for p, k in zip(prompts, new)]
    assert not sch.active and len(sch.waiting) == 7
    sch.run()
    for sid, p, k in zip(ids, prompts, new):
        assert sch.sequences[sid] == _expected(p, len(p) + k)[: len(p) + k]
    assert dec.prefill_batches and dec.prefill_batches[0] == ids[:3]            # the first three together, in arrival order
    assert max(len(c) for c in dec.decode_calls) == 3 and sch.stats["admitted"] == 7
    assert sum(len(c) for c in dec.decode_calls) == sum(new) == sch.stats["decode_rows"]
    assert len(sch.first_token_s) == 7 and sum(len(a) for a in sch.token_latency_s) == sum(new)
    assert sorted(dec.pool.free_blocks) == list(range(64)) and not sch.pending()


def test_abort_frees_a_request_wherever_it_is():
    dec = FakeDecoder(num_blocks=10, layers=2)
    sch = BatchScheduler(dec, max_length=40, eos_token_id=EOS, sampler=sample_greedy)
    ids = [sch.add_sequence([3 + i] * 15) for i in range(4)]
    queued = sch.submit([50] * 4, max_new_tokens=5)
    for _ in range(3):
        sch.step()                                                # past token 16: the two youngest are swapped out
    assert len(sch.swapped) == 2 and queued in [w[0] for w in sch.waiting]
    victim = next(iter(sch.swapped))
    assert sch.abort(victim) and victim not in sch.swapped and not dec.pool.swapped.get(victim)
    assert sch.abort(queued) and not sch.waiting and not sch.abort(queued) and not sch.abort(12345)
    running = next(iter(sch.active))
    n_before = len(sch.sequences[running])
    assert sch.abort(running) and running not in sch.active and len(sch.sequences[running]) == n_before
    sch.run()
    rest = [s for s in ids if s not in (victim, running)]
    assert all(sch.sequences[s] == _expected([3 + s] * 15, 40) for s in rest)
    assert sorted(dec.pool.free_blocks) == list(range(10)) and sorted(dec.pool._host_free) == list(range(dec.pool.host_blocks))
    assert not sch.pending() and not sch.last_logits


def test_a_request_that_can_never_fit_is_rejected_and_the_loop_goes_on():
    dec = FakeDecoder(num_blocks=8, layers=2, max_blocks_per_seq=8)                # 8 blocks: a 70-token prompt needs 2 x 5
    sch = BatchScheduler(dec, max_length=100, eos_token_id=EOS, sampler=sample_greedy)
    a = sch.submit([3] * 5, max_new_tokens=4)
    big = sch.submit([7] * 70, max_new_tokens=4)
    c = sch.submit([9] * 6, max_new_tokens=3)
    sch.run()
    assert big in sch.rejected and "free blocks" in sch.rejected[big] and sch.sequences[big] == [7] * 70
    assert sch.sequences[a] == _expected([3] * 5, 9) and sch.sequences[c] == _expected([9] * 6, 9)
    assert not sch.pending() and sorted(dec.pool.free_blocks) == list(range(8))


def test_out_of_blocks_drop_mode_is_the_reference_behaviour():
    # 2 layers, 16-token blocks: each sequence needs 2 blocks at prefill and 2 more at token 17
    dec = FakeDecoder(num_blocks=10, layers=2)
    sch = BatchScheduler(dec, max_length=40, eos_token_id=EOS, sampler=sample_greedy, preempt="drop")
    ids = [sch.add_sequence([3 + i] * 15) for i in range(4)]      # 8 blocks used, 2 free
    sch.run()
    assert sch.evicted, "pool of 10 blocks cannot hold four sequences past 16 tokens"
    assert sch.evicted[0] == ids[-1]                              # youngest goes first (scheduler.py:117-130)
    done = [s for s in ids if s not in sch.evicted]
    assert done and all(sch.sequences[s] == _expected([3 + s] * 15, 40) for s in done)
    assert sorted(dec.pool.free_blocks) == list(range(10))
    with pytest.raises(RuntimeError):
        BatchScheduler(FakeDecoder(num_blocks=1), 8, EOS).add_sequence([1])   # cannot even prefill (kv_cache.py:22-23)


def test_top_k_sampler_matches_reference_procedure():
    """Same steps as Scheduler.sample_next_token (scheduler.py:144-153) but batched."""
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(3, 500, generator=g)
    got = sample_top_k(logits, generator=torch.Generator().manual_seed(7))
    gen = torch.Generator().manual_seed(7)
    vals, idx = torch.topk(logits, 50)
    probs = torch.softmax(vals, dim=-1)
    choice = torch.multinomial(probs, 1, generator=gen)
    assert torch.equal(got, idx.gather(-1, choice).squeeze(-1))
    assert all(int(t) in set(torch.topk(logits[i], 50).indices.tolist()) for i, t in enumerate(got))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, tmpdir):
    import torch.distributed as dist

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    sys.path.insert(0, os.path.join(repo, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_scheduler import FakeDecoder, _expected
        from vllmini_amd import shard

        prompts = [[i + 1, i + 2] for i in range(5)]               # 5 requests over 2 ranks: 3 + 2
        mine = deal_requests(len(prompts), rank, world)
        sch = BatchScheduler(FakeDecoder(), max_length=10, eos_token_id=EOS, sampler=sample_greedy)
        local = {sch.add_sequence(prompts[i]): i for i in mine}
        width = -(-len(prompts) // world)
        for _ in range(8):
            sch.step()
            # per-step hand-back of the newest token of every request to every rank (8 B/sequence)
            newest = torch.full((width,), -1, dtype=torch.int64)
            for sid, gi in local.items():
                newest[mine.index(gi)] = sch.sequences[sid][-1]
            gathered = [torch.empty_like(newest) for _ in range(world)]
            dist.all_gather(gathered, newest)
            merged = {}
            for r in range(world):
                for j, gi in enumerate(deal_requests(len(prompts), r, world)):
                    merged[gi] = int(gathered[r][j])
            assert sorted(merged) == list(range(len(prompts)))
        for sid, gi in local.items():
            assert sch.sequences[sid] == _expected(prompts[gi], 10)
        assert merged == {gi: _expected(prompts[gi], 10)[-1] for gi in range(len(prompts))}
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_scheduling_gloo(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))
    assert deal_requests(5, 0, 2) == [0, 2, 4] and deal_requests(5, 1, 2) == [1, 3]


def test_max_length_beyond_the_block_table_is_refused_up_front():
    """One sequence that outgrows its block table would abort the decode step of every sequence batched with it
    (the reference crashes there too, block_manager.py:36-39): the scheduler checks the capacity when it is built."""
    import types

    dec = FakeDecoder()
    dec.pool = types.SimpleNamespace(max_blocks_per_seq=4, block_size=16)
    BatchScheduler(dec, max_length=48, eos_token_id=EOS)                 # (4 - 1) * 16 tokens: fits
    with pytest.raises(ValueError, match="does not fit"):
        BatchScheduler(dec, max_length=49, eos_token_id=EOS)


@pytest.mark.parametrize("seed", range(60))
def test_randomised_serving_runs_complete_every_request_with_the_expected_tokens(seed):
    """Seeded random traces through the serving loop on the stand-in decoder: arrivals between steps (submit and add_sequence
    mixed), ragged prompts, per-request lengths, pools from roomy to barely enough, host pools from ample to too small,
    admission cadence 1 ... 8 — whatever is preempted, resumed, or (only when the host pool is full) dropped, every surviving
    request ends with exactly the tokens of its solo run, and every block — device and host — is back at the end."""
    rng = np.random.default_rng(1000 + seed)
    layers = int(rng.integers(1, 4))
    mbs = 8
    nblocks = int(rng.integers(layers * mbs + 2, 6 * layers * mbs))
    ample = 40 * nblocks                                      # (room for every request's pages at once)
    host = int(rng.choice([ample, ample, layers * 3, layers * mbs]))
    dec = FakeDecoder(num_blocks=nblocks, layers=layers, max_blocks_per_seq=mbs, host_blocks=host)
    sch = BatchScheduler(dec, max_length=(mbs - 1) * 16, eos_token_id=EOS, max_batch=int(rng.integers(1, 7)), sampler=sample_greedy,
                         admit_every=int(rng.integers(1, 9)), headroom_blocks=int(rng.choice([0, layers])) if rng.random() < 0.7 else None,
                         record_latency=bool(seed % 2))
    reqs = {}
    pending = [(rng.integers(1, V - 1, int(rng.integers(1, 60))).tolist(), int(rng.integers(1, 50))) for _ in range(int(rng.integers(3, 25)))]
    steps = 0
    while pending or sch.pending():
        for _ in range(int(rng.integers(0, 4))):
            if pending:
                p, k = pending.pop()
                try:
                    sid = sch.submit(p, max_new_tokens=k) if rng.random() < 0.7 else sch.add_sequence(p, max_new_tokens=k)
                except RuntimeError as e:       # add_sequence with nothing left to preempt: the request is refused, nothing leaks
                    assert "free blocks" in str(e)
                    continue
                reqs[sid] = (p, k)
        sch.step()
        steps += 1
        owned = [b for s in dec.pool.allocated_blocks for b in dec.pool.allocated_blocks[s]]
        assert len(owned) == len(set(owned)) and not set(owned) & set(dec.pool.free_blocks)          # no block owned twice
        assert set(sch.active) == set(dec.pool.allocated_blocks) and set(sch.swapped) == set(dec.pool.swapped)
        assert steps < 5000, "the loop does not terminate"
    assert not sch.active and not sch.swapped and not sch.waiting and not sch.last_logits
    assert sorted(dec.pool.free_blocks) == list(range(nblocks)) and sorted(dec.pool._host_free) == list(range(dec.pool.host_blocks))
    cap = (mbs - 1) * 16
    for sid, (p, k) in reqs.items():
        want = _expected(p, min(cap, len(p) + k))[: min(cap, len(p) + k)] if len(p) < cap else p
        if sid in sch.evicted:
            assert sch.sequences[sid] == want[: len(sch.sequences[sid])] and host < ample   # dropped only when the host pool can fill up
        else:
            assert sch.sequences[sid] == want, (sid, len(p), k)
    assert sch.stats["resumes"] <= sch.stats["preemptions"]
