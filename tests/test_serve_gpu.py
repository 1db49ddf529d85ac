"""The serving loop on the GPU (round 6): preemption by swap (BlockManager.swap_to_cpu / swap_from_cpu,
vllmini/block_manager.py:70-87) under the batching scheduler (vllmini/scheduler.py:55-130), admission of several prompts in
one prefill call, and the decode step with its scatter deferred to ONE launch per token (the append-read kernels)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _model(num_blocks, max_blocks_per_seq=10, n_layer=3, seed=0, host_blocks=0, max_seqs=16, **dec_kw):
    from vllmini_amd.gpt2_decode import GPT2Dims, GPT2PagedDecoder, random_state_dict
    from vllmini_amd.kv_pool import PagedKVPool

    dims = GPT2Dims(vocab_size=512, n_positions=256, n_embd=128, n_layer=n_layer, n_head=2)
    pool = PagedKVPool(num_blocks, dims.n_head, dims.head_size, 16, max_blocks_per_seq, dims.n_layer, device=_dev(),
                       max_seqs=max_seqs, host_blocks=host_blocks)
    g = torch.Generator(device=_dev()).manual_seed(99)        # stale bytes in unowned blocks must never matter
    pool.key_cache.uniform_(-4, 4, generator=g)
    pool.value_cache.uniform_(-4, 4, generator=g)
    sd = random_state_dict(dims, _dev(), seed=seed)
    for k in sd:                                              # livelier logits than std 0.02 gives: decisive argmaxes
        if k.endswith("weight") and "ln_" not in k:
            sd[k] = sd[k] * 4
    sd["lm_head.weight"] = sd["transformer.wte.weight"]
    return dims, GPT2PagedDecoder(dims, sd, pool, **dec_kw)


def test_swap_blocks_batched_moves_the_bytes_swap_blocks_moves():
    from vllmini_amd import cache_ops

    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(1)
    NB, NH = 24, 16
    kc = torch.empty((NB, 3, 8, 16, 8), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
    vc = torch.empty((NB, 3, 64, 16), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
    hk = torch.zeros((NH,) + tuple(kc.shape[1:]), dtype=torch.float16).pin_memory()
    hv = torch.zeros((NH,) + tuple(vc.shape[1:]), dtype=torch.float16).pin_memory()
    out_map = torch.tensor([[3, 0], [17, 5], [4, 15], [23, 1], [0, 9]], dtype=torch.int64)
    cache_ops.swap_blocks_batched(kc, vc, hk, hv, out_map)                       # device -> pinned host, one launch
    ref_k, ref_v = torch.zeros_like(hk), torch.zeros_like(hv)
    cache_ops.swap_blocks(kc, ref_k, out_map)                                    # the reference's op: one memcpy per block
    cache_ops.swap_blocks(vc, ref_v, out_map)
    torch.cuda.synchronize()
    assert torch.equal(hk.view(torch.int16), ref_k.view(torch.int16)) and torch.equal(hv.view(torch.int16), ref_v.view(torch.int16))
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    in_map = torch.tensor([[0, 7], [5, 2], [15, 20], [1, 11], [9, 3]], dtype=torch.int64)
    cache_ops.swap_blocks_batched(hk, hv, kc2, vc2, in_map.to(dev))              # host -> device; mapping may live on the device
    torch.cuda.synchronize()
    for (s, h), (h2, d) in zip(out_map.tolist(), in_map.tolist()):
        assert h == h2 and torch.equal(kc2[d], kc[s]) and torch.equal(vc2[d], vc[s])
    untouched = [b for b in range(NB) if b not in in_map[:, 1].tolist()]
    assert not kc2[untouched].any() and not vc2[untouched].any()
    cache_ops.swap_blocks_batched(kc, vc, kc2, vc2, torch.tensor([[2, 2], [6, 0]], dtype=torch.int64))   # device -> device
    torch.cuda.synchronize()
    assert torch.equal(kc2[0], kc[6]) and torch.equal(vc2[2], vc[2])
    cache_ops.swap_blocks_batched(kc, vc, hk, hv, torch.zeros((0, 2), dtype=torch.int64))                 # empty: no-op
    with pytest.raises(RuntimeError, match="not pinned"):
        cache_ops.swap_blocks_batched(kc, vc, torch.zeros_like(hk), hv, out_map)
    with pytest.raises(RuntimeError, match="out of range"):
        cache_ops.swap_blocks_batched(kc, vc, hk, hv, torch.tensor([[0, NH]], dtype=torch.int64))
    with pytest.raises(RuntimeError, match="Invalid device combination"):
        cache_ops.swap_blocks_batched(hk, hv, hk.clone().pin_memory(), hv.clone().pin_memory(), out_map)


def test_pool_swap_roundtrip_leaves_the_sequence_bit_identical():
    """A sequence swapped out and back (onto OTHER blocks: the freed ones are taken in between) decodes on exactly as its
    undisturbed twin does: same batch, same logits bits."""
    toks = np.random.default_rng(0).integers(0, 500, (40, 3))
    outs = []
    for swap in (False, True):
        _, dec = _model(num_blocks=120)
        for sid, n in ((0, 5), (1, 23), (2, 40)):
            dec.prefill(sid, list(range(10 + sid, 10 + sid + n)))
        rows = []
        for step in range(30):
            if swap and step == 12:
                before = dec.pool.table(1)
                n = dec.pool.swap_out(1)
                assert n == 3 * 3
                dec.prefill(7, list(range(16 * 4)))          # takes the freed blocks (and overwrites them)
                assert dec.pool.swap_in(1)
                after = dec.pool.table(1)
                assert not set(before[before >= 0].tolist()) & set(after[after >= 0].tolist())
            rows.append(dec.decode([0, 1, 2], toks[step].tolist()).clone())
        torch.cuda.synchronize()
        outs.append(torch.stack(rows))
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


def _greedy_run(num_blocks, prompts, new_tokens, max_batch=6, host_blocks=0, preempt="swap", submit=True, **dec_kw):
    from vllmini_amd.scheduler import BatchScheduler, sample_greedy

    dims, dec = _model(num_blocks=num_blocks, host_blocks=host_blocks, **dec_kw)
    sch = BatchScheduler(dec, max_length=140, eos_token_id=dims.vocab_size + 7, max_batch=max_batch, sampler=sample_greedy,
                         preempt=preempt, record_latency=True)
    ids = [(sch.submit if submit else sch.add_sequence)(p, max_new_tokens=k) for p, k in zip(prompts, new_tokens)]
    steps = sch.run()
    torch.cuda.synchronize()
    return sch, dec, ids, steps


def _workload(n=10, seed=3):
    rng = np.random.default_rng(seed)
    prompts = [rng.integers(0, 500, int(rng.integers(3, 40))).tolist() for _ in range(n)]
    new = [int(k) for k in rng.integers(20, 90, n)]
    return prompts, new


def _same_tokens_up_to_near_ties(a, b, dec_factory, prompt):
    """Two greedy continuations of one prompt agree, or first differ where the model's two best logits are an fp16-GEMM coin
    flip (the linear layers' tiling depends on the batch size; the attention and the pages do not)."""
    if a == b:
        return True
    k = next(i for i in range(min(len(a), len(b))) if a[i] != b[i])
    _, dec = dec_factory()
    logits = dec.prefill(0, a[: len(prompt)])
    for t in a[len(prompt): k]:
        logits = dec.decode([0], [t])[0]
    top2 = torch.topk(logits.float(), 2).values
    return float(top2[0] - top2[1]) < 5e-2 and k >= len(prompt)


def test_scheduler_preemption_by_swap_reproduces_the_unconstrained_run():
    """A pool too small for the batch: sequences are swapped out and back (refills, ragged contexts, several prompts per
    prefill call) and EVERY request ends with the tokens of the run whose pool holds everything; nothing is dropped."""
    prompts, new = _workload()
    big, _, ids_b, _ = _greedy_run(2000, prompts, new)
    assert big.stats["preemptions"] == 0 and not big.evicted
    small, dec, ids_s, steps = _greedy_run(3 * 22, prompts, new)           # ~3.7 blocks x 3 layers per sequence at the end
    assert small.stats["preemptions"] > 0 and small.stats["resumes"] == small.stats["preemptions"] and not small.evicted
    assert dec.pool.swap_stats["blocks_out"] == dec.pool.swap_stats["blocks_in"] > 0
    assert sorted(dec.pool.free_blocks) == list(range(3 * 22)) and not dec.pool.swapped and not small.pending()
    exact = 0
    for sb, ss, p, k in zip(ids_b, ids_s, prompts, new):
        a, b = big.sequences[sb], small.sequences[ss]
        assert len(a) == len(b) == len(p) + k
        exact += a == b
        assert _same_tokens_up_to_near_ties(a, b, lambda: _model(num_blocks=200), p), (sb, a, b)
    assert exact >= len(prompts) - 2, exact
    assert sum(len(x) for x in small.token_latency_s) == sum(new) and len(small.first_token_s) == len(prompts)


def test_scheduler_drop_mode_loses_the_victims_like_the_reference():
    prompts, new = _workload()
    sch, dec, ids, _ = _greedy_run(3 * 22, prompts, new, preempt="drop")
    assert sch.evicted and sch.stats["dropped"] == len(sch.evicted) and sch.stats["preemptions"] == 0
    assert all(len(sch.sequences[s]) < len(p) + k for s, p, k in zip(ids, prompts, new) if s in sch.evicted)
    assert sorted(dec.pool.free_blocks) == list(range(3 * 22))


@pytest.mark.parametrize("paged", [True, False], ids=["paged", "eager"])
def test_prefill_batch_matches_prefill_one_by_one(paged):
    """prefill_batch — its causal attention as ONE paged_attention_v1 launch over the prompts' positions (paged_prefill, the
    default on the GPU), or eager on padded groups — against the reference-style prefill of one prompt at a time (eager masked
    attention, gpt2.py:46-58): same blocks, tables and slots; logits and caches within fp16-GEMM noise; decisive argmaxes equal."""
    rng = np.random.default_rng(5)
    prompts = [rng.integers(0, 500, n).tolist() for n in (1, 16, 17, 3, 90, 33, 64)]
    _, one = _model(num_blocks=400, paged_prefill=False)
    _, many = _model(num_blocks=400, paged_prefill=paged)
    assert many.paged_prefill == paged and (not paged or many._prefill_variant() > 0)     # (the temporal one-wave kernel, by id)
    many.PREFILL_SCORE_BYTES = 3 * 2 * 90 * 90 * 2            # forces several padded groups inside the one call
    ref = torch.stack([one.prefill(i, p) for i, p in enumerate(prompts)])
    got = many.prefill_batch(list(range(len(prompts))), prompts)
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    err = (got.float() - ref.float()).abs().max()
    assert float(err) <= 2e-2 + 2e-2 * float(ref.float().abs().max()), float(err)
    # the same blocks, tables and slots as one by one (the pools start equal): the caches agree up to GEMM tiling
    for i in range(len(prompts)):
        assert np.array_equal(one.pool.table(i), many.pool.table(i))
    owned = sorted(b for i in range(len(prompts)) for b in one.pool.allocated_blocks[i])
    dk = (one.pool.key_cache[owned].float() - many.pool.key_cache[owned].float()).abs().max()
    dv = (one.pool.value_cache[owned].float() - many.pool.value_cache[owned].float()).abs().max()
    assert float(dk) < 5e-2 and float(dv) < 5e-2
    # and decoding on from either pool gives the same next tokens wherever the argmax is decisive
    toks = [int(t) for t in ref.argmax(-1)]
    la, lb = one.decode(list(range(len(prompts))), toks), many.decode(list(range(len(prompts))), toks)
    top2 = torch.topk(la.float(), 2, dim=-1).values
    decisive = (top2[:, 0] - top2[:, 1]) > 5e-2
    assert torch.equal(la.argmax(-1)[decisive], lb.argmax(-1)[decisive]) and bool(decisive.any())
    # a group that does not fit leaves nothing behind
    _, tight = _model(num_blocks=3 * 3)
    with pytest.raises(RuntimeError, match="free blocks"):
        tight.prefill_batch([0, 1, 2], [[1] * 20, [2] * 20, [3] * 20])
    assert sorted(tight.pool.free_blocks) == list(range(9)) and not tight.pool.allocated_blocks


@pytest.mark.parametrize("native", [True, False])
def test_deferred_scatter_step_is_bit_identical_to_the_call_pair(native):
    """GPT2PagedDecoder(deferred_scatter=True): the layers attend through the append-read kernels and the token's rows are
    written by ONE reshape_and_cache — same logits bits, same cache bytes as reshape_and_cache + paged_attention_v1 per layer."""
    toks = np.random.default_rng(2).integers(0, 500, (24, 4))
    res = []
    for deferred in (False, True):
        _, dec = _model(num_blocks=200, native_layers=native, deferred_scatter=deferred, scatter_in_c_attn=False if native else None)
        for sid, n in ((0, 1), (1, 15), (2, 16), (3, 50)):
            dec.prefill(sid, list(range(3 + sid, 3 + sid + n)))
        rows = [dec.decode([0, 1, 2, 3], toks[s].tolist(), use_graph=(s >= 12)).clone() for s in range(24)]
        torch.cuda.synchronize()
        res.append((torch.stack(rows), dec.pool.key_cache.clone(), dec.pool.value_cache.clone()))
    i16 = torch.int16
    assert torch.equal(res[0][0].view(i16), res[1][0].view(i16))
    assert torch.equal(res[0][1].view(i16), res[1][1].view(i16)) and torch.equal(res[0][2].view(i16), res[1][2].view(i16))


def test_newest_entry_reads_the_rows_and_leaves_the_cache_alone():
    from vllmini_amd import cache_ops, ops

    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(4)
    for B, H, lens in ((256, 12, [1024] * 256), (7, 12, [1, 16, 17, 100, 333, 1024, 47])):
        lens_np = np.asarray(lens, dtype=np.int32)
        nblk = (lens_np + 15) // 16
        NB = int(nblk.sum()) + 4
        kc = torch.empty((NB, H, 8, 16, 8), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
        vc = torch.empty((NB, H, 64, 16), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
        qkv = torch.empty((B, 3 * H * 64), dtype=torch.float16, device=dev).normal_(0, 1, generator=g)
        q, k, v = (qkv[:, i * H * 64:(i + 1) * H * 64].view(B, H, 64) for i in range(3))
        tab = np.full((B, int(nblk.max()) + 1), -1, dtype=np.int32)
        perm, at = np.random.default_rng(B).permutation(NB).astype(np.int32), 0
        for s in range(B):
            tab[s, : nblk[s]] = perm[at: at + nblk[s]]
            at += nblk[s]
        slots = torch.from_numpy(tab[np.arange(B), (lens_np - 1) // 16].astype(np.int64) * 16 + (lens_np - 1) % 16).to(dev)
        tab_d, lens_d = torch.from_numpy(tab).to(dev), torch.from_numpy(lens_np).to(dev)
        kc0, vc0 = kc.clone(), vc.clone()
        out_n = torch.full((B, H, 64), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1_append(out_n, q, k, v, kc, vc, H, 0.125, tab_d, lens_d, 16, 1024, write_cache=False)
        torch.cuda.synchronize()
        assert torch.equal(kc.view(torch.int16), kc0.view(torch.int16)) and torch.equal(vc.view(torch.int16), vc0.view(torch.int16))
        out_p = torch.full_like(out_n, float("nan"))
        cache_ops.reshape_and_cache(k, v, kc, vc, slots, "auto", 1.0)
        prev = ops.set_workspace_enabled(False)
        try:
            ops.paged_attention_v1(out_p, q, kc, vc, H, 0.125, tab_d, lens_d, 16, 1024, None, "auto", 1.0)
        finally:
            ops.set_workspace_enabled(prev)
        torch.cuda.synchronize()
        assert torch.equal(out_n.view(torch.int16), out_p.view(torch.int16)), B
