#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference's PYTHON in this container.

Run here only (needs /root/reference, read-only); the fixtures it writes are data — inputs and
expected outputs — and are what travels to the GPU box.  Nothing of the reference's source is
copied: the reference modules are imported from where they lie and executed.

How the reference is made importable on a GPU-less box:
  1. a TorchFunctionMode maps every device='cuda' / .cuda() / .to('cuda') to CPU
     (the reference hard-codes 'cuda': vllmini/kv_cache.py:13-14,31,35; block_manager.py:56;
     model/helpers/generate_triangular_mask.py:9);
  2. a stub module is registered as sys.modules['paged_attention_cuda'] BEFORE importing
     vllmini.model.gpt2 (hard import at gpt2.py:5).  The stub records every call at the seam and
     executes the CPU oracle (oracle/) so that the reference's scheduler can keep running.

Fixtures written:
  ref_eager.npz     outputs of the reference's own eager attention (`GPT2Attention._vanilla_attention`,
                    gpt2.py:71-78) on seeded inputs, for the reference test's scenario
                    (tests/kernels/paged_attention.py:7-24: 1 seq, 3 tokens, H12, D64) and for
                    multi-block / multi-sequence scenarios -> pins oracle.eager and bounds
                    oracle.kernel_model at the reference test's own tolerance (atol 1e-2, :138).
  ref_eager_long.npz  (round 5) the same reference expression at the lengths the bench runs: 300 ... 1024 tokens at 12 x 64,
                    1999 / 2048 at 4 x 128, fp32 and fp16 outputs.  The input rows come from numpy's PCG64 stream
                    (long_rows below: bit-stable everywhere), so the fixture holds their SHA-256 and the outputs only.
  seam_trace.npz    every (reshape_and_cache | paged_attention_v1) call the reference's
                    Scheduler/BlockManager/GPT-2 make for config 1 (B=1, 5-token prompt ->
                    32 tokens, block 16, max_blocks_per_seq 4), with strides, plus the allocator state
                    (free list, block tables) after every decode step -> pins the host-side
                    mirror and replays reference-produced inputs through the HIP kernels.
  capacity_trace.npz  the same stack (tiny random-weight config) with max_length ABOVE the capacity of a sequence's block
                    table (max_blocks_per_seq * block_size = the max_seq_len of scheduler.py:97): every seam call and
                    allocator state up to the step where the reference itself fails (block_manager.py:36-39 finds no -1 in
                    a full row) -> pins what the reference's callers can and cannot hand to the kernel at capacity.
  gpt2_tiny_decode.npz  the reference's GPT2LMHeadModel + BlockManager (tiny random-weight config, fp16,
                    prompt of 6 tokens then 30 forced decode tokens crossing two block boundaries): weights,
                    tokens and the logits of every step -> pins the batched decode harness
                    (vllmini_amd/gpt2_decode.py) as a caller of the ops.
  ref_selftest.json result of running the reference's OWN unittest (tests/kernels/paged_attention.py)
                    against the oracle-backed stub on CPU.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys
import types
import unittest

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True  # the reference tree is read-only

import oracle  # noqa: E402  (test infrastructure: allowed here)


# ------------------------------------------------------------------------------------------------
# shim 1: 'cuda' -> cpu
# ------------------------------------------------------------------------------------------------
def _is_cuda(dev) -> bool:
    if isinstance(dev, str):
        return dev.startswith("cuda")
    if isinstance(dev, torch.device):
        return dev.type == "cuda"
    return False


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if "device" in kwargs and _is_cuda(kwargs["device"]):
            kwargs["device"] = "cpu"
        name = getattr(func, "__name__", "")
        if name == "cuda":
            return args[0]
        if name == "to":
            args = tuple("cpu" if _is_cuda(a) else a for a in args)
        return func(*args, **kwargs)


# ------------------------------------------------------------------------------------------------
# shim 2: recording stub for the extension module
# ------------------------------------------------------------------------------------------------
class SeamRecorder:
    def __init__(self):
        self.calls = []

    def _np(self, t: torch.Tensor) -> np.ndarray:
        return t.detach().cpu().numpy()

    def reshape_and_cache(self, key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype,
                          kv_scale):
        assert kv_cache_dtype == "auto" and kv_scale == 1.0
        k, v = self._np(key), self._np(value)  # strided views of the fused qkv output
        self.calls.append({
            "op": "reshape_and_cache",
            "key": np.array(k), "value": np.array(v),
            "key_strides": tuple(key.stride()), "value_strides": tuple(value.stride()),
            "slot_mapping": np.array(self._np(slot_mapping)),
        })
        kc, vc = self._np(key_cache), self._np(value_cache)  # share memory with the torch tensors
        oracle.reshape_and_cache(k, v, kc, vc, self._np(slot_mapping))

    def paged_attention_v1(self, out, query, key_cache, value_cache, num_kv_heads, scale,
                           block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                           kv_cache_dtype, kv_scale, tp_rank, bs_local, bs_vert, bs_block, bs_step):
        assert alibi_slopes is None and kv_cache_dtype == "auto"
        assert (tp_rank, bs_local, bs_vert, bs_block, bs_step) == (0, 0, 1, 1, 0)
        q = self._np(query)
        res = oracle.paged_attention_v1(q, self._np(key_cache), self._np(value_cache), num_kv_heads,
                                        scale, self._np(block_tables), self._np(seq_lens), block_size)
        self.calls.append({
            "op": "paged_attention_v1",
            "query": np.array(q), "query_strides": tuple(query.stride()),
            "out_shape": tuple(out.shape),
            "num_kv_heads": int(num_kv_heads), "scale": float(scale),
            "block_tables": np.array(self._np(block_tables)), "seq_lens": np.array(self._np(seq_lens)),
            "block_size": int(block_size), "max_seq_len": int(max_seq_len),
            "oracle_out": res,
        })
        out.copy_(torch.from_numpy(res).view(out.shape))


def install_stub(rec: SeamRecorder):
    mod = types.ModuleType("paged_attention_cuda")
    mod.paged_attention_v1 = rec.paged_attention_v1
    mod.paged_attention_v2 = None
    mod.cache_ops = types.SimpleNamespace(reshape_and_cache=rec.reshape_and_cache)
    sys.modules["paged_attention_cuda"] = mod
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)


# ------------------------------------------------------------------------------------------------
# fixture 1: the reference's eager attention
# ------------------------------------------------------------------------------------------------
def gen_ref_eager(out_path: str):
    from transformers import GPT2Config
    from vllmini.model.gpt2 import GPT2Attention  # reference code, imported in place

    fixtures = {}
    scenarios = [
        # name, num_seqs, seq_len(s), H, D  — first one is the reference test's own scenario
        ("reftest_s1_l3", [3], 12, 64),
        ("cfg1_s1_l32", [32], 12, 64),
        ("ragged_s4", [1, 15, 16, 17], 12, 64),
        ("multi_s3_l100", [100, 77, 33], 12, 64),
        ("d128_s2", [40, 129], 4, 128),
    ]
    g = torch.Generator().manual_seed(0)
    for name, lens, H, D in scenarios:
        cfg = GPT2Config(n_embd=H * D, n_head=H)
        attn = GPT2Attention(cfg)  # only .scale is used by _vanilla_attention
        assert abs(attn.scale - D ** -0.5) < 1e-12
        for s, L in enumerate(lens):
            key = torch.randn(L, H, D, generator=g).to(torch.float16)
            value = torch.randn(L, H, D, generator=g).to(torch.float16)
            query = torch.randn(1, H, D, generator=g).to(torch.float16)
            # reference expression on [B, H, T, D] tensors (gpt2.py:53-58, 71-78), no mask in decode
            q4 = query.view(1, 1, H, D).transpose(1, 2)
            k4 = key.view(1, L, H, D).transpose(1, 2)
            v4 = value.view(1, L, H, D).transpose(1, 2)
            with torch.no_grad():
                out32 = attn._vanilla_attention(q4.float(), k4.float(), v4.float(), None)
                out16 = attn._vanilla_attention(q4, k4, v4, None)  # fp16 as the reference test runs it
            fixtures[f"{name}/{s}/key"] = key.numpy()
            fixtures[f"{name}/{s}/value"] = value.numpy()
            fixtures[f"{name}/{s}/query"] = query.numpy()
            fixtures[f"{name}/{s}/ref_eager_fp32"] = out32.reshape(H, D).numpy()
            fixtures[f"{name}/{s}/ref_eager_fp16"] = out16.reshape(H, D).numpy()
            fixtures[f"{name}/{s}/scale"] = np.float64(attn.scale)
    np.savez_compressed(out_path, **fixtures)
    print(f"wrote {out_path}: {len(fixtures)} arrays")


# ------------------------------------------------------------------------------------------------
# fixture 1b (round 5): the reference's eager attention AT THE LENGTHS THE BENCH RUNS
# ------------------------------------------------------------------------------------------------
LONG_SCENARIOS = [
    # name, context lengths, H, D, numpy seed — BASELINE.json configs[1..3]: 512 / 1024 tokens at 12 x 64, 2048 at head size 128
    ("long512_h12_d64", [512, 513, 300], 12, 64, 5120),
    ("long1024_h12_d64", [1023, 1024, 513, 700], 12, 64, 10240),
    ("long2048_h4_d128", [2048, 1999], 4, 128, 20480),
]


def long_rows(seed: int, s: int, L: int, H: int, D: int):
    """The inputs of sequence `s` of a LONG_SCENARIO: float16 rows from numpy's PCG64 stream (bit-stable across platforms
    and numpy versions) — so the fixture only has to hold their checksums and the reference's outputs (14 MB of
    incompressible rows otherwise).  Used by this generator AND by the tests that replay the fixture."""
    rng = np.random.default_rng([seed, s])
    key = rng.standard_normal((L, H, D), dtype=np.float32).astype(np.float16)
    value = rng.standard_normal((L, H, D), dtype=np.float32).astype(np.float16)
    query = rng.standard_normal((1, H, D), dtype=np.float32).astype(np.float16)
    return key, value, query


def rows_checksum(key, value, query) -> np.ndarray:
    import hashlib
    h = hashlib.sha256()
    for a in (key, value, query):
        h.update(np.ascontiguousarray(a).view(np.uint8).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def gen_ref_eager_long(out_path: str):
    """`GPT2Attention._vanilla_attention` (gpt2.py:71-78 = tests/kernels/paged_attention.py:102-110) in fp32 and fp16 on
    contexts of 300 ... 2048 tokens: the oracle and the HIP kernels are pinned against the REFERENCE where the bench runs,
    not only on the short contexts of ref_eager.npz."""
    from transformers import GPT2Config
    from vllmini.model.gpt2 import GPT2Attention  # reference code, imported in place

    fixtures = {}
    for name, lens, H, D, seed in LONG_SCENARIOS:
        attn = GPT2Attention(GPT2Config(n_embd=H * D, n_head=H))
        assert abs(attn.scale - D ** -0.5) < 1e-12
        for s, L in enumerate(lens):
            key, value, query = long_rows(seed, s, L, H, D)
            q4 = torch.from_numpy(query).view(1, 1, H, D).transpose(1, 2)
            k4 = torch.from_numpy(key).view(1, L, H, D).transpose(1, 2)
            v4 = torch.from_numpy(value).view(1, L, H, D).transpose(1, 2)
            with torch.no_grad():
                out32 = attn._vanilla_attention(q4.float(), k4.float(), v4.float(), None)
                out16 = attn._vanilla_attention(q4, k4, v4, None)
            fixtures[f"{name}/{s}/len"] = np.int64(L)
            fixtures[f"{name}/{s}/rows_sha256"] = rows_checksum(key, value, query)
            fixtures[f"{name}/{s}/ref_eager_fp32"] = out32.reshape(H, D).numpy()
            fixtures[f"{name}/{s}/ref_eager_fp16"] = out16.reshape(H, D).numpy()
        fixtures[f"{name}/meta"] = np.array(json.dumps({"lens": lens, "H": H, "D": D, "seed": seed, "scale": float(attn.scale)}))
    np.savez_compressed(out_path, **fixtures)
    print(f"wrote {out_path}: {len(fixtures)} arrays, {os.path.getsize(out_path)} bytes")


# ------------------------------------------------------------------------------------------------
# fixture 2: the seam trace of config 1
# ------------------------------------------------------------------------------------------------
def gen_seam_trace(out_path: str, rec: SeamRecorder):
    from transformers import GPT2Config
    from vllmini.block_manager import BlockManager
    from vllmini.model.gpt2 import GPT2LMHeadModel
    from vllmini.scheduler import Scheduler

    torch.manual_seed(0)
    cfg = GPT2Config()  # defaults are GPT-2 small (768/12/12); no network needed
    model = GPT2LMHeadModel(cfg).eval()
    num_blocks, block_size, max_blocks_per_seq, max_length = 64, 16, 4, 32  # server.py:37-41 default MB=4;
    # the reference needs a trailing -1 in every table (block_manager.py:36-39), so MB must exceed the blocks in use
    bm = BlockManager(num_blocks, block_size, cfg.num_attention_heads,
                      cfg.hidden_size // cfg.num_attention_heads, max_blocks_per_seq)
    sched = Scheduler(model, bm, max_length=max_length)

    alloc_trace = []
    orig_decode_step = bm.decode_step

    def traced_decode_step(seq_id, input_len):
        tables, slots = orig_decode_step(seq_id, input_len)
        alloc_trace.append({
            "free_blocks": list(bm.kv_cache.free_blocks),
            "block_tables": np.stack([t.numpy().copy() for t in tables]),  # [layers, 1, MB]
            "slots": np.array([int(s.item()) for s in slots], dtype=np.int64),
            "filled": np.array([[b, f] for (b, f) in bm.kv_cache.block_tables[seq_id]], dtype=np.int64),
        })
        return tables, slots

    bm.decode_step = traced_decode_step

    rec.calls.clear()
    prompt = torch.tensor([[464, 3290, 318, 257, 922]], dtype=torch.long)  # 5 tokens
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        seq_id = sched.add_sequence(prompt)
        prefill_free = list(bm.kv_cache.free_blocks)
        prefill_tables = np.stack([t.numpy().copy() for t in bm.kv_cache.paged_attention_block_tables[seq_id]])
        sched.run()
    tokens = sched.sequences[seq_id].numpy()

    arrays = {
        "meta": np.array(json.dumps({
            "num_blocks": num_blocks, "block_size": block_size, "max_blocks_per_seq": max_blocks_per_seq,
            "max_length": max_length, "num_layers": cfg.num_hidden_layers, "num_heads": cfg.num_attention_heads,
            "head_size": cfg.hidden_size // cfg.num_attention_heads, "prompt_len": int(prompt.shape[1]),
            "num_calls": len(rec.calls), "num_decode_steps": len(alloc_trace),
            "final_free_blocks": list(bm.kv_cache.free_blocks),
        })),
        "tokens": tokens,
        "prefill_free_blocks": np.array(prefill_free, dtype=np.int64),
        "prefill_block_tables": prefill_tables,
        "final_key_cache": bm.kv_cache.key_cache.numpy(),
        "final_value_cache": bm.kv_cache.value_cache.numpy(),
    }
    for i, c in enumerate(rec.calls):
        if c["op"] == "reshape_and_cache":
            arrays[f"call{i:04d}/op"] = np.array("reshape_and_cache")
            arrays[f"call{i:04d}/key"] = c["key"]
            arrays[f"call{i:04d}/value"] = c["value"]
            arrays[f"call{i:04d}/key_strides"] = np.array(c["key_strides"], dtype=np.int64)
            arrays[f"call{i:04d}/value_strides"] = np.array(c["value_strides"], dtype=np.int64)
            arrays[f"call{i:04d}/slot_mapping"] = c["slot_mapping"]
        else:
            arrays[f"call{i:04d}/op"] = np.array("paged_attention_v1")
            arrays[f"call{i:04d}/query"] = c["query"]
            arrays[f"call{i:04d}/query_strides"] = np.array(c["query_strides"], dtype=np.int64)
            arrays[f"call{i:04d}/out_shape"] = np.array(c["out_shape"], dtype=np.int64)
            arrays[f"call{i:04d}/block_tables"] = c["block_tables"]
            arrays[f"call{i:04d}/seq_lens"] = c["seq_lens"]
            arrays[f"call{i:04d}/scalars"] = np.array(
                [c["num_kv_heads"], c["block_size"], c["max_seq_len"]], dtype=np.int64)
            arrays[f"call{i:04d}/scale"] = np.float64(c["scale"])
            arrays[f"call{i:04d}/oracle_out"] = c["oracle_out"]
    for i, a in enumerate(alloc_trace):
        arrays[f"alloc{i:03d}/free_blocks"] = np.array(a["free_blocks"], dtype=np.int64)
        arrays[f"alloc{i:03d}/block_tables"] = a["block_tables"]
        arrays[f"alloc{i:03d}/slots"] = a["slots"]
        arrays[f"alloc{i:03d}/filled"] = a["filled"]
    np.savez_compressed(out_path, **arrays)
    print(f"wrote {out_path}: {len(rec.calls)} seam calls, {len(alloc_trace)} decode steps, "
          f"tokens={tokens.shape}")


# ------------------------------------------------------------------------------------------------
# fixture 2b: a sequence that reaches the CAPACITY of its block table (round 4)
# ------------------------------------------------------------------------------------------------
def gen_capacity_trace(out_path: str, rec: SeamRecorder):
    """The reference's Scheduler passes max_seq_len = max_blocks_per_seq * block_size — the capacity of a sequence's block
    table — to every paged_attention_v1 call (scheduler.py:97) and lets a sequence grow until max_length.  With max_length
    ABOVE that capacity the reference never gets as far as seq_len > max_seq_len: BlockManager.decode_step looks for the
    first -1 of the table row to find the last block (block_manager.py:36-39), and once the row is full there is none — the
    step after the last block was appended dies in Python.  This fixture records that run to its end: every seam call, the
    allocator after every completed decode step, the largest seq_len the kernel ever saw against max_seq_len, and how the
    reference failed.  The drop-in's own behaviour past that point (seq_len > max_seq_len: truncated; the reference kernel
    would overflow its logits buffer) is therefore OUTSIDE what the reference's callers can produce; tests/ pin both facts."""
    from transformers import GPT2Config
    from vllmini.block_manager import BlockManager
    from vllmini.model.gpt2 import GPT2LMHeadModel
    from vllmini.scheduler import Scheduler

    torch.manual_seed(3)
    cfg = GPT2Config(vocab_size=512, n_positions=128, n_embd=256, n_layer=2, n_head=4, eos_token_id=511)
    model = GPT2LMHeadModel(cfg).eval()
    num_blocks, block_size, max_blocks_per_seq, max_length = 64, 16, 4, 80      # capacity 64 tokens < max_length
    H, D = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    bm = BlockManager(num_blocks, block_size, H, D, max_blocks_per_seq)
    sched = Scheduler(model, bm, max_length=max_length)
    alloc_trace = []
    orig_decode_step = bm.decode_step

    def traced_decode_step(seq_id, input_len):
        tables, slots = orig_decode_step(seq_id, input_len)
        alloc_trace.append({"free_blocks": list(bm.kv_cache.free_blocks),
                            "block_tables": np.stack([t.numpy().copy() for t in tables]),
                            "slots": np.array([int(s_.item()) for s_ in slots], dtype=np.int64)})
        return tables, slots

    bm.decode_step = traced_decode_step
    rec.calls.clear()
    prompt = torch.tensor([[17, 301, 45, 7, 222]], dtype=torch.long)
    failure = None
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        seq_id = sched.add_sequence(prompt)
        try:
            sched.run()
        except Exception as e:  # noqa: BLE001 — whatever the reference does at capacity IS the datum
            failure = {"type": type(e).__name__, "message": str(e)}
    assert failure is not None, "the reference was expected to fail once the table row is full"
    pa = [c for c in rec.calls if c["op"] == "paged_attention_v1"]
    arrays = {"meta": np.array(json.dumps({
        "num_blocks": num_blocks, "block_size": block_size, "max_blocks_per_seq": max_blocks_per_seq, "max_length": max_length,
        "num_layers": cfg.num_hidden_layers, "num_heads": H, "head_size": D, "prompt_len": int(prompt.shape[1]),
        "num_calls": len(rec.calls), "num_decode_steps": len(alloc_trace), "failure": failure,
        "sequence_length_at_failure": int(sched.sequence_lengths[seq_id]),
        "max_seq_len_passed": int(pa[-1]["max_seq_len"]), "largest_seq_len_passed": int(max(int(c["seq_lens"][0]) for c in pa)),
        "free_blocks_at_failure": list(bm.kv_cache.free_blocks)})),
        "final_key_cache": bm.kv_cache.key_cache.numpy(), "final_value_cache": bm.kv_cache.value_cache.numpy(),
        "final_table": np.stack([t.numpy().copy() for t in bm.kv_cache.paged_attention_block_tables[seq_id]])}
    for i, c in enumerate(rec.calls):
        for k, v in c.items():
            if k in ("key_strides", "value_strides", "query_strides", "out_shape"):
                v = np.array(v, dtype=np.int64)
            arrays[f"call{i:04d}/{k}"] = np.array(v)
    for i, a_ in enumerate(alloc_trace):
        for k, v in a_.items():
            arrays[f"alloc{i:03d}/{k}"] = np.array(v, dtype=np.int64) if k != "block_tables" else v
    np.savez_compressed(out_path, **arrays)
    print(f"wrote {out_path}: {len(rec.calls)} seam calls, {len(alloc_trace)} decode steps, failure={failure}, "
          f"largest seq_len {arrays['meta']}")


# ------------------------------------------------------------------------------------------------
# fixture 3: the reference model as the caller (tiny GPT-2, forced tokens)
# ------------------------------------------------------------------------------------------------
def gen_gpt2_tiny(out_path: str):
    from transformers import GPT2Config
    from vllmini.block_manager import BlockManager
    from vllmini.model.gpt2 import GPT2LMHeadModel
    from vllmini.model.helpers.generate_triangular_mask import generate_triangular_mask

    torch.manual_seed(1)
    cfg = GPT2Config(vocab_size=512, n_positions=64, n_embd=128, n_layer=2, n_head=2)
    model = GPT2LMHeadModel(cfg).eval().to(torch.float16)           # scheduler.py:13
    H, D = cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    num_blocks, block_size, mb = 32, 16, 4
    bm = BlockManager(num_blocks, block_size, H, D, mb)
    g = torch.Generator().manual_seed(2)
    prompt = torch.randint(0, cfg.vocab_size, (1, 6), generator=g)
    forced = torch.randint(0, cfg.vocab_size, (30,), generator=g)
    seq_id = 5
    logits_steps = []
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        # prefill exactly as Scheduler.add_sequence does (scheduler.py:33-47)
        _, _, slot_mappings, tables = bm.allocate_for_prefill(seq_id, cfg.num_hidden_layers, prompt.shape[1])
        mask = generate_triangular_mask(1, H, prompt.shape[1])
        logits, _ = model(input_ids=prompt, position_ids=torch.arange(prompt.shape[1]), attention_mask=mask,
                          use_cache=True, key_cache=bm.kv_cache.key_cache, value_cache=bm.kv_cache.value_cache,
                          slot_mappings=slot_mappings, block_tables=tables)
        logits_steps.append(logits[0, -1].float().numpy())
        cur_len = prompt.shape[1]
        for tok in forced:                                          # decode exactly as Scheduler.run does (:81-98)
            tables, new_slots = bm.decode_step(seq_id, 1)
            logits, _ = model(input_ids=tok.view(1, 1), position_ids=torch.tensor([cur_len]), attention_mask=None,
                              use_cache=True, is_prefill=False, key_cache=bm.kv_cache.key_cache,
                              value_cache=bm.kv_cache.value_cache, slot_mappings=new_slots, block_tables=tables,
                              seq_lens=torch.tensor([cur_len], dtype=torch.int32), max_seq_len=mb * block_size)
            logits_steps.append(logits[0, -1].float().numpy())
            cur_len += 1
    arrays = {"meta": np.array(json.dumps({
        "vocab_size": cfg.vocab_size, "n_positions": cfg.n_positions, "n_embd": cfg.n_embd, "n_layer": cfg.n_layer,
        "n_head": cfg.n_head, "layer_norm_epsilon": cfg.layer_norm_epsilon, "num_blocks": num_blocks,
        "block_size": block_size, "max_blocks_per_seq": mb, "seq_id": seq_id})),
        "prompt": prompt.numpy(), "forced": forced.numpy(), "logits": np.stack(logits_steps),
        "final_table": np.stack([t.numpy() for t in bm.kv_cache.paged_attention_block_tables[seq_id]])}
    for k, v in model.state_dict().items():
        arrays["sd/" + k] = v.numpy()
    np.savez_compressed(out_path, **arrays)
    print(f"wrote {out_path}: {len(logits_steps)} logit rows of {cfg.vocab_size}")


# ------------------------------------------------------------------------------------------------
# pin 3: the reference's own unittest, run against the oracle-backed stub
# ------------------------------------------------------------------------------------------------
def run_reference_selftest(out_path: str):
    import importlib.util

    path = os.path.join(REFERENCE, "vllmini", "tests", "kernels", "paged_attention.py")
    spec = importlib.util.spec_from_file_location("ref_kernel_test", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    buf = io.StringIO()
    results = []
    for seed in range(5):  # the reference test is unseeded; run it under several seeds
        torch.manual_seed(seed)
        suite = unittest.defaultTestLoader.loadTestsFromTestCase(mod.TestPagedAttention)  # single-use
        with contextlib.redirect_stdout(io.StringIO()):
            res = unittest.TextTestRunner(stream=buf, verbosity=0).run(suite)
        results.append({"seed": seed, "run": res.testsRun, "failures": len(res.failures),
                        "errors": len(res.errors)})
    ok = all(r["failures"] == 0 and r["errors"] == 0 and r["run"] >= 1 for r in results)
    with open(out_path, "w") as f:
        json.dump({"reference_test": "vllmini/tests/kernels/paged_attention.py::TestPagedAttention",
                   "backend": "oracle (kernel model) through a stub paged_attention_cuda, CPU",
                   "results": results, "all_passed": ok}, f, indent=1)
    print(f"wrote {out_path}: all_passed={ok} {results}")
    if not ok:
        print(buf.getvalue())
        raise SystemExit(1)


def main():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("needs /root/reference (run in the build container)")
    oracle.build()
    rec = SeamRecorder()
    install_stub(rec)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""     # e.g. `--only capacity`: leave the others as committed
    with CudaToCpu():
        if only in ("", "eager"):
            gen_ref_eager(os.path.join(HERE, "ref_eager.npz"))
        if only in ("", "eager_long"):
            gen_ref_eager_long(os.path.join(HERE, "ref_eager_long.npz"))
        if only in ("", "seam"):
            gen_seam_trace(os.path.join(HERE, "seam_trace.npz"), rec)
        if only in ("", "capacity"):
            gen_capacity_trace(os.path.join(HERE, "capacity_trace.npz"), rec)
        if only in ("", "gpt2"):
            gen_gpt2_tiny(os.path.join(HERE, "gpt2_tiny_decode.npz"))
        if only in ("", "selftest"):
            run_reference_selftest(os.path.join(HERE, "ref_selftest.json"))


if __name__ == "__main__":
    main()
