"""Batched GPT-2 decode harness around the paged-attention ops.

Counterpart of the reference's callers of the hot path — GPT2Attention.forward / _cache_kv /
_paged_attention (vllmini/model/gpt2.py:21-115), GPT2Block/GPT2Model/GPT2LMHeadModel
(gpt2.py:130-273) and the decode part of Scheduler.run (vllmini/scheduler.py:76-100) — written
for what the operators are actually batched over:

  * the reference decodes ONE sequence per step (scheduler.py:55-115) and rebuilds its metadata
    with per-element device syncs; here a step advances B sequences at once from host-side
    bookkeeping (kv_pool.PagedKVPool.decode_step_batch) with a single upload;
  * q/k/v are strided views of the fused c_attn output (row stride 3*hidden), exactly the views
    the reference hands to the ops (gpt2.py:35-41);
  * around the two ops a decode step runs the block's linear layers on this build's own gfx950 kernels
    (`native_layers`, vllmini_amd/gpt2_layer.py -> csrc/gpt2_layer.hip): LayerNorm + c_attn, c_proj + residual,
    LayerNorm + c_fc + GELU, mlp.c_proj + residual — four launches per layer where the torch modules take eleven
    (67 -> 23 us per layer at batch 256).  `native_layers=False` is the torch-module chain (F.linear -> hipBLASLt,
    layer_norm, gelu), which prefill, the embeddings and lm_head always use.

`reference_off_by_one=True` reproduces the reference caller's seq_lens (length BEFORE the new
token, scheduler.py:96, so the newest token is not attended); the default attends to it.

Weights use the reference's state_dict names (nn.Linear layout [out, in]; lm_head tied to wte,
gpt2.py:240-241) so a reference checkpoint loads unchanged.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import cache_ops, gpt2_layer, ops
from .kv_pool import PagedKVPool


@dataclasses.dataclass(frozen=True)
class GPT2Dims:
    vocab_size: int = 50257
    n_positions: int = 1024
    n_embd: int = 768
    n_layer: int = 12
    n_head: int = 12
    layer_norm_epsilon: float = 1e-5
    eos_token_id: int = 50256

    @property
    def head_size(self) -> int:
        return self.n_embd // self.n_head


def random_state_dict(dims: GPT2Dims, device, dtype=torch.float16, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights of the GPT-2 architecture (no checkpoints exist offline), std 0.02."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    E = dims.n_embd

    def w(*shape, std=0.02):
        return (torch.randn(*shape, generator=g) * std).to(device=device, dtype=dtype)

    sd = {"transformer.wte.weight": w(dims.vocab_size, E), "transformer.wpe.weight": w(dims.n_positions, E, std=0.01),
          "transformer.ln_f.weight": torch.ones(E, device=device, dtype=dtype),
          "transformer.ln_f.bias": torch.zeros(E, device=device, dtype=dtype)}
    for i in range(dims.n_layer):
        p = f"transformer.h.{i}."
        sd[p + "ln_1.weight"] = torch.ones(E, device=device, dtype=dtype)
        sd[p + "ln_1.bias"] = torch.zeros(E, device=device, dtype=dtype)
        sd[p + "attn.c_attn.weight"] = w(3 * E, E)
        sd[p + "attn.c_attn.bias"] = w(3 * E)
        sd[p + "attn.c_proj.weight"] = w(E, E)
        sd[p + "attn.c_proj.bias"] = w(E)
        sd[p + "ln_2.weight"] = torch.ones(E, device=device, dtype=dtype)
        sd[p + "ln_2.bias"] = torch.zeros(E, device=device, dtype=dtype)
        sd[p + "mlp.c_fc.weight"] = w(4 * E, E)
        sd[p + "mlp.c_fc.bias"] = w(4 * E)
        sd[p + "mlp.c_proj.weight"] = w(E, 4 * E)
        sd[p + "mlp.c_proj.bias"] = w(E)
    sd["lm_head.weight"] = sd["transformer.wte.weight"]     # tied (gpt2.py:241)
    return sd


class GPT2PagedDecoder:
    """Prefill + batched decode of GPT-2 over a PagedKVPool, calling the two hot-path ops."""

    NATIVE_LAYERS_MAX_BATCH = 512    # larger steps run the block's linear layers as torch modules (real GEMMs by then)
    SCATTER_IN_C_ATTN_MAX_BATCH = 64  # up to here the cache write rides in the q / k / v projection unless the caller says no

    def __init__(self, dims: GPT2Dims, state_dict: Dict[str, torch.Tensor], pool: PagedKVPool,
                 reference_off_by_one: bool = False, fused_append: bool = False, native_layers: Optional[bool] = None,
                 scatter_in_c_attn: Optional[bool] = None, deferred_scatter: bool = False, pad_batch_to: int = 0,
                 paged_prefill: Optional[bool] = None):
        assert pool.num_layers == dims.n_layer and pool.num_heads == dims.n_head
        assert pool.head_size == dims.head_size
        self.dims, self.sd, self.pool = dims, state_dict, pool
        wdt = next(iter(state_dict.values())).dtype
        if wdt != torch.float16 or pool.kv_cache_dtype == "fp8_e5m2":
            # bfloat16 / float32 weights (hence q, k, v) and E5M2 pages are outside the hot path: the product library holds no
            # kernel for them — fail in the constructor with the operators' message, not at the first launch
            from . import _lib
            _lib.require_extras(f"GPT2PagedDecoder over {wdt} weights / kv_cache_dtype='{pool.kv_cache_dtype}'")
        self.reference_off_by_one = reference_off_by_one
        # fused_append: one launch per layer (ops.paged_attention_v1_append) instead of the reference's call pair
        # reshape_and_cache + paged_attention_v1 (gpt2.py:44, :62); bit-identical caches and outputs.  It derives
        # the slot from seq_lens-1, so it cannot reproduce the reference caller's off-by-one seq_lens.
        if fused_append and pool.kv_cache_dtype != "auto":
            raise ValueError("fused_append is built for fp16/bf16 pages only")
        if fused_append and reference_off_by_one:
            raise ValueError("fused_append writes at position seq_lens-1; reference_off_by_one passes seq_lens-1 "
                             "as the length, so the two cannot be combined")
        self.fused_append = fused_append
        # deferred_scatter (round 6): every layer's attention takes the newest token from this step's k / v rows themselves
        # (ops.paged_attention_v1_append(write_cache=False): the append-read kernels, bit-identical `out`) and the token's
        # n_layer x B rows go into the cache with ONE reshape_and_cache at the end of the step — the pool is shared by all
        # layers (kv_cache.py:13-14), so the op takes them as [n_layer * B] rows with their [n_layer * B] slots — instead of
        # one launch in front of every attention (gpt2.py:44).  Same caches, same logits as the call pair.
        if deferred_scatter and (fused_append or reference_off_by_one or pool.kv_cache_dtype != "auto"):
            raise ValueError("deferred_scatter is built for float16 pages, the two-op step and seq_lens that count the new token")
        self.deferred_scatter = deferred_scatter
        # native_layers: None = wherever they apply (a HIP device, float16 weights, hidden size a multiple of 32); True insists
        # (RuntimeError otherwise); False = the torch modules.  The library is loaded HERE: a missing one fails in the constructor.
        E = dims.n_embd
        can = pool.device.type == "cuda" and wdt == torch.float16 and E % 32 == 0 and 4 * E <= 4608
        if native_layers and not can:
            raise RuntimeError("native_layers needs a HIP device, float16 weights and a hidden size that is a multiple of 32 "
                               f"(<= 1152); got {pool.device}, {wdt}, {E}")
        self.native_layers = can if native_layers is None else bool(native_layers)
        # scatter_in_c_attn: the q / k / v projection writes k and v into the paged cache itself (gpt2_layer.linear_qkv_cache:
        # reshape_and_cache's copy in the producer's epilogue, bit-identical caches) — the step runs paged_attention_v1 alone,
        # one launch fewer per layer than the reference's call pair and on the plain attention kernels (unlike fused_append)
        # None = where it is a measured win: steps of at most SCATTER_IN_C_ATTN_MAX_BATCH rows (launch-bound: -2 ... -6 % per token
        # from 1 to 32 sequences, +1 % at 256 where the scattered two-byte V pieces cost more in the projection's tail)
        can_scatter = self.native_layers and not fused_append and not deferred_scatter and pool.kv_cache_dtype == "auto"
        if scatter_in_c_attn and not can_scatter:
            raise ValueError("scatter_in_c_attn needs native_layers, float16 pages and the two-op attention (not fused_append)")
        self.scatter_in_c_attn = scatter_in_c_attn if scatter_in_c_attn is not None or not can_scatter else None
        if not can_scatter:
            self.scatter_in_c_attn = False
        self._packed: Dict[str, gpt2_layer.PackedWeight] = {}
        if self.native_layers:
            gpt2_layer.load()
            # the block weights once more as the MFMA tiles the kernels stream (include/vmi_gpt2_layer.h, w_layout 1): weights
            # are static, a wave's load of a k-step becomes one contiguous KiB (mlp.c_proj 15.0 -> 8.2 us at batch 256);
            # the state dict keeps the reference's nn.Linear layout for prefill and for loading checkpoints
            for i in range(dims.n_layer):
                for name in ("attn.c_attn", "attn.c_proj", "mlp.c_fc", "mlp.c_proj"):
                    key = f"transformer.h.{i}.{name}.weight"
                    self._packed[key] = gpt2_layer.pack_weight(state_dict[key])
        self.scale = dims.head_size ** -0.5                    # gpt2.py:13
        self.device = pool.device
        self.max_seq_len = pool.max_blocks_per_seq * pool.block_size   # capacity, like scheduler.py:97
        self._static: Optional[dict] = None        # the static buffers of the batch size used last (see _ensure_static)
        self._sets: Dict[int, dict] = {}
        self._cur: Optional[dict] = None
        # paged_prefill (round 6): prefill_batch runs its causal attention through paged_attention_v1 itself — every prompt
        # position is a "sequence" of length position + 1 over its prompt's block table, after the layer's K / V rows are in the
        # cache — instead of the reference's eager masked attention (gpt2.py:46-58): one launch per layer for all prompts of
        # the call, and prefill logits with the decode path's own arithmetic.  None = on a HIP device; False = eager.
        self.paged_prefill = (pool.device.type == "cuda") if paged_prefill is None else bool(paged_prefill)
        self.pad_batch_to = int(pad_batch_to)      # decode steps are laid out for the next multiple of this many rows (0: as they come)

    # ---- pieces shared by prefill and decode --------------------------------------------------------
    def _ln(self, x, prefix):
        return F.layer_norm(x, (self.dims.n_embd,), self.sd[prefix + ".weight"], self.sd[prefix + ".bias"],
                            self.dims.layer_norm_epsilon)

    def _mlp(self, x, p):                                      # gpt2.py:117-128 (nn.GELU = exact erf form)
        h = F.linear(x, self.sd[p + "mlp.c_fc.weight"], self.sd[p + "mlp.c_fc.bias"])
        return F.linear(F.gelu(h), self.sd[p + "mlp.c_proj.weight"], self.sd[p + "mlp.c_proj.bias"])

    def _qkv(self, h, p):
        E, H, D = self.dims.n_embd, self.dims.n_head, self.dims.head_size
        qkv = F.linear(h, self.sd[p + "attn.c_attn.weight"], self.sd[p + "attn.c_attn.bias"])   # [T, 3E]
        T = qkv.shape[0]
        # strided views of the fused projection, row stride 3E (gpt2.py:35-41)
        return (qkv[:, :E].view(T, H, D), qkv[:, E:2 * E].view(T, H, D), qkv[:, 2 * E:].view(T, H, D))

    # ---- prefill (one sequence; eager causal attention like the reference, gpt2.py:46-58, 71-78) ------
    @torch.no_grad()
    def prefill(self, seq_id: int, input_ids: Sequence[int]) -> torch.Tensor:
        """Allocates the sequence, writes its K/V through reshape_and_cache, returns last-token logits [V]."""
        ids = torch.as_tensor(list(input_ids), dtype=torch.long, device=self.device)
        T = ids.numel()
        _, slots, _ = self.pool.allocate_for_prefill(seq_id, T)
        slots_dev = torch.from_numpy(slots).to(self.device)
        H = self.dims.n_head
        x = self.sd["transformer.wte.weight"][ids] + self.sd["transformer.wpe.weight"][torch.arange(T, device=self.device)]
        mask = torch.triu(torch.full((T, T), float("-inf"), dtype=x.dtype, device=self.device), diagonal=1)
        for i in range(self.dims.n_layer):
            p = f"transformer.h.{i}."
            q, k, v = self._qkv(self._ln(x, p + "ln_1"), p)
            cache_ops.reshape_and_cache(k, v, self.pool.key_cache, self.pool.value_cache, slots_dev[i],
                                        self.pool.kv_cache_dtype, self.pool.kv_scale)
            qh, kh, vh = (t.transpose(0, 1) for t in (q, k, v))                       # [H, T, D]
            w = torch.matmul(qh, kh.transpose(-1, -2)) * self.scale + mask            # gpt2.py:72-74
            a = torch.matmul(F.softmax(w, dim=-1), vh)                                 # gpt2.py:76-78
            a = a.transpose(0, 1).reshape(T, H * self.dims.head_size)
            x = x + F.linear(a, self.sd[p + "attn.c_proj.weight"], self.sd[p + "attn.c_proj.bias"])
            x = x + self._mlp(self._ln(x, p + "ln_2"), p)
        x = self._ln(x[-1:], "transformer.ln_f")
        return F.linear(x, self.sd["lm_head.weight"])[0]

    PREFILL_SCORE_BYTES = 256 << 20   # a prefill_batch call pads its prompts to [n, H, T, T] scores: at most this many bytes

    @torch.no_grad()
    def prefill_batch(self, seq_ids: Sequence[int], prompts: Sequence[Sequence[int]]) -> torch.Tensor:
        """Admission of several prompts in ONE pass (round 6): the prompts' tokens run through the block's layers as one
        packed [sum T, E] matrix, every layer writes ALL their K / V rows with one reshape_and_cache (concatenated slots —
        the op is batched over tokens, cache_kernels.cu:219-260), and the causal attention is ONE paged_attention_v1 launch
        over sum T "sequences" — position t of a prompt attends to the first t + 1 tokens of that prompt's pages, a ragged
        batch the balanced kernels are built for (paged_prefill) — or, eager as the reference's prefill (gpt2.py:46-58,
        71-78), on the prompts padded to the longest, [n, H, T, T] scores at a time.
        Returns the last-token logits [n, V]; per sequence the arithmetic of prefill() up to GEMM tiling (eager) / of the
        decode steps that would have produced the same context (paged)."""
        n = len(seq_ids)
        lens = [len(p) for p in prompts]
        if n == 0:
            return torch.empty((0, self.dims.vocab_size), dtype=torch.float16, device=self.device)
        d, pool, dev = self.dims, self.pool, self.device
        H, D, E = d.n_head, d.head_size, d.n_embd
        done = []
        try:
            slots, tabs = [], []
            for sid, T in zip(seq_ids, lens):
                _, sl, tb = pool.allocate_for_prefill(sid, T)                # slots [layers, T], table [layers, MB]
                slots.append(sl)
                tabs.append(tb)
                done.append(sid)
        except RuntimeError:
            for sid in done:       # all or nothing: the caller retries with a smaller group
                pool.free(sid)
            raise
        slots_dev = torch.from_numpy(np.ascontiguousarray(np.concatenate(slots, axis=1))).to(dev)     # [layers, sum T]
        ids = torch.as_tensor([t for p in prompts for t in p], dtype=torch.long, device=dev)
        starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        pos = torch.from_numpy(np.concatenate([np.arange(T, dtype=np.int64) for T in lens])).to(dev)
        x = self.sd["transformer.wte.weight"][ids] + self.sd["transformer.wpe.weight"][pos]
        paged = self.paged_prefill
        if paged:
            mbe = -(-max(lens) // pool.block_size)
            seq_of_tok = torch.from_numpy(np.repeat(np.arange(n, dtype=np.int64), lens)).to(dev)
            tables_seq = torch.from_numpy(np.ascontiguousarray(np.stack(tabs, axis=1)[:, :, :mbe])).to(dev)   # [layers, n, mbe]
            tables_tok = tables_seq[:, seq_of_tok, :].contiguous()                                              # [layers, sum T, mbe]
            lens_tok = (pos + 1).to(torch.int32)
            pvar = self._prefill_variant()
        # groups of prompts whose padded scores fit the budget, longest first inside the call's own order
        groups, cur, cur_T = [], [], 0
        for i in range(n):
            T = max(cur_T, lens[i])
            if cur and (len(cur) + 1) * H * T * T * 2 > self.PREFILL_SCORE_BYTES:
                groups.append(cur)
                cur, T = [], lens[i]
            cur.append(i)
            cur_T = T
        groups.append(cur)
        plans = []
        for g in ([] if paged else groups):
            T = max(lens[i] for i in g)
            idx = np.zeros((len(g), T), dtype=np.int64)                      # packed row of (prompt, position); pads -> row 0
            valid = np.zeros((len(g), T), dtype=bool)
            for r, i in enumerate(g):
                idx[r, : lens[i]] = starts[i] + np.arange(lens[i])
                valid[r, : lens[i]] = True
            mask = torch.triu(torch.full((T, T), float("-inf"), dtype=x.dtype, device=dev), diagonal=1)
            plans.append((torch.from_numpy(idx).to(dev), torch.from_numpy(valid).to(dev),
                          torch.from_numpy(idx[valid]).to(dev), mask))
        for i in range(d.n_layer):
            p = f"transformer.h.{i}."
            q, k, v = self._qkv(self._ln(x, p + "ln_1"), p)                                       # [sum T, H, D] views
            cache_ops.reshape_and_cache(k, v, pool.key_cache, pool.value_cache, slots_dev[i], pool.kv_cache_dtype, pool.kv_scale)
            a = torch.empty((x.shape[0], E), dtype=x.dtype, device=dev)
            if paged:
                ops.paged_attention_v1(a.view(-1, H, D), q, pool.key_cache, pool.value_cache, H, self.scale, tables_tok[i], lens_tok,
                                       pool.block_size, mbe * pool.block_size, None, pool.kv_cache_dtype, pool.kv_scale, 0, 0, 1, 1, 0,
                                       _variant=pvar)
            for idx, valid, rows, mask in plans:
                qh, kh, vh = (t[idx].transpose(1, 2) for t in (q, k, v))                          # [g, H, T, D]
                w = torch.matmul(qh, kh.transpose(-1, -2)) * self.scale + mask                    # gpt2.py:72-74
                o = torch.matmul(F.softmax(w, dim=-1), vh)                                        # gpt2.py:76-78
                a[rows] = o.transpose(1, 2).reshape(idx.shape[0], idx.shape[1], E)[valid]
            x = x + F.linear(a, self.sd[p + "attn.c_proj.weight"], self.sd[p + "attn.c_proj.bias"])
            x = x + self._mlp(self._ln(x, p + "ln_2"), p)
        last = torch.from_numpy(starts[1:] - 1).to(dev)
        return F.linear(self._ln(x[last], "transformer.ln_f"), self.sd["lm_head.weight"])

    def _prefill_variant(self) -> int:
        """The work decomposition of the prefill launch.  Its "sequences" are the positions of a few prompts: thousands of short
        items that re-read each other's pages, so what wins is one wave per (position, head) with TEMPORAL page loads, four
        blocks deep — the prompt's pages stay in L2 (64 prompts U{4..512}: 530 us against 754 for the default pick, which sees
        10 GB of pages and streams them; 8 prompts 45 against 89: scripts/prefill_attention_variants.py).  0 = the library picks."""
        if getattr(self, "_pvar", None) is None:
            self._pvar = 0
            if self.pool.kv_cache_dtype == "auto" and self.pool.block_size == 16:
                hpw = 4 if self.dims.n_head % 4 == 0 else 1
                names = ops.variant_names()
                for u in (4, 2):
                    name = f"d{self.dims.head_size}_h{hpw}_w1_u{u}_nt0"
                    if name in names:
                        self._pvar = names.index(name) + 1
                        break
        return self._pvar

    # ---- batched decode -------------------------------------------------------------------------------
    def _forward_decode(self, st: dict) -> torch.Tensor:
        """One decode step for B sequences from static device buffers (graph-capturable: no allocation
        of metadata, no host sync)."""
        d, pool = self.dims, self.pool
        B = st["input_ids"].shape[0]
        # (the kernels are built for a decode batch: ahead of the torch modules from 1 to 256 rows — 23 - 31 us per layer against
        #  32 - 43 — level at 512, behind beyond: profiles/r05u_gpt2_layer_probe_batches.json)
        nat, sd, E, pw = self.native_layers and B <= self.NATIVE_LAYERS_MAX_BATCH, self.sd, d.n_embd, self._packed
        if nat:          # token + position embedding: one launch for two gathers and an add
            x = gpt2_layer.embed(st["input_ids"], st["position_ids"], sd["transformer.wte.weight"], sd["transformer.wpe.weight"])
        else:
            x = sd["transformer.wte.weight"][st["input_ids"]] + sd["transformer.wpe.weight"][st["position_ids"]]
        for i in range(d.n_layer):
            p = f"transformer.h.{i}."
            scat = nat and (B <= self.SCATTER_IN_C_ATTN_MAX_BATCH if self.scatter_in_c_attn is None else self.scatter_in_c_attn)
            if scat:     # ln_1 + c_attn + the cache write of k and v in one launch
                qkv = gpt2_layer.linear_qkv_cache(x, pw[p + "attn.c_attn.weight"], sd[p + "attn.c_attn.bias"], pool.key_cache,
                                                  pool.value_cache, st["slots"][i], d.n_head,
                                                  ln=(sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], d.layer_norm_epsilon))
                q, k, v = (qkv[:, j * E:(j + 1) * E].view(B, d.n_head, d.head_size) for j in range(3))
            elif nat:    # ln_1 + c_attn in one launch; q/k/v are the same 3E-strided views (gpt2.py:35-41)
                qkv = gpt2_layer.linear(x, pw[p + "attn.c_attn.weight"], sd[p + "attn.c_attn.bias"],
                                        ln=(sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], d.layer_norm_epsilon),
                                        out=st["qkv_all"][i] if self.deferred_scatter else None)   # (kept until the step's one scatter)
                q, k, v = (qkv[:, j * E:(j + 1) * E].view(B, d.n_head, d.head_size) for j in range(3))
            else:
                q, k, v = self._qkv(self._ln(x, p + "ln_1"), p)
                if self.deferred_scatter:
                    st["qkv_all"][i].copy_(torch.cat([q.reshape(B, E), k.reshape(B, E), v.reshape(B, E)], dim=1))
                    q, k, v = (st["qkv_all"][i][:, j * E:(j + 1) * E].view(B, d.n_head, d.head_size) for j in range(3))
            out = torch.empty((B, d.n_head, d.head_size), dtype=q.dtype, device=q.device)   # empty_like(q) is contiguous, gpt2.py:93
            var = st.get("variant", 0)   # work decomposition chosen on the host from the batch's lengths (decode())
            if self.fused_append or self.deferred_scatter:
                ops.paged_attention_v1_append(out, q, k, v, pool.key_cache, pool.value_cache, d.n_head, self.scale,
                                              st["tables"][i], st["seq_lens"], pool.block_size, self.max_seq_len,
                                              _variant=var, write_cache=not self.deferred_scatter)
            else:
                if not scat:
                    cache_ops.reshape_and_cache(k, v, pool.key_cache, pool.value_cache, st["slots"][i],
                                                pool.kv_cache_dtype, pool.kv_scale)                        # gpt2.py:44
                ops.paged_attention_v1(out, q, pool.key_cache, pool.value_cache, d.n_head, self.scale, st["tables"][i],
                                       st["seq_lens"], pool.block_size, self.max_seq_len, None,
                                       pool.kv_cache_dtype, pool.kv_scale, 0, 0, 1, 1, 0,
                                       _variant=var)
            if nat:      # c_proj + residual; ln_2 + c_fc + GELU; mlp.c_proj + residual (x is updated in place)
                gpt2_layer.linear(out.view(B, E), pw[p + "attn.c_proj.weight"], sd[p + "attn.c_proj.bias"], residual=x, out=x)
                h = gpt2_layer.linear(x, pw[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"], gelu=True,
                                      ln=(sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], d.layer_norm_epsilon))
                gpt2_layer.linear(h, pw[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"], residual=x, out=x)
            else:
                x = x + F.linear(out.view(B, d.n_embd), self.sd[p + "attn.c_proj.weight"], self.sd[p + "attn.c_proj.bias"])
                x = x + self._mlp(self._ln(x, p + "ln_2"), p)
        if self.deferred_scatter:    # the token's n_layer x B rows of k and v, one launch (the pool is shared by the layers)
            qa = st["qkv_all"]
            rows = qa.view(d.n_layer * B, 3 * E)     # [n_layer * B] rows of stride 3E: k and v are strided views, as per layer
            cache_ops.reshape_and_cache(rows[:, E:2 * E].view(d.n_layer * B, d.n_head, d.head_size),
                                        rows[:, 2 * E:].view(d.n_layer * B, d.n_head, d.head_size),
                                        pool.key_cache, pool.value_cache, st["slots"].view(-1), pool.kv_cache_dtype, pool.kv_scale)
        return F.linear(self._ln(x, "transformer.ln_f"), self.sd["lm_head.weight"])   # [B, V]

    def _ensure_static(self, B: int) -> dict:
        """The static device buffers, pinned staging buffers and captured graphs of batch size B (kept per size: a serving
        batch moves between a few padded sizes, and a graph must not be captured again every time it comes back to one)."""
        cur = self._sets.get(B)
        if cur is None:
            L, MB, dev = self.dims.n_layer, self.pool.max_blocks_per_seq, self.device
            # tables, slots, lengths and positions live in ONE device buffer (8-byte aligned sections, typed views): a step
            # uploads them with one copy instead of four (4 x ~4 us of copy kernels per token: 3 % of a batch-1 token)
            sections = (("slots", (L, B), torch.int64), ("position_ids", (B,), torch.int64),
                        ("tables", (L, B, MB), torch.int32), ("seq_lens", (B,), torch.int32))
            offs, total = {}, 0
            for name, shape, dt in sections:
                nbytes = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
                offs[name] = (total, nbytes)
                total += (nbytes + 7) // 8 * 8

            def views(buf):
                return {name: buf[offs[name][0]:offs[name][0] + offs[name][1]].view(dt).view(shape) for name, shape, dt in sections}

            meta = torch.zeros(total, dtype=torch.uint8, device=dev)
            static = {"input_ids": torch.zeros(B, dtype=torch.long, device=dev), **views(meta)}
            if self.deferred_scatter:
                static["qkv_all"] = torch.empty((L, B, 3 * self.dims.n_embd), dtype=torch.float16, device=dev)
            static["tables"].fill_(-1)
            cur = {"static": static, "meta": meta, "graphs": {}, "stage": None}
            # Two PINNED host staging buffers: an upload from pageable memory is staged synchronously by the
            # runtime behind everything already queued on the stream, which serialises host and GPU (the fp8 step, 1.7 ms
            # of GPU work, ran 3.06 ms that way).  From pinned memory the copy is truly asynchronous and the host
            # prepares step i+1 while the GPU runs step i; a buffer is reused only after its own copy has executed.
            if dev.type == "cuda":
                cur["stage_buf"] = [torch.empty(total, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
                cur["stage"] = [{k: v.numpy() for k, v in views(b).items()} for b in cur["stage_buf"]]
                cur["stage_ev"] = [None, None]
                cur["stage_i"] = 0
            self._sets[B] = cur
        self._cur = cur
        self._static = cur["static"]
        return self._static

    @property
    def _graph(self) -> Optional[dict]:
        """variant id -> (graph, static output) of the batch size used last."""
        return None if self._cur is None else self._cur["graphs"]

    def _padded(self, B: int) -> int:
        m = self.pad_batch_to
        return B if not m or B % m == 0 else (B // m + 1) * m

    def stage_step(self, seq_ids: Sequence[int], input_ids) -> dict:
        """Host bookkeeping for one step + ONE upload of all of it into the static device buffers.  With pad_batch_to the
        step is laid out for the next multiple of it: the padding rows are empty sequences (length 0: the attention writes
        zeros; slot -1: reshape_and_cache skips the row, cache_kernels.cu:165-169) that cost a row of the linear layers each
        and keep the launch geometry — hence the captured graph — one of a few."""
        B = len(seq_ids)
        Bp = self._padded(B)
        st = self._ensure_static(Bp)
        cur = self._cur
        positions = np.fromiter((self.pool.seq_len(s) for s in seq_ids), dtype=np.int64, count=B)  # scheduler.py:81
        tables, slots, ctx = self.pool.decode_step_batch(seq_ids)
        lens = ctx - 1 if self.reference_off_by_one else ctx
        host = {"tables": tables, "slots": slots, "seq_lens": lens.astype(np.int32), "position_ids": positions}
        if cur["stage"] is None:
            for k, a in host.items():
                dst = st[k][:, :B] if k in ("tables", "slots") else st[k][:B]
                dst.copy_(torch.from_numpy(np.ascontiguousarray(a)), non_blocking=True)
                if Bp > B:
                    pad = st[k][:, B:] if k in ("tables", "slots") else st[k][B:]
                    pad.fill_(-1 if k in ("tables", "slots") else 0)
        else:
            i = cur["stage_i"]
            if cur["stage_ev"][i] is not None:
                cur["stage_ev"][i].synchronize()          # this set's previous uploads have run
            sg = cur["stage"][i]
            sg["tables"][:, :B] = tables
            sg["slots"][:, :B] = slots
            sg["seq_lens"][:B] = host["seq_lens"]
            sg["position_ids"][:B] = positions
            if Bp > B:
                sg["tables"][:, B:] = -1
                sg["slots"][:, B:] = -1
                sg["seq_lens"][B:] = 0
                sg["position_ids"][B:] = 0
            cur["meta"].copy_(cur["stage_buf"][i], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            cur["stage_ev"][i] = ev
            cur["stage_i"] = i ^ 1
        # the lengths are known here on the host: let the library's heuristic see the batch's true longest and mean
        # length (a ragged batch gets the many-waves-per-head decomposition, vmi_paged_attention_v1_pick_variant_hint)
        st["variant"] = ops.pick_variant(Bp, self.dims.n_head, self.dims.head_size, max(int(lens.max()), 1),
                                         self.pool.block_size, mean_seq_len=max(int(lens.sum()) // Bp, 1),
                                         bf16=self.pool.key_cache.dtype == torch.bfloat16,
                                         fp8={"auto": False, "fp8_e5m2": "e5m2"}.get(self.pool.kv_cache_dtype, True))
        # few sequences x long contexts: with the wrapper's workspace the library spreads a head over several workgroups
        # (ops.pick_variant(workspace=True) names a split kernel, "_x<waves>", exactly where the default entry would run one)
        # (only when the launch WILL carry a workspace: the wrapper's switch is on and one exists for this stream — none is
        #  created under stream capture — else the split id would fail with VMI_E_WORKSPACE)
        if st["variant"] and not self.fused_append and not self.deferred_scatter and self.pool.kv_cache_dtype == "auto" and \
                self.pool.key_cache.dtype == torch.float16 and ops._ws_enabled and self.device.type == "cuda" and \
                ops.workspace_for(self.device.index if self.device.index is not None else torch.cuda.current_device()) is not None:
            ws_pick = ops.pick_variant(Bp, self.dims.n_head, self.dims.head_size, max(int(lens.max()), 1),
                                       self.pool.block_size, workspace=True)
            if ws_pick and ops.is_split(ws_pick):
                st["variant"] = ws_pick
        # ... but the launch reserves LDS for self.max_seq_len (the pool's capacity, as the reference's scheduler passes
        # it, scheduler.py:97) and may be the fused append: a hinted variant that cannot serve that is dropped
        if not ops.variant_fits(st["variant"], self.max_seq_len,
                                for_append="read" if self.deferred_scatter else self.fused_append):
            st["variant"] = 0
        ids_dst = st["input_ids"][:B]
        if isinstance(input_ids, torch.Tensor):
            ids_dst.copy_(input_ids.to(torch.long), non_blocking=True)
        else:
            ids_dst.copy_(torch.as_tensor(list(input_ids), dtype=torch.long), non_blocking=True)
        st["rows"] = B
        return st

    @torch.no_grad()
    def decode(self, seq_ids: Sequence[int], input_ids, use_graph: bool = False) -> torch.Tensor:
        """Feed one token per sequence; returns logits [B, V].  With use_graph the step's ~150 kernel
        launches are replayed from one hipGraph (captured on first use for this batch size)."""
        st = self.stage_step(seq_ids, input_ids)
        B = st["rows"]
        if not use_graph:
            out = self._forward_decode(st)
            return out if out.shape[0] == B else out[:B]
        # one captured graph per launch geometry (the variant is baked into a capture): a batch whose contexts grow across a
        # pick threshold — 512 tokens at batch 32, say — switches graphs instead of capturing again in the middle of a run
        # (a re-capture costs milliseconds: the step before 1 051 us per token, after 624)
        graphs = self._cur["graphs"]
        hit = graphs.get(st["variant"])
        if hit is None:
            s = torch.cuda.Stream(self.device)
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):          # warm-up outside capture (library handles, LDS attributes)
                self._forward_decode(st)
            torch.cuda.current_stream(self.device).wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):   # (the warm-up's stream: its paged_attention_v1 workspace exists)
                out = self._forward_decode(st)
            hit = graphs[st["variant"]] = (graph, out)
        hit[0].replay()
        return hit[1] if hit[1].shape[0] == B else hit[1][:B]

    def greedy(self, logits: torch.Tensor) -> torch.Tensor:
        """argmax over the vocabulary, on the device (int64 [B]); with the native layers one workgroup per row instead of
        torch's single-pass reduce (24 -> 7 us at batch 256, 19 -> 4 at batch 1)."""
        if self.native_layers and logits.dtype == torch.float16 and logits.dim() == 2:
            return gpt2_layer.argmax(logits)
        return logits.argmax(-1)

    # ---- sampling (scheduler.py:144-153: temperature 1.0, top-k 50, multinomial) -------------------------
    @staticmethod
    def sample_top_k(logits: torch.Tensor, top_k: int = 50, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        vals, idx = torch.topk(logits.float(), top_k, dim=-1)
        probs = F.softmax(vals, dim=-1)
        choice = torch.multinomial(probs, num_samples=1, generator=generator)
        return idx.gather(-1, choice).squeeze(-1)
