"""Eager attention over the paged cache — restatement of the reference's non-kernel path.

Reference expression (vllmini/model/gpt2.py:71-78 `_vanilla_attention`, identical to
vllmini/tests/kernels/paged_attention.py:108-110):

    attn_weights = matmul(q, k.transpose(-1, -2)) * scale
    attn_probs   = softmax(attn_weights, dim=-1)
    out          = matmul(attn_probs, v)

applied to decode: K/V rows are first gathered out of the paged caches by `block_tables`.
Test infrastructure; also the `cpu_baseline` that bench.py times ("the reference's CPU
fallback (PyTorch eager attention)").
"""
from __future__ import annotations

import numpy as np


def gather_kv(key_cache: np.ndarray, value_cache: np.ndarray, block_table: np.ndarray,
              seq_len: int) -> tuple[np.ndarray, np.ndarray]:
    """Rows [seq_len, H, D] of one sequence, read back through the reference layout
    (K[blk,h,d/8,tok,d%8], V[blk,h,d,tok]; cache_kernels.cu:187-194)."""
    bs = key_cache.shape[3]
    nblk = (seq_len + bs - 1) // bs
    blocks = np.asarray(block_table[:nblk], dtype=np.int64)
    kb = key_cache[blocks]                                   # [nblk, H, D/8, bs, 8]
    vb = value_cache[blocks]                                 # [nblk, H, D, bs]
    nb, H, dx, _, x = kb.shape
    k = kb.transpose(0, 3, 1, 2, 4).reshape(nb * bs, H, dx * x)[:seq_len]
    v = vb.transpose(0, 3, 1, 2).reshape(nb * bs, H, vb.shape[2])[:seq_len]
    return k, v


def eager_paged_attention(query: np.ndarray, key_cache: np.ndarray, value_cache: np.ndarray,
                          num_kv_heads: int, scale: float, block_tables: np.ndarray,
                          seq_lens: np.ndarray, dtype=np.float64,
                          alibi_slopes: np.ndarray | None = None) -> np.ndarray:
    """softmax(q.K^T*scale).V per (seq, head) in `dtype` (float64 = the 'exact' yardstick)."""
    S, H, D = query.shape
    out = np.zeros((S, H, D), dtype=dtype)
    rep = H // num_kv_heads
    for s in range(S):
        L = int(seq_lens[s])
        if L == 0:
            continue
        k, v = gather_kv(key_cache, value_cache, block_tables[s], L)
        k = np.repeat(k.astype(dtype), rep, axis=1) if rep > 1 else k.astype(dtype)
        v = np.repeat(v.astype(dtype), rep, axis=1) if rep > 1 else v.astype(dtype)
        q = query[s].astype(dtype)                                   # [H, D]
        w = np.einsum("hd,lhd->hl", q, k) * dtype(scale)
        if alibi_slopes is not None:
            pos = np.arange(L, dtype=dtype) - (L - 1)
            w = w + alibi_slopes.astype(dtype)[:, None] * pos[None, :]
        w = w - w.max(axis=-1, keepdims=True)
        p = np.exp(w)
        p = p / p.sum(axis=-1, keepdims=True)
        out[s] = np.einsum("hl,lhd->hd", p, v)
    return out


def torch_eager_decode(query, key_cache, value_cache, scale: float, block_tables, seq_len: int,
                       dtype=None):
    """The reference's eager math in torch on CPU for a batch with equal seq_len (bench cpu_baseline).

    query [S,H,D], caches in the reference layout, block_tables [S,MB] (torch CPU tensors).
    Gather is included in what callers time, as in BASELINE.md §3.
    """
    import torch

    S, H, D = query.shape
    bs = key_cache.shape[3]
    nblk = (seq_len + bs - 1) // bs
    idx = block_tables[:, :nblk].to(torch.int64)                      # [S, nblk]
    kb = key_cache[idx]                                               # [S, nblk, H, D/8, bs, 8]
    vb = value_cache[idx]                                             # [S, nblk, H, D, bs]
    k = kb.permute(0, 2, 1, 4, 3, 5).reshape(S, H, nblk * bs, D)[:, :, :seq_len]
    v = vb.permute(0, 2, 1, 4, 3).reshape(S, H, nblk * bs, D)[:, :, :seq_len]
    q = query.unsqueeze(2)                                            # [S, H, 1, D]
    if dtype is not None:
        q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    attn_weights = torch.matmul(q, k.transpose(-1, -2)) * scale      # gpt2.py:72
    attn_weights = torch.nn.functional.softmax(attn_weights, dim=-1)  # gpt2.py:76
    return torch.matmul(attn_weights, v).squeeze(2)                   # gpt2.py:78
