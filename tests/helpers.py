"""Shared builders for the parity tests (numpy side; the GPU tests upload these)."""
from __future__ import annotations

import numpy as np

BS = 16


def make_case(rng, num_seqs, num_heads, head_size, lens, num_blocks=None, max_blocks=None,
              num_kv_heads=None, kv="uniform", poison_tail=False, q_row_pad=0, block_size=BS):
    """Random paged-KV decode case in the reference layout.

    lens: list of per-sequence context lengths.  Every sequence gets distinct physical blocks in a
    shuffled order; table tails are -1.  With poison_tail the cache slots past each context length
    (and all unowned blocks) hold NaN, which must never leak into the output
    (attention_kernels.cu:302-303, 420-430).  q_row_pad > 0 makes `query` a strided view
    (row stride = (1+q_row_pad)*H*D, like the fused-qkv view at gpt2.py:35-39).
    """
    BS = block_size  # noqa: N806 (shadows the module default on purpose)
    lens = np.asarray(lens, dtype=np.int32)
    assert lens.shape == (num_seqs,)
    num_kv_heads = num_kv_heads or num_heads
    nblk = (lens + BS - 1) // BS
    need = int(nblk.sum())
    num_blocks = num_blocks or max(need + 3, 8)
    max_blocks = max_blocks or max(int(nblk.max()), 1)
    if kv == "uniform":
        kc = rng.uniform(-1, 1, (num_blocks, num_kv_heads, head_size // 8, BS, 8)).astype(np.float16)
        vc = rng.uniform(-1, 1, (num_blocks, num_kv_heads, head_size, BS)).astype(np.float16)
    else:
        kc = rng.standard_normal((num_blocks, num_kv_heads, head_size // 8, BS, 8)).astype(np.float16)
        vc = rng.standard_normal((num_blocks, num_kv_heads, head_size, BS)).astype(np.float16)
    perm = rng.permutation(num_blocks)[:need].astype(np.int32)
    tables = np.full((num_seqs, max_blocks), -1, dtype=np.int32)
    pos = 0
    owned = np.zeros(num_blocks, dtype=bool)
    for s in range(num_seqs):
        n = int(nblk[s])
        tables[s, :n] = perm[pos:pos + n]
        owned[perm[pos:pos + n]] = True
        pos += n
    if poison_tail:
        kc[~owned] = np.nan
        vc[~owned] = np.nan
        for s in range(num_seqs):
            L = int(lens[s])
            if L % BS:
                last = tables[s, nblk[s] - 1]
                kc[last, :, :, L % BS:, :] = np.nan
                vc[last, :, :, L % BS:] = np.nan
    width = (1 + q_row_pad) * num_heads * head_size
    qbuf = rng.standard_normal((num_seqs, width)).astype(np.float16)
    q = qbuf[:, : num_heads * head_size].reshape(num_seqs, num_heads, head_size)  # strided view if padded
    return dict(q=q, qbuf=qbuf, kc=kc, vc=vc, tables=tables, lens=lens, num_kv_heads=num_kv_heads,
                scale=float(head_size) ** -0.5, num_heads=num_heads, head_size=head_size, bs=BS)


def ulp16(x):
    """fp16 unit in the last place at |x| (float64 array)."""
    ax = np.maximum(np.abs(np.asarray(x, dtype=np.float64)), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(ax)) - 10)
