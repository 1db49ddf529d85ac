"""The append-read kernels on a FULL chip (round 6): vmi_paged_attention_v1_newest_f16 runs the balanced kernel's APP form
(csrc/pa_queue.hpp) — the attention reads the newest token from this step's key / value rows instead of the cache and writes
nothing, so the caller can store a token's rows of all layers with one reshape_and_cache.  `out` must be BIT-identical to the
reference's call pair, cache_ops.reshape_and_cache then paged_attention_v1 (vllmini/model/gpt2.py:44, :62), in every mode
the kernel chooses on the device: one wave per item (equal lengths), solo workers over ranked items (ragged), four-wave
teams (heavy tails), more items than waves.  The WRITING fused entry (vmi_paged_attention_v1_append_f16) is checked on the
same cases: same caches, same out, whatever kernel it picks."""
from __future__ import annotations

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BS = 16


def _dev():
    return torch.device("cuda:0")


def _case(B, H, Hkv, D, lens, seed):
    """Random pools with DISTINCT shuffled blocks per sequence; q / key / value as strided views of one fused row."""
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(seed)
    lens = np.asarray(lens, dtype=np.int32)
    nblk = (np.maximum(lens, 1) + BS - 1) // BS
    mb = int(nblk.max()) + 1
    NB = int(nblk.sum()) + 8
    kc = torch.empty((NB, Hkv, D // 8, BS, 8), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
    vc = torch.empty((NB, Hkv, D, BS), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
    qkv = torch.empty((B, (H + 2 * Hkv) * D), dtype=torch.float16, device=dev).normal_(0, 1, generator=g)
    q = qkv[:, : H * D].view(B, H, D)
    key = qkv[:, H * D: (H + Hkv) * D].view(B, Hkv, D)
    value = qkv[:, (H + Hkv) * D:].view(B, Hkv, D)
    perm = np.random.default_rng(seed).permutation(NB).astype(np.int32)
    tab = np.full((B, mb), -1, dtype=np.int32)
    at = 0
    for s in range(B):
        tab[s, : nblk[s]] = perm[at: at + nblk[s]]
        at += nblk[s]
    pos = np.maximum(lens.astype(np.int64) - 1, 0)
    slots = tab[np.arange(B), pos // BS].astype(np.int64) * BS + pos % BS
    slots[lens <= 0] = -1
    return dict(kc=kc, vc=vc, q=q, key=key, value=value, tab=torch.from_numpy(tab).to(dev), lens=torch.from_numpy(lens).to(dev),
                slots=torch.from_numpy(slots).to(dev), msl=max(int(lens.max()), 1), H=H, Hkv=Hkv, D=D, B=B)


def _check(c, what, variant=0, expect_balanced=True):
    from vllmini_amd import cache_ops, ops

    kc_a, vc_a, kc_b, vc_b = c["kc"], c["vc"], c["kc"].clone(), c["vc"].clone()
    kc_0, vc_0 = c["kc"].clone(), c["vc"].clone()
    nan = float("nan")
    out_a = torch.full((c["B"], c["H"], c["D"]), nan, dtype=torch.float16, device=_dev())
    out_b, out_n = torch.full_like(out_a, nan), torch.full_like(out_a, nan)
    scale = c["D"] ** -0.5
    # append-read first, on the caches BEFORE the rows are stored: it must not need them there, nor put them there
    ops.paged_attention_v1_append(out_n, c["q"], c["key"], c["value"], kc_a, vc_a, c["Hkv"], scale, c["tab"], c["lens"], BS,
                                  c["msl"], _variant=variant, write_cache=False)
    read_name = ops.variant_names()[ops.last_variant() - 1]
    torch.cuda.synchronize()
    i16 = torch.int16
    assert torch.equal(kc_a.view(i16), kc_0.view(i16)) and torch.equal(vc_a.view(i16), vc_0.view(i16)), f"{what}: append-read wrote"
    cache_ops.reshape_and_cache(c["key"], c["value"], kc_a, vc_a, c["slots"], "auto", 1.0)
    prev = ops.set_workspace_enabled(False)
    try:
        ops.paged_attention_v1(out_a, c["q"], kc_a, vc_a, c["Hkv"], scale, c["tab"], c["lens"], BS, c["msl"], None, "auto", 1.0,
                               _variant=variant)
        pair_name = ops.variant_names()[ops.last_variant() - 1]
    finally:
        ops.set_workspace_enabled(prev)
    if not (variant and pair_name.startswith("q_d")):      # (the writing entry has no balanced form: not by that id)
        ops.paged_attention_v1_append(out_b, c["q"], c["key"], c["value"], kc_b, vc_b, c["Hkv"], scale, c["tab"], c["lens"], BS,
                                      c["msl"], _variant=variant)
    else:
        with pytest.raises(RuntimeError, match="no fused-append twin|append-read form only"):
            ops.paged_attention_v1_append(out_b, c["q"], c["key"], c["value"], kc_b, vc_b, c["Hkv"], scale, c["tab"], c["lens"],
                                          BS, c["msl"], _variant=variant)
        ops.paged_attention_v1_append(out_b, c["q"], c["key"], c["value"], kc_b, vc_b, c["Hkv"], scale, c["tab"], c["lens"], BS, c["msl"])
    torch.cuda.synchronize()
    if expect_balanced:
        assert read_name.startswith("q_d") and read_name == pair_name, (what, pair_name, read_name)
    assert torch.equal(kc_a.view(i16), kc_b.view(i16)), f"{what}: key cache differs"
    assert torch.equal(vc_a.view(i16), vc_b.view(i16)), f"{what}: value cache differs"
    live = (c["lens"] > 0)
    assert torch.isfinite(out_n[live].float()).all(), f"{what}: non-finite"
    for name, o in (("append-read", out_n), ("fused", out_b)):
        same = torch.equal(out_a.view(i16), o.view(i16))
        if name == "fused" and expect_balanced and not same:
            # the writing entry runs another work decomposition than the pair on a full chip: fp32 summation order
            d = (out_a.float() - o.float()).abs()
            assert float(d.max()) <= 1e-3, f"{what}: fused differs by {float(d.max()):.3e}"
            continue
        assert same, f"{what}: {name} out differs, max {float((out_a.float() - o.float()).abs().max()):.3e} in " \
                     f"{int((out_a.view(i16) != o.view(i16)).any(-1).any(-1).sum())} sequences"


def test_append_balanced_equal_lengths_one_wave_per_item():
    for L in (1024, 1023, 1009, 528, 513):           # newest token at a block's end, start, middle
        _check(_case(256, 12, 12, 64, [L] * 256, seed=L), f"B256 L{L}")
    for L in (16, 17, 1):                             # short contexts: whatever kernel the pair runs there, the same bits
        _check(_case(256, 12, 12, 64, [L] * 256, seed=L), f"B256 L{L}", expect_balanced=False)


def test_append_balanced_ragged_solo_workers():
    rng = np.random.default_rng(5)
    lens = rng.integers(1, 1025, 256)
    lens[0] = 1024
    lens[7] = 0                                         # an empty row: no store, zero output
    _check(_case(256, 12, 12, 64, lens, seed=11), "B256 U{1..1024}")
    lens = rng.integers(1, 40, 256)                     # every item's first K group is its last: the next item's prefetch is patched
    lens[3] = 1024
    _check(_case(256, 12, 12, 64, lens, seed=12), "B256 short + one long")


def test_append_balanced_heavy_tail_teams():
    lens = np.full(256, 32)
    lens[:4] = 1024                                     # 4 long among 252 short: four-wave teams + solo quads
    lens[100] = 31
    lens[101] = 33
    _check(_case(256, 12, 12, 64, lens, seed=21), "B256 4 full, rest 1/32")
    lens = np.where(np.arange(256) % 2 == 0, 1024, 64)  # bimodal
    _check(_case(256, 12, 12, 64, lens, seed=22), "B256 bimodal")


def test_append_balanced_more_rows_than_waves():
    _check(_case(600, 12, 12, 64, [300] * 600, seed=31), "B600 equal")          # 7200 rows over 3072 waves: rows stored in strides
    rng = np.random.default_rng(32)
    _check(_case(600, 12, 12, 64, rng.integers(1, 400, 600), seed=32), "B600 ragged")


def test_append_balanced_by_variant_id_and_grouped_query_heads():
    from vllmini_amd import ops

    vid = ops.variant_names().index("q_d64_s1q2") + 1
    rng = np.random.default_rng(41)
    _check(_case(64, 12, 12, 64, rng.integers(1, 600, 64), seed=41), "B64 by id", variant=vid)
    _check(_case(256, 12, 4, 64, rng.integers(1, 700, 256), seed=42), "B256 12 q heads on 4 KV heads", variant=vid)
    _check(_case(256, 12, 4, 64, [512] * 256, seed=43), "B256 GQA equal", variant=vid)
    _check(_case(5, 3, 3, 64, [1, 16, 17, 0, 200], seed=44), "tiny by id", variant=vid)
