"""The north-star performance target as a test: paged_attention_v1 at >= 70 % of the MI355X HBM roofline on the
BASELINE roofline config (batch 256, seq_len 1024, 12 heads x 64, block 16, random-permutation block tables) — and on the
Llama-shaped configs[3] — measured the way bench.py measures `roofline.achieved`: the reference's call pair
(reshape_and_cache, then paged_attention_v1 through the drop-in surface), a HIP event pair around every attention launch,
two disjoint table sets alternating so that no launch re-reads what the previous one left in the Infinity Cache, median.

Margins: the kernels run at 0.82 / 0.85 by this measure (events read ~3 us above rocprofv3's kernel time); boxes differ by
+-3 %.  A failure here means a real regression (or a box that is not an idle MI355X), not noise.
Algorithmic bytes: SURVEY.md §8d — cfg3 806 159 360 B -> 70 % of 8 TB/s = 144.0 us; cfg4 4 297 130 496 B -> 767.3 us.
"""
from __future__ import annotations

import statistics

import pytest
import torch

pytestmark = pytest.mark.gpu

HBM_PEAK = 8.0e12
TARGET = 0.70


def _median_attention_us(cfg_name: str, n: int = 40, warm: int = 25, kv: str = "auto", ragged: bool = False) -> tuple:
    import paged_attention_cuda as ext
    from vllmini_amd import ops
    from vllmini_amd.workload import CONFIGS, make_workload

    dev = torch.device("cuda:0")
    cfg = CONFIGS[cfg_name] if isinstance(cfg_name, str) else cfg_name
    wl = make_workload(cfg, dev, seed=21, table_sets=2, ragged=ragged)
    out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    kc, vc, esz = wl.key_cache, wl.value_cache, 2
    if kv == "fp8":          # E4M3 pages, x = 16; random codes with magnitude < 2 (exponent field <= 7), no NaNs
        g = torch.Generator(device=dev).manual_seed(5)
        code = lambda shape: (torch.randint(0, 64, shape, dtype=torch.uint8, device=dev, generator=g)
                              | (torch.randint(0, 2, shape, dtype=torch.uint8, device=dev, generator=g) << 7))
        del wl.key_cache, wl.value_cache, kc, vc
        kc = code((cfg.num_blocks, cfg.kv_heads, cfg.head_size // 16, cfg.block_size, 16))
        vc = code((cfg.num_blocks, cfg.kv_heads, cfg.head_size, cfg.block_size))
        esz = 1

    def pair(i, ev=None):
        t = i % len(wl.tables)
        ext.cache_ops.reshape_and_cache(wl.key, wl.value, kc, vc, wl.slots[t], kv, 1.0)
        if ev:
            ev[0].record()
        ext.paged_attention_v1(out, wl.query, kc, vc, cfg.kv_heads, wl.scale, wl.tables[t], wl.seq_lens,
                               cfg.block_size, cfg.seq_len, None, kv, 1.0, 0, 0, 1, 1, 0)
        if ev:
            ev[1].record()

    for i in range(warm):            # (the first launches after a pause run on a lower clock)
        pair(i)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i, ev in enumerate(evs):
        pair(i, ev)
    torch.cuda.synchronize(dev)
    us = statistics.median(a.elapsed_time(b) for a, b in evs) * 1e3
    # SURVEY.md §8d's formula with the batch's own lengths and the cache element size: K and V, q and out, tables, lengths
    lens = wl.seq_lens.to(torch.int64)
    nbytes = (2 * int(lens.sum()) * cfg.kv_heads * cfg.head_size * esz + 2 * cfg.batch * cfg.num_heads * cfg.head_size * 2
              + int(((lens + cfg.block_size - 1) // cfg.block_size).sum()) * 4 + cfg.batch * 4)
    if kv == "auto" and not ragged:
        assert nbytes == cfg.algorithmic_bytes()
    return us, nbytes, ops.last_launch_label()


@pytest.mark.parametrize("cfg_name", ["cfg3", "cfg4"])
def test_paged_attention_v1_meets_the_north_star_roofline_target(cfg_name):
    if torch.cuda.get_device_properties(0).multi_processor_count < 200:
        pytest.skip("the target is stated for a whole MI355X (256 CUs)")
    us, nbytes, label = _median_attention_us(cfg_name)
    frac = nbytes / (us * 1e-6) / HBM_PEAK
    assert frac >= TARGET, (f"{cfg_name}: paged_attention_v1 {us:.1f} us = {frac:.3f} of the 8 TB/s HBM roofline "
                            f"(target {TARGET}); kernel: {label}")


# Regression floors for the two BASELINE-shaped runs that sit furthest below the roofline (the driver line's `fp8_kv_step`
# and `ragged_step`): measured 0.73-0.76 (fp8 pages: the 1-KiB gather line of the layout) and 0.73-0.75
# (seq_lens ~ U{1..1024}: balanced kernel) by this event-pair measure on most boxes, 0.74 / 0.71 on the
# slowest box seen (boxes differ by 2 - 4 %); the floors sit 10 % below the usual figures.
@pytest.mark.parametrize("kv,ragged,floor", [("fp8", False, 0.66), ("auto", True, 0.64)])
def test_fp8_pages_and_ragged_lengths_keep_their_measured_fraction(kv, ragged, floor):
    if torch.cuda.get_device_properties(0).multi_processor_count < 200:
        pytest.skip("the floors are stated for a whole MI355X (256 CUs)")
    us, nbytes, label = _median_attention_us("cfg3", kv=kv, ragged=ragged)
    frac = nbytes / (us * 1e-6) / HBM_PEAK
    assert frac >= floor, (f"cfg3 kv={kv} ragged={ragged}: {us:.1f} us = {frac:.3f} of the 8 TB/s HBM roofline "
                           f"(floor {floor}); kernel: {label}")


# The N = 1 anchor of the strong-scaling curve (BASELINE configs[4] on ONE GPU: 2048 sequences) and one sequence more.  2048 is
# QSORT_MAX, the most sequences the balanced kernel ranks in LDS; beyond it that kernel serves a batch in index order.  Since
# round 3 a batch with more items than resident waves takes eight waves per head instead (the hardware dispatcher hands the
# workgroups out), so neither size reaches the unranked path by default — this floor keeps it that way: measured 0.83 at 2048
# (`cfg5_strong_n1` in the bench line), the floor sits 10 % below for equal lengths and at the ragged figure of cfg3 for U{1..L}.
@pytest.mark.parametrize("batch", [2048, 2049])
@pytest.mark.parametrize("ragged,floor", [(False, 0.74), (True, 0.64)])
def test_batch_at_and_above_the_ranking_limit_keeps_its_fraction(batch, ragged, floor):
    import dataclasses

    from vllmini_amd.workload import CONFIGS

    if torch.cuda.get_device_properties(0).multi_processor_count < 200:
        pytest.skip("the floors are stated for a whole MI355X (256 CUs)")
    c5 = CONFIGS["cfg5"]
    cfg = dataclasses.replace(c5, name=f"cfg5_b{batch}", batch=batch, num_blocks=2 * batch * c5.blocks_per_seq)
    us, nbytes, label = _median_attention_us(cfg, n=24, warm=8, ragged=ragged)
    frac = nbytes / (us * 1e-6) / HBM_PEAK
    assert "q_d64" not in label, f"batch {batch}: the balanced kernel would serve {batch} sequences unranked; picked {label}"
    assert frac >= floor, (f"batch {batch} ragged={ragged}: {us:.1f} us = {frac:.3f} of the 8 TB/s HBM roofline "
                           f"(floor {floor}); kernel: {label}")


# Round 5: the split kernels' regimes.  Floors 15 - 25 % off the measured figures (event pairs read 3 - 4 us
# above rocprofv3 at these sizes): few sequences x long contexts (batch 1 x 16384 tokens: 18 - 19 us by rocprofv3 against 53 without
# a workspace — asserted as a ratio on the same box), grouped-query heads, and contexts past the plain kernels' LDS in rounds
# (batch 48 x 32768 tokens: 0.835 of the roofline; the one-wave fallback it replaces ran 0.13).
@pytest.mark.parametrize("name,kernel,floor", [("long_b1", "_x", None), ("long_gqa", "_gq4_x", None), ("long_32k", "_x", 0.70)])
def test_split_kernels_keep_their_measured_gain(name, kernel, floor):
    from vllmini_amd import ops

    if torch.cuda.get_device_properties(0).multi_processor_count < 200:
        pytest.skip("the figures are stated for a whole MI355X (256 CUs)")
    us, nbytes, label = _median_attention_us(name, n=24, warm=10)
    assert kernel in label, f"{name}: the default entry picked {label}"
    if floor is not None:
        frac = nbytes / (us * 1e-6) / HBM_PEAK
        assert frac >= floor, f"{name}: {us:.1f} us = {frac:.3f} of the 8 TB/s HBM roofline (floor {floor}); kernel: {label}"
        return
    prev = ops.set_workspace_enabled(False)
    try:
        plain_us, _, plain_label = _median_attention_us(name, n=24, warm=10)
    finally:
        ops.set_workspace_enabled(prev)
    assert "_x" not in plain_label
    assert us * 1.5 <= plain_us, (f"{name}: {label} {us:.1f} us with a workspace, {plain_label} {plain_us:.1f} us without "
                                  f"(measured 2.9 x / 4.2 x)")
