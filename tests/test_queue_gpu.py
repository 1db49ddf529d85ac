"""GPU parity of the balanced paged_attention_v1 kernels (vllmini_amd/csrc/pa_queue.hpp): every mode the kernel
can choose on the device — one item per wave (S), ranked work lists with solo workers or 4-wave teams (Q) — is forced
through vmi_debug_set_queue_flags (diagnostic build of the library) on small inputs and compared with the CPU oracle (checker only) and, for the
single-wave modes, bit for bit with the one-wave-per-head kernel whose operations they repeat; then the BASELINE cfg3
size with ragged lengths goes through the DEFAULT entry (no hint, no variant).

Reference semantics under test: attention_kernels.cu:115-136 (context bounds), 302-305 (masked logits -> 0),
334-342 (softmax), 420-430 (tail of V zeroed).  Nothing here reads /root/reference.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

import oracle
from helpers import BS, make_case
from test_parity_gpu import _dev, assert_close, run_hip, run_model

pytestmark = pytest.mark.gpu


def _names():
    from vllmini_amd import ops

    return {n: i + 1 for i, n in enumerate(ops.variant_names())}


def _flags(mode=0, wq=0, nosort=0, team=0, nohybrid=0):
    return mode | (wq << 2) | (nohybrid << 10) | (nosort << 11) | (team << 12)


# label -> (flags, bit-identical to the one-wave-per-head kernel?)
MODES = {
    "auto": (_flags(), None),                      # whatever the kernel chooses from seq_lens
    "S": (_flags(1), True),                        # one item per wave
    "Q solo": (_flags(2, 2, 0, 1), True),          # first round in index order, the rest ranked; 2 workers per workgroup
    "Q solo early sort": (_flags(2, 2, 0, 1) | (1 << 15), True),   # everything ranked before the first item
    "Q solo unranked": (_flags(2, 2, 1, 1), True),
    "Q solo 1 worker": (_flags(2, 1, 0, 1), True),
    "Q solo 4 workers": (_flags(2, 4, 0, 1), True),
    "Q team": (_flags(2, 0, 0, 2), False),         # long items by 4 waves (other fp32 summation order), short ones in solo quads
    "Q team only": (_flags(2, 0, 0, 2, 1), False),  # every item by 4 waves (what an unranked batch gets)
    "Q team unranked": (_flags(2, 0, 1, 2), False),
}


@pytest.fixture()
def queue_flags():
    """The mode knob exists in the DIAGNOSTIC build only (include/vmi_paged_attention_diag.h): tests that force a mode
    run the operators on that library — the product's sources with -DVMI_DIAG, same kernels — for their duration.
    The natural-trigger tests below (no fixture) drive the same modes in the product library through seq_lens."""
    from vllmini_amd import _lib

    with _lib.use_diag() as lib:
        yield lib.vmi_debug_set_queue_flags
        lib.vmi_debug_set_queue_flags(0)


def _check_all_modes(case, qname, ref_name, set_flags, what, **kw):
    names = _names()
    ref = run_model(case, alibi=kw.get("alibi"))
    plain = run_hip(case, variant=names[ref_name], **kw)
    assert_close(plain, ref, f"{what}: {ref_name}")
    for label, (flags, bitwise) in MODES.items():
        set_flags(flags)
        got = run_hip(case, variant=names[qname], **kw)
        again = run_hip(case, variant=names[qname], **kw)
        set_flags(0)
        assert_close(got, ref, f"{what}: {qname} [{label}]")
        assert np.array_equal(got.view(np.uint16), again.view(np.uint16)), f"{what}: {qname} [{label}] not deterministic"
        if bitwise:
            assert np.array_equal(got.view(np.uint16), plain.view(np.uint16)), \
                f"{what}: {qname} [{label}] differs from {ref_name} ({np.abs(got.astype(np.float64) - plain).max():.3e})"


@pytest.mark.parametrize("D,qname", [(64, "q_d64_s1q2"), (128, "q_d128_s1q1")])
def test_queue_kernel_every_mode_matches_model_and_single_wave_kernel(D, qname, queue_flags):
    """Block-boundary lengths, empty sequences, one long sequence among short ones, NaN-poisoned tails and unowned
    blocks, q as a strided view, more heads than a workgroup has waves."""
    H = 12 if D == 64 else 8
    lens = [1, 15, 16, 17, 31, 33, 0, 100, 64, 257, 1024, 513, 2, 700, 0, 48, 333, 1023, 5, 16]
    rng = np.random.default_rng(4200 + D)
    case = make_case(rng, len(lens), H, D, lens, q_row_pad=2, poison_tail=True)
    _check_all_modes(case, qname, f"d{D}_h1_w1_u1_nt1", queue_flags, f"D={D}")


@pytest.mark.parametrize("B,H", [(96, 12), (100, 7), (77, 5), (160, 12), (600, 4)])
def test_queue_kernel_several_rounds_of_solo_workers(B, H, queue_flags):
    """More items than solo workers, so the hand-out beyond the first round really runs: with a worker count that is a
    multiple of the head count (first round in index order, the rest ranked by a retired wave: 96x12, 100x7, 160x12 =
    three rounds and a part), with one that is not (77x5: everything ranked up front), and with more sequences than the
    late ranking takes (600).  Every row against the kernel model and bit for bit against the one-wave-per-head kernel."""
    rng = np.random.default_rng(4400 + B)
    lens = rng.integers(0, 97, B).tolist()
    lens[3], lens[B // 2] = 200, 0
    case = make_case(rng, B, H, 64, lens, poison_tail=True)
    names = _names()
    ref = run_model(case)
    plain = run_hip(case, variant=names["d64_h1_w1_u1_nt1"])
    assert_close(plain, ref, f"B={B} H={H}: plain kernel")
    for label in ("auto", "Q solo", "Q solo early sort", "Q solo 1 worker", "Q team", "Q team only"):
        flags, bitwise = MODES[label]
        queue_flags(flags)
        got = run_hip(case, variant=names["q_d64_s1q2"])
        queue_flags(0)
        assert_close(got, ref, f"B={B} H={H} [{label}]")
        if bitwise:
            assert np.array_equal(got.view(np.uint16), plain.view(np.uint16)), f"B={B} H={H} [{label}] differs from the plain kernel"


def test_queue_kernel_alibi_gqa_capacity_max_seq_len_and_truncation(queue_flags):
    rng = np.random.default_rng(4300)
    # ALiBi + grouped-query attention (every query head reads its KV head's pages; no tile sharing in these kernels)
    lens = [40, 300, 7, 128, 129, 1]
    case = make_case(rng, len(lens), 8, 64, lens, num_kv_heads=2, q_row_pad=1, poison_tail=True, max_blocks=40)
    alibi = (0.5 ** np.arange(1, 9)).astype(np.float32)
    _check_all_modes(case, "q_d64_s1q2", "d64_h1_w1_u1_nt1", queue_flags, "ALiBi + GQA", alibi=alibi,
                     max_seq_len=40 * BS)  # capacity-style max_seq_len, table rows wider than any sequence needs
    # seq_len > max_seq_len: truncated to the reserved logits (max_seq_len padded to 32), as the plain kernels do
    case = make_case(rng, 3, 4, 64, [100, 20, 64], max_blocks=8)
    trunc = dict(case)
    trunc["lens"] = np.array([64, 20, 64], dtype=np.int32)
    ref = run_model(trunc)
    names = _names()
    for label, (flags, _) in MODES.items():
        queue_flags(flags)
        got = run_hip(case, variant=names["q_d64_s1q2"], max_seq_len=40)
        queue_flags(0)
        assert_close(got, ref, f"truncated context [{label}]")


def test_queue_kernel_bf16(queue_flags):
    """bfloat16 elements (products rounded to bf16, fp32 sums — dtype_bfloat16.cuh): all modes agree with the plain
    bf16 kernel bit for bit (solo) / to 2 bf16 ulp (teams)."""
    from vllmini_amd import ops

    dev = _dev()
    names = _names()
    rng = np.random.default_rng(4400)
    lens = np.array([1, 16, 17, 100, 333, 1024, 47, 2, 0, 640], dtype=np.int32)
    S, H, D, NB = len(lens), 12, 64, 200
    kc = torch.from_numpy(rng.uniform(-1, 1, (NB, H, D // 8, BS, 8)).astype(np.float32)).to(dev).to(torch.bfloat16)
    vc = torch.from_numpy(rng.uniform(-1, 1, (NB, H, D, BS)).astype(np.float32)).to(dev).to(torch.bfloat16)
    q = torch.from_numpy(rng.standard_normal((S, H, D)).astype(np.float32)).to(dev).to(torch.bfloat16)
    nblk = (lens + BS - 1) // BS
    tab = np.full((S, 64), -1, dtype=np.int32)
    perm = rng.permutation(NB).astype(np.int32)
    pos = 0
    for s in range(S):
        tab[s, : nblk[s]] = perm[pos:pos + nblk[s]]
        pos += nblk[s]
    t_tab, t_len = torch.from_numpy(tab).to(dev), torch.from_numpy(lens).to(dev)

    def attend(variant):
        out = torch.full((S, H, D), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.paged_attention_v1(out, q, kc, vc, H, D ** -0.5, t_tab, t_len, BS, 1024, None, "auto", 1.0, 0, 0, 1, 1, 0,
                               _variant=variant)
        torch.cuda.synchronize()
        return out

    plain = attend(names["bf16_d64_bs16_h1_w1_u1_nt1"])
    assert torch.isfinite(plain.float()).all()
    for label, (flags, bitwise) in MODES.items():
        queue_flags(flags)
        got = attend(names["bf16_q_d64_s1q2"])
        queue_flags(0)
        if bitwise:
            assert torch.equal(got.view(torch.int16), plain.view(torch.int16)), label
        else:
            d = (got.float() - plain.float()).abs()
            assert float(d.max()) <= 2 * 2.0 ** -8, f"{label}: {float(d.max()):.3e}"   # 2 bf16 ulp at |x| < 1


def test_queue_kernel_more_sequences_than_the_ranking_holds(queue_flags):
    """num_seqs > 2048: too many sequences to rank in LDS -> items go out in index order; and a grid with far more items
    than waves (one head) still covers every row."""
    from vllmini_amd import ops

    dev = _dev()
    names = _names()
    rng = np.random.default_rng(4500)
    S, H, D, NB = 5000, 2, 64, 96
    kc = torch.from_numpy(rng.uniform(-1, 1, (NB, H, D // 8, BS, 8)).astype(np.float16)).to(dev)
    vc = torch.from_numpy(rng.uniform(-1, 1, (NB, H, D, BS)).astype(np.float16)).to(dev)
    q_np = rng.standard_normal((S, H, D)).astype(np.float16)
    tab_np = rng.integers(0, NB, (S, 3)).astype(np.int32)     # sequences may share pages: read-only
    lens_np = rng.integers(0, 49, S).astype(np.int32)
    t_q, t_tab, t_len = torch.from_numpy(q_np).to(dev), torch.from_numpy(tab_np).to(dev), torch.from_numpy(lens_np).to(dev)
    idx = np.r_[0:40, 2040:2060, S - 40:S]
    ref = oracle.paged_attention_v1(np.ascontiguousarray(q_np[idx]), kc.cpu().numpy(), vc.cpu().numpy(), H, D ** -0.5,
                                    tab_np[idx], lens_np[idx], BS, threads=8)
    outs = []
    for label in ("auto", "Q solo", "Q team"):
        queue_flags(MODES[label][0])
        out = torch.full((S, H, D), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1(out, t_q, kc, vc, H, D ** -0.5, t_tab, t_len, BS, 48, None, "auto", 1.0, 0, 0, 1, 1, 0,
                               _variant=names["q_d64_s1q2"])
        torch.cuda.synchronize()
        queue_flags(0)
        got = out.cpu().numpy()
        assert np.isfinite(got).all(), label
        assert_close(got[idx], ref, f"5000 sequences [{label}]")
        outs.append(got)
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))


def _pages_to_host(wl, table, idx, cfg):
    """The sampled sequences' pages, re-indexed into a small pool the CPU oracle can take."""
    dev = wl.key_cache.device
    tab_dev = table[torch.from_numpy(idx).to(dev)][:, : cfg.blocks_per_seq].clamp(min=0)
    flat = tab_dev.reshape(-1).to(torch.int64)
    kc = wl.key_cache[flat].cpu().numpy()
    vc = wl.value_cache[flat].cpu().numpy()
    small_tab = np.arange(flat.numel(), dtype=np.int32).reshape(len(idx), cfg.blocks_per_seq)
    return kc, vc, small_tab


def test_queue_kernel_full_size_ragged_cfg3_through_the_default_entry():
    """BASELINE cfg3 (B256 H12 D64, pool of 32768 blocks) with seq_lens ~ U{1..1024}: the DEFAULT entry — no hint, no
    variant — must run the balanced kernel in its ranked mode.  Checked: bit-identical to the one-wave-per-head kernel on
    all 3072 rows, determinism, permutation equivariance over sequences (a different ranking, the same rows), and 32
    sequences against the CPU kernel model — the longest and shortest, the first and last workgroups' items."""
    from vllmini_amd import ops
    from vllmini_amd.workload import CONFIGS, make_workload

    dev = _dev()
    names = _names()
    cfg = CONFIGS["cfg3"]
    wl = make_workload(cfg, dev, seed=11, table_sets=2, ragged=True)
    assert ops.variant_names()[ops.pick_variant(cfg.batch, cfg.num_heads, cfg.head_size, cfg.seq_len) - 1] == "q_d64_s1q2"

    def attend(q, table, lens, variant=0):
        out = torch.full((cfg.batch, cfg.num_heads, cfg.head_size), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1(out, q, wl.key_cache, wl.value_cache, cfg.num_heads, wl.scale, table, lens,
                               cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0, _variant=variant)
        torch.cuda.synchronize()
        return out

    base = attend(wl.query, wl.tables[0], wl.seq_lens)
    assert ops.variant_names()[ops.last_variant() - 1] == "q_d64_s1q2"     # what the launch really ran
    assert torch.isfinite(base).all()
    assert torch.equal(base, attend(wl.query, wl.tables[0], wl.seq_lens))
    plain = attend(wl.query, wl.tables[0], wl.seq_lens, names["d64_h4_w1_u1_nt1"])
    assert ops.last_variant() == names["d64_h4_w1_u1_nt1"]
    assert torch.equal(base.view(torch.int16), plain.view(torch.int16))
    perm = torch.randperm(cfg.batch, device=dev)
    permuted = attend(wl.qkv[perm][:, : cfg.num_heads * cfg.head_size].view(cfg.batch, cfg.num_heads, cfg.head_size),
                      wl.tables[0][perm], wl.seq_lens[perm])
    assert torch.equal(permuted, base[perm])
    lens = wl.seq_lens.cpu().numpy()
    order = np.argsort(-lens, kind="stable")
    idx = np.unique(np.r_[order[:8], order[-8:], order[124:132], np.arange(0, 4), np.arange(cfg.batch - 4, cfg.batch)])
    kc, vc, small_tab = _pages_to_host(wl, wl.tables[0], idx, cfg)
    qn = np.ascontiguousarray(wl.query.cpu().numpy()[idx])
    ref = oracle.paged_attention_v1(qn, kc, vc, cfg.num_heads, wl.scale, small_tab, lens[idx], cfg.block_size, threads=8)
    assert_close(base.cpu().numpy()[idx], ref, "cfg3 ragged, default entry, sampled vs model")


def test_queue_kernel_heavy_tailed_batch_runs_teams_and_matches_model(queue_flags):
    """A few long sequences among many short ones (the usual serving batch): the kernel's own choice is the team mode;
    rows match the model and the forced solo mode to summation-order effects."""
    from vllmini_amd import ops
    from vllmini_amd.workload import CONFIGS, make_workload

    dev = _dev()
    names = _names()
    cfg = CONFIGS["cfg3"]
    wl = make_workload(cfg, dev, seed=12, table_sets=1)
    g = torch.Generator().manual_seed(5)
    lens = torch.where(torch.rand(cfg.batch, generator=g) < 0.125, cfg.seq_len, cfg.seq_len // 8).to(torch.int32)
    lens[7] = 0
    t_len = lens.to(dev)
    outs = {}
    for label in ("auto", "Q solo", "Q team", "Q team only"):
        queue_flags(MODES[label][0])
        out = torch.full((cfg.batch, cfg.num_heads, cfg.head_size), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.num_heads, wl.scale, wl.tables[0], t_len,
                               cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0, _variant=names["q_d64_s1q2"])
        torch.cuda.synchronize()
        queue_flags(0)
        outs[label] = out
        assert torch.isfinite(out).all(), label
    assert torch.equal(outs["auto"].view(torch.int16), outs["Q team"].view(torch.int16)), "auto mode should be the team mode here"
    assert float((outs["Q team"].float() - outs["Q solo"].float()).abs().max()) <= 1e-3
    idx = np.unique(np.r_[np.nonzero(lens.numpy() == cfg.seq_len)[0][:6], 7, np.arange(0, 6), np.arange(250, 256)])
    kc, vc, small_tab = _pages_to_host(wl, wl.tables[0], idx, cfg)
    qn = np.ascontiguousarray(wl.query.cpu().numpy()[idx])
    ref = oracle.paged_attention_v1(qn, kc, vc, cfg.num_heads, wl.scale, small_tab, lens.numpy()[idx], cfg.block_size, threads=8)
    assert_close(outs["auto"].cpu().numpy()[idx], ref, "heavy-tailed batch, team mode vs model")


@pytest.mark.parametrize("kind", ["heavy tail", "bimodal", "exponential"])
def test_default_entry_on_heavy_tailed_batches_product_library(kind):
    """PRODUCT library, no knob: BASELINE cfg3 with a few long sequences among many short ones / half long half short /
    exponential lengths.  The kernel's own choice is teams for the long items (more than a quarter of the longest) and
    solo quads for the short ones.  Every short row is bit-identical to the one-wave-per-head kernel (a wave on its own
    repeats its operations), every long row is within the team tolerance of it, every row was written, and a sample of
    long and short sequences matches the CPU kernel model."""
    from vllmini_amd import _lib, ops
    from vllmini_amd.workload import CONFIGS, make_workload

    assert _lib.load().vmi_is_diag_build() == 0
    dev = _dev()
    names = _names()
    cfg = CONFIGS["cfg3"]
    wl = make_workload(cfg, dev, seed=13, table_sets=1)
    g = torch.Generator().manual_seed(6)
    Lm = cfg.seq_len
    if kind == "heavy tail":
        lens = torch.where(torch.rand(cfg.batch, generator=g) < 0.0625, Lm, Lm // 16)
        lens[3] = 1000
    elif kind == "bimodal":
        lens = torch.where(torch.rand(cfg.batch, generator=g) < 0.5, Lm, Lm // 16)
    else:
        lens = torch.clamp((torch.empty(cfg.batch).exponential_(1.0, generator=g) * Lm / 8).long() + 1, max=Lm)
        lens[0] = Lm
    lens[9] = 0
    lens = lens.to(torch.int32)
    t_len = lens.to(dev)

    def attend(variant=0):
        out = torch.full((cfg.batch, cfg.num_heads, cfg.head_size), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.num_heads, wl.scale, wl.tables[0], t_len,
                               cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0, _variant=variant)
        torch.cuda.synchronize()
        return out

    got = attend()
    assert ops.variant_names()[ops.last_variant() - 1] == "q_d64_s1q2"
    assert torch.isfinite(got).all()
    assert torch.equal(got, attend())                                  # deterministic
    plain = attend(names["d64_h4_w1_u1_nt1"])
    maxL = int(lens.max())
    short = (lens * 4 < maxL - 4)                                      # clearly below the long/short boundary
    long_ = (lens * 4 > maxL + 4)
    assert int(short.sum()) > 20 and int(long_.sum()) >= 4
    sd, ld = short.to(dev), long_.to(dev)
    assert torch.equal(got[sd].view(torch.int16), plain[sd].view(torch.int16)), "a short row differs from the one-wave kernel"
    assert float((got[ld].float() - plain[ld].float()).abs().max()) <= 5e-4
    if kind != "exponential":                                           # long rows really went through the 4-wave path
        assert not torch.equal(got[ld].view(torch.int16), plain[ld].view(torch.int16))
    li, si = np.nonzero(long_.numpy())[0], np.nonzero(short.numpy())[0]
    idx = np.unique(np.r_[li[:5], si[:5], si[-3:], 9])
    kc, vc, small_tab = _pages_to_host(wl, wl.tables[0], idx, cfg)
    qn = np.ascontiguousarray(wl.query.cpu().numpy()[idx])
    ref = oracle.paged_attention_v1(qn, kc, vc, cfg.num_heads, wl.scale, small_tab, lens.numpy()[idx], cfg.block_size, threads=8)
    assert_close(got.cpu().numpy()[idx], ref, f"{kind}: default entry, sampled vs model")


@pytest.mark.parametrize("kvd", ["auto", "fp8"])
@pytest.mark.parametrize("B,spread", [(257, 0.0), (288, 0.15), (320, 0.0), (512, 0.5)])
def test_default_entry_more_items_than_resident_waves_is_several_waves_per_head(B, spread, kvd):
    """PRODUCT library, no knob: 12 heads x (257 ...) sequences are more items than the chip holds waves (3072 on 256 CUs).
    Since the end of round 3 the default there is a plain several-waves-per-head kernel — eight over fp16 pages, four over
    fp8 pages: many times the resident waves, balanced by the hardware dispatcher — not the balanced kernel (whose overflow
    twin it replaced: level on equal lengths, ahead on ragged ones; profiles/r03z_eight_waves_per_head.md).  Every row
    written, deterministic, within the tolerance of the one-wave-per-head kernel, a sample against the CPU kernel model."""
    import dataclasses

    from vllmini_amd import _lib, ops
    from vllmini_amd.workload import CONFIGS, make_workload

    assert _lib.load().vmi_is_diag_build() == 0
    dev = _dev()
    if torch.cuda.get_device_properties(dev).multi_processor_count != 256:
        pytest.skip("sized for the 256 CUs of an MI355X")
    names = _names()
    cfg = dataclasses.replace(CONFIGS["cfg3"], name=f"b{B}", batch=B, num_blocks=B * 64 + 8)
    wl = make_workload(cfg, dev, seed=70 + B, table_sets=1)
    g = torch.Generator().manual_seed(B)
    lens = (cfg.seq_len - (torch.rand(B, generator=g) * spread * cfg.seq_len).long()).to(torch.int32)
    lens[0] = cfg.seq_len
    t_len = lens.to(dev)
    f8 = kvd == "fp8"
    if f8:
        g8 = torch.Generator(device=dev).manual_seed(9)
        D = cfg.head_size
        kc = torch.randint(0, 64, (cfg.num_blocks, cfg.kv_heads, D // 16, 16, 16), dtype=torch.uint8, device=dev, generator=g8)
        vc = torch.randint(0, 64, (cfg.num_blocks, cfg.kv_heads, D, 16), dtype=torch.uint8, device=dev, generator=g8)
    else:
        kc, vc = wl.key_cache, wl.value_cache

    def attend(variant=0):
        out = torch.full((B, cfg.num_heads, cfg.head_size), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1(out, wl.query, kc, vc, cfg.num_heads, wl.scale, wl.tables[0], t_len, cfg.block_size, cfg.seq_len,
                               None, kvd, 1.0, 0, 0, 1, 1, 0, _variant=variant)
        torch.cuda.synchronize()
        return out

    got = attend()
    assert ops.last_launch_label() == ("fp8_d64_bs16_h1_w4_u2_nt1" if f8 else "d64_h1_w8_u1_nt1")
    assert torch.isfinite(got).all()
    assert torch.equal(got, attend())
    plain = attend(names["fp8_d64_bs16_h1_w1_u2_nt1" if f8 else "d64_h4_w1_u1_nt1"])
    assert float((got.float() - plain.float()).abs().max()) <= 1e-3 * (2.0 if f8 else 1.0)
    idx = np.unique(np.r_[0, 1, B // 2, B - 2, B - 1])
    tab_dev = wl.tables[0][torch.from_numpy(idx).to(dev)][:, : cfg.blocks_per_seq].clamp(min=0)
    flat = tab_dev.reshape(-1).to(torch.int64)
    small_tab = np.arange(flat.numel(), dtype=np.int32).reshape(len(idx), cfg.blocks_per_seq)
    qn = np.ascontiguousarray(wl.query.cpu().numpy()[idx])
    if f8:
        ref = oracle.paged_attention_v1_fp8(qn, kc[flat].cpu().numpy(), vc[flat].cpu().numpy(), cfg.num_heads, wl.scale, small_tab,
                                            lens.numpy()[idx], cfg.block_size, kv_scale=1.0, threads=8)
    else:
        ref = oracle.paged_attention_v1(qn, kc[flat].cpu().numpy(), vc[flat].cpu().numpy(), cfg.num_heads, wl.scale, small_tab,
                                        lens.numpy()[idx], cfg.block_size, threads=8)
    assert_close(got.cpu().numpy()[idx], ref, f"batch {B} {kvd}: default entry, sampled vs model", vmax=2.0 if f8 else 1.0)


def test_head_128_default_entry_is_a_gated_double_launch():
    """BASELINE cfg4 (B128 H32 D128 L2048): the default entry launches the lockstep multi-head kernel AND the balanced
    kernel; each decides on the device, from the same statistics, whether the batch is its kind.  Equal lengths: rows
    bit-identical to the multi-head kernel alone.  Ragged lengths: rows bit-identical to the balanced kernel alone,
    every row written exactly once (the output starts as NaN), sampled sequences match the CPU kernel model."""
    from vllmini_amd import ops
    from vllmini_amd.workload import CONFIGS, make_workload

    dev = _dev()
    names = _names()
    cfg = CONFIGS["cfg4"]
    wl = make_workload(cfg, dev, seed=21, table_sets=1)

    def attend(lens, variant=0):
        out = torch.full((cfg.batch, cfg.num_heads, cfg.head_size), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.num_heads, wl.scale, wl.tables[0], lens,
                               cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0, _variant=variant)
        torch.cuda.synchronize()
        return out

    uniform = attend(wl.seq_lens)
    assert torch.isfinite(uniform).all()
    assert torch.equal(uniform.view(torch.int16), attend(wl.seq_lens, names["d128_mh4_h4_u1_nt1_lock"]).view(torch.int16))
    g = torch.Generator().manual_seed(3)
    for what, lens in (("U{1..2048}", torch.randint(1, cfg.seq_len + 1, (cfg.batch,), generator=g)),
                       ("just ragged enough", torch.full((cfg.batch,), int(0.79 * cfg.seq_len)).index_fill_(0, torch.tensor([5]), cfg.seq_len)),
                       ("just not ragged", torch.full((cfg.batch,), int(0.81 * cfg.seq_len)).index_fill_(0, torch.tensor([5]), cfg.seq_len))):
        t_len = lens.to(torch.int32).to(dev)
        got = attend(t_len)
        assert torch.isfinite(got).all(), f"{what}: a row was written by neither kernel"
        alone = attend(t_len, names["d128_mh4_h4_u1_nt1_lock" if what == "just not ragged" else "q_d128_s1q1"])
        assert torch.equal(got.view(torch.int16), alone.view(torch.int16)), what
    lens = torch.randint(1, cfg.seq_len + 1, (cfg.batch,), generator=g).to(torch.int32)
    got = attend(lens.to(dev))
    label = ops.last_launch_label()        # both kernels of the double launch are named, whichever did the work
    assert label.startswith("d128_mh4_h4_u1_nt1_lock | q_d128_s1q1") and "gated" in label, label
    order = np.argsort(-lens.numpy(), kind="stable")
    idx = np.unique(np.r_[order[:3], order[-3:], 0, cfg.batch - 1])
    kc, vc, small_tab = _pages_to_host(wl, wl.tables[0], idx, cfg)
    qn = np.ascontiguousarray(wl.query.cpu().numpy()[idx])
    ref = oracle.paged_attention_v1(qn, kc, vc, cfg.num_heads, wl.scale, small_tab, lens.numpy()[idx], cfg.block_size, threads=8)
    assert_close(got.cpu().numpy()[idx], ref, "cfg4 ragged, default entry, sampled vs model")


def test_two_host_threads_alternate_large_max_seq_len_launches():
    """Two host threads (each on its own stream) alternate launches whose logits need more than 48 KiB of LDS with two
    different max_seq_len values: the dynamic-LDS attribute is granted per launch, not remembered in a shared table
    (round 1 kept a per-process cache there: a data race), so no launch may fail or use a stale grant."""
    import threading

    from vllmini_amd import ops

    dev = _dev()
    rng = np.random.default_rng(4700)
    lens = [1500, 40, 13000, 7]
    case = make_case(rng, len(lens), 4, 64, lens, max_blocks=1024)
    ref = run_model(case)
    S, H, D = case["q"].shape
    kc, vc = torch.from_numpy(case["kc"]).to(dev), torch.from_numpy(case["vc"]).to(dev)
    q = torch.from_numpy(case["q"].copy()).to(dev)
    tab, ln = torch.from_numpy(case["tables"]).to(dev), torch.from_numpy(case["lens"]).to(dev)
    errors, results = [], {}

    def worker(k):
        try:
            stream = torch.cuda.Stream(dev)
            with torch.cuda.stream(stream):
                for i in range(40):
                    msl = (13056, 16384)[(i + k) % 2]
                    out = torch.full((S, H, D), float("nan"), dtype=torch.float16, device=dev)
                    ops.paged_attention_v1(out, q, kc, vc, H, case["scale"], tab, ln, BS, msl, None, "auto", 1.0, 0, 0, 1, 1, 0)
                stream.synchronize()
                results[k] = out.cpu().numpy()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for k in (0, 1):
        assert_close(results[k], ref, f"thread {k}")


@pytest.mark.parametrize("qname,kvd", [("fp8_q_d64_s2q4", "fp8"), ("fp8_q_d64_s1q2", "fp8"), ("fp8e5m2_q_d64_s2q4", "fp8_e5m2"),
                                       ("fp8_q_d128_s1q2", "fp8"), ("fp8_q_d64_s2q4m", "fp8"), ("fp8_q_d128_s1q2m", "fp8")])
def test_queue_kernel_over_fp8_pages_every_mode(qname, kvd, queue_flags):
    """fp8 pages (kv_scale 1: every cache element is half(float(fp8)), quant_utils.cuh:295-300) through every mode of the
    balanced kernel, against the CPU kernel model; single-wave modes agree with each other bit for bit; any other
    kv_scale is refused for these kernels (the operator's automatic choice then stays with pa_v1_kernel)."""
    from test_parity_gpu import _fp8_case
    from vllmini_amd import ops

    dev = _dev()
    names = _names()
    D = 128 if "d128" in qname else 64
    H = 8
    lens = [1, 16, 17, 100, 333, 47, 700, 2, 0, 1024, 513, 31]
    rng = np.random.default_rng(4800 + D + len(kvd))
    case = _fp8_case(rng, len(lens), H, D, lens, 16, num_kv_heads=4)
    if kvd == "fp8_e5m2":   # E5M2 bytes: keep exponent field <= 15 (|x| < 2) and drop the Inf / NaN codes
        for key in ("kq", "vq"):
            a = case[key]
            case[key] = np.where((a & 0x7f) >= 0x40, (a & 0x80) | 0x38 | (a & 3), a).astype(np.uint8)
    ref = oracle.paged_attention_v1_fp8(case["q"], case["kq"], case["vq"], 4, case["scale"], case["tables"], case["lens"],
                                        16, kv_scale=1.0, threads=8, e5m2=kvd == "fp8_e5m2")
    S = len(lens)
    q = torch.from_numpy(case["qbuf"]).to(dev)[:, : H * D].view(S, H, D)
    kq, vq = torch.from_numpy(case["kq"]).to(dev), torch.from_numpy(case["vq"]).to(dev)
    tab, ln = torch.from_numpy(case["tables"]).to(dev), torch.from_numpy(case["lens"]).to(dev)

    def attend(variant, kv_scale=1.0):
        out = torch.full((S, H, D), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1(out, q, kq, vq, 4, case["scale"], tab, ln, 16, 1024, None, kvd, kv_scale, 0, 0, 1, 1, 0,
                               _variant=variant)
        torch.cuda.synchronize()
        return out.cpu().numpy()

    solo = None
    for label, (flags, bitwise) in MODES.items():
        queue_flags(flags)
        got = attend(names[qname])
        queue_flags(0)
        assert_close(got, ref, f"{qname} [{label}]", vmax=2.0)
        if bitwise:
            solo = got if solo is None else solo
            assert np.array_equal(got.view(np.uint16), solo.view(np.uint16)), f"{qname} [{label}] differs from mode S"
    with pytest.raises(RuntimeError, match="kv_scale 1"):
        attend(names[qname], kv_scale=0.5)
    assert_close(attend(0), ref, "default entry, kv_scale 1", vmax=2.0)


def test_wave_timeline_of_the_diagnostic_library():
    """vmi_diag_set_wave_timeline (include/vmi_paged_attention_diag.h): every wave of a balanced-kernel launch leaves its
    start and end time, HW_ID and XCC_ID; switching it off stops the writes.  scripts/wave_timeline_probe.py is built on it."""
    from vllmini_amd import _lib, ops
    from vllmini_amd.workload import CONFIGS, make_workload

    dev = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    cfg = CONFIGS["cfg3"]
    wl = make_workload(cfg, dev, seed=2, table_sets=1)
    out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)

    def launch():
        ops.paged_attention_v1(out, wl.query, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, wl.tables[0], wl.seq_lens,
                               cfg.block_size, cfg.seq_len, None, "auto", 1.0, 0, 0, 1, 1, 0)

    with _lib.use_diag() as lib:
        launch()
        if not ops.last_launch_label().startswith("q_d64"):
            pytest.skip("the default entry did not pick the balanced kernel on this device")
        waves = 3 * cus * 4
        rec = torch.zeros((waves + 64, 4), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        assert lib.vmi_diag_set_wave_timeline(rec.data_ptr(), 0) == 0
        launch()
        torch.cuda.synchronize()
        assert lib.vmi_diag_set_wave_timeline(None, 0) == 0
        r = rec.cpu().numpy()
        assert (r[waves:] == 0).all()                                  # nothing past the launch's waves
        r = r[:waves]
        assert (r[:, 0] > 0).all() and (r[:, 1] > r[:, 0]).all()
        span_us = (r[:, 1].max() - r[:, 0].min()) * 0.01                # 100 MHz ticks
        assert 20.0 < span_us < 1000.0, span_us
        assert set(np.unique(r[:, 3] & 0xF).tolist()) <= set(range(8)) and len(np.unique(r[:, 3] & 0xF)) >= 2
        rec.zero_()
        launch()
        torch.cuda.synchronize()
        assert (rec == 0).all()                                        # switched off


@pytest.mark.parametrize("B,ragged,kernel", [(128, False, "fp8_d64_bs16_h1_w4_u2_nt1"), (176, True, "fp8_d64_bs16_h1_w4_u2_nt1"),
                                             (224, True, "fp8_q_d64_s2q4m"), (224, False, "fp8_q_d64_s2q4m")])
def test_default_entry_over_fp8_pages_between_half_a_chip_and_a_full_one(B, ragged, kernel):
    """PRODUCT library, no knob, fp8 pages (kv_scale 1), 12 heads: from half the resident waves to 85 % of them (batch 128 ..
    217) the default is FOUR waves per head (twice the resident waves: half-size tiles want the requests, and the dispatcher
    balances ragged batches), from there on the balanced kernel (profiles/r03x_fp8_four_solo_workers.md).  Every row finite
    and deterministic, within the tolerance of the one-wave-per-head fp8 kernel; a sample of sequences against the CPU
    kernel model."""
    import dataclasses

    from vllmini_amd import _lib, ops
    from vllmini_amd.workload import CONFIGS, make_workload

    assert _lib.load().vmi_is_diag_build() == 0
    dev = _dev()
    if torch.cuda.get_device_properties(dev).multi_processor_count != 256:
        pytest.skip("sized for the 256 CUs of an MI355X")
    names = _names()
    cfg = dataclasses.replace(CONFIGS["cfg3"], name=f"b{B}", batch=B, num_blocks=B * 64 + 8)
    wl = make_workload(cfg, dev, seed=90 + B, table_sets=1, ragged=ragged)
    lens = wl.seq_lens.cpu().numpy()
    D = cfg.head_size
    g8 = torch.Generator(device=dev).manual_seed(11)
    kc = torch.randint(0, 64, (cfg.num_blocks, cfg.kv_heads, D // 16, 16, 16), dtype=torch.uint8, device=dev, generator=g8)
    vc = torch.randint(0, 64, (cfg.num_blocks, cfg.kv_heads, D, 16), dtype=torch.uint8, device=dev, generator=g8)

    def attend(variant=0):
        out = torch.full((B, cfg.num_heads, D), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1(out, wl.query, kc, vc, cfg.num_heads, wl.scale, wl.tables[0], wl.seq_lens, cfg.block_size,
                               cfg.seq_len, None, "fp8", 1.0, 0, 0, 1, 1, 0, _variant=variant)
        torch.cuda.synchronize()
        return out

    got = attend()
    assert ops.last_launch_label() == kernel
    assert torch.isfinite(got).all() and torch.equal(got, attend())
    plain = attend(names["fp8_d64_bs16_h1_w1_u2_nt1"])    # (the "m" kernel's q.K^T runs on the matrix cores: other fp32 summation
    assert float((got.float() - plain.float()).abs().max()) <= 2e-3    #  order than the plain kernel's, same tolerance)
    idx = np.unique(np.r_[0, 1, B // 2, B - 1, int(np.argmax(lens)), int(np.argmin(lens))])
    tab_dev = wl.tables[0][torch.from_numpy(idx).to(dev)][:, : cfg.blocks_per_seq].clamp(min=0)
    flat = tab_dev.reshape(-1).to(torch.int64)
    small_tab = np.arange(flat.numel(), dtype=np.int32).reshape(len(idx), cfg.blocks_per_seq)
    ref = oracle.paged_attention_v1_fp8(np.ascontiguousarray(wl.query.cpu().numpy()[idx]), kc[flat].cpu().numpy(), vc[flat].cpu().numpy(),
                                        cfg.num_heads, wl.scale, small_tab, lens[idx], cfg.block_size, kv_scale=1.0, threads=8)
    assert_close(got.cpu().numpy()[idx], ref, f"fp8 default entry, batch {B}", vmax=2.0)
