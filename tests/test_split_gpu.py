"""The split kernels (vllmini_amd/csrc/pa_split.hpp): paged_attention_v1 with one (sequence, head) spread over several
workgroups of ONE launch that meet in a caller-owned workspace (C-ABI 21: vmi_paged_attention_v1_f16_ws).

Checked here, through the drop-in surface -> C-ABI, against the CPU oracle (checker only):
  * every split kernel of the menu, at the tight bound (2 fp16 ulp) — the probabilities are normalised with the item's
    GLOBAL max / exp-sum before they are rounded to fp16, as the reference does (attention_kernels.cu:334-346, 398-400);
  * the workspace contract: NULL / too small = yesterday's kernels bit for bit; a split kernel asked for by id without a
    workspace is an error; a launch leaves every control word zero again; a poisoned workspace is healed by the reset entry;
  * re-use: many launches back to back on one workspace, hipGraph replay, two streams, two host threads;
  * run-to-run determinism (the partial rows are added in workgroup order, whatever the arrival order).

Nothing here reads /root/reference.
"""
from __future__ import annotations

import threading

import numpy as np
import pytest
import torch

from helpers import make_case
from test_parity_gpu import _dev, assert_close, run_model

pytestmark = pytest.mark.gpu

LENS = [1, 16, 17, 100, 333, 1024, 47, 0, 700, 2, 513, 31]


def _split_names(D=None):
    from vllmini_amd import ops

    return [(i + 1, n) for i, n in enumerate(ops.variant_names())
            if "_x" in n and "_gq" not in n and n.startswith("d") and (D is None or n.startswith(f"d{D}_"))]


def _upload(case, dev):
    S, H, D = case["q"].shape
    qbuf = torch.from_numpy(case["qbuf"]).to(dev)
    return dict(q=qbuf[:, : H * D].view(S, H, D), kc=torch.from_numpy(case["kc"]).to(dev),
                vc=torch.from_numpy(case["vc"]).to(dev), tab=torch.from_numpy(case["tables"]).to(dev),
                lens=torch.from_numpy(case["lens"]).to(dev), msl=max(int(case["lens"].max()), 1), _keep=qbuf)


def _launch(case, t, variant=0, out=None):
    from vllmini_amd import ops

    S, H, D = case["q"].shape
    if out is None:
        out = torch.full((S, H, D), float("nan"), dtype=torch.float16, device=t["q"].device)
    ops.paged_attention_v1(out, t["q"], t["kc"], t["vc"], case["num_kv_heads"], case["scale"], t["tab"], t["lens"], 16,
                           t["msl"], None, "auto", 1.0, 0, 0, 1, 1, 0, _variant=variant)
    return out


def _control_words_are_zero(index=0):
    """status, arrival counters and granules of the current stream's workspace (everything in front of the partial rows)."""
    from vllmini_amd import ops

    ws = ops.workspace_for(index, create=False)
    assert ws is not None
    n = 256 + 8192 * 4 + 2048 * 4 * 8
    torch.cuda.synchronize()
    return int(ws[:n].view(torch.int64).ne(0).sum().item()) == 0


@pytest.mark.parametrize("D,H", [(64, 12), (128, 8)])
def test_every_split_kernel_matches_the_kernel_model(D, H):
    from vllmini_amd import ops

    dev = _dev()
    rng = np.random.default_rng(500 + D)
    case = make_case(rng, len(LENS), H, D, LENS, q_row_pad=2, poison_tail=True)
    t = _upload(case, dev)
    ref = run_model(case)
    names = _split_names(D)
    assert len(names) >= 12
    ran = 0
    for vid, name in names:
        got = _launch(case, t, vid).cpu().numpy()     # (144 items x 16+ workgroups: more than are resident -> in rounds)
        assert_close(got, ref, name)
        assert ops.workspace_status(0) == 0, name
        ran += 1
    assert ran == len(names)
    assert _control_words_are_zero()


@pytest.mark.parametrize("D", [64, 128])
def test_more_items_than_are_resident_go_in_rounds(D):
    """A launch of more workgroups than the chip holds (or than the workspace has words for) is served by the kernel's twin
    that goes in ROUNDS (Variant::fn_rounds): a grid of whole items that IS resident, every workgroup serving grid-strided
    items, the workspace indexed by the place in the grid with two halves alternating by round.  Ragged and empty sequences
    (workgroups that leave a round early), 4 ... 40 rounds, fp16 / grouped / fp8 kernels; against the kernel model, bit for
    bit against the one-round kernel on a sub-batch that fits, the same bits on every launch, workspace clean afterwards."""
    import oracle
    from test_parity_gpu import _fp8_case, _run_fp8
    from vllmini_amd import ops

    dev = _dev()
    names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
    rng = np.random.default_rng(1200 + D)
    S, H = 56, 12
    lens = rng.integers(0, 900, S)
    lens[[3, 17, 40]] = [3000, 0, 2049]
    lens[-1] = 1
    case = make_case(rng, S, H, D, lens.tolist(), q_row_pad=1, poison_tail=True)
    t = _upload(case, dev)
    ref = run_model(case)
    sub = 2                                                   # 24 items: resident for every width
    for name in (f"d{D}_x8_u2_nt0", f"d{D}_x32_u2_nt1", f"d{D}_x128_u2_nt0", f"d{D}_x256_u2_nt0"):
        x = int(name.split("_x")[1].split("_")[0])
        if S * H * (x // 4) <= 3 * 256:
            continue                                          # (would not need rounds)
        got = _launch(case, t, names[name])
        assert_close(got.cpu().numpy(), ref, name)
        for _ in range(2):
            assert torch.equal(_launch(case, t, names[name]).view(torch.int16), got.view(torch.int16)), name
        one = torch.full((sub, H, D), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1(one, t["q"][:sub], t["kc"], t["vc"], H, case["scale"], t["tab"][:sub], t["lens"][:sub], 16, t["msl"],
                               None, "auto", 1.0, 0, 0, 1, 1, 0, _variant=names[name])
        assert torch.equal(one.view(torch.int16), got[:sub].view(torch.int16)), f"{name}: rounds vs one round"
        assert ops.workspace_status(0) == 0 and _control_words_are_zero(), name
    # four query heads of a KV head per item
    hkv = 3
    case = make_case(rng, S, hkv * 4, D, lens.tolist(), num_kv_heads=hkv, poison_tail=True)
    t = _upload(case, dev)
    ref = run_model(case)
    for name in (f"d{D}_gq4_x16_u{2 if D == 64 else 1}_nt0", f"d{D}_gq4_x64_u{2 if D == 64 else 1}_nt1"):
        got = _launch(case, t, names[name])
        assert_close(got.cpu().numpy(), ref, name)
        assert torch.equal(_launch(case, t, names[name]).view(torch.int16), got.view(torch.int16)), name
    assert ops.workspace_status(0) == 0 and _control_words_are_zero()
    # fp8 pages
    fcase = _fp8_case(rng, S, 8, D, lens.tolist(), 16, num_kv_heads=4)
    fref = oracle.paged_attention_v1_fp8(fcase["q"], fcase["kq"], fcase["vq"], 4, fcase["scale"], fcase["tables"], fcase["lens"], 16,
                                         kv_scale=0.7, threads=8)
    for name in (f"fp8_d{D}_x16_u2_nt0", f"fp8_d{D}_x128_u2_nt1"):
        assert_close(_run_fp8(fcase, 0.7, variant=names[name]), fref, name, vmax=1.4)
    assert ops.workspace_status(0) == 0 and _control_words_are_zero()


@pytest.mark.parametrize("D", [64, 128])
def test_long_contexts_use_every_wave_of_the_widest_split_kernels(D):
    """An item uses 4 * floor(blocks / 16) of its waves, so only contexts of 4096+ tokens reach 64 ... 256 waves per item
    (several granules per lane in the poll, partial rows of 16 ... 64 workgroups added by the last one's four waves)."""
    from vllmini_amd import ops

    dev = _dev()
    rng = np.random.default_rng(900 + D)
    lens = [16384, 9000, 40, 4097]
    case = make_case(rng, len(lens), 2, D, lens)
    t = _upload(case, dev)
    ref = run_model(case)
    names = dict((n, i) for i, n in _split_names(D))
    first = None
    for x in (64, 128, 256):
        for u, nt in ((2, 0), (2, 1)):
            name = f"d{D}_x{x}_u{u}_nt{nt}"
            got = _launch(case, t, names[name])
            assert_close(got.cpu().numpy(), ref, name)
            for _ in range(3):      # the same bits every time, whichever workgroup arrives last
                assert torch.equal(_launch(case, t, names[name]).view(torch.int16), got.view(torch.int16)), name
    assert ops.workspace_status(0) == 0 and _control_words_are_zero()


@pytest.mark.parametrize("D", [64, 128])
def test_fp8_pages_through_the_split_kernels(D):
    """fp8 E4M3 pages ("fp8_d<head>_x<waves>_..."): every element becomes half(float(fp8) * kv_scale) first, then the fp16
    arithmetic of the fp16 kernels — against the oracle's fp8 restatement, kv_scale 1 and 0.6, long and short contexts, and the
    default entry (vmi_paged_attention_v1_fp8_ws) where it picks one."""
    import oracle
    from test_parity_gpu import _fp8_case, _run_fp8
    from vllmini_amd import ops

    rng = np.random.default_rng(40 + D)
    lens = [9000, 1, 700, 0, 4097, 33]
    H, hkv = 4, 2
    case = _fp8_case(rng, len(lens), H, D, lens, 16, num_kv_heads=hkv)
    names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
    mine = [n for n in names if n.startswith(f"fp8_d{D}_x")]
    assert len(mine) >= 10
    for kv_scale in (1.0, 0.6):
        ref = oracle.paged_attention_v1_fp8(case["q"], case["kq"], case["vq"], hkv, case["scale"], case["tables"], case["lens"], 16,
                                            kv_scale=kv_scale, threads=8)
        for n in mine:
            got = _run_fp8(case, kv_scale, variant=names[n])
            assert_close(got, ref, f"{n} kv_scale {kv_scale}", vmax=2 * kv_scale)
        got = _run_fp8(case, kv_scale)                          # default entry: 24 items x 9000 tokens -> a split kernel
        assert ops.last_launch_label().startswith(f"fp8_d{D}_x"), ops.last_launch_label()
        assert_close(got, ref, f"default entry kv_scale {kv_scale}", vmax=2 * kv_scale)
    assert ops.workspace_status(0) == 0 and _control_words_are_zero()


@pytest.mark.parametrize("D", [64, 128])
def test_four_query_heads_of_a_kv_head_per_item(D):
    """The "gq4" split kernels: an item is four query heads of one KV head — every K / V tile of a wave's blocks loaded once,
    q.K^T on the matrix cores (exact fp16 products, fp32 accumulation), every head with its own granules, probabilities and
    partial rows.  Against the kernel model at the tight bound; ragged, empty, ALiBi, 4 and 8 query heads per KV head; the
    same bits on a second launch; and the default entry where it picks this form."""
    import oracle
    from vllmini_amd import ops

    dev = _dev()
    names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
    mine = [n for n in names if n.startswith(f"d{D}_gq4_x")]
    assert len(mine) >= 8
    for (S, hkv, qpk, lens) in ((3, 2, 4, [4096, 100, 0]), (2, 1, 8, [1500, 17]), (1, 2, 4, [9000]), (5, 1, 4, [1, 16, 33, 700, 2048])):
        H = hkv * qpk
        rng = np.random.default_rng(D + S + H)
        case = make_case(rng, S, H, D, lens, num_kv_heads=hkv, q_row_pad=1, poison_tail=True)
        slopes = rng.uniform(0.01, 0.3, H).astype(np.float32) if S == 2 else None
        ref = oracle.paged_attention_v1(case["q"], case["kc"], case["vc"], hkv, case["scale"], case["tables"], case["lens"], 16,
                                        alibi_slopes=slopes, threads=8)
        t = _upload(case, dev)
        al = None if slopes is None else torch.from_numpy(slopes).to(dev)
        for n in mine:
            outs = []
            for _ in range(2):
                out = torch.full((S, H, D), float("nan"), dtype=torch.float16, device=dev)
                ops.paged_attention_v1(out, t["q"], t["kc"], t["vc"], hkv, case["scale"], t["tab"], t["lens"], 16, t["msl"], al,
                                       "auto", 1.0, 0, 0, 1, 1, 0, _variant=names[n])
                outs.append(out)
            assert_close(outs[0].cpu().numpy(), ref, f"{n} S{S} H{H}/{hkv} lens {lens}")
            assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), n
    # the default entry: 8 sequences x 32 / 8 heads = 256 query heads -> four per item
    rng = np.random.default_rng(7 + D)
    case = make_case(rng, 8, 32, D, [2048] * 8, num_kv_heads=8)
    t = _upload(case, dev)
    got = _launch(case, t).cpu().numpy()
    assert f"d{D}_gq4_x" in ops.last_launch_label(), ops.last_launch_label()
    assert_close(got, run_model(case), f"default entry ({ops.last_launch_label()})")
    assert ops.workspace_status(0) == 0 and _control_words_are_zero()


@pytest.mark.parametrize("D", [64, 128])
def test_four_query_heads_per_item_over_fp8_pages(D):
    """"fp8_d<head>_gq4_x...": the fp8 tile of a wave's block decoded ONCE for the four query heads of its KV head (q.K^T on the
    matrix cores from half(float(fp8) * kv_scale) operands, V decoded once per tile).  Against the oracle's fp8 restatement;
    kv_scale 1 and 0.6; ragged, empty; 4 and 8 query heads per KV head; ALiBi; the default entry where it picks this form."""
    import oracle
    from test_parity_gpu import _fp8_case, _run_fp8
    from vllmini_amd import ops

    names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
    mine = [n for n in names if n.startswith(f"fp8_d{D}_gq4_x")]
    assert len(mine) >= 8
    for (S, hkv, qpk, lens) in ((3, 2, 4, [4096, 100, 0]), (2, 1, 8, [1500, 17]), (5, 1, 4, [1, 16, 33, 700, 2048])):
        H = hkv * qpk
        rng = np.random.default_rng(3 * D + S + H)
        case = _fp8_case(rng, S, H, D, lens, 16, num_kv_heads=hkv)
        slopes = rng.uniform(0.01, 0.3, H).astype(np.float32) if S == 2 else None
        for kv_scale in (1.0, 0.6):
            ref = oracle.paged_attention_v1_fp8(case["q"], case["kq"], case["vq"], hkv, case["scale"], case["tables"], case["lens"], 16,
                                                kv_scale=kv_scale, alibi_slopes=slopes, threads=8)
            for n in mine:
                got = _run_fp8(case, kv_scale, variant=names[n], alibi=slopes)
                assert_close(got, ref, f"{n} S{S} H{H}/{hkv} kv_scale {kv_scale}", vmax=2 * kv_scale)
                assert np.array_equal(_run_fp8(case, kv_scale, variant=names[n], alibi=slopes).view(np.int16), got.view(np.int16)), n
    # the default entry: 8 sequences x 32 / 8 heads = 256 query heads -> four per item
    rng = np.random.default_rng(11 + D)
    case = _fp8_case(rng, 8, 32, D, [2048] * 8, 16, num_kv_heads=8)
    ref = oracle.paged_attention_v1_fp8(case["q"], case["kq"], case["vq"], 8, case["scale"], case["tables"], case["lens"], 16,
                                        kv_scale=0.8, threads=8)
    got = _run_fp8(case, 0.8)
    assert f"fp8_d{D}_gq4_x" in ops.last_launch_label(), ops.last_launch_label()
    assert_close(got, ref, f"default entry ({ops.last_launch_label()})", vmax=1.6)
    assert ops.workspace_status(0) == 0 and _control_words_are_zero()


def test_grouped_query_and_alibi_through_a_split_kernel():
    import oracle

    dev = _dev()
    rng = np.random.default_rng(77)
    lens = [5, 64, 300, 129]
    case = make_case(rng, len(lens), 8, 64, lens, num_kv_heads=2)
    t = _upload(case, dev)
    slopes = (0.5 ** np.arange(1, 9)).astype(np.float32)
    from vllmini_amd import ops

    vid = dict((n, i) for i, n in _split_names(64))["d64_x8_u2_nt0"]
    out = torch.full((4, 8, 64), float("nan"), dtype=torch.float16, device=dev)
    ops.paged_attention_v1(out, t["q"], t["kc"], t["vc"], 2, case["scale"], t["tab"], t["lens"], 16, t["msl"],
                           torch.from_numpy(slopes).to(dev), "auto", 1.0, 0, 0, 1, 1, 0, _variant=vid)
    ref = oracle.paged_attention_v1(case["q"], case["kc"], case["vc"], 2, case["scale"], case["tables"], case["lens"], 16,
                                    alibi_slopes=slopes, threads=8)
    assert_close(out.cpu().numpy(), ref, "gqa + alibi", vmax=1.0)


def test_without_a_workspace_the_entry_is_yesterdays_kernel_bit_for_bit():
    """NULL workspace, a workspace that is too small, and the plain entries all run the same kernel."""
    from vllmini_amd import _lib, ops

    dev = _dev()
    lib = _lib.load()
    rng = np.random.default_rng(3)
    lens = [512] * 32
    case = make_case(rng, 32, 12, 64, lens)
    t = _upload(case, dev)
    prev = ops.set_workspace_enabled(False)
    try:
        plain = _launch(case, t)
        plain_kernel = ops.last_launch_label()
    finally:
        ops.set_workspace_enabled(prev)
    args = ops._pa_common(torch.empty_like(plain), t["q"], t["kc"], t["vc"], 12, case["scale"], t["tab"], t["lens"], 16, 512,
                          None, "auto", 1.0, 0, 0, 1, 1, 0)
    small = torch.zeros(4096, dtype=torch.uint8, device=dev)
    for ws_ptr, ws_bytes in ((None, 0), (small.data_ptr(), small.numel()), (None, 1 << 30)):
        out = torch.full_like(plain, float("nan"))
        a = (out.data_ptr(),) + args[1:]
        assert lib.vmi_paged_attention_v1_f16_ws(*a, ws_ptr, ws_bytes, 0) == 0
        torch.cuda.synchronize()
        assert ops.last_launch_label() == plain_kernel
        assert torch.equal(out.view(torch.int16), plain.view(torch.int16))
    # ... and a split kernel asked for by id without one is refused, with a message that names the remedy
    vid = _split_names(64)[0][0]
    out = torch.full_like(plain, float("nan"))
    rc = lib.vmi_paged_attention_v1_f16_ws(*((out.data_ptr(),) + args[1:]), None, 0, vid)
    assert rc == 11 and b"workspace" in lib.vmi_last_error_string()
    with pytest.raises(RuntimeError, match="workspace"):
        prev = ops.set_workspace_enabled(False)
        try:
            _launch(case, t, vid)
        finally:
            ops.set_workspace_enabled(prev)


def test_with_a_workspace_the_default_entry_splits_an_underfilled_launch_and_agrees_with_the_model():
    from vllmini_amd import ops

    dev = _dev()
    for S, L in ((1, 1024), (8, 1024), (32, 512)):
        rng = np.random.default_rng(S + L)
        case = make_case(rng, S, 12, 64, [L] * S)
        t = _upload(case, dev)
        got = _launch(case, t).cpu().numpy()
        label = ops.last_launch_label()
        assert_close(got, run_model(case), f"default entry, batch {S} x {L} ({label})")
        assert ops.pick_variant(S, 12, 64, L, workspace=True) == ops.last_variant()
    assert _control_words_are_zero()
    # grouped-query heads (8 query heads on 2 KV heads, head size 128): the split kernels replace the gq kernels from 1024
    # tokens on — a query head's waves read their KV head's tiles themselves
    rng = np.random.default_rng(99)
    case = make_case(rng, 2, 8, 128, [2048, 1500], num_kv_heads=2)
    t = _upload(case, dev)
    got = _launch(case, t).cpu().numpy()
    assert "_x" in ops.last_launch_label(), ops.last_launch_label()
    assert ops.pick_variant(2, 8, 128, 2048, workspace=True, num_kv_heads=2) == ops.last_variant()
    assert_close(got, run_model(case), f"default entry, grouped-query ({ops.last_launch_label()})")


def test_a_workspace_is_reused_across_many_launches_and_shapes_and_results_do_not_depend_on_arrival_order():
    from vllmini_amd import ops

    dev = _dev()
    names = dict((n, i) for i, n in _split_names(64))
    rng = np.random.default_rng(11)
    cases = []
    for S, lens in ((8, [1024] * 8), (5, [100, 1000, 17, 512, 64]), (32, [512] * 32), (1, [2048])):
        c = make_case(rng, S, 12, 64, lens)
        cases.append((c, _upload(c, dev), run_model(c)))
    first = {}
    for rep in range(30):
        for k, (c, t, ref) in enumerate(cases):
            for name in ("d64_x8_u2_nt0", "d64_x16_u2_nt0"):
                got = _launch(c, t, names[name])
                if rep == 0:
                    assert_close(got.cpu().numpy(), ref, f"{name} case {k}")
                    first[(k, name)] = got.clone()
                else:
                    assert torch.equal(got.view(torch.int16), first[(k, name)].view(torch.int16)), (rep, k, name)
    assert ops.workspace_status(0) == 0 and _control_words_are_zero()


def test_a_poisoned_workspace_is_healed_by_the_reset_entry():
    """A launch killed in flight can leave granules and counters behind; they make the next launch wrong (never hang:
    every spin is bounded).  reset_workspaces() = vmi_paged_attention_v1_workspace_reset zeroes them."""
    from vllmini_amd import ops

    dev = _dev()
    rng = np.random.default_rng(5)
    case = make_case(rng, 4, 12, 64, [700, 512, 64, 1000])
    t = _upload(case, dev)
    ref = run_model(case)
    vid = dict((n, i) for i, n in _split_names(64))["d64_x16_u2_nt0"]
    assert_close(_launch(case, t, vid).cpu().numpy(), ref, "clean")
    ws = ops.workspace_for(0, create=False)
    ws[256: 256 + 64].fill_(3)                    # arrival counters of the first items: "killed mid-flight"
    ws[256 + 8192 * 4: 256 + 8192 * 4 + 512] = 0x3f  # ... and some of their granules
    torch.cuda.synchronize()
    ops.reset_workspaces(0)
    assert _control_words_are_zero()
    assert_close(_launch(case, t, vid).cpu().numpy(), ref, "after reset")
    assert ops.workspace_status(0) == 0


def test_a_launch_in_rounds_replays_from_a_graph():
    """The rounds twin keeps its per-place round counts in the workspace and zeroes them at the end: a captured launch of more
    items than are resident replays with the same bits as the eager launch, every time."""
    from vllmini_amd import ops

    dev = _dev()
    rng = np.random.default_rng(4242)
    S, H, D = 40, 12, 64
    lens = rng.integers(1, 700, S).tolist()
    case = make_case(rng, S, H, D, lens)
    t = _upload(case, dev)
    vid = {n: i + 1 for i, n in enumerate(ops.variant_names())}["d64_x64_u2_nt0"]      # 480 items x 16 workgroups: ten rounds
    eager = _launch(case, t, vid)
    assert_close(eager.cpu().numpy(), run_model(case), "rounds, eager")
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    out = torch.full((S, H, D), float("nan"), dtype=torch.float16, device=dev)
    with torch.cuda.stream(side):
        _launch(case, t, vid, out=out)             # (this stream's workspace exists before the capture)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            _launch(case, t, vid, out=out)
            _launch(case, t, vid, out=out)
        for _ in range(3):
            out.fill_(float("nan"))
            g.replay()
            side.synchronize()
            assert torch.equal(out.view(torch.int16), eager.view(torch.int16))
    torch.cuda.current_stream(dev).wait_stream(side)
    assert ops.workspace_status(0, side.cuda_stream) == 0


def test_graph_replay_two_streams_and_two_host_threads():
    from vllmini_amd import ops

    dev = _dev()
    rng = np.random.default_rng(9)
    case = make_case(rng, 6, 12, 64, [1024, 999, 512, 100, 17, 640])
    t = _upload(case, dev)
    ref = run_model(case)
    vid = dict((n, i) for i, n in _split_names(64))["d64_x16_u2_nt0"]

    # hipGraph: the capture stream's workspace must exist before the capture (nothing is allocated under capture)
    s = torch.cuda.Stream()
    out = torch.full((6, 12, 64), float("nan"), dtype=torch.float16, device=dev)
    with torch.cuda.stream(s):
        _launch(case, t, vid, out)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(4):
                _launch(case, t, vid, out)
        for _ in range(5):
            out.fill_(float("nan"))
            g.replay()
            s.synchronize()
            assert_close(out.cpu().numpy(), ref, "graph replay")

    # a stream without a workspace under capture: yesterday's kernel is captured, not an error
    s2 = torch.cuda.Stream()
    out2 = torch.full_like(out, float("nan"))
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g2, stream=s2):
            _launch(case, t, 0, out2)
        g2.replay()
        s2.synchronize()
    assert_close(out2.cpu().numpy(), ref, "captured without a workspace")

    # two streams at once: a workspace each
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    oa, ob = torch.full_like(out, float("nan")), torch.full_like(out, float("nan"))
    torch.cuda.synchronize()
    for _ in range(20):
        with torch.cuda.stream(sa):
            _launch(case, t, vid, oa)
        with torch.cuda.stream(sb):
            _launch(case, t, vid, ob)
    torch.cuda.synchronize()
    assert_close(oa.cpu().numpy(), ref, "stream a")
    assert_close(ob.cpu().numpy(), ref, "stream b")
    assert ops.workspace_for(0, sa.cuda_stream, create=False).data_ptr() != ops.workspace_for(0, sb.cuda_stream, create=False).data_ptr()

    # two host threads, each on its own stream
    errs, outs = [], {}

    def worker(k):
        try:
            st = torch.cuda.Stream()
            o = torch.full_like(out, float("nan"))
            with torch.cuda.stream(st):
                for _ in range(20):
                    _launch(case, t, vid, o)
                st.synchronize()
            outs[k] = o.cpu().numpy()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for k in range(2):
        assert_close(outs[k], ref, f"thread {k}")
