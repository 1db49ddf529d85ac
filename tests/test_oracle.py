"""CPU tests of the oracle itself: it must be pinned before it is trusted (task statement ③).

Pins available for this path: the reference ships NO golden vectors and its CUDA
kernel cannot run here, so the pins are
  * tests/golden/ref_eager.npz — outputs of the reference's OWN eager attention
    (vllmini/model/gpt2.py:71-78), produced by importing the reference Python (gen_golden.py);
  * tests/golden/ref_selftest.json — the reference's own unittest (tests/kernels/paged_attention.py)
    run against the oracle through a stub extension module;
  * the layout round trip that unittest asserts (:63-82), re-stated here with plain numpy indexing;
  * IEEE fp16 known answers for the software converters the kernel model is built on.
"""
from __future__ import annotations

import json
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import BS, make_case


# ---- fp16 helpers: known answers ------------------------------------------------------------------
@pytest.mark.parametrize("f,h", [
    (0.0, 0x0000), (-0.0, 0x8000), (1.0, 0x3C00), (-2.0, 0xC000), (65504.0, 0x7BFF), (65519.9, 0x7BFF),
    (65520.0, 0x7C00), (1e9, 0x7C00), (2.0 ** -14, 0x0400), (2.0 ** -24, 0x0001), (2.0 ** -25, 0x0000),
    (2.0 ** -25 * 1.0000001, 0x0001), (0.1, 0x2E66), (1.0 + 2.0 ** -11, 0x3C00), (1.0 + 3 * 2.0 ** -11, 0x3C02),
    (float("inf"), 0x7C00), (-float("inf"), 0xFC00),
])
def test_f2h_known_answers(f, h):
    assert oracle.f2h(f) == h


def test_f2h_h2f_match_numpy_everywhere():
    rng = np.random.default_rng(0)
    for h in range(0, 65536, 7):
        g = float(np.uint16(h).view(np.float16).astype(np.float32))
        f = oracle.h2f(h)
        assert f == g or (f != f and g != g)
    xs = (rng.standard_normal(5000) * rng.choice([1e-8, 1e-6, 1e-4, 1e-2, 1, 30, 3000, 60000], 5000)).astype(np.float32)
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([oracle.f2h(float(x)) for x in xs], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_fp16_mul_add_are_correctly_rounded():
    """fp16 op == exact op rounded once (numpy computes in fp32 then rounds: innocuous for p >= 2q+2)."""
    lib = oracle.kernel_model._load()
    rng = np.random.default_rng(1)
    a = rng.integers(0, 0x7C00, 4000).astype(np.uint16) | (rng.integers(0, 2, 4000).astype(np.uint16) << 15)
    b = rng.integers(0, 0x7C00, 4000).astype(np.uint16) | (rng.integers(0, 2, 4000).astype(np.uint16) << 15)
    fa, fb = a.view(np.float16).astype(np.float64), b.view(np.float16).astype(np.float64)
    with np.errstate(over="ignore"):
        want_mul = (fa * fb).astype(np.float16).view(np.uint16)   # float64 product of fp16s is exact
        want_add = (fa + fb).astype(np.float16).view(np.uint16)   # float64 sum of fp16s is exact
    for i in range(len(a)):
        assert lib.vmi_oracle_hmul(int(a[i]), int(b[i])) == int(want_mul[i])
        assert lib.vmi_oracle_hadd(int(a[i]), int(b[i])) == int(want_add[i])


# ---- reshape_and_cache: the layout the reference test reads back ------------------------------------
def test_reshape_and_cache_layout_roundtrip_like_reference_test():
    rng = np.random.default_rng(2)
    T, H, D, NB = 37, 12, 64, 8
    kc = np.zeros((NB, H, D // 8, BS, 8), dtype=np.float16)
    vc = np.zeros((NB, H, D, BS), dtype=np.float16)
    qkv = rng.standard_normal((T, 3 * H * D)).astype(np.float16)
    key = qkv[:, H * D:2 * H * D].reshape(T, H, D)      # strided views, row stride 3*H*D (gpt2.py:35-41)
    val = qkv[:, 2 * H * D:].reshape(T, H, D)
    slots = rng.permutation(NB * BS)[:T].astype(np.int64)
    slots[5] = -1
    oracle.reshape_and_cache(key, val, kc, vc, slots)
    for t in range(T):
        if slots[t] < 0:
            continue
        b, o = divmod(int(slots[t]), BS)
        for h in range(H):
            # exactly the reads at tests/kernels/paged_attention.py:73-80
            assert np.array_equal(kc[b, h, :, o, :].reshape(D), key[t, h])
            assert np.array_equal(vc[b, h, :, o], val[t, h])
    written = np.zeros(NB * BS, dtype=bool)
    written[slots[slots >= 0]] = True
    untouched = ~written.reshape(NB, BS)
    assert not kc.transpose(0, 3, 1, 2, 4)[untouched].any()
    assert not vc.transpose(0, 3, 1, 2)[untouched].any()


# ---- golden: the reference's eager attention ------------------------------------------------------
def _golden_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_eager.npz"))
    for name in sorted({k.rsplit("/", 2)[0] for k in z.files}):
        n_seq = len({k.split("/")[1] for k in z.files if k.startswith(name + "/")})
        yield name, z, n_seq


def _paged_from_rows(rng, keys, values, H, D):
    """Lay [L,H,D] rows of several sequences into paged caches through the oracle's reshape_and_cache."""
    lens = np.array([k.shape[0] for k in keys], dtype=np.int32)
    nblk = (lens + BS - 1) // BS
    NB = int(nblk.sum()) + 2
    kc = np.full((NB, H, D // 8, BS, 8), np.nan, dtype=np.float16)
    vc = np.full((NB, H, D, BS), np.nan, dtype=np.float16)
    perm = rng.permutation(NB)
    tables = np.full((len(keys), int(nblk.max()) + 1), -1, dtype=np.int32)
    pos = 0
    for s, (k, v) in enumerate(zip(keys, values)):
        tables[s, : nblk[s]] = perm[pos:pos + nblk[s]]
        pos += nblk[s]
        slots = tables[s, np.arange(lens[s]) // BS].astype(np.int64) * BS + np.arange(lens[s]) % BS
        oracle.reshape_and_cache(np.ascontiguousarray(k), np.ascontiguousarray(v), kc, vc, slots)
    return kc, vc, tables, lens


def test_oracle_pinned_against_reference_eager_golden(golden_dir):
    rng = np.random.default_rng(3)
    worst_model, worst_eager = 0.0, 0.0
    for name, z, n_seq in _golden_cases(golden_dir):
        keys = [z[f"{name}/{s}/key"] for s in range(n_seq)]
        vals = [z[f"{name}/{s}/value"] for s in range(n_seq)]
        H, D = keys[0].shape[1:]
        q = np.stack([z[f"{name}/{s}/query"][0] for s in range(n_seq)])
        scale = float(z[f"{name}/0/scale"])
        kc, vc, tables, lens = _paged_from_rows(rng, keys, vals, H, D)
        eager64 = oracle.eager_paged_attention(q, kc, vc, H, scale, tables, lens)
        model = oracle.paged_attention_v1(q, kc, vc, H, scale, tables, lens, BS).astype(np.float64)
        for s in range(n_seq):
            ref32 = z[f"{name}/{s}/ref_eager_fp32"].astype(np.float64)
            ref16 = z[f"{name}/{s}/ref_eager_fp16"].astype(np.float64)
            # oracle.eager (fp64) == the reference's eager in fp32, to fp32 round-off
            worst_eager = max(worst_eager, np.abs(eager64[s] - ref32).max())
            # kernel model vs the reference's eager: the reference test's own bar is atol 1e-2 vs fp16 eager
            assert np.abs(model[s] - ref16).max() <= 1e-2, name
            worst_model = max(worst_model, np.abs(model[s] - ref32).max())
    assert worst_eager <= 2e-6, worst_eager
    assert worst_model <= 2.5e-3, worst_model   # fp16 rounding of p and p*v (1.95e-3 @ L=3)


def _long_cases(golden_dir):
    """tests/golden/ref_eager_long.npz: the reference's eager attention at the lengths the bench runs.  The input rows are
    regenerated from the seeded numpy stream the generator used (gen_golden.long_rows) and checked against the fixture's
    SHA-256 before anything is compared."""
    import sys

    sys.path.insert(0, golden_dir)
    import gen_golden as g

    z = np.load(os.path.join(golden_dir, "ref_eager_long.npz"))
    for name, lens, H, D, seed in g.LONG_SCENARIOS:
        meta = json.loads(str(z[f"{name}/meta"]))
        assert meta["lens"] == lens and meta["H"] == H and meta["D"] == D and meta["seed"] == seed
        rows = []
        for s, L in enumerate(lens):
            key, value, query = g.long_rows(seed, s, L, H, D)
            assert np.array_equal(g.rows_checksum(key, value, query), z[f"{name}/{s}/rows_sha256"]), (name, s)
            rows.append((key, value, query))
        yield name, z, rows, H, D, float(meta["scale"])


def test_oracle_pinned_against_reference_eager_at_bench_lengths(golden_dir):
    """Round 5: the kernel model against the REFERENCE's eager attention at 300 ... 2048 tokens (BASELINE.json configs[1..3]
    run 512 / 1024 / 2048).  Bound: 5e-4 — SURVEY.md A.3 measured 2.4e-4 at 512 tokens and 1.2e-4 at 1024 for this
    restatement against exact arithmetic; the short-context fixture's 2.5e-3 is the L = 3 figure."""
    rng = np.random.default_rng(8)
    worst = {}
    for name, z, rows, H, D, scale in _long_cases(golden_dir):
        kc, vc, tables, lens = _paged_from_rows(rng, [r[0] for r in rows], [r[1] for r in rows], H, D)
        q = np.stack([r[2][0] for r in rows])
        eager64 = oracle.eager_paged_attention(q, kc, vc, H, scale, tables, lens)
        model = oracle.paged_attention_v1(q, kc, vc, H, scale, tables, lens, BS, threads=8).astype(np.float64)
        for s in range(len(rows)):
            ref32 = z[f"{name}/{s}/ref_eager_fp32"].astype(np.float64)
            ref16 = z[f"{name}/{s}/ref_eager_fp16"].astype(np.float64)
            assert np.abs(eager64[s] - ref32).max() <= 2e-6, (name, s)
            assert np.abs(model[s] - ref16).max() <= 1e-2, (name, s)           # the reference test's own bar
            worst[name] = max(worst.get(name, 0.0), np.abs(model[s] - ref32).max())
    assert len(worst) == 3 and max(worst.values()) <= 5e-4, worst


def test_reference_own_unittest_passed_against_oracle(golden_dir):
    with open(os.path.join(golden_dir, "ref_selftest.json")) as f:
        r = json.load(f)
    assert r["all_passed"] and len(r["results"]) >= 5
    assert all(x["run"] >= 1 and x["failures"] == 0 and x["errors"] == 0 for x in r["results"])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_reference_own_unittest_live_against_oracle():
    """Re-run the reference's kernel unittest right now against the oracle-backed stub (CPU)."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os, json, tempfile; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests', 'golden'));\n"
        "import gen_golden as g\n"
        "rec = g.SeamRecorder(); g.install_stub(rec)\n"
        "out = os.path.join(tempfile.mkdtemp(), 'r.json')\n"
        "with g.CudaToCpu(): g.run_reference_selftest(out)\n"
        "print(json.load(open(out))['all_passed'])\n" % (repo, repo))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.strip().endswith("True")


# ---- kernel model: semantics the reference kernel defines -----------------------------------------
def test_kernel_model_close_to_exact_and_error_shrinks_with_length():
    rng = np.random.default_rng(4)
    errs = {}
    for L in (3, 32, 512):
        case = make_case(rng, 4, 12, 64, [L] * 4, kv="normal")
        m = oracle.paged_attention_v1(case["q"], case["kc"], case["vc"], 12, case["scale"], case["tables"], case["lens"], BS)
        e = oracle.eager_paged_attention(case["q"], case["kc"], case["vc"], 12, case["scale"], case["tables"], case["lens"])
        errs[L] = np.abs(m.astype(np.float64) - e).max()
    assert errs[3] <= 3e-3 and errs[32] <= 1.5e-3 and errs[512] <= 6e-4
    assert errs[512] < errs[3]


def test_kernel_model_masks_garbage_past_context_and_zero_length():
    rng = np.random.default_rng(5)
    lens = [0, 1, 15, 16, 17, 40]
    clean = make_case(rng, len(lens), 12, 64, lens, max_blocks=4)
    dirty = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in clean.items()}
    for s, L in enumerate(lens):            # NaN in every slot past the context (reference zeroes them: :302-303, :420-430)
        nb = (L + BS - 1) // BS
        if L % BS and nb:
            b = dirty["tables"][s, nb - 1]
            dirty["kc"][b, :, :, L % BS:, :] = np.nan
            dirty["vc"][b, :, :, L % BS:] = np.nan
    a = oracle.paged_attention_v1(clean["q"], clean["kc"], clean["vc"], 12, clean["scale"], clean["tables"], clean["lens"], BS)
    b = oracle.paged_attention_v1(dirty["q"], dirty["kc"], dirty["vc"], 12, dirty["scale"], dirty["tables"], dirty["lens"], BS)
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16))
    assert not a[0].any()                   # seq_len 0 -> exp_sum 0 -> zeros (:342, 1e-6 keeps it finite)


def test_kernel_model_gqa_alibi_and_strided_query():
    rng = np.random.default_rng(6)
    case = make_case(rng, 3, 8, 64, [20, 33, 64], num_kv_heads=2, q_row_pad=2)
    assert case["q"].strides[0] == 3 * 8 * 64 * 2
    alibi = (2.0 ** -np.arange(1, 9)).astype(np.float32)
    m = oracle.paged_attention_v1(case["q"], case["kc"], case["vc"], 2, case["scale"], case["tables"], case["lens"], BS,
                                  alibi_slopes=alibi)
    e = oracle.eager_paged_attention(case["q"], case["kc"], case["vc"], 2, case["scale"], case["tables"], case["lens"],
                                     alibi_slopes=alibi)
    assert np.abs(m.astype(np.float64) - e).max() <= 2e-3
    m1 = oracle.paged_attention_v1(np.ascontiguousarray(case["q"]), case["kc"], case["vc"], 2, case["scale"],
                                   case["tables"], case["lens"], BS, alibi_slopes=alibi, threads=3)
    assert np.array_equal(m.view(np.uint16), m1.view(np.uint16))   # stride- and thread-split-independent


def test_kernel_model_other_block_sizes_follow_reference_formulas():
    """block_size 8/32 and head sizes 80..256 go through the reference's constexpr formulas (:138-168, :360-369)."""
    rng = np.random.default_rng(7)
    for bs, D in ((8, 64), (32, 128), (16, 80), (16, 256)):
        S, H, L = 2, 4, 45
        nb = (L + bs - 1) // bs
        kc = rng.uniform(-1, 1, (S * nb, H, D // 8, bs, 8)).astype(np.float16)
        vc = rng.uniform(-1, 1, (S * nb, H, D, bs)).astype(np.float16)
        q = rng.standard_normal((S, H, D)).astype(np.float16)
        tables = rng.permutation(S * nb).reshape(S, nb).astype(np.int32)
        lens = np.array([L, L - 7], dtype=np.int32)
        m = oracle.paged_attention_v1(q, kc, vc, H, D ** -0.5, tables, lens, bs).astype(np.float64)
        # exact reference through explicit gathering in this block size
        for s in range(S):
            n = int(lens[s])
            kb = kc[tables[s]].transpose(0, 3, 1, 2, 4).reshape(nb * bs, H, D)[:n].astype(np.float64)
            vb = vc[tables[s]].transpose(0, 3, 1, 2).reshape(nb * bs, H, D)[:n].astype(np.float64)
            w = np.einsum("hd,lhd->hl", q[s].astype(np.float64), kb) * D ** -0.5
            p = np.exp(w - w.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            assert np.abs(m[s] - np.einsum("hl,lhd->hd", p, vb)).max() <= 2e-3, (bs, D)
    with pytest.raises(RuntimeError):
        oracle.paged_attention_v1(np.zeros((1, 1, 72), np.float16), np.zeros((1, 1, 9, 16, 8), np.float16),
                                  np.zeros((1, 1, 72, 16), np.float16), 1, 1.0, np.zeros((1, 1), np.int32),
                                  np.ones(1, np.int32), 16)


def test_seam_trace_fixture_is_self_consistent(golden_dir):
    """The seam trace (what the reference's scheduler fed the ops for config 1) has the shapes, strides
    and bookkeeping SURVEY.md §8c describes, and replaying its reshape_and_cache calls through the oracle
    reproduces the recorded final caches."""
    z = np.load(os.path.join(golden_dir, "seam_trace.npz"))
    meta = json.loads(str(z["meta"]))
    H, D, NB, layers = meta["num_heads"], meta["head_size"], meta["num_blocks"], meta["num_layers"]
    assert meta["num_calls"] == layers + meta["num_decode_steps"] * 2 * layers     # 12 + 27*24 = 660
    assert z["tokens"].shape == (1, meta["max_length"])
    kc = np.zeros((NB, H, D // 8, BS, 8), dtype=np.float16)
    vc = np.zeros((NB, H, D, BS), dtype=np.float16)
    n_pa = 0
    for i in range(meta["num_calls"]):
        op = str(z[f"call{i:04d}/op"])
        if op == "reshape_and_cache":
            ks = tuple(z[f"call{i:04d}/key_strides"])
            T = z[f"call{i:04d}/key"].shape[0]
            # views of the fused qkv output: row stride 3*hidden whenever there is more than one row
            # (a size-1 leading dim reports an arbitrary stride; torch gives hidden)
            assert ks[1:] == (D, 1) and (ks[0] == 3 * H * D if T > 1 else ks[0] in (H * D, 3 * H * D))
            oracle.reshape_and_cache(z[f"call{i:04d}/key"], z[f"call{i:04d}/value"], kc, vc, z[f"call{i:04d}/slot_mapping"])
        else:
            n_pa += 1
            assert tuple(z[f"call{i:04d}/query_strides"])[1:] == (D, 1)
            assert z[f"call{i:04d}/block_tables"].dtype == np.int32 and z[f"call{i:04d}/seq_lens"].dtype == np.int32
            nkv, bs, msl = (int(v) for v in z[f"call{i:04d}/scalars"])
            assert (nkv, bs, msl) == (H, 16, meta["max_blocks_per_seq"] * 16)      # capacity, scheduler.py:97
            if n_pa % 97 == 1:  # spot-check the recorded oracle outputs against a fresh evaluation
                m = oracle.paged_attention_v1(z[f"call{i:04d}/query"], kc, vc, nkv, float(z[f"call{i:04d}/scale"]),
                                              z[f"call{i:04d}/block_tables"], z[f"call{i:04d}/seq_lens"], bs)
                assert np.array_equal(m.view(np.uint16), z[f"call{i:04d}/oracle_out"].view(np.uint16))
    assert np.array_equal(kc.view(np.uint16), z["final_key_cache"].view(np.uint16))
    assert np.array_equal(vc.view(np.uint16), z["final_value_cache"].view(np.uint16))
    # every block went back to the free list (kv_cache.py:81-86)
    assert sorted(meta["final_free_blocks"]) == list(range(NB))


# ------------------------------------------------------------------------------------------------
# fp8 E4M3 KV cache (kv_cache_dtype "fp8"): the reference compiles this path to assert(false)
# (ENABLE_FP8 is never defined, setup.py:30-45), so its SOURCE is the only definition.  The oracle's
# converters are pinned against torch's float8_e4m3fn (the same OCP format) on the CPU.
# ------------------------------------------------------------------------------------------------
def test_fp8_e4m3_converters_match_torch_float8_e4m3fn():
    bits = np.arange(256, dtype=np.uint8)
    mine = oracle.fp8e4m3_to_f32(bits)
    ref = torch.from_numpy(bits.copy()).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    nan = np.isnan(ref)
    assert np.array_equal(nan, np.isnan(mine)) and nan.sum() == 2                # 0x7f, 0xff
    assert np.array_equal(mine[~nan].view(np.uint32), ref[~nan].view(np.uint32))  # incl. -0.0
    assert np.nanmax(mine) == 448.0
    # encode: every float16 value (what reshape_and_cache converts), RNE; torch does not saturate (overflow -> NaN),
    # the reference does (__NV_SATFINITE): compare where |x| rounds into range, check saturation separately
    halves = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    enc = oracle.f32_to_fp8e4m3(halves)
    tref = torch.from_numpy(halves.copy()).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    in_range = np.abs(halves) < 464.0                                             # midpoint between 448 and the next step
    assert np.array_equal(enc[in_range], tref[in_range])
    big = np.isfinite(halves) & (np.abs(halves) >= 464.0)
    assert np.array_equal(enc[big], np.where(halves[big] > 0, 0x7e, 0xfe).astype(np.uint8))
    inf = np.isinf(halves)
    assert np.array_equal(enc[inf], np.where(halves[inf] > 0, 0x7e, 0xfe).astype(np.uint8))
    assert (enc[np.isnan(halves)] & 0x7f == 0x7f).all()
    # decode(encode(x)) is the nearest representable value: idempotent on representable inputs
    rep = mine[~nan]
    assert np.array_equal(oracle.fp8e4m3_to_f32(oracle.f32_to_fp8e4m3(rep)).view(np.uint32), rep.view(np.uint32))


def test_fp8_kernel_model_is_the_f16_model_on_dequantised_caches():
    """With an fp8 cache the kernel only changes how a cache element is fetched
    (attention_kernels.cu:283-289, 410-418): model(fp8 cache, kv_scale) == model(fp16 cache holding
    half(float(fp8) * kv_scale)) bit for bit; reshape_and_cache_fp8 == quantise-then-scatter."""
    rng = np.random.default_rng(8)
    S, H, Hkv, D, bs, NB = 5, 4, 2, 64, 16, 12
    lens = np.array([1, 16, 17, 40, 33], dtype=np.int32)
    tables = rng.permutation(NB)[: S * 2].reshape(S, 2).astype(np.int32) if False else \
        np.stack([rng.permutation(NB)[:3] for _ in range(S)]).astype(np.int32)
    q = rng.standard_normal((S, H, D)).astype(np.float16)
    for kv_scale in (1.0, 0.37, 2.0):
        kq = rng.integers(0, 256, (NB, Hkv, D // 16, bs, 16), dtype=np.uint8)
        vq = rng.integers(0, 256, (NB, Hkv, D, bs), dtype=np.uint8)
        kq[(kq & 0x7f) == 0x7f] = 0x3c          # no NaN codes
        vq[(vq & 0x7f) == 0x7f] = 0x3c
        got = oracle.paged_attention_v1_fp8(q, kq, vq, Hkv, D ** -0.5, tables, lens, bs, kv_scale=kv_scale)
        k16 = (oracle.fp8e4m3_to_f32(kq) * np.float32(kv_scale)).astype(np.float16)   # RNE, as float_to_half
        v16 = (oracle.fp8e4m3_to_f32(vq) * np.float32(kv_scale)).astype(np.float16)
        # same values, fp16 layout [NB,H,D/8,bs,8]: chunk c of 16 dims -> chunks 2c, 2c+1 of 8 dims
        k16 = k16.reshape(NB, Hkv, D // 16, bs, 2, 8).transpose(0, 1, 2, 4, 3, 5).reshape(NB, Hkv, D // 8, bs, 8)
        ref = oracle.paged_attention_v1(q, np.ascontiguousarray(k16), v16, Hkv, D ** -0.5, tables, lens, bs)
        assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), kv_scale
    # reshape
    T = 7
    key = (rng.standard_normal((T, Hkv, D)) * 3).astype(np.float16)
    val = (rng.standard_normal((T, Hkv, D)) * 300).astype(np.float16)      # some values saturate
    slots = rng.permutation(NB * bs)[:T].astype(np.int64)
    slots[2] = -1
    kc = np.zeros((NB, Hkv, D // 16, bs, 16), dtype=np.uint8)
    vc = np.zeros((NB, Hkv, D, bs), dtype=np.uint8)
    oracle.reshape_and_cache_fp8(key, val, kc, vc, slots, kv_scale=0.5)
    for t in range(T):
        if slots[t] < 0:
            continue
        b, o = divmod(int(slots[t]), bs)
        ek = oracle.f32_to_fp8e4m3(key[t].astype(np.float32) / np.float32(0.5))
        ev = oracle.f32_to_fp8e4m3(val[t].astype(np.float32) / np.float32(0.5))
        assert np.array_equal(kc[b, :, :, o, :].reshape(Hkv, D), ek)
        assert np.array_equal(vc[b, :, :, o], ev)
    assert (vc == 0x7e).any() or (vc == 0xfe).any()                          # saturation was exercised


# ------------------------------------------------------------------------------------------------
# block-sparse attention (blocksparse_vert_stride > 1): the restatement of attention_kernels.cu:209-254, 385-393
# against an independently written masked fp64 attention.  The reference holds no fixture for this mode (its callers
# never enable it, gpt2.py:109-112) — "parity unpinned" for these five arguments beyond this check.
# ------------------------------------------------------------------------------------------------
def masked_exact_attention(case, H, hkv, D, bs, sparse, tp_rank):
    loc, vert, bsz, step = sparse
    NB = case["kc"].shape[0]
    kk = case["kc"].astype(np.float64).transpose(0, 1, 3, 2, 4).reshape(NB, hkv, bs, D)
    vv = case["vc"].astype(np.float64).transpose(0, 1, 3, 2)
    out = np.zeros((len(case["lens"]), H, D))
    skipped = 0
    for s, L in enumerate(case["lens"]):
        if L == 0:
            continue
        nb = (L + bs - 1) // bs
        for h in range(H):
            kvh = h // (H // hkv)
            K = kk[case["tables"][s, :nb], kvh].reshape(-1, D)[:L]
            V = vv[case["tables"][s, :nb], kvh].reshape(-1, D)[:L]
            off = (tp_rank * H + h) * step + 1 if step >= 0 else (tp_rank * hkv + kvh) * (-step) + 1
            tok = np.arange(L)
            sparse_blk = (tok // bs) * bs // bsz                 # classified per cache block, by its first token
            att = ((sparse_blk + off) % vert == 0) | (sparse_blk > (L - 1) // bsz - loc)
            skipped += int((~att).sum())
            if att.any():
                lg = np.where(att, (K @ case["q"][s, h].astype(np.float64)) * case["scale"], -np.inf)
                pr = np.exp(lg - lg.max())
                out[s, h] = (pr / pr.sum()) @ V
    return out, skipped


@pytest.mark.parametrize("cfg", [(4, 4, 64, 16, (2, 4, 64, 1), 0), (8, 2, 128, 16, (1, 3, 32, -1), 1),
                                 (4, 2, 64, 8, (0, 2, 16, 2), 0), (3, 3, 80, 32, (4, 8, 64, 0), 0)],
                         ids=lambda c: f"H{c[0]}_{c[1]}_D{c[2]}_bs{c[3]}_sp{'_'.join(map(str, c[4]))}")
def test_blocksparse_kernel_model_matches_masked_attention(cfg):
    H, hkv, D, bs, sparse, tp = cfg
    rng = np.random.default_rng(H * D + bs)
    lens = np.array([1, 17, 200, 700, 64, 0, 333], np.int32)
    case = make_case(rng, len(lens), H, D, lens, num_kv_heads=hkv, block_size=bs)
    exact, skipped = masked_exact_attention(case, H, hkv, D, bs, sparse, tp)
    assert skipped > 500                                          # the pattern really drops tokens
    a = (case["q"], case["kc"], case["vc"], hkv, case["scale"], case["tables"], case["lens"], bs)
    v1 = oracle.paged_attention_v1(*a, blocksparse=sparse, tp_rank=tp, threads=4).astype(np.float64)
    v2 = oracle.paged_attention_v2(*a, 1024, blocksparse=sparse, tp_rank=tp)[0].astype(np.float64)
    dense = oracle.paged_attention_v1(*a, threads=4).astype(np.float64)
    assert np.abs(v1 - exact).max() < 1e-3 and np.abs(v2 - exact).max() < 1e-3
    assert np.abs(v1 - dense).max() > 1e-2                        # ... and that changes the answer
    # vert_stride <= 1 is the dense operator
    same = oracle.paged_attention_v1(*a, blocksparse=(sparse[0], 1, sparse[2], sparse[3]), tp_rank=tp, threads=4)
    assert np.array_equal(same, dense.astype(np.float16))


# ------------------------------------------------------------------------------------------------
# fp8 E5M2 KV cache (kv_cache_dtype "fp8_e5m2", __NV_E5M2: quant_utils.cuh:552-558)
# ------------------------------------------------------------------------------------------------
def test_fp8_e5m2_converters_match_torch_float8_e5m2():
    bits = np.arange(256, dtype=np.uint8)
    mine = oracle.fp8e5m2_to_f32(bits)
    ref = torch.from_numpy(bits.copy()).view(torch.float8_e5m2).to(torch.float32).numpy()
    nan = np.isnan(ref)
    assert np.array_equal(nan, np.isnan(mine)) and nan.sum() == 6                # 0x7d-0x7f, 0xfd-0xff
    assert np.array_equal(mine[~nan].view(np.uint32), ref[~nan].view(np.uint32))  # incl. -0.0 and +-inf
    # encode every float16 value; torch rounds to nearest even without saturating (overflow -> inf):
    # compare below the midpoint past the largest finite value, check saturation separately
    halves = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    enc = oracle.f32_to_fp8e5m2(halves)
    tref = torch.from_numpy(halves.copy()).to(torch.float8_e5m2).view(torch.uint8).numpy()
    in_range = np.abs(halves) < 61440.0
    assert np.array_equal(enc[in_range], tref[in_range])
    big = ~np.isnan(halves) & (np.abs(halves) >= 61440.0)                         # incl. +-inf: __NV_SATFINITE
    assert np.array_equal(enc[big], np.where(halves[big] > 0, 0x7b, 0xfb).astype(np.uint8))
    assert (enc[np.isnan(halves)] & 0x7f > 0x7c).all()                            # a NaN code
    fin = np.isfinite(mine)
    assert np.array_equal(oracle.fp8e5m2_to_f32(oracle.f32_to_fp8e5m2(mine[fin])).view(np.uint32), mine[fin].view(np.uint32))


def test_fp8_e5m2_kernel_model_is_the_f16_model_on_dequantised_caches():
    rng = np.random.default_rng(18)
    S, H, Hkv, D, bs, NB = 5, 4, 2, 64, 16, 12
    lens = np.array([1, 16, 17, 40, 33], dtype=np.int32)
    tables = np.stack([rng.permutation(NB)[:3] for _ in range(S)]).astype(np.int32)
    q = rng.standard_normal((S, H, D)).astype(np.float16)
    for kv_scale in (1.0, 0.37, 2.0):
        kq = rng.integers(0, 256, (NB, Hkv, D // 16, bs, 16), dtype=np.uint8)
        vq = rng.integers(0, 256, (NB, Hkv, D, bs), dtype=np.uint8)
        # |x| < 2 (exponent field <= 15): no inf / NaN codes, finite logits and outputs
        kq = np.where((kq & 0x7c) > 0x3c, (kq & 0x83) | 0x38, kq).astype(np.uint8)
        vq = np.where((vq & 0x7c) > 0x3c, (vq & 0x83) | 0x38, vq).astype(np.uint8)
        got = oracle.paged_attention_v1_fp8(q, kq, vq, Hkv, D ** -0.5, tables, lens, bs, kv_scale=kv_scale, e5m2=True)
        k16 = (oracle.fp8e5m2_to_f32(kq) * np.float32(kv_scale)).astype(np.float16)
        v16 = (oracle.fp8e5m2_to_f32(vq) * np.float32(kv_scale)).astype(np.float16)
        k16 = k16.reshape(NB, Hkv, D // 16, bs, 2, 8).transpose(0, 1, 2, 4, 3, 5).reshape(NB, Hkv, D // 8, bs, 8)
        ref = oracle.paged_attention_v1(q, np.ascontiguousarray(k16), v16, Hkv, D ** -0.5, tables, lens, bs)
        assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), kv_scale
        assert np.isfinite(got.astype(np.float32)).all()
    T = 7
    key = (rng.standard_normal((T, Hkv, D)) * 3).astype(np.float16)
    val = (rng.standard_normal((T, Hkv, D)) * 300).astype(np.float16)
    slots = rng.permutation(NB * bs)[:T].astype(np.int64)
    slots[2] = -1
    kc = np.zeros((NB, Hkv, D // 16, bs, 16), np.uint8)
    vc = np.zeros((NB, Hkv, D, bs), np.uint8)
    oracle.reshape_and_cache_fp8(key, val, kc, vc, slots, kv_scale=0.5, e5m2=True)
    for t in range(T):
        if slots[t] < 0:
            continue
        b, o = divmod(int(slots[t]), bs)
        assert np.array_equal(kc[b, :, :, o, :].reshape(Hkv, D), oracle.f32_to_fp8e5m2(key[t].astype(np.float32) / np.float32(0.5)))
        assert np.array_equal(vc[b, :, :, o], oracle.f32_to_fp8e5m2(val[t].astype(np.float32) / np.float32(0.5)))


# ------------------------------------------------------------------------------------------------
# float32 tensors: the (float, float) branch of the dispatch (x = 4 cache layout, all-fp32 arithmetic)
# ------------------------------------------------------------------------------------------------
def f32_case(rng, S, H, hkv, D, bs, lens):
    lens = np.asarray(lens, np.int32)
    nb = max((int(lens.max()) + bs - 1) // bs, 1)
    NB = S * nb + 2
    return dict(q=rng.standard_normal((S, H, D)).astype(np.float32),
                kc=rng.standard_normal((NB, hkv, D // 4, bs, 4)).astype(np.float32),
                vc=rng.standard_normal((NB, hkv, D, bs)).astype(np.float32),
                tables=rng.permutation(NB)[: S * nb].reshape(S, nb).astype(np.int32), lens=lens, hkv=hkv, bs=bs,
                scale=float(D) ** -0.5)


def f32_exact(case, alibi=None):
    q, kc, vc, tables, lens, hkv, bs = (case[k] for k in ("q", "kc", "vc", "tables", "lens", "hkv", "bs"))
    S, H, D = q.shape
    NB = kc.shape[0]
    kk = kc.astype(np.float64).transpose(0, 1, 3, 2, 4).reshape(NB, hkv, bs, D)
    vv = vc.astype(np.float64).transpose(0, 1, 3, 2)
    out = np.zeros((S, H, D))
    for s, L in enumerate(lens):
        n = (L + bs - 1) // bs
        for h in range(H):
            if L == 0:
                continue
            kvh = h // (H // hkv)
            K = kk[tables[s, :n], kvh].reshape(-1, D)[:L]
            V = vv[tables[s, :n], kvh].reshape(-1, D)[:L]
            lg = (K @ q[s, h].astype(np.float64)) * case["scale"]
            if alibi is not None:
                lg = lg + alibi[h] * (np.arange(L) - L + 1)
            pr = np.exp(lg - lg.max())
            out[s, h] = (pr / pr.sum()) @ V
    return out


@pytest.mark.parametrize("D,bs,H,hkv", [(64, 16, 4, 2), (80, 8, 3, 3), (128, 32, 4, 1), (256, 16, 2, 2), (112, 16, 2, 1)])
def test_f32_kernel_model_matches_exact_attention(D, bs, H, hkv):
    rng = np.random.default_rng(D + bs)
    case = f32_case(rng, 6, H, hkv, D, bs, [1, bs, bs + 1, 100, 333, 0])
    al = rng.uniform(0, 0.2, H).astype(np.float32)
    got = oracle.paged_attention_v1_f32(case["q"], case["kc"], case["vc"], hkv, case["scale"], case["tables"], case["lens"],
                                        bs, alibi_slopes=al, threads=4)
    assert np.abs(got - f32_exact(case, al)).max() < 2e-5
    assert (got[5] == 0).all()                                       # seq_len 0
    # reshape: rows land where the attention reads them
    T = 5
    key = rng.standard_normal((T, hkv, D)).astype(np.float32)
    val = rng.standard_normal((T, hkv, D)).astype(np.float32)
    slots = np.array([3, 0, -1, 2 * bs + 1, bs - 1], np.int64)
    kc2, vc2 = np.zeros_like(case["kc"]), np.zeros_like(case["vc"])
    oracle.reshape_and_cache_f32(key, val, kc2, vc2, slots)
    for t, sl in enumerate(slots):
        if sl < 0:
            continue
        b, o = divmod(int(sl), bs)
        assert np.array_equal(kc2[b, :, :, o, :].reshape(hkv, D), key[t]) and np.array_equal(vc2[b, :, :, o], val[t])
