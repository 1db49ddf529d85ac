"""The decode harness's own library (include/vmi_gpt2_layer.h): CPU checks of its C-ABI, GPU checks of
the linear-layer kernels against a torch fp32 restatement of the module chain they replace (vllmini/model/gpt2.py:14-15,
117-128, 130-135) with the chain's own rounding points, and of the harness that runs on them."""
from __future__ import annotations

import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "vmi_gpt2_layer.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vmi_gpt2_[a-z0-9_]+)\s*\(", src)))


def test_layer_library_exports_its_header_and_nothing_else():
    from vllmini_amd import build, gpt2_layer

    path = build.build_layer()
    declared = _declared()
    assert "vmi_gpt2_linear_f16" in declared and set(declared) == set(gpt2_layer.SIGNATURES)
    r = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True)
    funcs = {ln.split()[-1] for ln in r.stdout.splitlines() if ln.strip() and ln.split()[-2] in "TtWw"}
    assert funcs == set(declared), funcs ^ set(declared)
    lib = gpt2_layer.load()
    assert lib.vmi_gpt2_layer_abi_version() == gpt2_layer.ABI_VERSION == 1
    assert lib.vmi_gpt2_layer_target_arch() == b"gfx950"


def test_layer_entry_refuses_bad_arguments_without_touching_a_device():
    from vllmini_amd import gpt2_layer

    lib = gpt2_layer.load()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf) // 16 * 16 + 16
    call = lambda x, w, y, M, N, K, epi, res=None: lib.vmi_gpt2_linear_f16(x, K, w, None, None, None, 0.0, res, N, y, N, M, N, K,  # noqa: E731
                                                                           epi, 0, 0, None)
    assert call(None, p, p, 1, 16, 32, 0) == 1                      # null x
    assert call(p, p, p, 1, 16, 32, 3) == 1                         # unknown epilogue
    assert call(p, p, p, 1, 16, 32, 2) == 1                         # residual epilogue without a residual
    assert b"residual" in lib.vmi_gpt2_layer_last_error()
    assert call(p, p, p, 1, 16, 48, 0) == 2                         # K % 32
    assert call(p, p, p, 1, 24, 32, 0) == 2                         # N % 16
    assert call(p, p, p, 1, 16, 4640, 0) == 2                       # row tile past a CU's LDS
    assert b"K % 32" in lib.vmi_gpt2_layer_last_error()
    assert not gpt2_layer.supports(4, 24, 32) and gpt2_layer.supports(4, 384, 128)
    assert gpt2_layer.kernel_name(4, 64, 2080, True) is None and gpt2_layer.kernel_name(4, 64, 2080, False) is not None


def test_layer_picks_for_the_gpt2_small_decode_step():
    """The four launches of a layer at batch 256: 32-row tiles where that leaves workgroups for most of the chip, 16-row
    tiles otherwise, the K range of a slab over two waves from 48 k-steps on."""
    from vllmini_amd import gpt2_layer as gl

    assert gl.kernel_name(256, 2304, 768, True, gl.EPI_BIAS) == "bm32_nw4_ks1_r2_ln_bias"
    assert gl.kernel_name(256, 768, 768, False, gl.EPI_BIAS_RESIDUAL) == "bm16_nw4_ks1_r2_residual"
    assert gl.kernel_name(256, 3072, 768, True, gl.EPI_BIAS_GELU) == "bm32_nw4_ks1_r2_ln_gelu"
    assert gl.kernel_name(256, 768, 3072, False, gl.EPI_BIAS_RESIDUAL) == "bm16_nw4_ks2_r4_residual"
    assert gl.kernel_name(3, 384, 128, True, gl.EPI_BIAS) == "bm16_nw4_ks1_r2_ln_bias"
    # few rows (one sequence per step) x a long K: 16-column workgroups with the K range over their four waves
    assert gl.kernel_name(1, 768, 3072, False, gl.EPI_BIAS_RESIDUAL) == "bm16_nw1_ks4_r2_residual"
    assert gl.kernel_name(64, 768, 3072, False, gl.EPI_BIAS_RESIDUAL) == "bm16_nw1_ks4_r2_residual"
    assert gl.kernel_name(64, 768, 768, False, gl.EPI_BIAS_RESIDUAL) == "bm16_nw4_ks1_r2_residual"


# ---- GPU ---------------------------------------------------------------------------------------------------------------

def _chain(x, w, b, ln=None, gelu=False, residual=None):
    """The torch module chain in fp32 with ITS rounding points: layer_norm -> half, linear (fp32 sums) + bias -> half,
    GELU (erf) -> half | residual add -> half."""
    xf = x.float()
    if ln is not None:
        xf = F.layer_norm(xf, (x.shape[1],), ln[0].float(), ln[1].float(), ln[2]).half().float()
    y = xf.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    y = y.float().half()
    if gelu:
        y = F.gelu(y.float()).half()
    if residual is not None:
        y = (residual.float() + y.float()).half()
    return y


def _close(got, ref, what):
    g, r = got.float().cpu().numpy().astype(np.float64), ref.float().cpu().numpy().astype(np.float64)
    assert np.isfinite(g).all(), what
    # fp32 sums in another order may cross a rounding boundary of the half output: one half ulp of the value (2^-10 relative),
    # twice over for the second rounding of the GELU / residual epilogues
    tol = 2e-3 * np.abs(r) + 2e-3 * max(1e-3, np.abs(r).max()) * 0.25 + 1e-4
    bad = np.abs(g - r) > tol
    assert not bad.any(), (what, int(bad.sum()), float(np.abs(g - r).max()))


SHAPES = [  # (N, K, ln, epi)    GPT-2 small's four layers, the tiny fixture's, and shapes that leave partial tiles
    (2304, 768, True, "bias"), (768, 768, False, "res"), (3072, 768, True, "gelu"), (768, 3072, False, "res"),
    (384, 128, True, "bias"), (128, 128, False, "res"), (512, 128, True, "gelu"), (128, 512, False, "res"),
    (80, 96, True, "gelu"), (128, 2048, False, "gelu"), (16, 32, False, "bias"), (208, 1568, False, "res"), (2320, 800, True, "bias"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 5, 16, 33, 100, 256, 300])
def test_linear_kernels_match_the_module_chain(M):
    from vllmini_amd import gpt2_layer as gl

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(M)
    for N, K, ln, epi in SHAPES:
        x = (torch.randn(M, K, generator=g) * 1.5 + 0.3).half().to(dev)
        w = (torch.randn(N, K, generator=g) * (0.6 / K ** 0.5)).half().to(dev)
        b = (torch.randn(N, generator=g) * 0.1).half().to(dev)
        lnp = ((1 + 0.2 * torch.randn(K, generator=g)).half().to(dev), (0.1 * torch.randn(K, generator=g)).half().to(dev), 1e-5) \
            if ln else None
        res = torch.randn(M, N, generator=g).half().to(dev) if epi == "res" else None
        got = gl.linear(x, w, b, ln=lnp, gelu=epi == "gelu", residual=res)
        ref = _chain(x, w, b, lnp, epi == "gelu", res)
        _close(got, ref, (M, N, K, ln, epi, gl.kernel_name(M, N, K, ln)))
        # the packed weight form holds the same values in another order: the same sums in the same order, the same bits
        assert torch.equal(gl.linear(x, gl.pack_weight(w), b, ln=lnp, gelu=epi == "gelu", residual=res), got), (M, N, K)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_linear_layout_is_not_transposed_anywhere():
    """x = one-hot rows and a weight whose entry names its (n, k): every output element must be the entry its row's hot k and
    its column n select — a swapped row/column map of the MFMA result, or a k-group permutation, cannot pass."""
    from vllmini_amd import gpt2_layer as gl

    dev = torch.device("cuda:0")
    for M, N, K in ((48, 64, 64), (256, 768, 3072), (40, 2304, 768)):
        hot = (torch.arange(M) * 37 + 5) % K
        x = torch.zeros(M, K, dtype=torch.float16)
        x[torch.arange(M), hot] = 1
        n, k = torch.meshgrid(torch.arange(N), torch.arange(K), indexing="ij")
        w = (((n * 7 + k * 13) % 1024).float() / 64 - 8).half()       # exactly representable, asymmetric in (n, k)
        got = gl.linear(x.to(dev), w.to(dev), None).cpu()
        assert torch.equal(got, w[:, hot].t().contiguous()), (M, N, K)
        assert torch.equal(gl.linear(x.to(dev), gl.pack_weight(w.to(dev)), None).cpu(), got), (M, N, K)


@pytest.mark.gpu
def test_linear_in_place_residual_strided_input_no_bias_and_graph_replay():
    from vllmini_amd import gpt2_layer as gl

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    M, E = 64, 768
    qkv = torch.randn(M, 3 * E, generator=g).half().to(dev)
    w = (torch.randn(E, E, generator=g) * 0.02).half().to(dev)
    b = (torch.randn(E, generator=g) * 0.1).half().to(dev)
    x0 = torch.randn(M, E, generator=g).half().to(dev)
    xs = qkv[:, E:2 * E]                                  # rows 3E apart
    ref = _chain(xs, w, b, residual=x0)
    x = x0.clone()
    assert gl.linear(xs, w, b, residual=x, out=x) is x    # out aliases the residual
    _close(x, ref, "in place")
    _close(gl.linear(xs, w, None), _chain(xs, w, None), "no bias")
    with pytest.raises(RuntimeError, match="alternative"):
        gl.linear(xs, w, b, gelu=True, residual=x0)
    with pytest.raises(RuntimeError, match="must not alias x"):
        gl.linear(x0, w, b, out=x0)
    with pytest.raises(RuntimeError, match="K % 32"):
        gl.linear(qkv[:, :48], w[:, :48].contiguous(), b)
    with pytest.raises(RuntimeError, match="no CPU path"):
        gl.linear(x0.cpu(), w.cpu(), b.cpu())
    # captured in a hipGraph: replays give the eager bits
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    y = torch.empty(M, E, dtype=torch.float16, device=dev)
    with torch.cuda.stream(s):
        gl.linear(xs, w, b, residual=x0, out=y)
    torch.cuda.current_stream(dev).wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        gl.linear(xs, w, b, residual=x0, out=y)
    y.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, x)


@pytest.mark.gpu
def test_gpt2_small_decode_on_native_layers_tracks_the_torch_modules():
    """GPT-2 small, random weights, 24 sequences: the step on the native layer kernels against the same step on torch modules
    (same pools, same tokens): logits agree to fp16 GEMM rounding, the argmax wherever the top-2 gap is not tiny; and the
    hipGraph replay of the native step gives the eager native bits."""
    from vllmini_amd.gpt2_decode import GPT2Dims, GPT2PagedDecoder, random_state_dict
    from vllmini_amd.kv_pool import PagedKVPool

    dev = torch.device("cuda:0")
    dims = GPT2Dims()
    sd = random_state_dict(dims, dev, seed=1)
    B = 24

    def make(native):
        pool = PagedKVPool(B * dims.n_layer * 4 + 8, dims.n_head, dims.head_size, 16, 5, dims.n_layer, device=dev, max_seqs=B)
        dec = GPT2PagedDecoder(dims, sd, pool, native_layers=native)
        for s in range(B):
            dec.prefill(s, [(7 * s + j) % dims.vocab_size for j in range(1 + s % 13)])
        return dec

    nat, ref, rep = make(True), make(False), make(True)
    assert nat.native_layers and not ref.native_layers
    rng = np.random.default_rng(0)
    ids = list(range(B))
    for step in range(6):
        toks = rng.integers(0, dims.vocab_size, B).tolist()
        a = nat.decode(ids, toks).float()
        b = ref.decode(ids, toks).float()
        c = rep.decode(ids, toks, use_graph=True).float()
        assert torch.equal(a, c)
        err = (a - b).abs().max().item()
        assert err <= 2e-2 + 1e-2 * b.abs().max().item(), err
        top2 = b.topk(2, dim=1).values
        decisive = (top2[:, 0] - top2[:, 1]) > 5e-2
        assert (a.argmax(1) == b.argmax(1))[decisive].all()


@pytest.mark.gpu
@pytest.mark.parametrize("M,H,D,BS", [(1, 12, 64, 16), (37, 12, 64, 16), (256, 12, 64, 16), (20, 2, 64, 16), (9, 4, 128, 8)])
def test_qkv_projection_with_the_cache_write_leaves_reshape_and_caches_bytes(M, H, D, BS):
    """vmi_gpt2_linear_qkv_cache_f16 = linear() + cache_ops.reshape_and_cache on its k / v views: the same qkv bits, the same
    cache bytes (rows with a negative slot skipped, everything else in the caches untouched)."""
    from vllmini_amd import cache_ops, gpt2_layer as gl

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(M + D)
    E = H * D
    K = E
    NB = max(4, (M + BS - 1) // BS * 3)
    x = torch.randn(M, K, generator=g).half().to(dev)
    w = (torch.randn(3 * E, K, generator=g) * (0.6 / K ** 0.5)).half().to(dev)
    b = (torch.randn(3 * E, generator=g) * 0.1).half().to(dev)
    lnp = ((1 + 0.2 * torch.randn(K, generator=g)).half().to(dev), (0.1 * torch.randn(K, generator=g)).half().to(dev), 1e-5)
    kc0 = torch.randn(NB, H, D // 8, BS, 8, generator=g).half().to(dev)
    vc0 = torch.randn(NB, H, D, BS, generator=g).half().to(dev)
    slots = torch.randperm(NB * BS, generator=g)[:M].to(torch.int64)
    if M > 4:
        slots[3] = -1
    slots = slots.to(dev)
    for weight in (w, gl.pack_weight(w)):
        ref_qkv = gl.linear(x, weight, b, ln=lnp)
        kr, vr = kc0.clone(), vc0.clone()
        k, v = (ref_qkv[:, j * E:(j + 1) * E].view(M, H, D) for j in (1, 2))
        cache_ops.reshape_and_cache(k, v, kr, vr, slots, "auto", 1.0)
        kg, vg = kc0.clone(), vc0.clone()
        got = gl.linear_qkv_cache(x, weight, b, kg, vg, slots, H, ln=lnp)
        torch.cuda.synchronize()
        assert torch.equal(got, ref_qkv)
        assert torch.equal(kg.view(torch.int16), kr.view(torch.int16)) and torch.equal(vg.view(torch.int16), vr.view(torch.int16))
        assert not torch.equal(kg, kc0)
    with pytest.raises(RuntimeError, match="float16 caches"):
        gl.linear_qkv_cache(x, w, b, kc0.to(torch.uint8), vc0.to(torch.uint8), slots, H, ln=lnp)


@pytest.mark.gpu
def test_decoder_with_the_scatter_in_c_attn_is_bit_identical_to_the_call_pair(golden_dir):
    """The harness with reshape_and_cache's copy folded into the q / k / v projection: same logits, same caches, bit for bit,
    as with the reference's call pair — eager and replayed from a hipGraph."""
    from vllmini_amd.gpt2_decode import GPT2Dims, GPT2PagedDecoder, random_state_dict
    from vllmini_amd.kv_pool import PagedKVPool

    dev = torch.device("cuda:0")
    dims = GPT2Dims()
    sd = random_state_dict(dims, dev, seed=2)
    B = 9

    def make(scatter):
        pool = PagedKVPool(B * dims.n_layer * 4 + 8, dims.n_head, dims.head_size, 16, 5, dims.n_layer, device=dev, max_seqs=B)
        dec = GPT2PagedDecoder(dims, sd, pool, scatter_in_c_attn=scatter)
        for s in range(B):
            dec.prefill(s, [(11 * s + j) % dims.vocab_size for j in range(1 + (5 * s) % 17)])
        return dec

    pair, scat, scat_g = make(False), make(True), make(True)
    rng = np.random.default_rng(1)
    ids = list(range(B))
    for step in range(8):
        toks = rng.integers(0, dims.vocab_size, B).tolist()
        a = pair.decode(ids, toks)
        b = scat.decode(ids, toks)
        c = scat_g.decode(ids, toks, use_graph=True)
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(pair.pool.key_cache, scat.pool.key_cache) and torch.equal(pair.pool.value_cache, scat.pool.value_cache)
    assert torch.equal(pair.pool.key_cache, scat_g.pool.key_cache)
    with pytest.raises(ValueError, match="scatter_in_c_attn"):
        GPT2PagedDecoder(dims, sd, pair.pool, scatter_in_c_attn=True, fused_append=True)


@pytest.mark.gpu
def test_linear_kernels_over_random_shapes():
    """Every pick — 32 / 16-row tiles, K over one, two or four waves, two- and four-chunk rings, partial chunks, partial row and
    column tiles — on 80 random (M, N, K, LayerNorm, epilogue) against the module chain, plain and packed weights."""
    from vllmini_amd import gpt2_layer as gl

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    g = torch.Generator(device="cpu").manual_seed(5)
    seen = set()
    for case in range(80):
        M = int(rng.choice([1, 2, 7, 16, 17, 31, 32, 33, 63, 64, 100, 129, 255, 256, 257, 400, 600]))
        N = 16 * int(rng.integers(1, 200))
        ln = bool(rng.integers(0, 2))
        K = 32 * int(rng.integers(1, (2048 if ln else 4608) // 32 + 1))
        epi = ["bias", "gelu", "res"][int(rng.integers(0, 3))]
        x = (torch.randn(M, K, generator=g) * 1.5 + 0.3).half().to(dev)
        w = (torch.randn(N, K, generator=g) * (0.6 / K ** 0.5)).half().to(dev)
        b = (torch.randn(N, generator=g) * 0.1).half().to(dev) if rng.integers(0, 4) else None
        lnp = ((1 + 0.2 * torch.randn(K, generator=g)).half().to(dev), (0.1 * torch.randn(K, generator=g)).half().to(dev), 1e-5) \
            if ln else None
        res = torch.randn(M, N, generator=g).half().to(dev) if epi == "res" else None
        name = gl.kernel_name(M, N, K, ln)
        seen.add(name.rsplit("_", 1)[0])
        got = gl.linear(x, w, b, ln=lnp, gelu=epi == "gelu", residual=res)
        _close(got, _chain(x, w, b, lnp, epi == "gelu", res), (case, M, N, K, ln, epi, name))
        assert torch.equal(gl.linear(x, gl.pack_weight(w), b, ln=lnp, gelu=epi == "gelu", residual=res), got), (case, M, N, K)
    torch.cuda.synchronize()
    assert len(seen) >= 6, seen      # the draw reached most of the kernel family


@pytest.mark.gpu
def test_embed_and_argmax_equal_torch():
    from vllmini_amd import gpt2_layer as gl

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(9)
    V, P, E = 50257, 1024, 768
    wte = torch.randn(V, E, generator=g).half().to(dev)
    wpe = torch.randn(P, E, generator=g).half().to(dev)
    for T in (1, 5, 256):
        ids = torch.randint(0, V, (T,), generator=g).to(dev)
        pos = torch.randint(0, P, (T,), generator=g).to(dev)
        assert torch.equal(gl.embed(ids, pos, wte, wpe), wte[ids] + wpe[pos])
    # argmax: rows of 50 257 halves start at every alignment; ties go to the first maximum; a row slice of a wider tensor
    for B in (1, 3, 9, 256):
        logits = torch.randn(B, V, generator=g).half().to(dev)
        assert torch.equal(gl.argmax(logits), logits.argmax(-1))
    tie = torch.zeros(4, V, dtype=torch.float16, device=dev)
    tie[0, 777] = tie[0, 40000] = 3.0
    tie[1, V - 1] = 1.0
    tie[2, 0] = tie[2, 1] = 2.0
    tie[3] = -5.0
    assert gl.argmax(tie).tolist() == [777, V - 1, 0, 0]
    wide = torch.randn(6, V + 37, generator=g).half().to(dev)
    assert torch.equal(gl.argmax(wide[:, 5:5 + 1000]), wide[:, 5:5 + 1000].argmax(-1))
    small = torch.randn(7, 13, generator=g).half().to(dev)
    assert torch.equal(gl.argmax(small), small.argmax(-1))
    with pytest.raises(RuntimeError, match="half logits"):
        gl.argmax(logits.float())


@pytest.mark.gpu
def test_graph_replay_across_a_pick_threshold_keeps_one_graph_per_launch_geometry():
    """32 sequences whose contexts grow from 505 to 520 tokens cross a threshold of the attention pick (the variant id is baked
    into a captured step): the harness keeps ONE graph per variant and switches between them — never a capture in the middle of
    a run once both exist — and the replayed logits are the eager ones."""
    from vllmini_amd.gpt2_decode import GPT2Dims, GPT2PagedDecoder, random_state_dict
    from vllmini_amd.kv_pool import PagedKVPool

    dev = torch.device("cuda:0")
    dims = GPT2Dims(n_layer=2)
    sd = random_state_dict(dims, dev, seed=4)
    B, ctx = 32, 505

    def make():
        pool = PagedKVPool(B * dims.n_layer * 34 + 8, dims.n_head, dims.head_size, 16, 35, dims.n_layer, device=dev, max_seqs=B,
                           multi_block_prefill=True)
        dec = GPT2PagedDecoder(dims, sd, pool)
        for s in range(B):
            dec.prefill(s, [(3 * s + j) % dims.vocab_size for j in range(ctx)])
        return dec

    eager, graph = make(), make()
    rng = np.random.default_rng(2)
    ids = list(range(B))
    variants = []
    for step in range(15):
        toks = rng.integers(0, dims.vocab_size, B).tolist()
        a = eager.decode(ids, toks)
        b = graph.decode(ids, toks, use_graph=True)
        variants.append(graph._static["variant"])
        assert torch.equal(a, b), step
    assert len(set(variants)) == len(graph._graph)
    captured = dict(graph._graph)
    for step in range(3):      # more steps on variants already seen: the same graph objects, nothing captured
        toks = rng.integers(0, dims.vocab_size, B).tolist()
        assert torch.equal(eager.decode(ids, toks), graph.decode(ids, toks, use_graph=True))
    assert all(graph._graph[v][0] is captured[v][0] for v in captured)


@pytest.mark.gpu
def test_top_k_sampling_kernel_draws_from_the_reference_distribution():
    """vmi_gpt2_sample_top_k_f16 against the torch chain the reference spells out (scheduler.py:144-153): the k survivors are
    torch.topk's, the draw at a given uniform number is the inverse CDF over their softmax in descending order, ties go to the
    smaller index, rows of equal logits (the block-wide path) work, and the frequencies of many draws match the probabilities."""
    from vllmini_amd import gpt2_layer as gl

    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    V = 50257
    for B, k, temp in ((1, 50, 1.0), (7, 50, 0.7), (256, 50, 1.0), (5, 1, 1.0), (4, 64, 1.3), (3, 13, 1.0)):
        logits = (torch.randn(B, V, generator=g) * 2).half().to(dev)
        vals, idx = torch.topk(logits.float() / temp, k, dim=-1)
        cdf = torch.softmax(vals, -1).double().cumsum(-1)
        for u in (0.0, 0.173, 0.5, 0.9371, 0.999999):
            uu = torch.full((B,), u, dtype=torch.float32, device=dev)
            got = gl.sample_top_k(logits, k, temp, uniform=uu)
            j = (cdf >= u * cdf[:, -1:]).int().argmax(-1)                       # first survivor whose cumulative mass reaches u
            want = idx.gather(-1, j[:, None]).squeeze(-1)
            # (a draw that sits within float rounding of a CDF step may land on the neighbour: compare there by mass)
            bad = got != want
            if bad.any():
                pos = (idx == got[:, None]).int().argmax(-1)
                assert ((idx == got[:, None]).any(-1))[bad].all()
                near = (cdf.gather(-1, pos[:, None]).squeeze(-1) - u * cdf[:, -1]).abs() < 1e-5
                near |= (cdf.gather(-1, (pos - 1).clamp(min=0)[:, None]).squeeze(-1) - u * cdf[:, -1]).abs() < 1e-5
                assert near[bad].all(), (B, k, u)
    # ties: equal logits everywhere -> the k smallest indices, equal mass each; a few equal maxima -> smaller index first
    flat = torch.zeros(3, V, dtype=torch.float16, device=dev)
    for u, want in ((0.0, 0), (0.51, 25), (0.999, 49)):
        assert gl.sample_top_k(flat, 50, 1.0, uniform=torch.full((3,), u, device=dev)).tolist() == [want] * 3
    peaks = torch.full((2, V), -4.0, dtype=torch.float16, device=dev)
    peaks[:, 40000] = peaks[:, 123] = peaks[:, 9999] = 6.0
    assert gl.sample_top_k(peaks, 3, 1.0, uniform=torch.tensor([0.1, 0.5], device=dev)).tolist() == [123, 9999]
    small = torch.randn(4, 37, generator=g).half().to(dev)           # fewer columns than threads, top_k larger than the row
    got = gl.sample_top_k(small, 50, 1.0, uniform=torch.zeros(4, device=dev))
    assert torch.equal(got, small.argmax(-1))
    # frequencies: 20 000 draws of one row against its top-8 probabilities
    row = (torch.randn(1, V, generator=g) * 3).half().to(dev)
    vals, idx = torch.topk(row.float(), 8, dim=-1)
    p = torch.softmax(vals, -1)[0].cpu().numpy()
    gen = torch.Generator(device=dev).manual_seed(3)
    draws = gl.sample_top_k(row.expand(20000, V).contiguous(), 8, 1.0, generator=gen).cpu().numpy()
    freq = np.array([(draws == int(i)).mean() for i in idx[0].cpu().numpy()])
    assert abs(freq.sum() - 1) < 1e-9 and np.abs(freq - p).max() < 0.012, (freq, p)
    # the scheduler's sampler takes this path for the decoder's half logits
    from vllmini_amd.scheduler import sample_top_k
    out = sample_top_k(row.expand(64, V).contiguous(), generator=torch.Generator(device=dev).manual_seed(1))
    assert out.dtype == torch.int64 and set(out.tolist()) <= set(torch.topk(row.float(), 50).indices[0].tolist())

