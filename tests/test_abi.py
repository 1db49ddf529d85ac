"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/*.h
declares (no compute calls without a GPU), the Python mirror carries the reference's operator
surface, and the product path fails loudly instead of falling back."""
from __future__ import annotations

import ctypes
import glob
import inspect
import os
import re
import subprocess

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="vmi_paged_attention.h"):
    src = open(os.path.join(REPO, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vmi_[a-z0-9_]+)\s*\(", src)))


def _exported(path):
    """Dynamic symbol table of a shared library (defined symbols only)."""
    r = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True)
    return {ln.split()[-1] for ln in r.stdout.splitlines() if ln.strip()}


def test_library_builds_loads_and_exports_every_declared_symbol():
    from vllmini_amd import _lib, build

    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    assert sorted(os.path.basename(h) for h in glob.glob(os.path.join(REPO, "include", "*.h"))) == \
        ["vmi_gpt2_layer.h", "vmi_paged_attention.h", "vmi_paged_attention_diag.h", "vmi_paged_attention_extras.h"]
    declared = _declared_symbols()
    assert {"vmi_paged_attention_v1_f16", "vmi_reshape_and_cache_f16", "vmi_last_error_string",
            "vmi_abi_version"} <= set(declared)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in vllmini_amd/_lib.py"
    assert set(_lib.SIGNATURES) <= set(declared)
    typed = _lib.load()
    assert typed.vmi_abi_version() == _lib.ABI_VERSION == 22
    assert typed.vmi_target_arch() == b"gfx950"
    assert typed.vmi_is_diag_build() == 0 and typed.vmi_has_extras() == 0


def test_libraries_export_the_c_abi_and_nothing_else_that_can_be_called():
    """Built with -fvisibility=hidden: the dynamic symbol table of each library holds the extern "C" entries of include/*.h
    (visibility pushed to default there) and no other FUNCTION — none of the ~900 vmi:: launchers / pick functions that used
    to leak (round-4 advisor finding: the extras library's wrappers reached them through the PLT, so loading two libraries
    RTLD_GLOBAL could bind one's entries to the other's menus).  What remains besides `vmi_*` are the weak data handles the HIP
    runtime registers kernels by (type V)."""
    from vllmini_amd import build

    build.build()
    for path in (build.LIB_PATH, build.EXTRAS_LIB_PATH, build.DIAG_LIB_PATH, build.LAYER_LIB_PATH):
        if not os.path.exists(path):
            continue
        r = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True)
        rows = [ln.split() for ln in r.stdout.splitlines() if ln.strip()]
        funcs = [x[-1] for x in rows if x[-2] in "TtWw"]
        assert funcs and all(f.startswith("vmi_") for f in funcs), [f for f in funcs if not f.startswith("vmi_")][:5]
        assert not [x[-1] for x in rows if "launch_pa" in x[-1] or "pick_variant" in x[-1] and not x[-1].startswith("vmi_")]


def test_head_128_long_context_picks_fit_their_lds():
    """Round-4 advisor finding: `lock_ok` had no LDS term, so head size 128 x 16 / 32 heads at 3400 ... 16384 tokens was handed
    the 16-heads-per-workgroup lockstep kernel, whose logits do not fit there (the launcher then fell back to one wave per
    head).  Every pick over that grid must fit its LDS."""
    from vllmini_amd import ops

    names = ops.variant_names()
    for H in (16, 32):
        for B in (128, 256, 1024, 2048):
            for L in (2048, 2560, 3400, 4096, 8192, 16384):
                v = ops.pick_variant(B, H, 128, L)
                assert v and ops.variant_fits(v, L), (H, B, L, names[v - 1])
                if L >= 3400:
                    assert "_lock" not in names[v - 1], (H, B, L, names[v - 1])


def test_a_worker_thread_starts_on_the_product_library_and_can_opt_in_for_itself():
    """`_lib.use_extras()` / `use_diag()` switch the library of the calling CONTEXT (a ContextVar since round 4): a thread
    started inside the `with` runs on the product library unless it opts in itself or is started through
    contextvars.copy_context().run."""
    import contextvars
    import threading

    from vllmini_amd import _lib, build

    if not os.path.exists(build.EXTRAS_LIB_PATH):
        pytest.skip("extras library not built")
    seen = {}

    def worker(tag):
        seen[tag] = _lib.load().vmi_has_extras()

    def opting_in():
        with _lib.use_extras():
            seen["opted"] = _lib.load().vmi_has_extras()

    with _lib.use_extras():
        assert _lib.load().vmi_has_extras() == 1
        t = [threading.Thread(target=worker, args=("plain",)), threading.Thread(target=opting_in),
             threading.Thread(target=contextvars.copy_context().run, args=(worker, "copied"))]
        [x.start() for x in t]
        [x.join() for x in t]
    assert seen == {"plain": 0, "opted": 1, "copied": 1}
    assert _lib.load().vmi_has_extras() == 0


OUT_OF_SCOPE = re.compile(r"bf16|e5m2|_sp_|sparse|f32_kernel|convert_fp8_kernel|flash_kernel", re.I)


def test_product_library_holds_the_hot_path_and_nothing_else():
    """libvmi_paged_attention.so = SURVEY.md §8: float16 tensors over float16 / fp8-E4M3 pages.  The out-of-scope corners of
    the reference's dispatch (bfloat16 / float32 tensors, E5M2 pages, block-sparse attention, reshape_and_cache_flash,
    convert_fp8: SURVEY.md §2 rows 8-10) are NOT in it — no kernel in its symbol table, no row in its menus, and their
    C-ABI entries (include/vmi_paged_attention_extras.h) are not exported by it; the Python operators raise RuntimeError with a
    message that says where they live.  libvmi_paged_attention_extras.so holds the same objects plus those."""
    from vllmini_amd import _lib, build, ops

    path = build.build()
    syms = _exported(path)
    assert not [s for s in syms if re.search(r"f32_kernel|convert_fp8_kernel|flash_kernel", s)]
    targs = lambda sym: [int(a[2:-1]) for a in re.findall(r"L[ib]\d+E", sym.split("_kernelI", 1)[1])]      # noqa: E731
    v1 = [targs(s) for s in syms if "12pa_v1_kernelI" in s]          # <D,HPW,WPH,U,NT,LO,PART,BS,LOCK,BF,HPT,APP,UMAX,F8,GQS,FPV,SPARSE>
    q = [targs(s) for s in syms if "11pa_q_kernelI" in s]            # <D,BF,NT,US,UQ,F8,KM,UT>
    sc = [targs(s) for s in syms if "reshape_and_cache_fp8_kernelI" in s]   # <VEC,BF,E5>
    assert len(v1) > 200 and len(q) >= 7 and len(sc) >= 2      # (kernel + its host stub: two symbols each)
    assert all(a[9] == 0 and a[13] in (0, 1) and a[16] == 0 for a in v1)     # float16 query, fp16 / E4M3 pages, not block-sparse
    assert all(a[1] == 0 and a[5] in (0, 1) for a in q)
    assert all(a[1] == 0 and a[2] == 0 for a in sc)
    names = ops.variant_names() + ops.variant_names_v2()
    assert names and not [n for n in names if OUT_OF_SCOPE.search(n) or n.startswith("bf16_")]
    assert any(n.startswith("q_d64") for n in names) and any(n.startswith("fp8_q_d64") for n in names)
    lib = _lib.load()
    assert lib.vmi_has_extras() == 0
    # ABI 20: the out-of-scope entries are declared in vmi_paged_attention_extras.h and are NOT exported by the product library
    # (no entry that could only answer VMI_E_NOT_BUILT); the Python operators raise in front of them
    extras_only = set(_declared_symbols("vmi_paged_attention_extras.h")) - set(_declared_symbols())
    assert extras_only == set(_lib.EXTRAS_SIGNATURES) and len(extras_only) == 17
    assert not (extras_only & syms), extras_only & syms
    assert {s_ for s_ in syms if s_.startswith("vmi_")} == set(_declared_symbols())
    with pytest.raises(RuntimeError, match="convert_fp8: not in this build.*extras"):
        _lib.require_extras("convert_fp8")
    # every bfloat16 / E5M2 pick says "no kernel"
    assert lib.vmi_paged_attention_v1_pick_variant_gqa(256, 12, 12, 64, 16, 1024, 1, 0) == 0
    assert ops.pick_variant(256, 12, 64, 1024, fp8="e5m2") == 0 and ops.pick_variant(256, 12, 64, 1024, fp8=True, bf16=True) == 0
    assert ops.pick_variant(256, 12, 64, 1024, fp8=True) > 0
    assert lib.vmi_paged_attention_v1_pick_variant_gqa(256, 12, 12, 64, 16, 1024, 0, 0) > 0


def test_extras_library_is_the_product_plus_the_out_of_scope_operators():
    from vllmini_amd import _lib, build, ops

    path = build.build(extras=True)
    assert path.endswith("libvmi_paged_attention_extras.so") and os.path.exists(build.LIB_PATH)
    syms = _exported(path)
    assert {s for s in syms if s.startswith("vmi_")} == set(_declared_symbols()) | set(_declared_symbols("vmi_paged_attention_extras.h"))
    product_names, product_v2 = ops.variant_names(), ops.variant_names_v2()
    with _lib.use_extras() as lib:
        assert lib.vmi_has_extras() == 1 and lib.vmi_is_diag_build() == 0 and _lib.load() is lib
        names, names_v2 = ops.variant_names(), ops.variant_names_v2()
        assert lib.vmi_paged_attention_v1_pick_variant_gqa(256, 12, 12, 64, 16, 1024, 1, 0) > 0
    assert _lib.load().vmi_has_extras() == 0               # the switch ends with the context
    extra = [n for n in names + names_v2 if n not in product_names + product_v2]
    assert extra and all(OUT_OF_SCOPE.search(n) or n.startswith("bf16_") for n in extra), [n for n in extra if not OUT_OF_SCOPE.search(n)][:5]
    assert [n for n in names if n in product_names] == product_names          # same kernels, same order
    assert [n for n in names_v2 if n in product_v2] == product_v2


def test_product_library_exports_no_diagnostic_symbol_and_no_diagnostic_kernel():
    """The product .so carries nothing that is wrong by design or untested: no vmi_diag_* / vmi_debug_* entry, no
    "loads only" variant, no LDS-staging experiment kernel — in its C-ABI, in its variant names and in its dynamic
    symbol table (kernel stubs included).  The diagnostic build (-DVMI_DIAG) is where those live."""
    from vllmini_amd import _lib, build, ops

    path = build.build()
    syms = _exported(path)
    vmi = {s for s in syms if s.startswith("vmi_")}
    assert vmi == set(_declared_symbols()), vmi ^ set(_declared_symbols())
    diag_only = set(_declared_symbols("vmi_paged_attention_diag.h")) - set(_declared_symbols())
    assert diag_only == set(_lib.DIAG_SIGNATURES) and diag_only
    assert not (diag_only & syms)
    bad = [s for s in syms - {"vmi_is_diag_build"} if re.search(r"diag|debug|stage|gather_read|stream_read|LOADSONLY", s, re.I)]
    assert not bad, bad[:5]
    names = ops.variant_names() + ops.variant_names_v2()
    assert not [n for n in names if "LOADSONLY" in n or n.startswith("stage_")]


def test_diagnostic_library_is_the_extras_library_plus_the_diagnostics():
    from vllmini_amd import _lib, build, ops

    path = build.build(diag=True)
    assert path.endswith("libvmi_paged_attention_diag.so") and os.path.exists(build.LIB_PATH)
    syms = _exported(path)
    for name in [*_declared_symbols(), *_declared_symbols("vmi_paged_attention_extras.h"), *_declared_symbols("vmi_paged_attention_diag.h")]:
        assert name in syms, name
    with _lib.use_extras():
        full_names = ops.variant_names()                   # (the diagnostic build is the EXTRAS library under -DVMI_DIAG)
    with _lib.use_diag() as lib:
        assert lib.vmi_is_diag_build() == 1 and lib.vmi_has_extras() == 1 and _lib.load() is lib
        diag_names = ops.variant_names()
    assert _lib.load().vmi_is_diag_build() == 0            # the switch ends with the context
    extra = [n for n in diag_names if n not in full_names]
    # the bandwidth probes, the LDS-staging experiment and (round 6) the work decompositions no pick rule returns — the
    # offline sweeps' comparison points: fp16 / fp8-E4M3 menus of head size 64 / 128 and the one-block split kernels
    probes = [n for n in extra if "LOADSONLY" in n or n.startswith("stage_")]
    compare = [n for n in extra if n not in probes]
    assert probes and 50 <= len(compare) <= 90, (probes, len(compare))
    assert all(re.match(r"^(fp8_)?d(64|128)_", n) for n in compare), compare
    assert [n for n in diag_names if n in full_names] == full_names           # same kernels, same order


def test_library_is_gfx950_only_and_has_no_torch_dependency(tmp_path):
    import shutil

    from vllmini_amd import build

    path = build.build()
    r = subprocess.run(["ldd", path], capture_output=True, text=True)
    out = r.stdout
    assert "libamdhip64" in out, (r.returncode, out, r.stderr)
    deps = [ln.split()[0] for ln in out.splitlines() if ln.strip()]        # sonames only, not resolved paths
    assert not any(("torch" in d) or ("c10" in d) or ("python" in d.lower()) for d in deps), deps
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if os.path.exists(objdump):
        copy = str(tmp_path / "lib.so")          # --offloading extracts the code objects next to its input
        shutil.copy(path, copy)
        r = subprocess.run([objdump, "--offloading", copy], capture_output=True, text=True, cwd=str(tmp_path))
        archs = set(re.findall(r"gfx[0-9a-f]+", r.stdout))
        if archs:
            assert archs == {"gfx950"}, (archs, r.stdout[-2000:], r.stderr[-2000:])


def test_python_surface_matches_reference_signatures():
    """Names, arity and order of paged_attention_cuda.paged_attention_v1 / cache_ops.reshape_and_cache
    (paged_attention_cuda.cpp:7-25, cache_kernels.h:11-14)."""
    import paged_attention_cuda as ext

    assert set(ext.__all__) == {"paged_attention_v1", "paged_attention_v2", "cache_ops"}     # ext/__init__.py:4-8
    pa = [p.name for p in inspect.signature(ext.paged_attention_v1).parameters.values()
          if p.kind is not inspect.Parameter.KEYWORD_ONLY]
    assert pa == ["out", "query", "key_cache", "value_cache", "num_kv_heads", "scale", "block_tables", "seq_lens",
                  "block_size", "max_seq_len", "alibi_slopes", "kv_cache_dtype", "kv_scale", "tp_rank",
                  "blocksparse_local_blocks", "blocksparse_vert_stride", "blocksparse_block_size",
                  "blocksparse_head_sliding_step"]
    rc = list(inspect.signature(ext.cache_ops.reshape_and_cache).parameters)
    assert rc == ["key", "value", "key_cache", "value_cache", "slot_mapping", "kv_cache_dtype", "kv_scale"]
    for name in ("swap_blocks", "copy_blocks", "reshape_and_cache_flash", "convert_fp8"):          # .cpp:56-61
        assert hasattr(ext.cache_ops, name)
    assert list(inspect.signature(ext.cache_ops.swap_blocks).parameters) == ["src", "dst", "block_mapping"]
    assert list(inspect.signature(ext.cache_ops.copy_blocks).parameters) == ["key_caches", "value_caches", "block_mapping"]
    assert list(inspect.signature(ext.cache_ops.convert_fp8).parameters) == ["dst_cache", "src_cache", "kv_scale", "kv_cache_dtype"]
    with pytest.raises(RuntimeError, match="Unsupported data type: int8"):                  # cache_kernels.cu:389-391
        ext.cache_ops.convert_fp8(torch.zeros(4, dtype=torch.uint8), torch.zeros(4, dtype=torch.float16), 1.0, "int8")
    with pytest.raises(RuntimeError, match="src must be on a GPU"):                         # :345
        ext.cache_ops.convert_fp8(torch.zeros(4, dtype=torch.uint8), torch.zeros(4, dtype=torch.float16), 1.0, "fp8")
    v2 = [p.name for p in inspect.signature(ext.paged_attention_v2).parameters.values()
          if p.kind is not inspect.Parameter.KEYWORD_ONLY]
    assert v2 == ["out", "exp_sums", "max_logits", "tmp_out"] + pa[1:]                      # .cpp:27-47


def _cpu_args(D=64, dtype=torch.float16):
    S, H, NB = 2, 4, 8
    return dict(out=torch.empty(S, H, D, dtype=dtype), q=torch.zeros(S, H, D, dtype=dtype),
                kc=torch.zeros(NB, H, D // 8, 16, 8, dtype=torch.float16), vc=torch.zeros(NB, H, D, 16, dtype=torch.float16),
                tab=torch.zeros(S, 4, dtype=torch.int32), lens=torch.ones(S, dtype=torch.int32), H=H)


def test_no_cpu_fallback_ops_raise_on_host_tensors():
    import paged_attention_cuda as ext

    a = _cpu_args()
    with pytest.raises(RuntimeError, match="no CPU path"):
        ext.paged_attention_v1(a["out"], a["q"], a["kc"], a["vc"], a["H"], 0.125, a["tab"], a["lens"], 16, 64, None,
                               "auto", 1.0, 0, 0, 1, 1, 0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ext.cache_ops.reshape_and_cache(a["q"], a["q"], a["kc"], a["vc"], torch.zeros(2, dtype=torch.int64), "auto", 1.0)


def test_argument_validation_before_any_launch():
    import paged_attention_cuda as ext

    a = _cpu_args()
    call = lambda **o: ext.paged_attention_v1(  # noqa: E731
        o.get("out", a["out"]), o.get("q", a["q"]), a["kc"], a["vc"], a["H"], 0.125, a["tab"], a["lens"], 16, 64, None,
        o.get("kvd", "auto"), 1.0, 0, 0, o.get("vert", 1), 1, 0)
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache: fp8_e3m4"):
        call(kvd="fp8_e3m4")                                 # "fp8" / "fp8_e4m3" / "fp8_e5m2" are the reference's fp8 names
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache: int8"):
        call(kvd="int8")                                     # quant_utils.cuh:564
    with pytest.raises(RuntimeError, match="Unsupported input type"):
        call(q=a["q"].double())                              # float / half / bfloat16 are the reference's element types
    with pytest.raises(RuntimeError, match="block-sparse"):
        call(vert=2, kvd="fp8")                              # block-sparse attention: 16-bit caches only
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        ext.cache_ops.reshape_and_cache(a["q"], a["q"], a["kc"], a["vc"], torch.zeros(2, dtype=torch.int64), "fp8_e3m4", 1.0)


def test_native_library_missing_is_a_loud_error(tmp_path, monkeypatch):
    from vllmini_amd import _lib, build

    monkeypatch.setattr(build, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_product", None)
    with pytest.raises(_lib.NativeLibraryError, match="no CPU/torch fallback"):
        _lib.load()
    import paged_attention_cuda as ext                      # the operators do not route around it either

    x = torch.zeros(1)
    with pytest.raises(RuntimeError):
        ext.paged_attention_v1(x, x, x, x, 1, 1.0, x, x, 16, 16, None, "auto", 1.0, 0, 0, 1, 1, 0)


def test_build_returns_at_once_on_a_tree_with_current_libraries_and_no_objects(tmp_path, monkeypatch):
    """The GPU box: the .so files travel with the snapshot, the objects do not (.gpurunignore).  build() — what
    __graft_entry__.build() and _lib.load(build_if_missing=True) call — must then return the library WITHOUT invoking
    hipcc (round 3's ADVICE: the per-object staleness test would have recompiled everything there); with a stale library
    and no objects it must compile."""
    import shutil

    from vllmini_amd import build

    out = tmp_path / "_C"
    out.mkdir()
    for kind in ("LIB_PATH", "EXTRAS_LIB_PATH", "DIAG_LIB_PATH"):
        dst = out / os.path.basename(getattr(build, kind))
        shutil.copy(getattr(build, kind), dst)
        os.utime(dst, None)                                  # newer than every source
        monkeypatch.setattr(build, kind, str(dst))
    monkeypatch.setattr(build, "OUT_DIR", str(out))
    calls = []
    monkeypatch.setattr(build, "_compile", lambda units, verbose: calls.append(("compile", len(units))))
    monkeypatch.setattr(build, "_link", lambda objs, lib, verbose: calls.append(("link", lib)))
    monkeypatch.setattr(build, "_hipcc", lambda: (_ for _ in ()).throw(AssertionError("hipcc asked for")))
    assert not build._have_objects()
    assert build.build() == str(out / "libvmi_paged_attention.so") and not calls
    assert build.build(extras=True).endswith("libvmi_paged_attention_extras.so") and not calls
    assert build.build(diag=True).endswith("libvmi_paged_attention_diag.so") and not calls
    os.utime(out / "libvmi_paged_attention.so", (1, 1))      # a library older than its sources IS rebuilt, objects or not
    build.build()
    assert calls and calls[0][0] == "compile" and calls[0][1] == len(build.PRODUCT_UNITS) and calls[-1][0] == "link"


def test_c_abi_validation_codes_without_gpu():
    """Argument validation in the C entry points runs before any HIP call, so it is testable here."""
    from vllmini_amd import _lib

    lib = _lib.load()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    args = lambda hs=64, bs=16, nkv=4, ptr=p16, qs=256: (  # noqa: E731
        p16, ptr, p16, p16, 1, 4, hs, nkv, 0.125, p16, p16, bs, 16, 1, None, qs, 4096, 1024, 0, None)
    assert lib.vmi_paged_attention_v1_f16(*args(hs=72)) == 2
    assert b"Unsupported head size: 72" in lib.vmi_last_error_string()
    assert lib.vmi_paged_attention_v1_f16(*args(bs=64)) == 3
    assert b"Unsupported block size: 64" in lib.vmi_last_error_string()
    assert lib.vmi_paged_attention_v1_f16(*args(nkv=3)) == 4
    assert lib.vmi_paged_attention_v1_f16(*args(ptr=p16 + 2)) == 5
    assert lib.vmi_paged_attention_v1_f16(*args(qs=257)) == 5
    assert lib.vmi_paged_attention_v1_f16(None, *args()[1:]) == 1
    a = list(args())
    a[4] = 0                                                   # num_seqs == 0: nothing to do, no HIP call
    assert lib.vmi_paged_attention_v1_f16(*a) == 0
    assert lib.vmi_paged_attention_v1_f16_variant(*args(), 9999) == 8
    assert lib.vmi_reshape_and_cache_f16(p16, p16, p16, p16, p16, 1, 4, 64, 16, 4, 256, 256, 0, None) == 9
    assert lib.vmi_reshape_and_cache_f16(p16, p16, p16, p16, p16, 0, 4, 64, 16, 8, 256, 256, 0, None) == 0
    assert lib.vmi_paged_attention_v1_pick_variant(256, 12, 64, 16, 1024) >= 1
    assert lib.vmi_paged_attention_v1_pick_variant(1, 12, 80, 8, 64) >= 1
    assert lib.vmi_paged_attention_v1_pick_variant(1, 12, 72, 16, 64) == 0
    assert lib.vmi_paged_attention_v1_pick_variant(1, 12, 64, 64, 64) == 0
    n = lib.vmi_paged_attention_v1_variant_count()
    names = [lib.vmi_paged_attention_v1_variant_name(i + 1).decode() for i in range(n)]
    assert len(set(names)) == n and all(re.match(r"^(stage_)?(bf16_)?(fp8(e5m2)?_)?(q_)?d\d+_", nm) for nm in names), names
    # every (head size, block size) of the reference's dispatch set has a kernel
    for d in (64, 80, 96, 112, 128, 192, 256):
        for bs in (8, 16, 32):
            assert lib.vmi_paged_attention_v1_pick_variant(4, 8, d, bs, 2048) >= 1, (d, bs)


def test_workload_builder_matches_survey_byte_counts():
    from vllmini_amd.workload import CONFIGS, DecodeConfig, make_workload

    assert CONFIGS["cfg2"].algorithmic_bytes() == 50_434_176
    assert CONFIGS["cfg3"].algorithmic_bytes() == 806_159_360
    assert CONFIGS["cfg4"].algorithmic_bytes() == 4_297_130_496
    wl = make_workload(DecodeConfig("t", 3, 12, 64, 40, 64), "cpu", seed=0, table_sets=2, ragged=True)
    c = wl.cfg
    assert wl.key_cache.shape == (c.num_blocks, 12, 8, 16, 8) and wl.value_cache.shape == (c.num_blocks, 12, 64, 16)
    assert wl.query.stride(0) == 3 * 12 * 64 and wl.key.data_ptr() - wl.query.data_ptr() == 12 * 64 * 2
    t0, t1 = (t[t >= 0] for t in wl.tables)
    assert len(set(t0.tolist())) == len(t0) and not set(t0.tolist()) & set(t1.tolist())   # distinct, disjoint
    lens = wl.seq_lens.to(torch.int64)
    for tab, slot in zip(wl.tables, wl.slots):
        blk = tab[torch.arange(c.batch), (lens - 1) // 16].to(torch.int64)
        assert torch.equal(slot, blk * 16 + (lens - 1) % 16)


def test_heuristic_picks_follow_the_host_hint():
    """vmi_paged_attention_v1_pick_variant[_hint] need no GPU: head size 64 on a full chip gets the balanced kernel
    (it reads the lengths on the device, so the host hint changes nothing there) — unless max_seq_len leaves no room
    for three workgroups' logits per CU; head size 128 keeps the hint: a batch whose mean length is well below its
    longest gets eight waves per head; small batches get as many waves per head as it takes to fill the chip."""
    from vllmini_amd import ops

    names = ops.variant_names()
    pick = lambda *a, **k: names[ops.pick_variant(*a, **k) - 1]          # noqa: E731
    assert pick(256, 12, 64, 1024) == "q_d64_s1q2"
    assert pick(256, 12, 64, 1024, mean_seq_len=1024) == "q_d64_s1q2"
    assert pick(256, 12, 64, 1024, mean_seq_len=512) == "q_d64_s1q2"
    assert pick(2048, 12, 64, 1024) == "d64_h1_w8_u1_nt1"              # more items than resident waves: many waves per head,
    assert pick(320, 12, 64, 512) == "d64_h1_w4_u1_nt1"                #   the hardware dispatcher balances (end of round 3)
    assert pick(240, 12, 64, 1024) == "q_d64_s1q2" and pick(192, 12, 64, 1024) == "d64_h1_w8_u1_nt1"
    # the balanced kernel's LDS (4 waves' logits + the ranking, three workgroups per CU) ends at ~2400 tokens; past it the
    # choice is by how well the launch's workgroups fill the resident slots: 4-head workgroups of one-wave heads ...
    assert pick(256, 12, 64, 3000) == "d64_h4_w1_u1a4_nt1"            # 768 workgroups on 768 slots
    assert pick(320, 12, 64, 3000) == "d64_h1_w2_u1_nt1"              # 960 on 768 would idle 37 % of the second round
    assert pick(256, 12, 64, 4096) == "d64_h1_w2_u1_nt1"              # two 64-KiB workgroups per CU: 768 on 512 slots
    assert pick(320, 12, 64, 4096) == "d64_h1_w2_u1_nt1"              # (within 15 % on paper: the finer form, for ragged batches)
    assert pick(256, 12, 64, 2500) == "d64_h4_w1_u1a4_nt1"            # 768 on 768 slots against 3072 on 2304: stays
    assert pick(256, 12, 64, 8192) == "d64_h1_w4_u1_nt1"              # one 4-wave workgroup per CU is too few waves
    assert pick(256, 12, 64, 16384) == "d64_h1_w8_u1_nt1"
    assert pick(256, 12, 64, 8192, mean_seq_len=2000) == "d64_h1_w8_u1_nt1"
    lib = ops._lib.load()
    q = names.index("q_d64_s1q2") + 1
    assert lib.vmi_paged_attention_v1_variant_fits(q, 1024, 0) == 1
    assert lib.vmi_paged_attention_v1_variant_fits(q, 1024, 1) == 0   # no fused-append twin
    assert lib.vmi_paged_attention_v1_variant_fits(q, 40000, 0) == 0
    assert lib.vmi_paged_attention_v1_variant_fits(names.index("d64_h4_w1_u1_nt1") + 1, 1024, 1) == 1
    assert lib.vmi_paged_attention_v1_variant_fits(0, 1024, 0) == 0
    assert pick(128, 32, 128, 2048) == "d128_mh4_h4_u1_nt1_lock"
    assert pick(128, 32, 128, 2048, mean_seq_len=1000) == "d128_h1_w8_u1_nt1"
    assert "_w16_" in pick(1, 12, 64, 1024)
    assert ops.pick_variant(256, 12, 64, 1024, bf16=True) == 0         # bfloat16 kernels are not in the product library
    assert pick(256, 5, 80, 1024, 32) == "d80_bs32_h1_w4_u1_nt1"       # 1280 (seq, head) units do not fill 256 CUs
    assert pick(1024, 5, 80, 1024, 32) == "d80_bs32_h1_w1_u1_nt1"
    with ops._lib.use_extras():                                        # (the same rules for bfloat16, in the extras library)
        names = ops.variant_names()
        assert pick(256, 12, 64, 1024) == "q_d64_s1q2"
        assert pick(256, 12, 64, 1024, bf16=True) == "bf16_q_d64_s1q2"
        assert pick(256, 12, 64, 8192, bf16=True) == "bf16_d64_bs16_h1_w4_u1_nt1"
        assert pick(256, 12, 64, 8192, mean_seq_len=300, bf16=True) == "bf16_d64_bs16_h1_w8_u1_nt1"


def test_driver_entry_points_compile_and_parse_arguments():
    """bench.py / __graft_entry__.py are run by the driver on a GPU box: at least make sure they compile and that
    bench.py's argument parser accepts the contract's flags (a SyntaxError there would lose the round's numbers)."""
    import py_compile
    import sys

    for f in ("bench.py", "__graft_entry__.py"):
        py_compile.compile(os.path.join(REPO, f), doraise=True)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout


def test_makefile_lists_the_same_translation_units_as_build_py():
    """`make` and `python -m vllmini_amd.build` must produce the same libraries."""
    import re

    from vllmini_amd import build as b

    mk = open(os.path.join(REPO, "Makefile")).read()

    def units(var):
        return re.search(var + r"\s*:=\s*((?:.*\\\n)*.*)\n", mk).group(1).replace("\\\n", " ").split()

    base = lambda srcs: sorted(os.path.basename(s)[: -len(".hip")] for s in srcs)      # noqa: E731
    assert sorted(units("CORE_UNITS")) == base(b.CORE)
    assert sorted(units("ABSENT_UNITS")) == base([b.SRC_ABSENT])
    assert sorted(units("EXTRAS_UNITS")) == base(b.EXTRAS)
    assert sorted(units("DIAG_UNITS")) == base([*b.DIAG_UNITS, *b.DIAG_ONLY])
    # the product library: the core units + the stub that stands in for the out-of-scope ones — and none of those
    assert [s for s, _ in b.PRODUCT_UNITS] == [*b.CORE, b.SRC_ABSENT]
    assert not set(b.EXTRAS) & {s for s, _ in b.PRODUCT_UNITS}
    assert b.SRC_ABSENT not in [s for s, _ in b.EXTRAS_UNITS]

def test_c_abi_is_callable_from_several_host_threads():
    """Error strings and the opt-in switches are per thread: two threads that fail differently each read their own
    message, and one thread's vmi_set_pv_mfma (and, in the diagnostic build, vmi_debug_set_queue_flags) does not leak
    into the other's calls."""
    import threading

    from vllmini_amd import _lib

    lib = _lib.load_diag()          # a superset of the product's entries: the same host code with the mode knob
    buf = (ctypes.c_char * 4096)()
    p16 = (ctypes.addressof(buf) + 15) & ~15

    def args(hs=64, bs=16):
        return (p16, p16, p16, p16, 2, 4, hs, 4, 0.125, p16, p16, bs, 64, 4, None, 256, 4096, 1024, 0, None)

    errors, barrier = [], threading.Barrier(2)

    def worker(kind):
        try:
            barrier.wait()
            for _ in range(300):
                if kind == 0:
                    assert lib.vmi_paged_attention_v1_f16(*args(hs=72)) == 2
                    assert b"head size: 72" in lib.vmi_last_error_string()
                    assert lib.vmi_set_pv_mfma(1) in (0, 1)
                    assert lib.vmi_debug_set_queue_flags(5) in (0, 5)
                else:
                    assert lib.vmi_paged_attention_v1_f16(*args(bs=64)) == 3
                    assert b"block size: 64" in lib.vmi_last_error_string()
                    assert lib.vmi_set_pv_mfma(0) == 0                  # never sees thread 0's opt-in
                    assert lib.vmi_debug_set_queue_flags(0) == 0
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
