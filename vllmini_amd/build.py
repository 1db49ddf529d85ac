"""Build the gfx950 shared libraries (C-ABI, include/vmi_paged_attention.h) in-tree with hipcc.

The reference builds its kernels as a torch CUDAExtension for sm_70..sm_89
(paged_attention_ext/setup.py:21-46, build.sh:3-5).  Here there is no torch/pybind in the
native code at all: hipcc translation units compiled concurrently and linked into

  vllmini_amd/_C/libvmi_paged_attention.so          the PRODUCT library — what ops.py / cache_ops.py load (ctypes,
                                                    vllmini_amd/_lib.py): the hot path of SURVEY.md §8 and nothing else —
                                                    float16 tensors over float16 or fp8-E4M3 pages (every head / block
                                                    size of the reference's dispatch, grouped-query heads, ALiBi,
                                                    paged_attention_v2, the fused append, copy/swap_blocks).  Seven units.
                                                    No diagnostic entry, no kernel that is wrong by design, no experiment
                                                    kernel, and none of the out-of-scope element types and operators: their
                                                    kernel menus are EMPTY here (pa_extras_absent.hip, chosen at link time)
                                                    and their C-ABI entries (include/vmi_paged_attention_extras.h) do not
                                                    exist in it.
  vllmini_amd/_C/libvmi_paged_attention_extras.so   (`--extras`) the product's objects plus the rest of the reference's
                                                    dispatch surface (out of the path's scope):
                                                    bfloat16 and float32 tensors, fp8-E5M2 pages, block-sparse attention,
                                                    reshape_and_cache_flash, convert_fp8.  Opt-in: `_lib.use_extras()`;
                                                    the operators raise RuntimeError("... not in this build ...") without it.
  vllmini_amd/_C/libvmi_paged_attention_diag.so     (`--diag`) the extras library with -DVMI_DIAG — adds
                                                    include/vmi_paged_attention_diag.h's entries (read-bandwidth probes, the
                                                    balanced kernels' mode knob), the "loads only" variants and the
                                                    LDS-staging experiment (pa_stage.hip).  Only the units -DVMI_DIAG changes
                                                    are compiled a second time (pa_queue.hip a third: its kernels record a
                                                    per-wave timeline there).  Used by tests that force kernel modes,
                                                    scripts/ and scripts/bench_diag.py — never by the operators.

Every object is compiled once and shared by the libraries that hold it (pa_queue.hip twice: its bfloat16 / E5M2 rows are
behind -DVMI_EXTRAS).  hipcc cross-compiles without a GPU, so this runs in the build container; the .so files are
git-ignored but travel to the GPU box with the repo snapshot (the objects do not: .gpurunignore).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
SRC = os.path.join(CSRC, "paged_attention.hip")              # core kernels + host code + C-ABI
SRC_EXTRA = os.path.join(CSRC, "pa_variants_extra.hip")       # remaining head/block-size instantiations
SRC_BF16 = os.path.join(CSRC, "pa_variants_bf16.hip")         # bfloat16 instantiations
SRC_APPEND = [os.path.join(CSRC, f"pa_append_{t}.hip") for t in ("core", "extra", "bf16")]   # fused-append twins
TABLES = [os.path.join(CSRC, f"pa_table_{t}.inc") for t in ("core", "extra", "bf16", "fp8", "sparse")] + \
    [os.path.join(CSRC, f"pa_variants_fp8{t}_body.inc") for t in ("", "_bf16")]         # shared kernel menus
SRC_FP8 = os.path.join(CSRC, "pa_variants_fp8.hip")           # fp8 E4M3 KV-cache instantiations
SRC_FP8_BF16 = os.path.join(CSRC, "pa_variants_fp8_bf16.hip")  # ... with a bfloat16 query
SRC_SPARSE = [os.path.join(CSRC, f"pa_variants_sparse{t}.hip") for t in ("", "_bf16")]   # block-sparse attention
SRC_FP8_E5M2 = [os.path.join(CSRC, f"pa_variants_fp8_e5m2{t}.hip") for t in ("", "_bf16")]   # ... over E5M2 bytes
SRC_F32 = os.path.join(CSRC, "pa_f32.hip")                    # float32 tensors (x = 4)
SRC_QUEUE = os.path.join(CSRC, "pa_queue.hip")                # balanced (work-queue) kernels for ragged batches
SRC_SPLIT = os.path.join(CSRC, "pa_split.hip")                # split kernels: one (sequence, head) over several workgroups + workspace
SRC_STAGE = os.path.join(CSRC, "pa_stage.hip")                # experiment: pages staged through LDS (diagnostic library only)
SRC_ABSENT = os.path.join(CSRC, "pa_extras_absent.hip")      # product library: empty out-of-scope menus, no out-of-scope entry
SRC_EXTRAS_CACHE = os.path.join(CSRC, "pa_extras_cache.hip")  # extras: convert_fp8, reshape_and_cache_flash, bf16 / E5M2 fp8 scatter
SRC_EXTRAS_ABI = os.path.join(CSRC, "pa_extras_abi.hip")      # extras: the C-ABI entries of include/vmi_paged_attention_extras.h
# (source, flavor): flavor "" = plain, "extras" = -DVMI_EXTRAS, "diag" = -DVMI_DIAG (+ -DVMI_EXTRAS)
CORE = [SRC, SRC_EXTRA, SRC_APPEND[0], SRC_APPEND[1], SRC_FP8, SRC_QUEUE, SRC_SPLIT]          # the hot path
EXTRAS = [SRC_BF16, SRC_APPEND[2], SRC_FP8_BF16, *SRC_FP8_E5M2, *SRC_SPARSE, SRC_F32, SRC_EXTRAS_CACHE, SRC_EXTRAS_ABI]
PRODUCT_UNITS = [(s, "") for s in CORE] + [(SRC_ABSENT, "")]
EXTRAS_UNITS = [(s, "extras" if s == SRC_QUEUE else "") for s in CORE] + [(s, "") for s in EXTRAS]
# the diagnostic library: these units are compiled again with -DVMI_DIAG (it changes their variant tables / entries),
# pa_stage.hip exists only there, every other object is the extras library's
DIAG_UNITS = [SRC, SRC_APPEND[0], SRC_FP8, SRC_QUEUE, SRC_SPLIT]
DIAG_ONLY = [SRC_STAGE]
DIAG_LIB_UNITS = [(s, "diag") if s in DIAG_UNITS else (s, f) for s, f in EXTRAS_UNITS] + [(s, "diag") for s in DIAG_ONLY]
SOURCES = [*CORE, SRC_ABSENT, *EXTRAS]                        # every unit of the product and extras libraries
HDR = os.path.join(CSRC, "pa_kernel.hpp")
HDR_QUEUE = os.path.join(CSRC, "pa_queue.hpp")
HDR_SPLIT = os.path.join(CSRC, "pa_split.hpp")
HDRS_HOST = [os.path.join(CSRC, "pa_host.hpp"), os.path.join(CSRC, "pa_cache_fp8.hpp")]
INCLUDE = os.path.join(REPO_ROOT, "include")
OUT_DIR = os.path.join(PKG_DIR, "_C")
LIB_NAME = "libvmi_paged_attention.so"
LIB_PATH = os.path.join(OUT_DIR, LIB_NAME)
DIAG_LIB_PATH = os.path.join(OUT_DIR, "libvmi_paged_attention_diag.so")
EXTRAS_LIB_PATH = os.path.join(OUT_DIR, "libvmi_paged_attention_extras.so")
# the decode harness's own library (include/vmi_gpt2_layer.h): the GPT-2 block's linear layers — one unit, no shared objects,
# not part of the operators' boundary; loaded by vllmini_amd/gpt2_layer.py only
SRC_LAYER = os.path.join(CSRC, "gpt2_layer.hip")
LAYER_LIB_PATH = os.path.join(OUT_DIR, "libvmi_gpt2_layer.so")

ARCH = "gfx950"
# -ffp-contract=off: the fp16 p*v products must be rounded before the fp16 adds (reference
# rounding points, dtype_float16.cuh:252-260, 451-457) — no v_pk_fma_f16 contraction.
HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fPIC",
    "-fvisibility=hidden",          # only the extern "C" entries (VMI_API in the headers) leave the library
    "-fno-gpu-rdc", *os.environ.get("VMI_EXTRA_FLAGS", "").split(),
    f"-I{INCLUDE}",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (expected on PATH or at /opt/rocm/bin/hipcc)")
    return exe


def _lib_of(kind: str) -> str:
    return {"product": LIB_PATH, "extras": EXTRAS_LIB_PATH, "diag": DIAG_LIB_PATH}[kind]


def _units_of(kind: str):
    return {"product": PRODUCT_UNITS, "extras": EXTRAS_UNITS, "diag": DIAG_LIB_UNITS}[kind]


def _deps(kind: str = "product") -> list[str]:
    return [*(s for s, _ in _units_of(kind)), *TABLES, HDR, HDR_QUEUE, HDR_SPLIT, *HDRS_HOST,
            os.path.join(INCLUDE, "vmi_paged_attention.h"), os.path.join(INCLUDE, "vmi_paged_attention_extras.h"),
            os.path.join(INCLUDE, "vmi_paged_attention_diag.h"),
            os.path.abspath(__file__)]


def is_stale(kind="product") -> bool:
    kind = {False: "product", True: "diag"}.get(kind, kind)      # (older callers pass diag: bool)
    lib = _lib_of(kind)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps(kind))


def _obj_of(src: str, flavor="") -> str:
    flavor = {False: "", True: "diag"}.get(flavor, flavor)
    return os.path.join(OUT_DIR, os.path.basename(src) + (f".{flavor}.o" if flavor else ".o"))


def _flags(flavor="") -> list[str]:
    flavor = {False: "", True: "diag"}.get(flavor, flavor)
    extra = {"": [], "extras": ["-DVMI_EXTRAS"], "diag": ["-DVMI_DIAG", "-DVMI_EXTRAS"]}[flavor]
    return [*HIPCC_FLAGS, *extra]


def _obj_stale(src: str, flavor="") -> bool:
    """An object is rebuilt when it is missing, older than anything its depfile (hipcc -MD) lists, or was
    compiled with other flags."""
    obj = _obj_of(src, flavor)
    dep, stamp = obj + ".d", obj + ".flags"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(stamp)):
        return True
    if open(stamp).read() != " ".join(_flags(flavor)):
        return True
    t = os.path.getmtime(obj)
    text = open(dep).read().replace("\\\n", " ")
    files = text.split(":", 1)[1].split() if ":" in text else []
    for f in [*files, os.path.abspath(__file__)]:
        if not os.path.exists(f) or os.path.getmtime(f) > t:
            return True
    return False


def _compile(units, verbose: bool) -> None:
    """units: [(src, flavor)] — compiled concurrently."""
    procs = []
    for src, flavor in units:
        obj = _obj_of(src, flavor)
        cmd = [_hipcc(), *_flags(flavor), "-MD", "-MF", obj + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, flavor, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    failed = None
    for src, flavor, cmd, proc in procs:
        out, err = proc.communicate()
        if proc.returncode != 0:
            failed = failed or RuntimeError(f"hipcc failed ({proc.returncode}): {' '.join(cmd)}\n{out}\n{err}")
            continue                                   # (let the other units finish: their objects stay valid)
        with open(_obj_of(src, flavor) + ".flags", "w") as f:
            f.write(" ".join(_flags(flavor)))
    if failed:
        raise failed


def _link(objs, lib: str, verbose: bool) -> None:
    tmp = lib + ".tmp"
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc",
            *os.environ.get("VMI_EXTRA_FLAGS", "").split(), *objs, "-o", tmp]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    proc = subprocess.run(link, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"link failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    os.replace(tmp, lib)


def layer_is_stale() -> bool:
    if not os.path.exists(LAYER_LIB_PATH):
        return True
    t = os.path.getmtime(LAYER_LIB_PATH)
    return any(os.path.getmtime(d) > t for d in (SRC_LAYER, os.path.join(INCLUDE, "vmi_gpt2_layer.h")))


def build_layer(force: bool = False, verbose: bool = False) -> str:
    """libvmi_gpt2_layer.so: one translation unit, compiled and linked in one hipcc call (~20 s)."""
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(LAYER_LIB_PATH) and not _have_objects():
        return LAYER_LIB_PATH      # a tree with libraries and no objects is the GPU box: complete as it is (see build())
    if force or layer_is_stale():
        tmp = LAYER_LIB_PATH + ".tmp"
        cmd = [_hipcc(), *[f for f in HIPCC_FLAGS if f != "-ffp-contract=off"], "-shared", SRC_LAYER, "-o", tmp]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed ({proc.returncode}): {' '.join(cmd)}\n{proc.stdout}\n{proc.stderr}")
        os.replace(tmp, LAYER_LIB_PATH)
    return LAYER_LIB_PATH


def _have_objects() -> bool:
    return os.path.isdir(OUT_DIR) and any(f.endswith(".o") for f in os.listdir(OUT_DIR))


def build(force: bool = False, verbose: bool = False, diag: bool = False, extras: bool = False) -> str:
    """Compile the translation units that are missing or stale (all of them with force=True), link; return the path of
    the last library asked for: the product library, then (extras=True) libvmi_paged_attention_extras.so, then (diag=True)
    the diagnostic library — which is the extras library under -DVMI_DIAG, so diag=True builds all three.

    A tree that holds current libraries but NO objects — the GPU box: objects do not travel (.gpurunignore) — is
    complete as it is: nothing is recompiled there."""
    os.makedirs(OUT_DIR, exist_ok=True)
    build_layer(force, verbose)
    kinds = ["product"] + (["extras"] if (extras or diag) else []) + (["diag"] if diag else [])
    if not force and not _have_objects() and not any(is_stale(k) for k in kinds):
        return _lib_of(kinds[-1])
    wanted = []
    for k in kinds:
        for u in _units_of(k):
            if u not in wanted:
                wanted.append(u)
    units = [u for u in wanted if force or _obj_stale(*u)]
    _compile(units, verbose)
    for k in kinds:
        if force or is_stale(k) or any(u in units for u in _units_of(k)):
            _link([_obj_of(*u) for u in _units_of(k)], _lib_of(k), verbose)
    return _lib_of(kinds[-1])


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv, extras="--extras" in sys.argv))
