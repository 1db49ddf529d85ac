"""Build the gfx950 shared libraries (C-ABI, include/vmi_paged_attention.h) in-tree with hipcc.

The reference builds its kernels as a torch CUDAExtension for sm_70..sm_89
(paged_attention_ext/setup.py:21-46, build.sh:3-5).  Here there is no torch/pybind in the
native code at all: hipcc translation units compiled concurrently and linked into

  vllmini_amd/_C/libvmi_paged_attention.so        the PRODUCT library — what ops.py / cache_ops.py load (ctypes,
                                                  vllmini_amd/_lib.py).  No diagnostic entry, no kernel that is wrong
                                                  by design, no experiment kernel.
  vllmini_amd/_C/libvmi_paged_attention_diag.so   the DIAGNOSTIC library (`--diag`): the same sources with -DVMI_DIAG —
                                                  adds include/vmi_paged_attention_diag.h's entries (read-bandwidth
                                                  probes, the balanced kernels' mode knob), the "loads only" variants
                                                  and the LDS-staging experiment (pa_stage.hip).  Only the three units
                                                  -DVMI_DIAG changes are compiled a second time; every other object
                                                  is shared with the product library.  Used by tests that force kernel
                                                  modes, scripts/ and scripts/bench_diag.py — never by the operators.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so files are
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
SRC_STAGE = os.path.join(CSRC, "pa_stage.hip")                # experiment: pages staged through LDS (diagnostic library only)
SOURCES = [SRC, SRC_EXTRA, SRC_BF16, *SRC_APPEND, SRC_FP8, SRC_FP8_BF16, *SRC_FP8_E5M2, *SRC_SPARSE, SRC_F32, SRC_QUEUE]
# the diagnostic library: these units are compiled again with -DVMI_DIAG (it changes their variant tables / entries),
# pa_stage.hip exists only there, every other unit's object is the product's
DIAG_UNITS = [SRC, SRC_APPEND[0]]
DIAG_ONLY = [SRC_STAGE]
HDR = os.path.join(CSRC, "pa_kernel.hpp")
HDR_QUEUE = os.path.join(CSRC, "pa_queue.hpp")
INCLUDE = os.path.join(REPO_ROOT, "include")
OUT_DIR = os.path.join(PKG_DIR, "_C")
LIB_NAME = "libvmi_paged_attention.so"
LIB_PATH = os.path.join(OUT_DIR, LIB_NAME)
DIAG_LIB_PATH = os.path.join(OUT_DIR, "libvmi_paged_attention_diag.so")

ARCH = "gfx950"
# -ffp-contract=off: the fp16 p*v products must be rounded before the fp16 adds (reference
# rounding points, dtype_float16.cuh:252-260, 451-457) — no v_pk_fma_f16 contraction.
HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fPIC",
    "-fno-gpu-rdc", *os.environ.get("VMI_EXTRA_FLAGS", "").split(),
    f"-I{INCLUDE}",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (expected on PATH or at /opt/rocm/bin/hipcc)")
    return exe


def _deps(diag: bool = False) -> list[str]:
    return [*SOURCES, *(DIAG_ONLY if diag else []), *TABLES, HDR, HDR_QUEUE,
            os.path.join(INCLUDE, "vmi_paged_attention.h"), os.path.join(INCLUDE, "vmi_paged_attention_diag.h"),
            os.path.abspath(__file__)]


def is_stale(diag: bool = False) -> bool:
    lib = DIAG_LIB_PATH if diag else LIB_PATH
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps(diag))


def _obj_of(src: str, diag: bool = False) -> str:
    return os.path.join(OUT_DIR, os.path.basename(src) + (".diag.o" if diag else ".o"))


def _flags(diag: bool) -> list[str]:
    return [*HIPCC_FLAGS, *(["-DVMI_DIAG"] if diag else [])]


def _obj_stale(src: str, diag: bool = False) -> bool:
    """An object is rebuilt when it is missing, older than anything its depfile (hipcc -MD) lists, or was
    compiled with other flags."""
    obj = _obj_of(src, diag)
    dep, stamp = obj + ".d", obj + ".flags"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(stamp)):
        return True
    if open(stamp).read() != " ".join(_flags(diag)):
        return True
    t = os.path.getmtime(obj)
    text = open(dep).read().replace("\\\n", " ")
    files = text.split(":", 1)[1].split() if ":" in text else []
    for f in [*files, os.path.abspath(__file__)]:
        if not os.path.exists(f) or os.path.getmtime(f) > t:
            return True
    return False


def _compile(units, verbose: bool) -> None:
    """units: [(src, diag)] — compiled concurrently."""
    procs = []
    for src, diag in units:
        obj = _obj_of(src, diag)
        cmd = [_hipcc(), *_flags(diag), "-MD", "-MF", obj + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, diag, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    failed = None
    for src, diag, cmd, proc in procs:
        out, err = proc.communicate()
        if proc.returncode != 0:
            failed = failed or RuntimeError(f"hipcc failed ({proc.returncode}): {' '.join(cmd)}\n{out}\n{err}")
            continue                                   # (let the other units finish: their objects stay valid)
        with open(_obj_of(src, diag) + ".flags", "w") as f:
            f.write(" ".join(_flags(diag)))
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


def _have_objects() -> bool:
    return os.path.isdir(OUT_DIR) and any(f.endswith(".o") for f in os.listdir(OUT_DIR))


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """Compile the translation units that are missing or stale (all of them with force=True), link; return the
    library's path.  diag=True builds BOTH libraries (the diagnostic one shares the product's objects) and returns
    the diagnostic library's path.

    A tree that holds current libraries but NO objects — the GPU box: objects do not travel (.gpurunignore) — is
    complete as it is: nothing is recompiled there."""
    os.makedirs(OUT_DIR, exist_ok=True)
    want = [LIB_PATH] + ([DIAG_LIB_PATH] if diag else [])
    if not force and not _have_objects() and not is_stale(False) and (not diag or not is_stale(True)):
        return want[-1]
    units = [(s, False) for s in SOURCES if force or _obj_stale(s)]
    if diag:
        units += [(s, True) for s in [*DIAG_UNITS, *DIAG_ONLY] if force or _obj_stale(s, True)]
    if not units and all(os.path.exists(w) for w in want) and not is_stale(False) and (not diag or not is_stale(True)):
        return want[-1]
    _compile(units, verbose)
    if any(not d for _, d in units) or not os.path.exists(LIB_PATH) or is_stale(False):
        _link([_obj_of(s) for s in SOURCES], LIB_PATH, verbose)
    if diag:
        objs = [_obj_of(s, s in DIAG_UNITS) for s in SOURCES] + [_obj_of(s, True) for s in DIAG_ONLY]
        _link(objs, DIAG_LIB_PATH, verbose)
    return want[-1]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
