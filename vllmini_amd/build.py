"""Build the gfx950 shared library (C-ABI, include/vmi_paged_attention.h) in-tree with hipcc.

The reference builds its kernels as a torch CUDAExtension for sm_70..sm_89
(paged_attention_ext/setup.py:21-46, build.sh:3-5).  Here there is no torch/pybind in the
native code at all: eight hipcc translation units (core kernels + C-ABI; extra head/block-size instantiations; bfloat16;
the fused-append twin of each of those three kernel menus; the fp8 KV-cache menu)
compiled concurrently and linked into vllmini_amd/_C/libvmi_paged_attention.so, loaded through ctypes (vllmini_amd/_lib.py).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the repo snapshot.
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
SRC_STAGE = os.path.join(CSRC, "pa_stage.hip")                # experiment: pages staged through LDS (global_load_lds)
SOURCES = [SRC, SRC_EXTRA, SRC_BF16, *SRC_APPEND, SRC_FP8, SRC_FP8_BF16, *SRC_FP8_E5M2, *SRC_SPARSE, SRC_F32, SRC_QUEUE,
           SRC_STAGE]
HDR = os.path.join(CSRC, "pa_kernel.hpp")
HDR_QUEUE = os.path.join(CSRC, "pa_queue.hpp")
INCLUDE = os.path.join(REPO_ROOT, "include")
OUT_DIR = os.path.join(PKG_DIR, "_C")
LIB_NAME = "libvmi_paged_attention.so"
LIB_PATH = os.path.join(OUT_DIR, LIB_NAME)

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


def _deps() -> list[str]:
    return [*SOURCES, *TABLES, HDR, HDR_QUEUE, os.path.join(INCLUDE, "vmi_paged_attention.h"), os.path.abspath(__file__)]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps())


def _obj_of(src: str) -> str:
    return os.path.join(OUT_DIR, os.path.basename(src) + ".o")


def _flags_stamp() -> str:
    return " ".join(HIPCC_FLAGS)


def _obj_stale(src: str) -> bool:
    """An object is rebuilt when it is missing, older than anything its depfile (hipcc -MD) lists, or was
    compiled with other flags."""
    obj, dep, stamp = _obj_of(src), _obj_of(src) + ".d", _obj_of(src) + ".flags"
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(stamp)):
        return True
    if open(stamp).read() != _flags_stamp():
        return True
    t = os.path.getmtime(obj)
    text = open(dep).read().replace("\\\n", " ")
    files = text.split(":", 1)[1].split() if ":" in text else []
    for f in [*files, os.path.abspath(__file__)]:
        if not os.path.exists(f) or os.path.getmtime(f) > t:
            return True
    return False


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the translation units that are missing or stale (all of them with force=True), link; return the
    library's path."""
    os.makedirs(OUT_DIR, exist_ok=True)
    todo = [s for s in SOURCES if force or _obj_stale(s)]
    if not todo and os.path.exists(LIB_PATH) and not is_stale():
        return LIB_PATH
    tmp = LIB_PATH + ".tmp"
    procs = []
    for src in todo:                                  # the units compile concurrently
        obj = _obj_of(src)
        cmd = [_hipcc(), *HIPCC_FLAGS, "-MD", "-MF", obj + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    failed = None
    for src, cmd, proc in procs:
        out, err = proc.communicate()
        if proc.returncode != 0:
            failed = failed or RuntimeError(f"hipcc failed ({proc.returncode}): {' '.join(cmd)}\n{out}\n{err}")
            continue                                   # (let the other units finish: their objects stay valid)
        with open(_obj_of(src) + ".flags", "w") as f:
            f.write(_flags_stamp())
    if failed:
        raise failed
    objs = [_obj_of(s) for s in SOURCES]
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", *os.environ.get("VMI_EXTRA_FLAGS", "").split(), *objs, "-o", tmp]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    proc = subprocess.run(link, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"link failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
