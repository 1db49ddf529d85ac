"""ctypes binding of the C-ABI shared libraries (include/vmi_paged_attention.h; the extras library also
include/vmi_paged_attention_extras.h).

There is NO fallback: if the HIP library is missing or does not export the declared
symbols, loading raises.  The operators in ops.py never route around it.
"""
from __future__ import annotations

import contextvars
import ctypes
import os
import threading

from . import build as _build

_c_void_p = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_f32 = ctypes.c_float

# name -> (restype, argtypes); mirrors include/vmi_paged_attention.h one to one
_PA_ARGS = [
    _c_void_p, _c_void_p, _c_void_p, _c_void_p,      # out, query, key_cache, value_cache
    _i32, _i32, _i32, _i32,                          # num_seqs, num_heads, head_size, num_kv_heads
    _f32,                                            # scale
    _c_void_p, _c_void_p,                            # block_tables, seq_lens
    _i32, _i32, _i32,                                # block_size, max_seq_len, max_num_blocks_per_seq
    _c_void_p,                                       # alibi_slopes
    _i64, _i64, _i64,                                # q_stride, kv_block_stride, kv_head_stride
    _i32, _c_void_p,                                 # device, stream
]

SIGNATURES = {
    "vmi_abi_version": (ctypes.c_int, []),
    "vmi_last_error_string": (ctypes.c_char_p, []),
    "vmi_target_arch": (ctypes.c_char_p, []),
    "vmi_paged_attention_v1_f16": (ctypes.c_int, list(_PA_ARGS)),
    "vmi_paged_attention_v1_f16_variant": (ctypes.c_int, list(_PA_ARGS) + [_i32]),
    "vmi_paged_attention_v1_f16_ws": (ctypes.c_int, list(_PA_ARGS) + [_c_void_p, _i64, _i32]),
    "vmi_paged_attention_v1_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "vmi_paged_attention_v1_workspace_reset": (ctypes.c_int, [_c_void_p, _i64, _i32, _c_void_p]),
    "vmi_paged_attention_v1_pick_variant_ws": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "vmi_paged_attention_v1_append_f16": (ctypes.c_int, list(_PA_ARGS) + [_c_void_p, _c_void_p, _i64, _i64, _i32]),
    "vmi_paged_attention_v1_newest_f16": (ctypes.c_int, list(_PA_ARGS) + [_c_void_p, _c_void_p, _i64, _i64, _i32]),
    "vmi_paged_attention_v1_fp8": (ctypes.c_int, list(_PA_ARGS) + [_f32, _i32]),
    "vmi_paged_attention_v1_fp8_ws": (ctypes.c_int, list(_PA_ARGS) + [_f32, _c_void_p, _i64, _i32]),
    "vmi_paged_attention_v2_fp8": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p] + list(_PA_ARGS) + [_f32, _i32]),
    "vmi_reshape_and_cache_fp8": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                                 _i32, _i32, _i32, _i32, _i32, _i64, _i64, _f32, _i32, _c_void_p]),
    "vmi_paged_attention_v1_pick_variant_fp8": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "vmi_paged_attention_v1_variant_count": (ctypes.c_int, []),
    "vmi_paged_attention_v1_variant_name": (ctypes.c_char_p, [_i32]),
    "vmi_paged_attention_v1_pick_variant": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "vmi_paged_attention_v1_pick_variant_gqa": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "vmi_set_pv_mfma": (ctypes.c_int, [_i32]),
    "vmi_paged_attention_v1_last_variant": (ctypes.c_int, []),
    "vmi_paged_attention_v1_last_partner": (ctypes.c_int, []),
    "vmi_paged_attention_v1_variant_fits": (ctypes.c_int, [_i32, _i32, _i32]),
    "vmi_paged_attention_v1_pick_variant_hint": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "vmi_paged_attention_v2_f16": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p] + list(_PA_ARGS) + [_i32]),
    "vmi_paged_attention_v2_variant_count": (ctypes.c_int, []),
    "vmi_paged_attention_v2_variant_name": (ctypes.c_char_p, [_i32]),
    "vmi_copy_blocks": (ctypes.c_int, [_c_void_p, _c_void_p, _i32, _c_void_p, _i32, _i64, _i32, _c_void_p]),
    "vmi_swap_blocks": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _i32, _i64, _i32, _i32, _c_void_p]),
    "vmi_swap_blocks_batched": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _i32, _i64, _i32, _c_void_p]),
    "vmi_is_diag_build": (ctypes.c_int, []),
    "vmi_has_extras": (ctypes.c_int, []),
    "vmi_reshape_and_cache_f16": (ctypes.c_int, [
        _c_void_p, _c_void_p, _c_void_p, _c_void_p,  # key, value, key_cache, value_cache
        _c_void_p,                                   # slot_mapping
        _i32, _i32, _i32, _i32, _i32,                # num_tokens, num_heads, head_size, block_size, x
        _i64, _i64,                                  # key_stride, value_stride
        _i32, _c_void_p,                             # device, stream
    ]),
}

# entries of include/vmi_paged_attention_extras.h: exported by libvmi_paged_attention_extras.so (and the diagnostic build) only —
# the out-of-scope corners of the reference's dispatch
EXTRAS_SIGNATURES = {
    "vmi_paged_attention_v1_bf16": (ctypes.c_int, list(_PA_ARGS) + [_i32]),
    "vmi_paged_attention_v2_bf16": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p] + list(_PA_ARGS) + [_i32]),
    "vmi_paged_attention_v1_blocksparse": (ctypes.c_int, list(_PA_ARGS) + [_i32] * 6),
    "vmi_paged_attention_v2_blocksparse": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p] + list(_PA_ARGS) + [_i32] * 6),
    "vmi_paged_attention_v1_append_bf16": (ctypes.c_int, list(_PA_ARGS) + [_c_void_p, _c_void_p, _i64, _i64, _i32]),
    "vmi_paged_attention_v1_fp8_bf16": (ctypes.c_int, list(_PA_ARGS) + [_f32, _i32]),
    "vmi_reshape_and_cache_fp8_bf16": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                                      _i32, _i32, _i32, _i32, _i32, _i64, _i64, _f32, _i32, _c_void_p]),
    "vmi_paged_attention_v1_pick_variant_fp8_bf16": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32]),
    "vmi_paged_attention_v1_f32": (ctypes.c_int, list(_PA_ARGS)),
    "vmi_reshape_and_cache_f32": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                                 _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _c_void_p]),
    "vmi_convert_fp8": (ctypes.c_int, [_c_void_p, _c_void_p, _i64, _f32, _i32, _i32, _i32, _c_void_p]),
    "vmi_paged_attention_v1_fp8_e5m2": (ctypes.c_int, list(_PA_ARGS) + [_f32, _i32, _i32]),
    "vmi_paged_attention_v2_fp8_bf16": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p] + list(_PA_ARGS) + [_f32, _i32, _i32]),
    "vmi_paged_attention_v2_fp8_e5m2": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p] + list(_PA_ARGS) + [_f32, _i32]),
    "vmi_reshape_and_cache_fp8_e5m2": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                                      _i32, _i32, _i32, _i32, _i32, _i64, _i64, _f32, _i32, _c_void_p, _i32]),
    "vmi_paged_attention_v1_pick_variant_fp8_e5m2": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "vmi_reshape_and_cache_flash_16": (ctypes.c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                                      _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _c_void_p]),
}

# entries of include/vmi_paged_attention_diag.h: exported by the diagnostic build only
DIAG_SIGNATURES = {
    "vmi_debug_set_queue_flags": (ctypes.c_int, [_i32]),
    "vmi_diag_gather_read": (ctypes.c_int, [_c_void_p, _i64, _c_void_p, _i32, _i32, _i32, _i32, _i32, _c_void_p]),
    "vmi_diag_stream_read": (ctypes.c_int, [_c_void_p, _i64, _c_void_p, _i32, _i32, _i32, _c_void_p]),
    "vmi_diag_set_wave_timeline": (ctypes.c_int, [_c_void_p, _i32]),
    "vmi_diag_set_stage_stamps": (ctypes.c_int, [_c_void_p, _i32]),
    "vmi_diag_set_split_stamps": (ctypes.c_int, [_c_void_p, _i32]),
    "vmi_debug_set_split_flags": (ctypes.c_int, [_i32]),
}

ABI_VERSION = 22

_lock = threading.Lock()
_product = None      # libvmi_paged_attention.so
_extras = None       # libvmi_paged_attention_extras.so (opt-in: the reference's out-of-scope dispatch corners)
_diag = None         # libvmi_paged_attention_diag.so (tests / probes only)
# What load() hands to the operators: the product library unless THIS CONTEXT switched (use_extras / use_diag).  A context
# variable, not a process global: a `with use_extras():` on one thread does not move another thread's launches to the other
# library (whose variant ids, thread-local last_variant and pv_mfma state are its own) — a new thread starts on the product
# library and opts in for itself.  Variant ids, cached picks and hipGraphs captured under one library belong to it: resolve
# names and replay graphs inside the same context.
_active = contextvars.ContextVar("vmi_active_library", default=None)


class NativeLibraryError(RuntimeError):
    """The HIP extension is missing / stale / does not match the header."""


def lib_path() -> str:
    return _build.LIB_PATH


def _open(path: str, signatures: dict, want_diag: int, want_extras: int = 0) -> ctypes.CDLL:
    if not os.path.exists(path):
        flag = " --diag" if want_diag else " --extras" if want_extras else ""
        raise NativeLibraryError(
            f"{path} not found: build it with `python -m vllmini_amd.build{flag}` "
            "(or __graft_entry__.build()). There is no CPU/torch fallback for these ops.")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # missing libamdhip64 etc.
        raise NativeLibraryError(f"cannot load {path}: {e}") from e
    for name, (restype, argtypes) in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{path} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.vmi_abi_version()
    if got != ABI_VERSION:
        raise NativeLibraryError(f"{path}: ABI version {got}, Python side expects {ABI_VERSION}")
    if lib.vmi_is_diag_build() != want_diag:
        raise NativeLibraryError(f"{path}: vmi_is_diag_build() = {lib.vmi_is_diag_build()}, expected {want_diag}")
    if lib.vmi_has_extras() != want_extras:
        raise NativeLibraryError(f"{path}: vmi_has_extras() = {lib.vmi_has_extras()}, expected {want_extras}")
    lib._vmi_has_extras = bool(want_extras)
    return lib


def _load_product(build_if_missing: bool = False) -> ctypes.CDLL:
    global _product
    if _product is None:
        with _lock:
            if _product is None:
                if not os.path.exists(_build.LIB_PATH) and build_if_missing:
                    _build.build()
                _product = _open(_build.LIB_PATH, SIGNATURES, 0)
    return _product


def load(build_if_missing: bool = False) -> ctypes.CDLL:
    """The library the operators call: the PRODUCT library (loaded and typed once), unless the calling context switched
    to the extras or the diagnostic build (use_extras / use_diag).  Raises NativeLibraryError if unavailable."""
    lib = _active.get()
    return lib if lib is not None else _load_product(build_if_missing)


def require_extras(what: str) -> ctypes.CDLL:
    """The active library if it holds the out-of-scope operators (include/vmi_paged_attention_extras.h), else the
    RuntimeError every such call on the product library ends in — raised HERE, in front of any use of an entry the product
    library does not export."""
    lib = load()
    if not getattr(lib, "_vmi_has_extras", False):
        raise RuntimeError(
            f"{what}: not in this build of the library (libvmi_paged_attention.so holds the float16 / fp8-E4M3 hot path; "
            "`python -m vllmini_amd.build --extras` builds libvmi_paged_attention_extras.so, "
            "`with vllmini_amd._lib.use_extras():` runs the operators on it)")
    return lib


def load_diag() -> ctypes.CDLL:
    """The diagnostic build (include/vmi_paged_attention_diag.h): the product's entries plus the probes and knobs.
    Loading it does not change what the operators call; use_diag() does."""
    global _diag
    with _lock:
        if _diag is None:
            _diag = _open(_build.DIAG_LIB_PATH, {**SIGNATURES, **EXTRAS_SIGNATURES, **DIAG_SIGNATURES}, 1, 1)
        return _diag


def load_extras() -> ctypes.CDLL:
    """libvmi_paged_attention_extras.so: the product's objects plus the corners of the reference's dispatch that lie
    outside the hot path (bfloat16 / float32 tensors, fp8-E5M2 pages, block-sparse attention, reshape_and_cache_flash,
    convert_fp8 — SURVEY.md §2 rows 8-10).  Loading it does not change what the operators call; use_extras() does."""
    global _extras
    with _lock:
        if _extras is None:
            _extras = _open(_build.EXTRAS_LIB_PATH, {**SIGNATURES, **EXTRAS_SIGNATURES}, 0, 1)
        return _extras


class use_extras:
    """Opt in to the out-of-scope operators: inside this context (or after `use_extras().__enter__()` for a whole process)
    the operators run on libvmi_paged_attention_extras.so.  The product library answers those calls with
    RuntimeError("... not in this build of the library ...").  Variant ids differ between the libraries (the extras one has
    more rows): resolve names inside."""

    def __enter__(self):
        _load_product()
        lib = load_extras()
        self._token = _active.set(lib)
        return lib

    def __exit__(self, *exc):
        _active.reset(self._token)
        return False


class use_diag:
    """Context manager for tests and probes: inside it the operators of this process run on the diagnostic library
    (the extras library's sources under -DVMI_DIAG), so that kernel modes can be forced and the experiment kernels selected by variant id.
    Variant ids differ between the two libraries (the diagnostic one has more rows): resolve names inside."""

    def __enter__(self):
        _load_product()
        lib = load_diag()
        self._token = _active.set(lib)
        return lib

    def __exit__(self, *exc):
        _active.reset(self._token)
        return False


def last_error() -> str:
    s = load().vmi_last_error_string()
    return s.decode("utf-8", "replace") if s else ""
