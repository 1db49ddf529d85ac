"""ctypes wrapper around oracle/pa_kernel_model.c (numpy in, numpy out).  Test infrastructure."""
from __future__ import annotations

import ctypes
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_DIR, "pa_kernel_model.c")
_OUT = os.path.join(_DIR, "_build", "liboracle_pa.so")
_lib = None


def build(force: bool = False) -> str:
    """gcc-compile the C restatement if missing/stale; returns the .so path."""
    if force or not os.path.exists(_OUT) or os.path.getmtime(_OUT) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        tmp = _OUT + ".tmp"
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
               _SRC, "-o", tmp, "-lm"]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"oracle build failed:\n{proc.stderr}")
        os.replace(tmp, _OUT)
    return _OUT


def _load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        lib.vmi_oracle_paged_attention_v1_f16.restype = ctypes.c_int
        lib.vmi_oracle_paged_attention_v1_f16.argtypes = [
            vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, i32, i32, vp, i64, i64, i64, i32, i32]
        lib.vmi_oracle_paged_attention_v1.restype = ctypes.c_int
        lib.vmi_oracle_paged_attention_v1.argtypes = lib.vmi_oracle_paged_attention_v1_f16.argtypes + [i32]
        lib.vmi_oracle_paged_attention_v1_blocksparse.restype = ctypes.c_int
        lib.vmi_oracle_paged_attention_v1_blocksparse.argtypes = lib.vmi_oracle_paged_attention_v1.argtypes + [i32] * 5
        lib.vmi_oracle_paged_attention_v2.restype = ctypes.c_int
        lib.vmi_oracle_paged_attention_v2.argtypes = [
            vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, i32, i32, i32, vp, i64, i64, i64, i32]
        lib.vmi_oracle_paged_attention_v2_blocksparse.restype = ctypes.c_int
        lib.vmi_oracle_paged_attention_v2_blocksparse.argtypes = lib.vmi_oracle_paged_attention_v2.argtypes + [i32] * 5
        lib.vmi_oracle_f2b.restype = ctypes.c_uint16
        lib.vmi_oracle_f2b.argtypes = [f32]
        lib.vmi_oracle_b2f.restype = f32
        lib.vmi_oracle_b2f.argtypes = [ctypes.c_uint16]
        lib.vmi_oracle_paged_attention_v2_f16.restype = ctypes.c_int
        lib.vmi_oracle_paged_attention_v2_f16.argtypes = [
            vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, i32, i32, i32, vp, i64, i64, i64]
        lib.vmi_oracle_reshape_and_cache_f16.restype = ctypes.c_int
        lib.vmi_oracle_reshape_and_cache_f16.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64]
        lib.vmi_oracle_f2h.restype = ctypes.c_uint16
        lib.vmi_oracle_f2h.argtypes = [f32]
        lib.vmi_oracle_h2f.restype = f32
        lib.vmi_oracle_h2f.argtypes = [ctypes.c_uint16]
        lib.vmi_oracle_hmul.restype = ctypes.c_uint16
        lib.vmi_oracle_hmul.argtypes = [ctypes.c_uint16, ctypes.c_uint16]
        lib.vmi_oracle_hadd.restype = ctypes.c_uint16
        lib.vmi_oracle_hadd.argtypes = [ctypes.c_uint16, ctypes.c_uint16]
        _lib = lib
    return _lib


def f2h(x: float) -> int:
    return int(_load().vmi_oracle_f2h(float(x)))


def h2f(h: int) -> float:
    return float(_load().vmi_oracle_h2f(int(h)))


def _elem_strides(a: np.ndarray) -> tuple[int, ...]:
    return tuple(s // a.itemsize for s in a.strides)


def _base_ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def paged_attention_v1(query: np.ndarray, key_cache: np.ndarray, value_cache: np.ndarray,
                       num_kv_heads: int, scale: float, block_tables: np.ndarray,
                       seq_lens: np.ndarray, block_size: int,
                       alibi_slopes: np.ndarray | None = None, threads: int = 1, bf16: bool = False,
                       blocksparse: tuple | None = None, tp_rank: int = 0) -> np.ndarray:
    """Kernel-model output [num_seqs, num_heads, head_size] float16 (uint16 bit patterns when bf16=True).
    blocksparse = (local_blocks, vert_stride, block_size, head_sliding_step), the operator's last four arguments
    (attention_kernels.cu:209-254, 385-393); vert_stride <= 1 or None = dense.

    `query` may be a strided view (row stride = query.strides[0]); caches are float16 arrays in
    the reference layout (uint16 arrays holding bfloat16 bit patterns when bf16=True);
    block_tables int32 [S, MB]; seq_lens int32 [S].
    """
    et = np.uint16 if bf16 else np.float16
    assert query.dtype == et and key_cache.dtype == et and value_cache.dtype == et
    assert query.ndim == 3 and key_cache.ndim == 5 and value_cache.ndim == 4
    S, H, D = query.shape
    qs = _elem_strides(query)
    assert qs[2] == 1 and qs[1] == D, "query must be dense in its last two dims"
    key_cache = np.ascontiguousarray(key_cache)
    value_cache = np.ascontiguousarray(value_cache)
    block_tables = np.ascontiguousarray(block_tables, dtype=np.int32)
    seq_lens = np.ascontiguousarray(seq_lens, dtype=np.int32)
    assert block_tables.shape[0] == S and seq_lens.shape[0] == S
    kb, kh = _elem_strides(key_cache)[:2]
    out = np.zeros((S, H, D), dtype=et)
    alibi = None
    if alibi_slopes is not None:
        alibi = np.ascontiguousarray(alibi_slopes, dtype=np.float32)
    lib = _load()

    def run(lo: int, hi: int) -> int:
        args = (_base_ptr(out), _base_ptr(query), _base_ptr(key_cache), _base_ptr(value_cache),
                S, H, D, int(num_kv_heads), float(scale), _base_ptr(block_tables), _base_ptr(seq_lens),
                int(block_size), int(block_tables.shape[1]),
                None if alibi is None else _base_ptr(alibi), int(qs[0]), int(kb), int(kh), lo, hi, 1 if bf16 else 0)
        if blocksparse is not None:
            loc, vert, bsz, step = (int(v) for v in blocksparse)
            return lib.vmi_oracle_paged_attention_v1_blocksparse(*args, int(tp_rank), loc, vert, bsz, step)
        return lib.vmi_oracle_paged_attention_v1(*args)

    threads = max(1, min(int(threads), S))
    if threads == 1:
        rcs = [run(0, S)]
    else:
        bounds = np.linspace(0, S, threads + 1).astype(int)
        with ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL during the call
            rcs = list(ex.map(lambda i: run(int(bounds[i]), int(bounds[i + 1])), range(threads)))
    for rc in rcs:
        if rc == 1:
            raise RuntimeError(f"Unsupported head size / block size: {D} / {block_size}")
        if rc != 0:
            raise MemoryError("oracle allocation failed")
    return out


def paged_attention_v2(query: np.ndarray, key_cache: np.ndarray, value_cache: np.ndarray,
                       num_kv_heads: int, scale: float, block_tables: np.ndarray, seq_lens: np.ndarray,
                       block_size: int, max_seq_len: int, alibi_slopes: np.ndarray | None = None,
                       bf16: bool = False, blocksparse: tuple | None = None, tp_rank: int = 0):
    """Kernel model of the split-KV operator (attention_kernels.cu:966-990): returns
    (out [S,H,D] f16, exp_sums [S,H,P] f32, max_logits [S,H,P] f32, tmp_out [S,H,P,D] f16) with
    P = ceil(max_seq_len / 512).  Partitions past a sequence's context keep their fill value (NaN)."""
    et = np.uint16 if bf16 else np.float16
    assert query.dtype == et and query.ndim == 3
    S, H, D = query.shape
    qs = _elem_strides(query)
    assert qs[2] == 1 and qs[1] == D
    key_cache = np.ascontiguousarray(key_cache)
    value_cache = np.ascontiguousarray(value_cache)
    block_tables = np.ascontiguousarray(block_tables, dtype=np.int32)
    seq_lens = np.ascontiguousarray(seq_lens, dtype=np.int32)
    kb, kh = _elem_strides(key_cache)[:2]
    P = (int(max_seq_len) + 511) // 512
    out = np.zeros((S, H, D), dtype=et)
    exp_sums = np.full((S, H, P), np.nan, dtype=np.float32)
    max_logits = np.full((S, H, P), np.nan, dtype=np.float32)
    tmp_out = np.full((S, H, P, D), 0x7FC0 if bf16 else np.nan, dtype=et)   # NaN fill in either encoding
    alibi = None if alibi_slopes is None else np.ascontiguousarray(alibi_slopes, dtype=np.float32)
    args = (_base_ptr(out), _base_ptr(exp_sums), _base_ptr(max_logits), _base_ptr(tmp_out), _base_ptr(query),
            _base_ptr(key_cache), _base_ptr(value_cache), S, H, D, int(num_kv_heads), float(scale),
            _base_ptr(block_tables), _base_ptr(seq_lens), int(block_size), int(max_seq_len),
            int(block_tables.shape[1]), None if alibi is None else _base_ptr(alibi), int(qs[0]), int(kb), int(kh),
            1 if bf16 else 0)
    if blocksparse is not None:
        loc, vert, bsz, step = (int(v) for v in blocksparse)
        rc = _load().vmi_oracle_paged_attention_v2_blocksparse(*args, int(tp_rank), loc, vert, bsz, step)
    else:
        rc = _load().vmi_oracle_paged_attention_v2(*args)
    if rc == 1:
        raise RuntimeError(f"Unsupported head size / block size: {D} / {block_size}")
    if rc != 0:
        raise MemoryError("oracle allocation failed")
    return out, exp_sums, max_logits, tmp_out


def paged_attention_v1_f32(query: np.ndarray, key_cache: np.ndarray, value_cache: np.ndarray, num_kv_heads: int,
                           scale: float, block_tables: np.ndarray, seq_lens: np.ndarray, block_size: int,
                           alibi_slopes: np.ndarray | None = None, threads: int = 1) -> np.ndarray:
    """Kernel model for float32 tensors (the (float, float) dispatch branch): key_cache [NB, H, D/4, BS, 4],
    value_cache [NB, H, D, BS], everything fp32 (dtype_float32.cuh)."""
    assert query.dtype == key_cache.dtype == value_cache.dtype == np.float32
    assert query.ndim == 3 and key_cache.ndim == 5 and key_cache.shape[4] == 4 and value_cache.ndim == 4
    S, H, D = query.shape
    qs = _elem_strides(query)
    assert qs[2] == 1 and qs[1] == D
    key_cache, value_cache = np.ascontiguousarray(key_cache), np.ascontiguousarray(value_cache)
    block_tables = np.ascontiguousarray(block_tables, dtype=np.int32)
    seq_lens = np.ascontiguousarray(seq_lens, dtype=np.int32)
    kb, kh = _elem_strides(key_cache)[:2]
    out = np.zeros((S, H, D), dtype=np.float32)
    alibi = None if alibi_slopes is None else np.ascontiguousarray(alibi_slopes, dtype=np.float32)
    fn = _load().vmi_oracle_paged_attention_v1_f32
    fn.restype = ctypes.c_int
    fn.argtypes = ([ctypes.c_void_p] * 4 + [ctypes.c_int32] * 4 + [ctypes.c_float] + [ctypes.c_void_p] * 2 +
                   [ctypes.c_int32] * 2 + [ctypes.c_void_p] + [ctypes.c_int64] * 3 + [ctypes.c_int32] * 2)

    def run(lo: int, hi: int) -> int:
        return fn(_base_ptr(out), _base_ptr(query), _base_ptr(key_cache), _base_ptr(value_cache), S, H, D,
                  int(num_kv_heads), float(scale), _base_ptr(block_tables), _base_ptr(seq_lens), int(block_size),
                  int(block_tables.shape[1]), None if alibi is None else _base_ptr(alibi), int(qs[0]), int(kb), int(kh),
                  lo, hi)

    threads = max(1, min(int(threads), S))
    bounds = np.linspace(0, S, threads + 1).astype(int)
    with ThreadPoolExecutor(threads) as ex:
        rcs = list(ex.map(lambda i: run(int(bounds[i]), int(bounds[i + 1])), range(threads)))
    if any(rc == 1 for rc in rcs):
        raise RuntimeError(f"Unsupported head size / block size: {D} / {block_size}")
    if any(rcs):
        raise MemoryError("oracle allocation failed")
    return out


def reshape_and_cache_f32(key: np.ndarray, value: np.ndarray, key_cache: np.ndarray, value_cache: np.ndarray,
                          slot_mapping: np.ndarray) -> None:
    """cache_kernels.cu:164-199 for float32 tensors (x = 4): a pure copy, written with numpy indexing."""
    assert key.dtype == value.dtype == key_cache.dtype == value_cache.dtype == np.float32 and key_cache.shape[4] == 4
    T, H, D = key.shape
    bs = key_cache.shape[3]
    for t, slot in enumerate(np.asarray(slot_mapping, dtype=np.int64)):
        if slot < 0:
            continue                                                     # :165-169
        blk, off = divmod(int(slot), bs)
        key_cache[blk, :, :, off, :] = key[t].reshape(H, D // 4, 4)
        value_cache[blk, :, :, off] = value[t]


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 bit patterns (uint16), round-to-nearest-even (numpy side of f2b)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r.reshape(np.shape(x))


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(np.shape(b))


def reshape_and_cache(key: np.ndarray, value: np.ndarray, key_cache: np.ndarray,
                      value_cache: np.ndarray, slot_mapping: np.ndarray) -> None:
    """In-place scatter into contiguous 2-byte-element caches (reference layout); float16, or uint16
    arrays holding bfloat16 bit patterns (the op is a pure copy)."""
    assert key.dtype == value.dtype == key_cache.dtype == value_cache.dtype and key.itemsize == 2
    assert key_cache.flags.c_contiguous and value_cache.flags.c_contiguous
    T, H, D = key.shape
    ks, vs = _elem_strides(key), _elem_strides(value)
    assert ks[2] == 1 and ks[1] == D and vs[2] == 1 and vs[1] == D
    slot_mapping = np.ascontiguousarray(slot_mapping, dtype=np.int64)
    assert slot_mapping.shape == (T,)
    rc = _load().vmi_oracle_reshape_and_cache_f16(
        _base_ptr(key), _base_ptr(value), _base_ptr(key_cache), _base_ptr(value_cache),
        _base_ptr(slot_mapping), T, H, D, int(key_cache.shape[3]), int(key_cache.shape[4]),
        int(ks[0]), int(vs[0]))
    assert rc == 0


# ---- fp8 (E4M3) KV cache: kv_cache_dtype "fp8" / "fp8_e4m3" of the reference surface ------------------------
def fp8e4m3_to_f32(bits: np.ndarray) -> np.ndarray:
    """Decode fp8 E4M3 ("fn") bytes exactly (quant_utils.cuh:295-300: __nv_cvt_fp8_to_halfraw)."""
    lib = _load()
    lib.vmi_oracle_fp8e4m3_to_f32.restype = ctypes.c_float
    lib.vmi_oracle_fp8e4m3_to_f32.argtypes = [ctypes.c_uint8]
    flat = np.asarray(bits, dtype=np.uint8).ravel()
    return np.array([lib.vmi_oracle_fp8e4m3_to_f32(int(b)) for b in flat], dtype=np.float32).reshape(np.shape(bits))


def f32_to_fp8e4m3(x: np.ndarray) -> np.ndarray:
    """Encode float32 -> fp8 E4M3 bytes: round to nearest even, saturate to +-448 (__NV_SATFINITE,
    quant_utils.cuh:458-464), NaN -> 0x7f | sign."""
    lib = _load()
    lib.vmi_oracle_f32_to_fp8e4m3.restype = ctypes.c_uint8
    lib.vmi_oracle_f32_to_fp8e4m3.argtypes = [ctypes.c_float]
    flat = np.asarray(x, dtype=np.float32).ravel()
    return np.array([lib.vmi_oracle_f32_to_fp8e4m3(float(v)) for v in flat], dtype=np.uint8).reshape(np.shape(x))


def fp8e5m2_to_f32(bits: np.ndarray) -> np.ndarray:
    """Decode fp8 E5M2 bytes exactly: the upper byte of an IEEE half (kv_cache_dtype "fp8_e5m2", __NV_E5M2)."""
    b = np.ascontiguousarray(bits, dtype=np.uint8)
    return (b.astype(np.uint16) << 8).view(np.float16).astype(np.float32).reshape(np.shape(bits))


def f32_to_fp8e5m2(x: np.ndarray) -> np.ndarray:
    """Encode float32 -> fp8 E5M2 bytes: round to nearest even, saturate to +-57344 (__NV_SATFINITE), NaN -> 0x7f | sign."""
    lib = _load()
    lib.vmi_oracle_f32_to_fp8e5m2.restype = ctypes.c_uint8
    lib.vmi_oracle_f32_to_fp8e5m2.argtypes = [ctypes.c_float]
    flat = np.asarray(x, dtype=np.float32).ravel()
    return np.array([lib.vmi_oracle_f32_to_fp8e5m2(float(v)) for v in flat], dtype=np.uint8).reshape(np.shape(x))


def reshape_and_cache_fp8(key: np.ndarray, value: np.ndarray, key_cache: np.ndarray, value_cache: np.ndarray,
                          slot_mapping: np.ndarray, kv_scale: float = 1.0, bf16: bool = False,
                          e5m2: bool = False) -> None:
    """In-place quantising scatter (cache_kernels.cu:200-205): float16 rows -> uint8 caches
    key_cache [NB, H, D/16, BS, 16], value_cache [NB, H, D, BS]."""
    assert key.dtype == value.dtype == (np.uint16 if bf16 else np.float16)   # bf16: bit patterns
    assert key_cache.dtype == value_cache.dtype == np.uint8
    assert key_cache.flags.c_contiguous and value_cache.flags.c_contiguous and key_cache.shape[4] == 16
    T, H, D = key.shape
    ks, vs = _elem_strides(key), _elem_strides(value)
    assert ks[2] == 1 and ks[1] == D and vs[2] == 1 and vs[1] == D
    slot_mapping = np.ascontiguousarray(slot_mapping, dtype=np.int64)
    lib = _load()
    args = (_base_ptr(key), _base_ptr(value), _base_ptr(key_cache), _base_ptr(value_cache), _base_ptr(slot_mapping),
            T, H, D, int(key_cache.shape[3]), 16, int(ks[0]), int(vs[0]), float(kv_scale))
    base_types = [ctypes.c_void_p] * 5 + [ctypes.c_int32] * 5 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_float]
    if e5m2:
        fn = lib.vmi_oracle_reshape_and_cache_fp8x
        fn.restype, fn.argtypes = ctypes.c_int, base_types + [ctypes.c_int32] * 2
        rc = fn(*args, 1 if bf16 else 0, 2)
    else:
        fn = lib.vmi_oracle_reshape_and_cache_fp8_bf16 if bf16 else lib.vmi_oracle_reshape_and_cache_fp8
        fn.restype, fn.argtypes = ctypes.c_int, base_types
        rc = fn(*args)
    assert rc == 0


def paged_attention_v1_fp8(query: np.ndarray, key_cache: np.ndarray, value_cache: np.ndarray, num_kv_heads: int,
                           scale: float, block_tables: np.ndarray, seq_lens: np.ndarray, block_size: int,
                           kv_scale: float = 1.0, alibi_slopes: np.ndarray | None = None,
                           threads: int = 1, bf16: bool = False, e5m2: bool = False) -> np.ndarray:
    """Kernel model with an fp8 cache (E4M3, or E5M2 with e5m2=True): every cache element is first turned into
    float_to_half(float(fp8) * kv_scale) (quant_utils.cuh:295-300), then the fp16 arithmetic of
    paged_attention_v1 applies unchanged (attention_kernels.cu:283-289, 410-418)."""
    et = np.uint16 if bf16 else np.float16          # bf16: query / out are bit patterns
    assert query.dtype == et and key_cache.dtype == value_cache.dtype == np.uint8
    assert key_cache.ndim == 5 and key_cache.shape[4] == 16 and value_cache.ndim == 4
    S, H, D = query.shape
    qs = _elem_strides(query)
    assert qs[2] == 1 and qs[1] == D
    key_cache, value_cache = np.ascontiguousarray(key_cache), np.ascontiguousarray(value_cache)
    block_tables = np.ascontiguousarray(block_tables, dtype=np.int32)
    seq_lens = np.ascontiguousarray(seq_lens, dtype=np.int32)
    kb, kh = _elem_strides(key_cache)[:2]
    out = np.zeros((S, H, D), dtype=et)
    alibi = None if alibi_slopes is None else np.ascontiguousarray(alibi_slopes, dtype=np.float32)
    lib = _load()
    base_types = ([ctypes.c_void_p] * 4 + [ctypes.c_int32] * 4 + [ctypes.c_float] + [ctypes.c_void_p] * 2 +
                  [ctypes.c_int32] * 2 + [ctypes.c_void_p] + [ctypes.c_int64] * 3 + [ctypes.c_int32] * 2 + [ctypes.c_float])
    extra = ()
    if e5m2:
        fn8 = lib.vmi_oracle_paged_attention_v1_fp8x
        fn8.restype, fn8.argtypes = ctypes.c_int, base_types + [ctypes.c_int32] * 2
        extra = (1 if bf16 else 0, 2)
    else:
        fn8 = lib.vmi_oracle_paged_attention_v1_fp8_bf16 if bf16 else lib.vmi_oracle_paged_attention_v1_fp8
        fn8.restype, fn8.argtypes = ctypes.c_int, base_types

    def run(lo: int, hi: int) -> int:
        return fn8(
            _base_ptr(out), _base_ptr(query), _base_ptr(key_cache), _base_ptr(value_cache), S, H, D,
            int(num_kv_heads), float(scale), _base_ptr(block_tables), _base_ptr(seq_lens), int(block_size),
            int(block_tables.shape[1]), None if alibi is None else _base_ptr(alibi), int(qs[0]), int(kb), int(kh),
            lo, hi, float(kv_scale), *extra)

    threads = max(1, min(int(threads), S))
    bounds = np.linspace(0, S, threads + 1).astype(int)
    with ThreadPoolExecutor(threads) as ex:
        rcs = list(ex.map(lambda i: run(int(bounds[i]), int(bounds[i + 1])), range(threads)))
    if any(rcs):
        raise RuntimeError(f"oracle fp8 attention failed: {rcs}")
    return out


def paged_attention_v2_fp8(query: np.ndarray, key_cache: np.ndarray, value_cache: np.ndarray, num_kv_heads: int,
                           scale: float, block_tables: np.ndarray, seq_lens: np.ndarray, block_size: int,
                           max_seq_len: int, kv_scale: float = 1.0, alibi_slopes: np.ndarray | None = None,
                           e5m2: bool = False, bf16: bool = False):
    """Split-KV kernel model over an fp8 E4M3 (e5m2=True: E5M2) cache: (out, exp_sums, max_logits, tmp_out) as paged_attention_v2."""
    et = np.uint16 if bf16 else np.float16          # bf16: query / out / tmp_out are bit patterns
    assert query.dtype == et and key_cache.dtype == value_cache.dtype == np.uint8 and key_cache.shape[4] == 16
    S, H, D = query.shape
    qs = _elem_strides(query)
    assert qs[2] == 1 and qs[1] == D
    key_cache, value_cache = np.ascontiguousarray(key_cache), np.ascontiguousarray(value_cache)
    block_tables = np.ascontiguousarray(block_tables, dtype=np.int32)
    seq_lens = np.ascontiguousarray(seq_lens, dtype=np.int32)
    kb, kh = _elem_strides(key_cache)[:2]
    P = (int(max_seq_len) + 511) // 512
    out = np.zeros((S, H, D), dtype=et)
    exp_sums = np.full((S, H, P), np.nan, dtype=np.float32)
    max_logits = np.full((S, H, P), np.nan, dtype=np.float32)
    tmp_out = np.full((S, H, P, D), 0x7FC0 if bf16 else np.nan, dtype=et)
    alibi = None if alibi_slopes is None else np.ascontiguousarray(alibi_slopes, dtype=np.float32)
    lib = _load()
    base_types = ([ctypes.c_void_p] * 7 + [ctypes.c_int32] * 4 + [ctypes.c_float] + [ctypes.c_void_p] * 2 +
                  [ctypes.c_int32] * 3 + [ctypes.c_void_p] + [ctypes.c_int64] * 3 + [ctypes.c_float])
    args = (_base_ptr(out), _base_ptr(exp_sums), _base_ptr(max_logits), _base_ptr(tmp_out), _base_ptr(query),
            _base_ptr(key_cache), _base_ptr(value_cache), S, H, D, int(num_kv_heads), float(scale),
            _base_ptr(block_tables), _base_ptr(seq_lens), int(block_size), int(max_seq_len),
            int(block_tables.shape[1]), None if alibi is None else _base_ptr(alibi), int(qs[0]), int(kb), int(kh),
            float(kv_scale))
    if e5m2 or bf16:
        fn = lib.vmi_oracle_paged_attention_v2_fp8x
        fn.restype, fn.argtypes = ctypes.c_int, base_types + [ctypes.c_int32] * 2
        rc = fn(*args, 1 if bf16 else 0, 2 if e5m2 else 1)
    else:
        lib.vmi_oracle_paged_attention_v2_fp8.restype = ctypes.c_int
        lib.vmi_oracle_paged_attention_v2_fp8.argtypes = base_types
        rc = lib.vmi_oracle_paged_attention_v2_fp8(*args)
    if rc:
        raise RuntimeError(f"oracle fp8 v2 failed: {rc}")
    return out, exp_sums, max_logits, tmp_out
