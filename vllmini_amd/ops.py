"""Python operator surface of the MI355X paged-attention decode path.

Mirrors, name for name and argument for argument, what the reference's pybind module
`paged_attention_cuda` exports and its Python stack calls:

  paged_attention_v1(...18 positional args...)      paged_attention_cuda.cpp:7-25,:52 ;
                                                    call site vllmini/model/gpt2.py:94-113
  cache_ops.reshape_and_cache(...7 positional...)   cache_kernels.h:11-14 ; paged_attention_cuda.cpp:58 ;
                                                    call site vllmini/model/gpt2.py:81-89

Each function reads sizes and strides off the tensors at the same places the reference
launchers do (attention_kernels.cu:701-707, cache_kernels.cu:265-272) and forwards raw device
pointers to the C-ABI (include/vmi_paged_attention.h) on torch's CURRENT stream, without
synchronising — same async contract as the reference (attention_kernels.cu:736-737).

Error behaviour: the reference raises RuntimeError through TORCH_CHECK for an unsupported
head size / block size / dtype / kv-cache dtype (attention_kernels.cu:764, 801; quant_utils.cuh:538,
564) and validates nothing else.  Here the same cases raise RuntimeError with the same message
stem, and malformed shapes/dtypes/devices (undefined behaviour in the reference) raise too.

There is no CPU or torch fallback: tensors must live on a HIP device and the HIP library must
be loadable, otherwise these functions raise.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

__all__ = [
    "paged_attention_v1",
    "paged_attention_v2",
    "reshape_and_cache",
    "pick_variant",
    "workspace_for",
    "reset_workspaces",
    "variant_fits",
    "variant_names",
    "is_split",
    "check_workspaces",
]

_SUPPORTED_KV_CACHE_DTYPES = ("auto",)
_FP8_KV_CACHE_DTYPES = ("fp8", "fp8_e4m3", "fp8_e5m2")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()




def _check_kv_cache_dtype(kv_cache_dtype: str) -> int:
    """-> 0 for 16-bit caches ("auto"), 1 when the caches hold fp8 E4M3 bytes, 2 for fp8 E5M2 bytes.  The reference maps
    "fp8" and "fp8_e4m3" to E4M3 and "fp8_e5m2" to E5M2 (quant_utils.cuh:529-566); both are built here (the reference's
    own build never defines ENABLE_FP8, so its fp8 path is assert(false) — paged_attention_ext/setup.py:30-45 — and its
    source is the spec)."""
    if kv_cache_dtype in _SUPPORTED_KV_CACHE_DTYPES:
        return 0
    if kv_cache_dtype in ("fp8", "fp8_e4m3"):
        return 1
    if kv_cache_dtype == "fp8_e5m2":
        return 2
    raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")


def _check_fp8_cache_dtype(fp8: int, key_cache: torch.Tensor, value_cache: torch.Tensor, kv_cache_dtype: str) -> None:
    ok = (torch.uint8, torch.float8_e5m2 if fp8 == 2 else torch.float8_e4m3fn)
    if key_cache.dtype not in ok or value_cache.dtype not in ok:
        raise RuntimeError(f"key_cache/value_cache must be uint8 (or {ok[1]}) for kv_cache_dtype='{kv_cache_dtype}'")


def _check_device(name: str, t: torch.Tensor, device: torch.device) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} must be a HIP device tensor (got {t.device}); this operator has no CPU path")
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")


def _raise_native(code: int) -> None:
    raise RuntimeError(f"{_lib.last_error()} (vmi code {code})")


_extras = _lib.require_extras      # the active library if it holds the out-of-scope operators, else RuntimeError("<what>: not in this build ...")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_of(index: int) -> int:
    """The current HIP stream of device `index` as an integer handle (what the C-ABI takes)."""
    if _raw_stream is not None:
        return _raw_stream(index)
    return torch.cuda.current_stream(index).cuda_stream


def _on(name: str, t: torch.Tensor, index: int, dev) -> None:
    """`t` lives on HIP device `index` (there is no CPU path)."""
    if t.get_device() != index:
        _check_device(name, t, dev)      # raises with the full message


def _pa_common(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens,
               block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale, tp_rank,
               blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
               blocksparse_head_sliding_step):
    """Validate and flatten the 18 reference arguments into the C-ABI argument tuple.

    Runs on every call (a decode loop hands in fresh views each step, so nothing about a tensor can be remembered
    safely); written to touch each tensor as few times as possible — shape and stride tuples are fetched once and
    compared as tuples: 11.7 -> about 5 us per call on the GPU box's host (profiles/r02_host_overhead.md)."""
    qs = query.shape
    if len(qs) != 3:
        raise RuntimeError(f"query must be [num_seqs, num_heads, head_size], got {tuple(qs)}")
    qdt = query.dtype
    if qdt not in (torch.float16, torch.bfloat16, torch.float32):
        # the reference dispatches float / half / bf16 (quant_utils.cuh:529-566); its callers only use half
        # (gpt2.py, scheduler.py:13)
        raise RuntimeError(f"Unsupported input type of paged attention: {qdt}")
    fp8 = _check_kv_cache_dtype(kv_cache_dtype)
    f32 = qdt == torch.float32          # x = 4 cache layout, plain kernels: v1 over float32 caches only
    sparse = int(blocksparse_vert_stride) > 1
    if f32 and (fp8 or sparse):
        raise RuntimeError("Unsupported input type of paged attention: torch.float32 is built for kv_cache_dtype='auto' "
                           "without block-sparse attention")
    if sparse:          # is_block_sparse, attention_kernels.cu:822 — kernels of their own
        if fp8:
            raise RuntimeError("block-sparse paged attention (blocksparse_vert_stride > 1) is built for "
                               "kv_cache_dtype='auto' (fp16 / bf16 caches) only")
        if int(blocksparse_block_size) <= 0:
            raise RuntimeError(f"blocksparse_block_size must be positive, got {blocksparse_block_size}")
    index = query.get_device()
    dev = query.device
    if index < 0:
        _check_device("query", query, dev)
    _on("out", out, index, dev)
    _on("key_cache", key_cache, index, dev)
    _on("value_cache", value_cache, index, dev)
    _on("block_tables", block_tables, index, dev)
    _on("seq_lens", seq_lens, index, dev)
    kdt, vdt = key_cache.dtype, value_cache.dtype
    if fp8:
        _check_fp8_cache_dtype(fp8, key_cache, value_cache, kv_cache_dtype)
    elif kdt != qdt or vdt != qdt:
        raise RuntimeError(f"key_cache/value_cache must be {qdt} for kv_cache_dtype='auto'")
    if out.dtype != qdt:
        raise RuntimeError(f"out must be {qdt}, got {out.dtype}")
    if block_tables.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("block_tables and seq_lens must be int32")

    num_seqs, num_heads, head_size = qs  # attention_kernels.cu:701-703
    ks, vs = key_cache.shape, value_cache.shape
    if len(ks) != 5 or len(vs) != 4:
        raise RuntimeError("key_cache must be [num_blocks, num_kv_heads, head_size/x, block_size, x] "
                           "and value_cache [num_blocks, num_kv_heads, head_size, block_size]")
    x = ks[4]
    want_x = 16 if fp8 else (4 if f32 else 8)                     # x = 16 / sizeof(cache_t), attention_kernels.cu:200
    if x != want_x:
        raise RuntimeError(f"key_cache innermost dimension must be {want_x} (16 bytes per chunk), got {x}")
    block_size = int(block_size)
    if ks[3] != block_size or vs[3] != block_size:
        raise RuntimeError(f"block_size={block_size} does not match the cache tensors "
                           f"({ks[3]}, {vs[3]})")
    if ks[2] * x != head_size or vs[2] != head_size:
        raise RuntimeError("cache head_size does not match query head_size")
    num_kv_heads = int(num_kv_heads)
    if ks[1] != num_kv_heads or vs[1] != num_kv_heads:
        raise RuntimeError("cache num_kv_heads does not match num_kv_heads argument")
    qst = query.stride()
    if qst[2] != 1 or qst[1] != head_size:
        raise RuntimeError("query must be contiguous in its last two dimensions")
    kst, vst = key_cache.stride(), value_cache.stride()
    tile = head_size * block_size
    # every (block, head) tile dense; blocks and heads may be strided (size-1 dimensions carry no layout)
    if (kst[4] != 1 and x != 1) or (kst[3] != x and block_size != 1) or (kst[2] != block_size * x and ks[2] != 1) or \
            (kst[1] != tile and ks[1] != 1) or (vst[3] != 1 and block_size != 1) or \
            (vst[2] != block_size and head_size != 1) or (vst[1] != tile and vs[1] != 1):
        raise RuntimeError("each cache block must be dense")
    if vst[0] != kst[0] or vst[1] != kst[1]:
        # the reference applies key_cache's strides to both tensors (attention_kernels.cu:706-707)
        raise RuntimeError("key_cache and value_cache must have identical block/head strides")
    if not out.is_contiguous() or out.numel() != num_seqs * num_heads * head_size:
        # reference tests pass out as [S, H, 1, D] (tests/kernels/paged_attention.py:114)
        raise RuntimeError("out must be contiguous with num_seqs*num_heads*head_size elements")
    ts, tst = block_tables.shape, block_tables.stride()
    if len(ts) != 2 or ts[0] != num_seqs or tst[1] != 1:
        raise RuntimeError("block_tables must be [num_seqs, max_num_blocks_per_seq] with unit inner stride")
    if tst[0] != ts[1] and num_seqs > 1:
        raise RuntimeError("block_tables must be row-contiguous")
    ls = seq_lens.shape
    if len(ls) != 1 or ls[0] != num_seqs or not seq_lens.is_contiguous():
        raise RuntimeError("seq_lens must be a contiguous [num_seqs] tensor")
    alibi_ptr = None
    if alibi_slopes is not None:
        _check_device("alibi_slopes", alibi_slopes, dev)
        if alibi_slopes.dtype != torch.float32 or alibi_slopes.numel() != num_heads or \
                not alibi_slopes.is_contiguous():
            raise RuntimeError("alibi_slopes must be a contiguous float32 [num_heads] tensor")
        alibi_ptr = alibi_slopes.data_ptr()

    return (
        out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        num_seqs, num_heads, head_size, num_kv_heads, float(scale),
        block_tables.data_ptr(), seq_lens.data_ptr(),
        block_size, int(max_seq_len), ts[1],   # attention_kernels.cu:704
        alibi_ptr,
        qst[0], kst[0], kst[1],  # :705-707
        index, _stream_of(index),
    )


# ---- the wrapper-owned workspace of paged_attention_v1 (ownership row: "if a split-KV path needs
# scratch, the Python wrapper allocates it with torch on the same stream").  One tensor per (device, stream): launches on
# one stream are ordered, so they can share it; launches on different streams get one each.  The native library keeps
# nothing — the pointer travels with every call.  Under stream capture nothing is allocated (an allocation would belong
# to the graph's private pool): a stream that has no workspace yet captures today's kernels; call workspace_for() before
# capturing to let the graph hold the split kernels.
_WS: dict = {}
_WS_HEAD_SIZES = (64, 128)
_ws_enabled = True


def _capturing() -> bool:
    try:
        return torch.cuda.is_current_stream_capturing()
    except Exception:
        return False


def workspace_for(index: int, stream: Optional[int] = None, create: bool = True) -> Optional[torch.Tensor]:
    """The workspace the operators hand to the library for launches on (device `index`, its current stream) — allocated
    and zeroed on first use (never during stream capture)."""
    if stream is None:
        stream = _stream_of(index)
    ws = _WS.get((index, stream))
    if ws is None and create and not _capturing():
        lib = _lib.load()
        nbytes = max(int(lib.vmi_paged_attention_v1_workspace_bytes(1, 1, d, 16)) for d in _WS_HEAD_SIZES)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=torch.device("cuda", index))   # zeroed on the current stream
        _WS[(index, stream)] = ws
    return ws


def reset_workspaces(index: Optional[int] = None) -> None:
    """Zero the control words of every cached workspace (of device `index`): after a launch that died in flight."""
    lib = _lib.load()
    for (i, st), ws in list(_WS.items()):
        if index is None or i == index:
            rc = lib.vmi_paged_attention_v1_workspace_reset(ws.data_ptr(), ws.numel(), i, st)
            if rc != 0:
                _raise_native(rc)


def workspace_status(index: int, stream: Optional[int] = None) -> int:
    """Polls that gave up in launches on this workspace (0 on a healthy run).  Synchronises."""
    ws = workspace_for(index, stream, create=False)
    return 0 if ws is None else int(ws[:4].view(torch.int32).item())


def set_workspace_enabled(on: bool) -> bool:
    """Process-wide switch (default on): off = the operators pass no workspace, i.e. yesterday's kernels.  Returns the
    previous setting."""
    global _ws_enabled
    prev, _ws_enabled = _ws_enabled, bool(on)
    return prev


def paged_attention_v1(
    out: torch.Tensor,
    query: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    kv_scale: float,
    tp_rank: int = 0,
    blocksparse_local_blocks: int = 0,
    blocksparse_vert_stride: int = 1,
    blocksparse_block_size: int = 1,
    blocksparse_head_sliding_step: int = 0,
    *,
    _variant: int = 0,
) -> None:
    """Decode attention over the paged KV cache; writes `out` in place, returns None.

    Reference: attention_kernels.cu:805-826 (host), :86-496 (kernel).  `_variant` (keyword only,
    not part of the reference surface) forces a work decomposition for tuning/tests.
    """
    args = _pa_common(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                      seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale,
                      tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
                      blocksparse_block_size, blocksparse_head_sliding_step)
    lib = _lib.load()
    # (the first five branches are the out-of-scope corners of the reference's dispatch: they exist in the extras library
    #  only, include/vmi_paged_attention_extras.h — on the product library _extras raises before any entry is touched)
    if query.dtype == torch.float32:               # the (float, float) dispatch branch: plain kernels, no variants
        if _variant:
            raise RuntimeError("_variant does not apply to float32 tensors")
        rc = _extras("paged_attention_v1 over float32 tensors").vmi_paged_attention_v1_f32(*args)
    elif int(blocksparse_vert_stride) > 1:           # block-sparse attention: its own kernels, no tuning variants
        if _variant:
            raise RuntimeError("_variant does not apply to block-sparse attention")
        rc = _extras("paged_attention_v1 with block-sparse attention").vmi_paged_attention_v1_blocksparse(
            *args, int(query.dtype == torch.bfloat16), int(tp_rank), int(blocksparse_local_blocks),
            int(blocksparse_vert_stride), int(blocksparse_block_size), int(blocksparse_head_sliding_step))
    elif _check_kv_cache_dtype(kv_cache_dtype) == 2:   # fp8 E5M2 cache
        rc = _extras("paged_attention_v1 over fp8-E5M2 pages").vmi_paged_attention_v1_fp8_e5m2(
            *args, float(kv_scale), int(_variant), int(query.dtype == torch.bfloat16))
    elif _check_kv_cache_dtype(kv_cache_dtype):      # fp8 E4M3 cache, float16 or bfloat16 query
        ws = workspace_for(args[18], args[19]) if (_ws_enabled and query.dtype == torch.float16 and args[6] in _WS_HEAD_SIZES) else None
        if ws is not None:
            rc = lib.vmi_paged_attention_v1_fp8_ws(*args, float(kv_scale), ws.data_ptr(), ws.numel(), int(_variant))
        else:
            fn = _extras("paged_attention_v1 over bfloat16 tensors").vmi_paged_attention_v1_fp8_bf16 \
                if query.dtype == torch.bfloat16 else lib.vmi_paged_attention_v1_fp8
            rc = fn(*args, float(kv_scale), int(_variant))
    elif query.dtype == torch.bfloat16:
        rc = _extras("paged_attention_v1 over bfloat16 tensors").vmi_paged_attention_v1_bf16(*args, int(_variant))
    else:
        ws = workspace_for(args[18], args[19]) if (_ws_enabled and args[6] in _WS_HEAD_SIZES) else None
        if ws is not None:
            rc = lib.vmi_paged_attention_v1_f16_ws(*args, ws.data_ptr(), ws.numel(), int(_variant))
        elif _variant:
            rc = lib.vmi_paged_attention_v1_f16_variant(*args, int(_variant))
        else:
            rc = lib.vmi_paged_attention_v1_f16(*args)
    if rc != 0:
        _raise_native(rc)
    return None


def paged_attention_v1_append(
    out: torch.Tensor,
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor] = None,
    kv_cache_dtype: str = "auto",
    kv_scale: float = 1.0,
    *,
    _variant: int = 0,
    write_cache: bool = True,
) -> None:
    """Fused decode step (extension, include/vmi_paged_attention.h: vmi_paged_attention_v1_append_*):

        cache_ops.reshape_and_cache(key, value, key_cache, value_cache, slot_of_position(seq_lens-1), ...)
        paged_attention_v1(out, query, key_cache, value_cache, ...)

    in one launch — the pair the reference issues per layer (gpt2.py:87-112).  `seq_lens` already counts this
    step's token (as in gpt2.py:99-104); its slot is derived from the block table, so no slot_mapping is passed.
    Caches and `out` are bit-identical to the two-op sequence (tests/test_parity_gpu.py).

    write_cache=False (vmi_paged_attention_v1_newest_f16): the same attention — the newest token is read from key / value —
    WITHOUT the cache write; the caller stores the rows later (one reshape_and_cache for all the layers of a token:
    GPT2PagedDecoder(deferred_scatter=True)).  `out` is unchanged by it, the caches are not touched.
    """
    if _check_kv_cache_dtype(kv_cache_dtype):
        raise RuntimeError("paged_attention_v1_append is not built for an fp8 KV cache")
    if query.dtype == torch.float32:
        raise RuntimeError("paged_attention_v1_append is not built for float32 tensors")
    args = _pa_common(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                      seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale, 0, 0, 1, 1, 0)
    num_seqs, _, head_size = (int(s) for s in query.shape)
    for name, t in (("key", key), ("value", value)):
        _check_device(name, t, query.device)
        if t.dtype != query.dtype:
            raise RuntimeError(f"{name} must be {query.dtype}, got {t.dtype}")
        if t.dim() != 3 or tuple(int(x) for x in t.shape) != (num_seqs, int(num_kv_heads), head_size):
            raise RuntimeError(f"{name} must be [num_seqs, num_kv_heads, head_size], got {tuple(t.shape)}")
        if t.stride(2) != 1 or t.stride(1) != head_size:
            raise RuntimeError(f"{name} must be contiguous in its last two dimensions")
    if not write_cache and query.dtype != torch.float16:
        raise RuntimeError("paged_attention_v1_append(write_cache=False) is built for float16 tensors")
    fn = _extras("paged_attention_v1_append over bfloat16 tensors").vmi_paged_attention_v1_append_bf16 \
        if query.dtype == torch.bfloat16 else (_lib.load().vmi_paged_attention_v1_append_f16 if write_cache else
                                               _lib.load().vmi_paged_attention_v1_newest_f16)
    rc = fn(*args, key.data_ptr(), value.data_ptr(), int(key.stride(0)), int(value.stride(0)), int(_variant))
    if rc != 0:
        _raise_native(rc)
    return None


def paged_attention_v2(
    out: torch.Tensor,
    exp_sums: torch.Tensor,
    max_logits: torch.Tensor,
    tmp_out: torch.Tensor,
    query: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    kv_scale: float,
    tp_rank: int = 0,
    blocksparse_local_blocks: int = 0,
    blocksparse_vert_stride: int = 1,
    blocksparse_block_size: int = 1,
    blocksparse_head_sliding_step: int = 0,
    *,
    _variant: int = 0,
) -> None:
    """Split-KV decode attention (512-token partitions + merge); writes out/exp_sums/max_logits/tmp_out.

    Reference: paged_attention_cuda.cpp:27-47 (signature), attention_kernels.cu:966-990 (host),
    :529-562 + :564-669 (kernels).  The reference exports it but no Python caller exists
    (SURVEY.md §2 #7); it is the right operator when num_seqs*num_heads is far below the CU count.
    """
    fp8 = _check_kv_cache_dtype(kv_cache_dtype)
    if query.dtype == torch.float32:
        raise RuntimeError("Unsupported input type of paged attention: torch.float32 is built for paged_attention_v1 only")
    args = _pa_common(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                      seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, kv_scale,
                      tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
                      blocksparse_block_size, blocksparse_head_sliding_step)
    num_seqs, num_heads, head_size = (int(x) for x in query.shape)
    parts = (int(max_seq_len) + 511) // 512                       # attention_kernels.cu:885
    dev = query.device
    for name, t, dt, shape in (("exp_sums", exp_sums, torch.float32, (num_seqs, num_heads, parts)),
                               ("max_logits", max_logits, torch.float32, (num_seqs, num_heads, parts)),
                               ("tmp_out", tmp_out, query.dtype, (num_seqs, num_heads, parts, head_size))):
        _check_device(name, t, dev)
        if t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous {dt} tensor of shape {shape} "
                               f"(max_num_partitions = ceil(max_seq_len/512) = {parts})")
    if int(blocksparse_vert_stride) > 1:
        if _variant:
            raise RuntimeError("_variant does not apply to block-sparse attention")
        rc = _extras("paged_attention_v2 with block-sparse attention").vmi_paged_attention_v2_blocksparse(
            args[0], exp_sums.data_ptr(), max_logits.data_ptr(), tmp_out.data_ptr(), *args[1:],
            int(query.dtype == torch.bfloat16), int(tp_rank), int(blocksparse_local_blocks),
            int(blocksparse_vert_stride), int(blocksparse_block_size), int(blocksparse_head_sliding_step))
    elif fp8 and query.dtype == torch.bfloat16:
        rc = _extras("paged_attention_v2 over bfloat16 tensors").vmi_paged_attention_v2_fp8_bf16(
            args[0], exp_sums.data_ptr(), max_logits.data_ptr(), tmp_out.data_ptr(), *args[1:], float(kv_scale), int(_variant),
            int(fp8 == 2))
    elif fp8:
        fn8 = _extras("paged_attention_v2 over fp8-E5M2 pages").vmi_paged_attention_v2_fp8_e5m2 if fp8 == 2 else \
            _lib.load().vmi_paged_attention_v2_fp8
        rc = fn8(args[0], exp_sums.data_ptr(), max_logits.data_ptr(), tmp_out.data_ptr(), *args[1:],
                 float(kv_scale), int(_variant))
    else:
        fn = _extras("paged_attention_v2 over bfloat16 tensors").vmi_paged_attention_v2_bf16 if query.dtype == torch.bfloat16 else \
            _lib.load().vmi_paged_attention_v2_f16
        rc = fn(args[0], exp_sums.data_ptr(), max_logits.data_ptr(), tmp_out.data_ptr(), *args[1:], int(_variant))
    if rc != 0:
        _raise_native(rc)
    return None


def reshape_and_cache(
    key: torch.Tensor,
    value: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    slot_mapping: torch.Tensor,
    kv_cache_dtype: str,
    kv_scale: float,
) -> None:
    """Scatter new-token K/V rows into the paged caches at `slot_mapping`; in place, returns None.

    Reference: cache_kernels.cu:256-281 (host), :152-207 (kernel).
    """
    fp8 = _check_kv_cache_dtype(kv_cache_dtype)
    kshape = key.shape
    if len(kshape) != 3 or kshape != value.shape:
        raise RuntimeError("key and value must both be [num_tokens, num_heads, head_size]")
    kdt = key.dtype
    if kdt not in (torch.float16, torch.bfloat16, torch.float32) or value.dtype != kdt:
        raise RuntimeError(f"Unsupported input type of reshape_and_cache: {kdt}")
    if kdt == torch.float32 and fp8:
        raise RuntimeError("Unsupported input type of reshape_and_cache: torch.float32 rows with an fp8 cache are not built")
    index = key.get_device()
    dev = key.device
    if index < 0:
        _check_device("key", key, dev)
    _on("value", value, index, dev)
    _on("key_cache", key_cache, index, dev)
    _on("value_cache", value_cache, index, dev)
    _on("slot_mapping", slot_mapping, index, dev)
    if fp8:
        _check_fp8_cache_dtype(fp8, key_cache, value_cache, kv_cache_dtype)
    elif key_cache.dtype != kdt or value_cache.dtype != kdt:
        raise RuntimeError(f"key_cache/value_cache must be {kdt} for kv_cache_dtype='auto'")
    if slot_mapping.dtype != torch.int64:
        raise RuntimeError("slot_mapping must be int64")
    num_tokens, num_heads, head_size = kshape       # cache_kernels.cu:265-267
    ks, vs = key_cache.shape, value_cache.shape
    if len(ks) != 5 or len(vs) != 4:
        raise RuntimeError("key_cache must be [num_blocks, num_heads, head_size/x, block_size, x] "
                           "and value_cache [num_blocks, num_heads, head_size, block_size]")
    block_size = ks[3]                                  # cache_kernels.cu:268
    x = ks[4]                                           # cache_kernels.cu:269
    if not key_cache.is_contiguous() or not value_cache.is_contiguous():
        # the reference computes dense offsets (cache_kernels.cu:187-194)
        raise RuntimeError("key_cache and value_cache must be contiguous")
    if ks[1] != num_heads or ks[2] * x != head_size or tuple(vs[1:]) != (num_heads, head_size, block_size):
        raise RuntimeError("cache shapes do not match key/value shapes")
    kst, vst = key.stride(), value.stride()
    if kst[2] != 1 or kst[1] != head_size or vst[2] != 1 or vst[1] != head_size:
        raise RuntimeError("key/value must be contiguous in their last two dimensions")
    if slot_mapping.numel() != num_tokens or not slot_mapping.is_contiguous():
        raise RuntimeError("slot_mapping must be a contiguous [num_tokens] tensor")
    stream = _stream_of(index)
    if fp8:                                                               # cache_kernels.cu:200-205
        a8 = (key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
              slot_mapping.data_ptr(), num_tokens, num_heads, head_size, block_size, x,
              kst[0], vst[0], float(kv_scale), index, stream)
        if fp8 == 2:
            rc = _extras("reshape_and_cache over fp8-E5M2 pages").vmi_reshape_and_cache_fp8_e5m2(*a8, int(kdt == torch.bfloat16))
        else:
            fn = _extras("reshape_and_cache (fp8) over bfloat16 rows").vmi_reshape_and_cache_fp8_bf16 if kdt == torch.bfloat16 else \
                _lib.load().vmi_reshape_and_cache_fp8
            rc = fn(*a8)
        if rc != 0:
            _raise_native(rc)
        return None
    fn16 = _extras("reshape_and_cache over float32 tensors").vmi_reshape_and_cache_f32 if kdt == torch.float32 else \
        _lib.load().vmi_reshape_and_cache_f16
    rc = fn16(
        key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        slot_mapping.data_ptr(), num_tokens, num_heads, head_size, block_size, x,
        kst[0], vst[0],                                                   # cache_kernels.cu:271-272
        index, stream)
    if rc != 0:
        _raise_native(rc)
    return None


# ---- tuning helpers (not part of the reference surface) --------------------------------------

_NAMES: dict = {}     # per loaded library (product / extras / diag have different menus): its variant names, asked for once


def variant_names() -> list[str]:
    lib = _lib.load()
    names = _NAMES.get(id(lib))
    if names is None:   # (~300 ctypes calls: 170 us — a decode step asks every time)
        n = lib.vmi_paged_attention_v1_variant_count()
        names = _NAMES[id(lib)] = [lib.vmi_paged_attention_v1_variant_name(i + 1).decode() for i in range(n)]
    return list(names)


def is_split(variant: int) -> bool:
    """Is `variant` a split kernel (pa_split.hpp: a (sequence, head) spread over several workgroups that meet in a workspace)?"""
    lib = _lib.load()
    if id(lib) not in _NAMES:
        variant_names()
    names = _NAMES[id(lib)]
    return 1 <= variant <= len(names) and "_x" in names[variant - 1]


def check_workspaces(index: Optional[int] = None) -> None:
    """Raise if any launch on a cached workspace (of device `index`) gave up waiting for another workgroup (the kernel then
    wrote NaN rows and counted the event in the workspace's first word).  Synchronises; for sync points: the end of a decode
    loop, a bench, a test.  Concurrent split launches on ONE workspace — two streams replaying graphs captured on the same
    side stream — are not supported: the workspace is per (device, stream at call or capture time)."""
    for (i, st), ws in list(_WS.items()):
        if index is None or i == index:
            n = int(ws[:4].view(torch.int32).item())
            if n:
                raise RuntimeError(f"paged_attention_v1: {n} poll(s) of a split kernel gave up on the workspace of device {i}, "
                                   f"stream {st:#x}: its workgroups were not co-resident (another kernel on the CUs, or two "
                                   "launches sharing the workspace); the affected rows are NaN.  ops.reset_workspaces() clears it")


def variant_names_v2() -> list[str]:
    lib = _lib.load()
    n = lib.vmi_paged_attention_v2_variant_count()
    return [lib.vmi_paged_attention_v2_variant_name(i + 1).decode() for i in range(n)]


def set_pv_mfma(on: bool) -> bool:
    """Opt-in (PER HOST THREAD since C-ABI 17 — thread-local in the library: set it on the thread that issues the
    launches; a setting made on the main thread does not reach a worker thread — default off; returns the calling
    thread's previous setting): with grouped-query attention let the
    operators pick the "_pvm" kernels, which also run probabilities x V on the matrix cores.  Results then match
    the reference kernel to the north-star 1e-3 instead of 1-2 fp16 ulp (include/vmi_paged_attention.h,
    vmi_set_pv_mfma); 1.2x faster with 8 query heads per KV head."""
    return bool(_lib.load().vmi_set_pv_mfma(int(bool(on))))


def variant_fits(variant: int, max_seq_len: int, for_append=False) -> bool:
    """Can `variant` (from pick_variant) serve a launch with this max_seq_len — and the fused append, if asked
    (for_append=True; "read": its form without the cache write, paged_attention_v1_append(write_cache=False))?
    False -> pass `_variant=0` and let the library choose."""
    mode = 2 if for_append == "read" else int(bool(for_append))
    return bool(_lib.load().vmi_paged_attention_v1_variant_fits(int(variant), int(max_seq_len), mode))


def last_variant() -> int:
    """Id of the variant the calling thread's last paged_attention_v1 launch ran (0: none yet / block-sparse)."""
    return int(_lib.load().vmi_paged_attention_v1_last_variant())


def last_launch_label() -> str:
    """Name(s) of what the calling thread's last paged_attention_v1 launch ran: "a", or "a | b (gated double launch: one
    of the two, chosen on the device from seq_lens)" — for labelling measurements."""
    lib = _lib.load()
    a, b = int(lib.vmi_paged_attention_v1_last_variant()), int(lib.vmi_paged_attention_v1_last_partner())
    if not a:
        return ""
    names = variant_names()
    return names[a - 1] if not b else f"{names[a - 1]} | {names[b - 1]} (gated double launch: the device picks one from seq_lens)"


def pick_variant(num_seqs: int, num_heads: int, head_size: int, max_seq_len: int, block_size: int = 16,
                 mean_seq_len: int = 0, bf16: bool = False, fp8=False, num_kv_heads: int = 0,
                 workspace: bool = False) -> int:
    """The library's work-decomposition heuristic (what `_variant=0` runs).  A caller that knows the batch's
    lengths on the host may pass their mean: a ragged batch (mean well below max_seq_len) then gets the
    many-waves-per-head decomposition; pass the result as `_variant`.  fp8: False / True (E4M3) / "e5m2".
    workspace=True: what the operators run when they hand the library a workspace (fp16 pages; the default of
    paged_attention_v1 outside stream capture)."""
    lib = _lib.load()
    fp8 = 2 if fp8 in (2, "e5m2", "fp8_e5m2") else int(bool(fp8))
    if workspace and not (mean_seq_len or bf16 or fp8):
        return int(lib.vmi_paged_attention_v1_pick_variant_ws(num_seqs, num_heads, int(num_kv_heads or 0), head_size, block_size,
                                                              max_seq_len))
    if num_kv_heads and num_kv_heads != num_heads:      # grouped-query attention: what the operators pick themselves
        return int(lib.vmi_paged_attention_v1_pick_variant_gqa(num_seqs, num_heads, int(num_kv_heads), head_size,
                                                               block_size, max_seq_len, int(bool(bf16)), fp8))
    if (fp8 == 2 or (fp8 and bf16)) and not getattr(lib, "_vmi_has_extras", False):
        return 0                                        # (the product library has no such kernel: what its menus answer for bf16)
    if fp8 == 2:
        return int(lib.vmi_paged_attention_v1_pick_variant_fp8_e5m2(num_seqs, num_heads, head_size, block_size,
                                                                    max_seq_len, int(mean_seq_len), int(bool(bf16))))
    if fp8:
        fn = lib.vmi_paged_attention_v1_pick_variant_fp8_bf16 if bf16 else lib.vmi_paged_attention_v1_pick_variant_fp8
        return int(fn(num_seqs, num_heads, head_size, block_size, max_seq_len, int(mean_seq_len)))
    if mean_seq_len or bf16:
        return int(lib.vmi_paged_attention_v1_pick_variant_hint(num_seqs, num_heads, head_size, block_size,
                                                                max_seq_len, int(mean_seq_len), int(bool(bf16))))
    return int(lib.vmi_paged_attention_v1_pick_variant(num_seqs, num_heads, head_size, block_size, max_seq_len))
