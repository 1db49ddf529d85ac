"""`cache_ops` submodule of the reference's extension (paged_attention_cuda.cpp:55-61).

  reshape_and_cache   the one op the reference's Python calls (vllmini/model/gpt2.py:81)   -> ops.py
  copy_blocks         cache_kernels.cu:96-148    multi-layer block copy (copy-on-write / forking)
  swap_blocks         cache_kernels.cu:24-63     block moves device<->device / device<->host (preemption)
  reshape_and_cache_flash  cache_kernels.cu:283-317  scatter into the flash layout [NB, block_size, H, D]
  convert_fp8              cache_kernels.cu:320-392  whole-cache conversion fp8 E4M3 <-> float / half / bfloat16
                           ("only for testing" in the reference, compiled to assert(false) in its shipped build)
                           (these two are outside the hot path: include/vmi_paged_attention_extras.h — they run inside
                           `_lib.use_extras()` and raise RuntimeError("... not in this build ...") on the product library)
"""
from __future__ import annotations

import ctypes
from typing import List

import torch

from . import _lib
from .ops import _check_device, _raise_native, reshape_and_cache  # noqa: F401

__all__ = ["reshape_and_cache", "reshape_and_cache_flash", "swap_blocks", "swap_blocks_batched", "copy_blocks", "convert_fp8"]


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: torch.Tensor) -> None:
    """For every layer and every (src, dst) row of `block_mapping` copy one K and one V block in place.
    Reference: cache_kernels.cu:96-148.  block_mapping: int64 [num_pairs, 2] on the caches' device."""
    num_layers = len(key_caches)
    if num_layers != len(value_caches):
        raise RuntimeError("key_caches and value_caches must have the same length")      # :100
    if num_layers == 0:
        return None                                                                      # :101-103
    dev = key_caches[0].device
    if not key_caches[0].is_cuda:
        raise RuntimeError("copy_blocks: caches must be HIP device tensors")             # :105
    if block_mapping.dim() != 2 or block_mapping.shape[1] != 2 or block_mapping.dtype != torch.int64:
        raise RuntimeError("block_mapping must be an int64 [num_pairs, 2] tensor")
    _check_device("block_mapping", block_mapping, dev)
    block_mapping = block_mapping.contiguous()
    block_bytes = key_caches[0].element_size() * key_caches[0][0].numel()                # :129
    for k, v in zip(key_caches, value_caches):
        _check_device("key_cache", k, dev)
        _check_device("value_cache", v, dev)
        if not k.is_contiguous() or not v.is_contiguous():
            raise RuntimeError("copy_blocks: caches must be contiguous")
        if k.element_size() * k[0].numel() != block_bytes or v.element_size() * v[0].numel() != block_bytes:
            raise RuntimeError("copy_blocks: all caches must have the same block size in bytes")
    kp = (ctypes.c_void_p * num_layers)(*[k.data_ptr() for k in key_caches])
    vp = (ctypes.c_void_p * num_layers)(*[v.data_ptr() for v in value_caches])
    rc = _lib.load().vmi_copy_blocks(kp, vp, num_layers, block_mapping.data_ptr(), int(block_mapping.shape[0]),
                                     block_bytes, dev.index if dev.index is not None else torch.cuda.current_device(),
                                     torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        _raise_native(rc)
    return None


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: torch.Tensor) -> None:
    """dst[block_mapping[i,1]] = src[block_mapping[i,0]] for every row, async on the current stream.
    Reference: cache_kernels.cu:24-63.  block_mapping must be a CPU int64 [n, 2] tensor (:45)."""
    if block_mapping.is_cuda:
        raise RuntimeError("block_mapping must be on CPU")                               # :45
    if block_mapping.dim() != 2 or block_mapping.shape[1] != 2 or block_mapping.dtype != torch.int64:
        raise RuntimeError("block_mapping must be an int64 [num_pairs, 2] tensor")
    if src.is_cuda and dst.is_cuda:
        if src.device != dst.device:
            raise RuntimeError("src and dst must be on the same GPU")                    # :30-31
        kind, dev = 0, src.device
    elif src.is_cuda and not dst.is_cuda:
        kind, dev = 1, src.device
    elif not src.is_cuda and dst.is_cuda:
        kind, dev = 2, dst.device
    else:
        raise RuntimeError("Invalid device combination")                                 # :39
    if not src.is_contiguous() or not dst.is_contiguous():
        raise RuntimeError("swap_blocks: src and dst must be contiguous")
    block_bytes = src.element_size() * src[0].numel()                                    # :50
    if dst.element_size() * dst[0].numel() != block_bytes:
        raise RuntimeError("swap_blocks: src and dst blocks differ in size")
    bm = block_mapping.contiguous()
    if bm.numel():
        if int(bm[:, 0].max()) >= src.shape[0] or int(bm[:, 1].max()) >= dst.shape[0] or int(bm.min()) < 0:
            raise RuntimeError("swap_blocks: block number out of range")
    rc = _lib.load().vmi_swap_blocks(src.data_ptr(), dst.data_ptr(), bm.data_ptr(), int(bm.shape[0]), block_bytes,
                                     kind, dev.index if dev.index is not None else torch.cuda.current_device(),
                                     torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        _raise_native(rc)
    return None


def swap_blocks_batched(src_key: torch.Tensor, src_value: torch.Tensor, dst_key: torch.Tensor, dst_value: torch.Tensor,
                        block_mapping: torch.Tensor) -> None:
    """dst_key[m[i,1]] = src_key[m[i,0]] and dst_value[m[i,1]] = src_value[m[i,0]] for every row of `block_mapping` in ONE
    launch on the current stream (include/vmi_paged_attention.h: vmi_swap_blocks_batched) — the pool's preemption move,
    counterpart of BlockManager.swap_to_cpu / swap_from_cpu (vllmini/block_manager.py:70-87), which the reference runs as one
    copy per block (cache_kernels.cu:56-62).  Each side is a pair of HIP device tensors or a pair of PINNED host tensors (the
    GPU addresses those pages directly); block_mapping int64 [n, 2], CPU or device.  The bytes moved are swap_blocks'."""
    if block_mapping.dim() != 2 or block_mapping.shape[1] != 2 or block_mapping.dtype != torch.int64:
        raise RuntimeError("block_mapping must be an int64 [num_pairs, 2] tensor")
    devs = [t.device for t in (src_key, src_value, dst_key, dst_value) if t.is_cuda]
    if not devs:
        raise RuntimeError("Invalid device combination")                                 # cache_kernels.cu:39
    dev = devs[0]
    for name, t in (("src_key", src_key), ("src_value", src_value), ("dst_key", dst_key), ("dst_value", dst_value)):
        if t.is_cuda:
            if t.device != dev:
                raise RuntimeError("src and dst must be on the same GPU")                # :30-31
        elif not t.is_pinned():
            raise RuntimeError(f"swap_blocks_batched: {name} is host memory that is not pinned (the GPU moves the blocks itself)")
        if not t.is_contiguous():
            raise RuntimeError(f"swap_blocks_batched: {name} must be contiguous")
    if src_key.is_cuda != src_value.is_cuda or dst_key.is_cuda != dst_value.is_cuda:
        raise RuntimeError("swap_blocks_batched: the key and value cache of a side must live in the same memory")
    block_bytes = src_key.element_size() * src_key[0].numel()
    for t in (src_value, dst_key, dst_value):
        if t.element_size() * t[0].numel() != block_bytes:
            raise RuntimeError("swap_blocks_batched: all four caches must have the same block size in bytes")
    n = int(block_mapping.shape[0])
    if n == 0:
        return None
    bm_host = block_mapping if not block_mapping.is_cuda else None
    if bm_host is not None:
        if int(bm_host[:, 0].max()) >= min(src_key.shape[0], src_value.shape[0]) or \
                int(bm_host[:, 1].max()) >= min(dst_key.shape[0], dst_value.shape[0]) or int(bm_host.min()) < 0:
            raise RuntimeError("swap_blocks_batched: block number out of range")
    bm = block_mapping.contiguous().to(dev)
    rc = _lib.load().vmi_swap_blocks_batched(src_key.data_ptr(), src_value.data_ptr(), dst_key.data_ptr(), dst_value.data_ptr(),
                                             bm.data_ptr(), n, block_bytes,
                                             dev.index if dev.index is not None else torch.cuda.current_device(),
                                             torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        _raise_native(rc)
    return None


def reshape_and_cache_flash(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                            slot_mapping: torch.Tensor, kv_cache_dtype: str) -> None:
    """Scatter rows into flash-layout caches [num_blocks, block_size, num_heads, head_size].
    Reference: cache_kernels.cu:283-317.  Pure copy; fp16 or bf16."""
    if kv_cache_dtype != "auto":
        raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")          # :291-293
    if key.dim() != 3 or key.shape != value.shape or key.dtype != value.dtype or key.element_size() != 2:
        raise RuntimeError("key/value must be [num_tokens, num_heads, head_size] tensors of one 2-byte dtype")
    dev = key.device
    for name, t in (("key", key), ("value", value), ("k_cache", k_cache), ("v_cache", v_cache),
                    ("slot_mapping", slot_mapping)):
        _check_device(name, t, dev)
    num_tokens, num_heads, head_size = (int(s) for s in key.shape)
    if k_cache.dim() != 4 or tuple(k_cache.shape[2:]) != (num_heads, head_size) or k_cache.shape != v_cache.shape \
            or k_cache.dtype != key.dtype or v_cache.dtype != key.dtype:
        raise RuntimeError("k_cache/v_cache must be [num_blocks, block_size, num_heads, head_size] of key's dtype")
    if k_cache.stride(0) != v_cache.stride(0):
        raise RuntimeError("k_cache and v_cache must have the same block stride")           # :302
    if not k_cache[0].is_contiguous() or not v_cache[0].is_contiguous():
        raise RuntimeError("cache blocks must be dense")
    if key.stride(2) != 1 or key.stride(1) != head_size or value.stride(2) != 1 or value.stride(1) != head_size:
        raise RuntimeError("key/value must be contiguous in their last two dimensions")
    if slot_mapping.dtype != torch.int64 or slot_mapping.numel() != num_tokens or not slot_mapping.is_contiguous():
        raise RuntimeError("slot_mapping must be a contiguous int64 [num_tokens] tensor")
    rc = _lib.require_extras("reshape_and_cache_flash").vmi_reshape_and_cache_flash_16(
        key.data_ptr(), value.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), slot_mapping.data_ptr(),
        num_tokens, num_heads, head_size, int(k_cache.shape[1]), int(k_cache.stride(0)), int(key.stride(0)),
        int(value.stride(0)), dev.index if dev.index is not None else torch.cuda.current_device(),
        torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        _raise_native(rc)
    return None


def convert_fp8(dst_cache: torch.Tensor, src_cache: torch.Tensor, kv_scale: float = 1.0,
                kv_cache_dtype: str = "auto") -> None:
    """dst_cache = scaled_convert(src_cache, kv_scale), elementwise (cache_kernels.cu:320-392): a float / half /
    bfloat16 cache to fp8 E4M3 bytes (fp8(x / kv_scale), RNE, saturating) or back (float(fp8) * kv_scale, rounded to
    dst's type).  kv_cache_dtype "fp8" / "fp8_e4m3"; the reference's "auto" branch selects a conversion that does not
    exist (kAuto, quant_utils.cuh:512-525) and every other name is its "Unsupported data type"."""
    if kv_cache_dtype not in ("fp8", "fp8_e4m3"):
        raise RuntimeError(f"Unsupported data type: {kv_cache_dtype}")                    # :389-391
    if not src_cache.is_cuda:
        raise RuntimeError("src must be on a GPU")                                        # :345
    if not dst_cache.is_cuda:
        raise RuntimeError("dst must be on a GPU")                                        # :346
    if src_cache.device != dst_cache.device:
        raise RuntimeError("src and dst must be on the same GPU")                         # :347-348
    kinds = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}
    fp8_types = (torch.uint8, torch.float8_e4m3fn)
    if src_cache.dtype in kinds and dst_cache.dtype in fp8_types:
        kind, to_fp8 = kinds[src_cache.dtype], 1
    elif src_cache.dtype in fp8_types and dst_cache.dtype in kinds:
        kind, to_fp8 = kinds[dst_cache.dtype], 0
    else:
        raise RuntimeError(f"convert_fp8: unsupported dtype pair {src_cache.dtype} -> {dst_cache.dtype}")
    if src_cache.numel() != dst_cache.numel() or not src_cache.is_contiguous() or not dst_cache.is_contiguous():
        raise RuntimeError("convert_fp8: src and dst must be contiguous and hold the same number of elements")
    dev = src_cache.device
    rc = _lib.require_extras("convert_fp8").vmi_convert_fp8(
        dst_cache.data_ptr(), src_cache.data_ptr(), int(src_cache.numel()), float(kv_scale), kind, to_fp8,
        dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        _raise_native(rc)
    return None
