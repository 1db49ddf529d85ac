"""`cache_ops` submodule of the reference's extension (paged_attention_cuda.cpp:55-61).

  reshape_and_cache   the one op the reference's Python calls (vllmini/model/gpt2.py:81)   -> ops.py
  copy_blocks         cache_kernels.cu:96-148    multi-layer block copy (copy-on-write / forking)
  swap_blocks         cache_kernels.cu:24-63     block moves device<->device / device<->host (preemption)
  reshape_and_cache_flash, convert_fp8           exported by the reference, no caller, not built
"""
from __future__ import annotations

import ctypes
from typing import List

import torch

from . import _lib
from .ops import _check_device, _raise_native, reshape_and_cache  # noqa: F401

__all__ = ["reshape_and_cache", "reshape_and_cache_flash", "swap_blocks", "copy_blocks", "convert_fp8"]


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


def _not_built(name: str, where: str):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            f"cache_ops.{name} is outside the decode hot path (reference {where} has no Python "
            "caller) and is not built")
    fn.__name__ = name
    return fn


reshape_and_cache_flash = _not_built("reshape_and_cache_flash", "cache_kernels.cu:283-317")
convert_fp8 = _not_built("convert_fp8", "cache_kernels.cu:335-392")
