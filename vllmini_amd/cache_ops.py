"""`cache_ops` submodule of the reference's extension (paged_attention_cuda.cpp:55-61).

Only `reshape_and_cache` has a Python caller in the reference (vllmini/model/gpt2.py:81); it is
the one implemented.  The other four names are exported by the reference but never called
(SURVEY.md §2 #8); they exist here so `hasattr` checks behave, and raise when invoked.
"""
from __future__ import annotations

from .ops import reshape_and_cache  # noqa: F401

__all__ = ["reshape_and_cache", "reshape_and_cache_flash", "swap_blocks", "copy_blocks", "convert_fp8"]


def _not_built(name: str, where: str):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            f"cache_ops.{name} is outside the decode hot path (reference {where} has no Python "
            "caller) and is not built")
    fn.__name__ = name
    return fn


reshape_and_cache_flash = _not_built("reshape_and_cache_flash", "cache_kernels.cu:283-317")
swap_blocks = _not_built("swap_blocks", "cache_kernels.cu:24-63")
copy_blocks = _not_built("copy_blocks", "cache_kernels.cu:96-148")
convert_fp8 = _not_built("convert_fp8", "cache_kernels.cu:335-392")
