"""vllmini_amd — MI355X (gfx950) native paged-attention decode path.

One hot path, behind the reference's own operator surface:
    paged_attention_v1, cache_ops.reshape_and_cache      (vllmini_amd.ops / vllmini_amd.cache_ops)
over the reference KV layout  K:[NB, H, D/8, 16, 8]  V:[NB, H, D, 16]  (vllmini/kv_cache.py:13-14).

Native code: vllmini_amd/csrc/paged_attention.hip -> vllmini_amd/_C/libvmi_paged_attention.so,
C-ABI in include/vmi_paged_attention.h.  Importing this package does not load the library;
calling an operator does, and fails loudly if it is absent.
"""
from . import cache_ops  # noqa: F401
from .ops import paged_attention_v1, paged_attention_v2  # noqa: F401

__all__ = ["paged_attention_v1", "paged_attention_v2", "cache_ops"]
__version__ = "0.1.0"
