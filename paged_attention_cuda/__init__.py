"""Drop-in replacement for the reference's `paged_attention_cuda` extension package.

The reference stack does `from paged_attention_cuda import paged_attention_v1, cache_ops`
(vllmini/model/gpt2.py:5, vllmini/tests/kernels/paged_attention.py:4; package definition
paged_attention_ext/paged_attention_cuda/__init__.py:1-8).  With this directory's parent on
sys.path that import resolves here and runs the MI355X HIP kernels instead.
"""
from vllmini_amd import cache_ops, paged_attention_v1, paged_attention_v2

__all__ = ["paged_attention_v1", "paged_attention_v2", "cache_ops"]
