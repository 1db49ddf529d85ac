"""CPU oracle for the paged-attention decode path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package,
and only as the checker.  vllmini_amd/ (the product) never does.

  kernel_model   C restatement of the reference CUDA kernel's arithmetic and rounding points
                 (oracle/pa_kernel_model.c; attention_kernels.cu:86-496, cache_kernels.cu:152-207)
  eager          numpy/torch restatement of the eager attention the reference tests its kernel
                 against and uses where it has no kernel (vllmini/model/gpt2.py:71-78,
                 vllmini/tests/kernels/paged_attention.py:102-110)

Pin status (pa_kernel_model.c header): PINNED by fixtures generated from the imported reference Python
(tests/golden/: its eager attention, its own unittest run against this model, its scheduler's op-call trace).  The CUDA
kernel's own outputs cannot be produced in this pipeline; fidelity to its rounding points rests on the restatement.
"""
from .kernel_model import (  # noqa: F401
    bf16_bits_to_f32,
    build,
    f32_to_bf16_bits,
    f2h,
    h2f,
    paged_attention_v1,
    paged_attention_v1_f32,
    reshape_and_cache_f32,
    paged_attention_v2,
    reshape_and_cache,
    # fp8 (E4M3) KV cache — kv_cache_dtype "fp8"
    f32_to_fp8e4m3,
    f32_to_fp8e5m2,
    fp8e5m2_to_f32,
    fp8e4m3_to_f32,
    paged_attention_v1_fp8,
    paged_attention_v2_fp8,
    reshape_and_cache_fp8,
)
from .eager import (  # noqa: F401
    eager_paged_attention,
    gather_kv,
    torch_eager_decode,
)
