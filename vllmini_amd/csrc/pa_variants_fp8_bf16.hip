// pa_variants_fp8_bf16.hip — paged_attention_v1 over an fp8 E4M3 KV cache with a bfloat16 query (the reference
// dispatches bf16 x uint8 too: quant_utils.cuh:545-551; element = __float2bfloat16(float(fp8) * kv_scale), :350-359).
// One-wave and four-wave-per-head kernels for every head size x block size {16, 32}, plus four heads per workgroup
// for the sizes the reference's callers use.
#define VMI_F8_FMT 1
#define VMI_F8_PFX "fp8_"
#define VMI_F8_SYM(x) x
#include "pa_kernel.hpp"
#include "pa_variants_fp8_bf16_body.inc"

