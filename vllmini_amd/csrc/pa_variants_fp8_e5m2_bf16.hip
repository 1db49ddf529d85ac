// pa_variants_fp8_e5m2_bf16.hip — bfloat16 query over an fp8 E5M2 cache (element = __float2bfloat16(float(fp8) * kv_scale))
#define VMI_F8_FMT 2
#define VMI_F8_PFX "fp8e5m2_"
#define VMI_F8_SYM(x) x##_e5m2
#include "pa_kernel.hpp"
#include "pa_variants_fp8_bf16_body.inc"
