// pa_variants_fp8_e5m2.hip — the fp8-cache kernels of pa_variants_fp8.hip over fp8 E5M2 bytes (kv_cache_dtype
// "fp8_e5m2", __NV_E5M2 in the reference: quant_utils.cuh:552-558).  An E5M2 byte is the upper byte of an IEEE half.
#define VMI_F8_FMT 2
#define VMI_F8_PFX "fp8e5m2_"
#define VMI_F8_SYM(x) x##_e5m2
#include "pa_kernel.hpp"
#include "pa_variants_fp8_body.inc"
