// pa_variants_fp8_bf16.hip — paged_attention_v1 over an fp8 E4M3 KV cache with a bfloat16 query (the reference
// dispatches bf16 x uint8 too: quant_utils.cuh:545-551; element = __float2bfloat16(float(fp8) * kv_scale), :350-359).
// One-wave and four-wave-per-head kernels for every head size x block size {16, 32}, plus four heads per workgroup
// for the sizes the reference's callers use.
#include "pa_kernel.hpp"

namespace vmi {

#define VMI_F8B(D, BS, HPW, WPH, U) \
  VMI_ROW_F8B("bf16_fp8_d" #D "_bs" #BS "_h" #HPW "_w" #WPH "_u" #U "_nt1", D, BS, HPW, WPH, U, 1, false, 1, 0, true)
Variant g_fp8bf_variants_v1[] = {
    VMI_F8B(64, 16, 4, 1, 2) VMI_F8B(64, 16, 1, 1, 2) VMI_F8B(64, 16, 1, 4, 2) VMI_F8B(64, 16, 1, 16, 1)
    VMI_F8B(128, 16, 4, 1, 1) VMI_F8B(128, 16, 1, 1, 1) VMI_F8B(128, 16, 1, 4, 2) VMI_F8B(128, 16, 1, 16, 1)
    VMI_F8B(64, 32, 1, 1, 1) VMI_F8B(64, 32, 1, 4, 1)
    VMI_F8B(80, 16, 1, 1, 2) VMI_F8B(80, 16, 1, 4, 2) VMI_F8B(80, 32, 1, 1, 1) VMI_F8B(80, 32, 1, 4, 1)
    VMI_F8B(96, 16, 1, 1, 2) VMI_F8B(96, 16, 1, 4, 2) VMI_F8B(96, 32, 1, 1, 1) VMI_F8B(96, 32, 1, 4, 1)
    VMI_F8B(112, 16, 1, 1, 2) VMI_F8B(112, 16, 1, 4, 2) VMI_F8B(112, 32, 1, 1, 1) VMI_F8B(112, 32, 1, 4, 1)
    VMI_F8B(128, 32, 1, 1, 1) VMI_F8B(128, 32, 1, 4, 1)
    VMI_F8B(192, 16, 1, 1, 1) VMI_F8B(192, 16, 1, 4, 1) VMI_F8B(192, 32, 1, 1, 1) VMI_F8B(192, 32, 1, 4, 1)
    VMI_F8B(256, 16, 1, 1, 1) VMI_F8B(256, 16, 1, 4, 1) VMI_F8B(256, 32, 1, 1, 1) VMI_F8B(256, 32, 1, 4, 1)
};
const int g_fp8bf_nvariants_v1 = (int)(sizeof(g_fp8bf_variants_v1) / sizeof(g_fp8bf_variants_v1[0]));

}  // namespace vmi
