// pa_variants_bf16.hip — bfloat16 instantiations of the paged-attention kernels (the reference dispatches
// on the element type: quant_utils.cuh:529-566; arithmetic dtype_bfloat16.cuh).  Block 16 x head 64/128 get
// the decompositions the heuristic uses on a full chip and for small batches; every other (head size, block
// size) of the reference's dispatch set gets one-wave and four-wave-per-head kernels, v1 and v2.
#include "pa_kernel.hpp"

namespace vmi {

#define VMI_B2(D, BS, WPH, U) \
  {"bf16_v2_d" #D "_bs" #BS "_h1_w" #WPH "_u" #U "_nt1", D, BS, 1, WPH, U, true, 1, true, \
   (pa_kernel_t)pa_v1_kernel<D, 1, WPH, U, true, false, true, BS, false, true>, 0}

#define VMI_APP false
Variant g_bf16_variants_v1[] = {
#include "pa_table_bf16.inc"
};
const int g_bf16_nvariants_v1 = (int)(sizeof(g_bf16_variants_v1) / sizeof(g_bf16_variants_v1[0]));

Variant g_bf16_variants_v2[] = {
    VMI_B2(64, 8, 1, 8), VMI_B2(64, 8, 4, 8),
    VMI_B2(64, 16, 1, 4), VMI_B2(64, 16, 4, 4),
    VMI_B2(64, 32, 1, 2), VMI_B2(64, 32, 4, 2),
    VMI_B2(80, 8, 1, 4), VMI_B2(80, 8, 4, 4),
    VMI_B2(80, 16, 1, 2), VMI_B2(80, 16, 4, 2),
    VMI_B2(80, 32, 1, 1), VMI_B2(80, 32, 4, 1),
    VMI_B2(96, 8, 1, 4), VMI_B2(96, 8, 4, 4),
    VMI_B2(96, 16, 1, 2), VMI_B2(96, 16, 4, 2),
    VMI_B2(96, 32, 1, 1), VMI_B2(96, 32, 4, 1),
    VMI_B2(112, 8, 1, 4), VMI_B2(112, 8, 4, 4),
    VMI_B2(112, 16, 1, 2), VMI_B2(112, 16, 4, 2),
    VMI_B2(112, 32, 1, 1), VMI_B2(112, 32, 4, 1),
    VMI_B2(128, 8, 1, 4), VMI_B2(128, 8, 4, 4),
    VMI_B2(128, 16, 1, 2), VMI_B2(128, 16, 4, 2),
    VMI_B2(128, 32, 1, 1), VMI_B2(128, 32, 4, 1),
    VMI_B2(192, 8, 1, 2), VMI_B2(192, 8, 4, 2),
    VMI_B2(192, 16, 1, 1), VMI_B2(192, 16, 4, 1),
    VMI_B2(192, 32, 1, 1), VMI_B2(192, 32, 4, 1),
    VMI_B2(256, 8, 1, 2), VMI_B2(256, 8, 4, 2),
    VMI_B2(256, 16, 1, 1), VMI_B2(256, 16, 4, 1),
    VMI_B2(256, 32, 1, 1), VMI_B2(256, 32, 4, 1),
};
const int g_bf16_nvariants_v2 = (int)(sizeof(g_bf16_variants_v2) / sizeof(g_bf16_variants_v2[0]));

pa_reduce_t bf16_reduce_kernel(int head_size) {
  switch (head_size) {
    case 64: return (pa_reduce_t)pa_v2_reduce_kernel<64, true>;
    case 80: return (pa_reduce_t)pa_v2_reduce_kernel<80, true>;
    case 96: return (pa_reduce_t)pa_v2_reduce_kernel<96, true>;
    case 112: return (pa_reduce_t)pa_v2_reduce_kernel<112, true>;
    case 128: return (pa_reduce_t)pa_v2_reduce_kernel<128, true>;
    case 192: return (pa_reduce_t)pa_v2_reduce_kernel<192, true>;
    case 256: return (pa_reduce_t)pa_v2_reduce_kernel<256, true>;
    default: return nullptr;
  }
}

}  // namespace vmi
