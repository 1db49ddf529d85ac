// pa_variants_extra.hip — instantiations of the paged-attention kernels for the (head size, block size)
// combinations of the reference's dispatch set (attention_kernels.cu:738-766 x :789-803) that its own
// callers never use: everything except block 16 x head {64, 128}, which lives in paged_attention.hip.
// Two decompositions each (one wave per head; four waves per head), for v1 and for the v2 partitions.
#include "pa_kernel.hpp"

namespace vmi {

#define VMI_X2(D, BS, WPH, U) \
  {"v2_d" #D "_bs" #BS "_h1_w" #WPH "_u" #U "_nt1", D, BS, 1, WPH, U, true, 1, false, \
   (pa_kernel_t)pa_v1_kernel<D, 1, WPH, U, true, false, true, BS>, 0}

#define VMI_APP false
Variant g_extra_variants_v1[] = {
#include "pa_table_extra.inc"
};
const int g_extra_nvariants_v1 = (int)(sizeof(g_extra_variants_v1) / sizeof(g_extra_variants_v1[0]));

Variant g_extra_variants_v2[] = {
    VMI_X2(64, 8, 1, 8), VMI_X2(64, 8, 4, 8),
    VMI_X2(64, 32, 1, 2), VMI_X2(64, 32, 4, 2),
    VMI_X2(80, 8, 1, 4), VMI_X2(80, 8, 4, 4),
    VMI_X2(80, 16, 1, 2), VMI_X2(80, 16, 4, 2),
    VMI_X2(80, 32, 1, 1), VMI_X2(80, 32, 4, 1),
    VMI_X2(96, 8, 1, 4), VMI_X2(96, 8, 4, 4),
    VMI_X2(96, 16, 1, 2), VMI_X2(96, 16, 4, 2),
    VMI_X2(96, 32, 1, 1), VMI_X2(96, 32, 4, 1),
    VMI_X2(112, 8, 1, 4), VMI_X2(112, 8, 4, 4),
    VMI_X2(112, 16, 1, 2), VMI_X2(112, 16, 4, 2),
    VMI_X2(112, 32, 1, 1), VMI_X2(112, 32, 4, 1),
    VMI_X2(128, 8, 1, 4), VMI_X2(128, 8, 4, 4),
    VMI_X2(128, 32, 1, 1), VMI_X2(128, 32, 4, 1),
    VMI_X2(192, 8, 1, 2), VMI_X2(192, 8, 4, 2),
    VMI_X2(192, 16, 1, 1), VMI_X2(192, 16, 4, 1),
    VMI_X2(192, 32, 1, 1), VMI_X2(192, 32, 4, 1),
    VMI_X2(256, 8, 1, 2), VMI_X2(256, 8, 4, 2),
    VMI_X2(256, 16, 1, 1), VMI_X2(256, 16, 4, 1),
    VMI_X2(256, 32, 1, 1), VMI_X2(256, 32, 4, 1),
};
const int g_extra_nvariants_v2 = (int)(sizeof(g_extra_variants_v2) / sizeof(g_extra_variants_v2[0]));

pa_reduce_t extra_reduce_kernel(int head_size, bool bf16) {
  if (bf16) return nullptr;
  switch (head_size) {
    case 80: return (pa_reduce_t)pa_v2_reduce_kernel<80>;
    case 96: return (pa_reduce_t)pa_v2_reduce_kernel<96>;
    case 112: return (pa_reduce_t)pa_v2_reduce_kernel<112>;
    case 192: return (pa_reduce_t)pa_v2_reduce_kernel<192>;
    case 256: return (pa_reduce_t)pa_v2_reduce_kernel<256>;
    // block sizes 8 / 32 with the core head sizes reuse the reduce kernels of the core unit
    default: return nullptr;
  }
}

}  // namespace vmi
