// pa_variants_fp8.hip — paged_attention_v1 instantiations for an fp8 E4M3 KV cache (pa_table_fp8.inc) and the
// quantising reshape_and_cache that fills it.  SURVEY.md section 8 row f-4.
#include "pa_kernel.hpp"

namespace vmi {

Variant g_fp8_variants_v1[] = {
#include "pa_table_fp8.inc"
};
const int g_fp8_nvariants_v1 = (int)(sizeof(g_fp8_variants_v1) / sizeof(g_fp8_variants_v1[0]));

// split-KV partitions over an fp8 cache (paged_attention_v2): same body, PART = true
#define VMI_F8V2(D, BS, HPW, WPH, U)                                                                                \
  {"fp8_v2_d" #D "_bs" #BS "_h" #HPW "_w" #WPH "_u" #U "_nt1", D, BS, HPW, WPH, U, true, 1, false,                    \
   (pa_kernel_t)pa_v1_kernel<D, HPW, WPH, U, true, false, true, BS, false, false, 1, false, 0, true>, 0, 0, 0, true}
Variant g_fp8_variants_v2[] = {
    VMI_F8V2(64, 16, 4, 1, 2), VMI_F8V2(64, 16, 1, 1, 2), VMI_F8V2(64, 16, 1, 4, 2), VMI_F8V2(64, 32, 1, 1, 1), VMI_F8V2(64, 32, 1, 4, 1),
    VMI_F8V2(80, 16, 1, 1, 2), VMI_F8V2(80, 16, 1, 4, 2), VMI_F8V2(80, 32, 1, 1, 1), VMI_F8V2(80, 32, 1, 4, 1),
    VMI_F8V2(96, 16, 1, 1, 2), VMI_F8V2(96, 16, 1, 4, 2), VMI_F8V2(96, 32, 1, 1, 1), VMI_F8V2(96, 32, 1, 4, 1),
    VMI_F8V2(112, 16, 1, 1, 2), VMI_F8V2(112, 16, 1, 4, 2), VMI_F8V2(112, 32, 1, 1, 1), VMI_F8V2(112, 32, 1, 4, 1),
    VMI_F8V2(128, 16, 4, 1, 1), VMI_F8V2(128, 16, 1, 1, 1), VMI_F8V2(128, 16, 1, 4, 1), VMI_F8V2(128, 32, 1, 1, 1), VMI_F8V2(128, 32, 1, 4, 1),
    VMI_F8V2(192, 16, 1, 1, 1), VMI_F8V2(192, 16, 1, 4, 1), VMI_F8V2(192, 32, 1, 1, 1), VMI_F8V2(192, 32, 1, 4, 1),
    VMI_F8V2(256, 16, 1, 1, 1), VMI_F8V2(256, 16, 1, 4, 1), VMI_F8V2(256, 32, 1, 1, 1), VMI_F8V2(256, 32, 1, 4, 1),
};
const int g_fp8_nvariants_v2 = (int)(sizeof(g_fp8_variants_v2) / sizeof(g_fp8_variants_v2[0]));

}  // namespace vmi
