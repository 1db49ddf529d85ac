// pa_variants_fp8.hip — paged_attention_v1 instantiations for an fp8 E4M3 KV cache (pa_table_fp8.inc) and the
// quantising reshape_and_cache that fills it.  SURVEY.md section 8 row f-4.
#include "pa_kernel.hpp"

namespace vmi {

Variant g_fp8_variants_v1[] = {
#include "pa_table_fp8.inc"
};
const int g_fp8_nvariants_v1 = (int)(sizeof(g_fp8_variants_v1) / sizeof(g_fp8_variants_v1[0]));

}  // namespace vmi
