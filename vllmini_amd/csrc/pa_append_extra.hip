// pa_append_extra.hip — fused-append (APP = true) instantiations of the paged_attention_v1 menu for the remaining fp16 head/block sizes:
// row i here is row i of the plain table built from the same pa_table_extra.inc, so a variant id means the same work
// decomposition for vmi_paged_attention_v1_* and vmi_paged_attention_v1_append_*.  A translation unit of its own
// only so that the six units compile concurrently.
#define VMI_APP true
#include "pa_kernel.hpp"

namespace vmi {

Variant g_app_extra_variants[] = {
#include "pa_table_extra.inc"
};
const int g_app_extra_nvariants = (int)(sizeof(g_app_extra_variants) / sizeof(g_app_extra_variants[0]));

}  // namespace vmi
