// pa_extras_absent.hip — what the PRODUCT library (libvmi_paged_attention.so) links in place of the out-of-scope units
// (bfloat16 / float32 tensors, E5M2 pages, block-sparse attention, reshape_and_cache_flash, convert_fp8: SURVEY.md §2 rows
// 8-10; pa_variants_bf16 / _sparse* / _fp8_bf16 / _fp8_e5m2*, pa_append_bf16, pa_f32, pa_extras_cache): EMPTY kernel menus,
// so no heuristic can pick and no variant id can name a kernel that is not there.  No C-ABI entry: the out-of-scope entries are
// declared in include/vmi_paged_attention_extras.h and exist in the extras library only (pa_extras_abi.hip, pa_f32.hip,
// pa_extras_cache.hip).  The choice is made at LINK time — no #ifdef in the host code of the path.
#include "pa_kernel.hpp"
#include "pa_host.hpp"
#include "pa_cache_fp8.hpp"

namespace vmi {

const bool g_has_extras = false;

#define VMI_EMPTY_MENU(TAB, N) \
  Variant TAB[1] = {};         \
  const int N = 0;
VMI_EMPTY_MENU(g_sparse_variants, g_sparse_nvariants)
VMI_EMPTY_MENU(g_sparse_bf16_variants, g_sparse_bf16_nvariants)
VMI_EMPTY_MENU(g_bf16_variants_v1, g_bf16_nvariants_v1)
VMI_EMPTY_MENU(g_bf16_variants_v2, g_bf16_nvariants_v2)
VMI_EMPTY_MENU(g_app_bf16_variants, g_app_bf16_nvariants)
VMI_EMPTY_MENU(g_fp8bf_variants_v1, g_fp8bf_nvariants_v1)
VMI_EMPTY_MENU(g_fp8bf_variants_v2, g_fp8bf_nvariants_v2)
VMI_EMPTY_MENU(g_fp8_variants_v1_e5m2, g_fp8_nvariants_v1_e5m2)
VMI_EMPTY_MENU(g_fp8_variants_v2_e5m2, g_fp8_nvariants_v2_e5m2)
VMI_EMPTY_MENU(g_fp8bf_variants_v1_e5m2, g_fp8bf_nvariants_v1_e5m2)
VMI_EMPTY_MENU(g_fp8bf_variants_v2_e5m2, g_fp8bf_nvariants_v2_e5m2)

pa_reduce_t bf16_reduce_kernel(int) { return nullptr; }
fp8_scatter_fn fp8_scatter_extra_kernel(bool, bool, bool) { return nullptr; }

}  // namespace vmi
