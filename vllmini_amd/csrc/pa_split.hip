// pa_split.hip — instantiations of the split kernels (pa_split.hpp): one (sequence, head) over several workgroups of one
// launch, meeting in a caller-owned workspace.  Block size 16, fp16 query and pages.
// Names: [fp8_]d<head>[_gq4]_x<waves per item>_u<blocks per register group>_nt<non-temporal page loads>.
// The waves per item are a launch parameter, so the rows of one (head, u, nt) share a kernel.  A workgroup holds the logits
// and probabilities of ITS waves' blocks only (6 bytes per token / workgroups per item), so max_seq_len is bounded by
// 27 000 tokens x workgroups per item — where the plain kernels stop at ~27 000 and fall back to one wave per head.
// Every row has a twin (Variant::fn_rounds) that serves more items than fit the chip at once in rounds.
#include "pa_split.hpp"

namespace vmi {

#define VMI_ROW_X(D, X, U, NT, VA)                                                                                  \
  {"d" #D "_x" #X "_u" #U "_nt" #NT, D, 16, 1, 4, U, (bool)(NT), 1, false, (pa_kernel_t)pa_split_kernel<D, U, (bool)(NT), VA>, \
   0, 0, 0, 0, false, false, false, false, false, false, X, (pa_kernel_t)pa_split_kernel<D, U, (bool)(NT), VA, 0, 1, true>},
#define VMI_ROWS_X(D, U, NT, VA) \
  VMI_ROW_X(D, 8, U, NT, VA) VMI_ROW_X(D, 16, U, NT, VA) VMI_ROW_X(D, 32, U, NT, VA) VMI_ROW_X(D, 64, U, NT, VA) VMI_ROW_X(D, 128, U, NT, VA) VMI_ROW_X(D, 256, U, NT, VA)

#define VMI_ROW_X8(D, X, U, NT, VA) /* fp8 E4M3 pages */                                                             \
  {"fp8_d" #D "_x" #X "_u" #U "_nt" #NT, D, 16, 1, 4, U, (bool)(NT), 1, false,                                       \
   (pa_kernel_t)pa_split_kernel<D, U, (bool)(NT), VA, 1>, 0, 0, 0, 1, false, false, false, false, false, false, X,    \
   (pa_kernel_t)pa_split_kernel<D, U, (bool)(NT), VA, 1, 1, true>},
#define VMI_ROWS_X8(D, U, NT, VA) \
  VMI_ROW_X8(D, 8, U, NT, VA) VMI_ROW_X8(D, 16, U, NT, VA) VMI_ROW_X8(D, 32, U, NT, VA) VMI_ROW_X8(D, 64, U, NT, VA) VMI_ROW_X8(D, 128, U, NT, VA)

#define VMI_ROW_XG(D, X, U, NT, VA) /* grouped-query: four query heads of one KV head per item, every tile loaded once */ \
  {"d" #D "_gq4_x" #X "_u" #U "_nt" #NT, D, 16, 1, 4, U, (bool)(NT), 4, false,                                       \
   (pa_kernel_t)pa_split_kernel<D, U, (bool)(NT), VA, 0, 4>, 0, 0, 0, 0, true, false, false, false, false, false, X,  \
   (pa_kernel_t)pa_split_kernel<D, U, (bool)(NT), VA, 0, 4, true>},
#define VMI_ROWS_XG(D, U, NT, VA) VMI_ROW_XG(D, 8, U, NT, VA) VMI_ROW_XG(D, 16, U, NT, VA) VMI_ROW_XG(D, 32, U, NT, VA) VMI_ROW_XG(D, 64, U, NT, VA)

#define VMI_ROW_XG8(D, X, U, NT, VA) /* ... over fp8 E4M3 pages */                                                    \
  {"fp8_d" #D "_gq4_x" #X "_u" #U "_nt" #NT, D, 16, 1, 4, U, (bool)(NT), 4, false,                                   \
   (pa_kernel_t)pa_split_kernel<D, U, (bool)(NT), VA, 1, 4>, 0, 0, 0, 1, true, false, false, false, false, false, X,  \
   (pa_kernel_t)pa_split_kernel<D, U, (bool)(NT), VA, 1, 4, true>},
#define VMI_ROWS_XG8(D, U, NT, VA) VMI_ROW_XG8(D, 8, U, NT, VA) VMI_ROW_XG8(D, 16, U, NT, VA) VMI_ROW_XG8(D, 32, U, NT, VA) VMI_ROW_XG8(D, 64, U, NT, VA)

Variant g_split_variants[] = {
    VMI_ROWS_X(64, 2, 0, 2) VMI_ROWS_X(64, 2, 1, 1)
    VMI_ROWS_X(128, 2, 0, 2) VMI_ROWS_X(128, 2, 1, 1)
    // grouped-query attention (num_heads / num_kv_heads a multiple of 4)
    VMI_ROWS_XG(64, 2, 0, 2) VMI_ROWS_XG(64, 2, 1, 1) VMI_ROWS_XG(128, 1, 0, 2) VMI_ROWS_XG(128, 1, 1, 1)
#ifdef VMI_DIAG   // (diagnostic library: the one-block-per-group forms the split sweeps compared against; no pick rule returns them)
    VMI_ROWS_X(64, 1, 0, 2) VMI_ROWS_X(64, 1, 1, 1) VMI_ROWS_X(128, 1, 0, 2) VMI_ROWS_X(128, 1, 1, 1) VMI_ROWS_XG(128, 2, 0, 2)
#endif
    // fp8 E4M3 pages (any kv_scale): half the bytes per tile, so two / four blocks per register group
    VMI_ROWS_X8(64, 2, 0, 2) VMI_ROWS_X8(64, 4, 0, 2) VMI_ROWS_X8(64, 2, 1, 1) VMI_ROWS_X8(128, 2, 0, 2) VMI_ROWS_X8(128, 2, 1, 1)
    // ... and grouped-query heads over fp8 pages: the tile decoded once for its four query heads
    VMI_ROWS_XG8(64, 2, 0, 2) VMI_ROWS_XG8(64, 2, 1, 1) VMI_ROWS_XG8(128, 2, 0, 2) VMI_ROWS_XG8(128, 2, 1, 1)
};
const int g_split_nvariants = (int)(sizeof(g_split_variants) / sizeof(g_split_variants[0]));

}  // namespace vmi

#ifdef VMI_DIAG
#include "vmi_paged_attention_diag.h"
// include/vmi_paged_attention_diag.h: where the split kernels write ten words per wave (nullptr: nowhere)
extern "C" int vmi_diag_set_split_stamps(void* records, int32_t device) {
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return -1;
  uint64_t* ptr = static_cast<uint64_t*>(records);
  const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(vmi::g_split_stamps), &ptr, sizeof(ptr));   // (synchronous)
  (void)hipSetDevice(prev);
  return e == hipSuccess ? 0 : -(int)e;
}
#endif
