// pa_queue.hip — instantiations of the balanced paged_attention_v1 kernels (pa_queue.hpp), block size 16.
// Names: [bf16_|fp8_]q_d<head>_s<U of mode S>q<U of mode Q>[m]; the mode is chosen on the device; m = K pass on MFMA.
#include "pa_queue.hpp"

namespace vmi {

#define VMI_ROW_Q(NAME, D, BF, US, UQ, UT)                                                                        \
  {NAME, D, 16, 4, 1, US, true, 1, BF, (pa_kernel_t)pa_q_kernel<D, BF, true, US, UQ, 0, false, UT>, 0, 0, 0, 0, false, false, false, true},
// ... with its append-read form (vmi_paged_attention_v1_newest_f16 on a full chip: pa_queue.hpp APP)
#define VMI_ROW_QA(NAME, D, BF, US, UQ, UT)                                                                       \
  {NAME, D, 16, 4, 1, US, true, 1, BF, (pa_kernel_t)pa_q_kernel<D, BF, true, US, UQ, 0, false, UT>, 0, 0, 0, 0, false, false, false, true, \
   false, false, 0, nullptr, (pa_kernel_t)pa_q_kernel<D, BF, true, US, UQ, 0, false, UT, true>},
#define VMI_ROW_Q8(NAME, D, US, UQ, F8, UT) /* fp8 pages (1 = E4M3, 2 = E5M2), float16 query, kv_scale 1 */        \
  {NAME, D, 16, 4, 1, US, true, 1, false, (pa_kernel_t)pa_q_kernel<D, false, true, US, UQ, F8, false, UT>, 0, 0, 0, F8, false, false, false, true},
#define VMI_ROW_Q8M(NAME, D, US, UQ, F8, UT) /* ... with q.K^T of the K pass on the matrix cores (pa_queue.hpp, KM) */ \
  {NAME, D, 16, 4, 1, US, true, 1, false, (pa_kernel_t)pa_q_kernel<D, false, true, US, UQ, F8, true, UT>, 0, 0, 0, F8, false, false, false, true, false, true},

// (last argument: UT, blocks per register group of a 4-wave team — a long item's waves on a chip that is mostly idle
//  are bound by their own bytes in flight: "4 full, rest 1/32" 22.5 -> 19.4 us with 2 instead of 1 at head size 64;
//  fp8 pages 20.6 -> 17.5 -> 16.2 us with 1 / 2 / 4; head size 128 has no registers for 2 (spills: 58.7 -> 66.3 us);
//  profiles/r03c_heavy_tailed_batches.md)
Variant g_queue_variants[] = {
    // head size 64: one block per group when every item has its own wave, two when workers run items in turn
    VMI_ROW_QA("q_d64_s1q2", 64, false, 1, 2, 2)
#ifdef VMI_EXTRAS   // (bfloat16 / E5M2 rows: libvmi_paged_attention_extras.so only — this unit is compiled once for each library)
    VMI_ROW_Q("bf16_q_d64_s1q2", 64, true, 1, 2, 2)
#endif
    // head size 128: twice the registers per block -> one block per group in both modes, 2 workgroups per CU
    VMI_ROW_Q("q_d128_s1q1", 128, false, 1, 1, 1)
#ifdef VMI_EXTRAS
    VMI_ROW_Q("bf16_q_d128_s1q1", 128, true, 1, 1, 1)
#endif
    // fp8 pages, kv_scale == 1 (any other scale stays with pa_v1_kernel): a tile is half the bytes, so twice the blocks
    // per register group keep the bytes in flight where the 16-bit kernels have them
    VMI_ROW_Q8("fp8_q_d64_s2q4", 64, 2, 4, 1, 4)
    VMI_ROW_Q8("fp8_q_d64_s1q2", 64, 1, 2, 1, 2)
#ifdef VMI_EXTRAS
    VMI_ROW_Q8("fp8e5m2_q_d64_s2q4", 64, 2, 4, 2, 4)
#endif
    VMI_ROW_Q8("fp8_q_d128_s1q2", 128, 1, 2, 1, 1)
    // the same with q.K^T of the K pass on the matrix cores ("m", pa_queue.hpp KM): the default over fp8 pages — equal
    // lengths unchanged (the 1-KiB tile request pattern is the bound there), ragged batches 47.8 -> 45.1 us on cfg3
    // (mode Q runs half the waves, each with twice the VALU work), head size 128 370.8 -> 360.5 / 205.5 -> 197.5 us;
    // over fp16 pages the same change is neutral and is not built (profiles/r03b_k_pass_on_mfma.md)
    VMI_ROW_Q8M("fp8_q_d64_s2q4m", 64, 2, 4, 1, 4)
    VMI_ROW_Q8M("fp8_q_d128_s1q2m", 128, 1, 2, 1, 1)
};
const int g_queue_nvariants = (int)(sizeof(g_queue_variants) / sizeof(g_queue_variants[0]));

#ifdef VMI_DIAG
__device__ uint64_t* g_wave_timeline = nullptr;
#endif

}  // namespace vmi

#ifdef VMI_DIAG
#include "vmi_paged_attention_diag.h"
// include/vmi_paged_attention_diag.h: where the balanced kernels write one record per wave (nullptr: nowhere)
extern "C" int vmi_diag_set_wave_timeline(void* records, int32_t device) {
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return -1;
  uint64_t* ptr = static_cast<uint64_t*>(records);
  const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(vmi::g_wave_timeline), &ptr, sizeof(ptr));   // (synchronous)
  (void)hipSetDevice(prev);
  return e == hipSuccess ? 0 : -(int)e;
}
#endif
