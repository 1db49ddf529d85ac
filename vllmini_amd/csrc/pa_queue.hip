// pa_queue.hip — instantiations of the balanced paged_attention_v1 kernels (pa_queue.hpp), block size 16.
// Names: q_d<head>[_bf16]; the mode (one item per wave / work queue) is chosen on the device.
#include "pa_queue.hpp"

namespace vmi {

#define VMI_ROW_Q(NAME, D, BF, US, UQ)                                                                           \
  {NAME, D, 16, 4, 1, US, true, 1, BF, (pa_kernel_t)pa_q_kernel<D, BF, true, US, UQ>, 0, UQ, 0, 0, false, false, false, true},

Variant g_queue_variants[] = {
    VMI_ROW_Q("q_d64_s1q2", 64, false, 1, 2)
    VMI_ROW_Q("q_d64_s1q1", 64, false, 1, 1)
    VMI_ROW_Q("q_d64_s2q4", 64, false, 2, 4)
    VMI_ROW_Q("q_d64_s4q4", 64, false, 4, 4)
};
const int g_queue_nvariants = (int)(sizeof(g_queue_variants) / sizeof(g_queue_variants[0]));

}  // namespace vmi
