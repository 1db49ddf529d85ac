// pa_variants_sparse_bf16.hip — the block-sparse kernels of pa_variants_sparse.hip for bfloat16 tensors
#include "pa_kernel.hpp"

namespace vmi {

#define VMI_SP(D, BS, U)                                                        \
  VMI_ROW_SP("sp_bf16_d" #D "_bs" #BS "_w1", D, BS, 1, U, true, false)          \
  VMI_ROW_SP("sp_bf16_d" #D "_bs" #BS "_w4", D, BS, 4, U, true, false)          \
  VMI_ROW_SP("sp_bf16_v2_d" #D "_bs" #BS "_w1", D, BS, 1, U, true, true)        \
  VMI_ROW_SP("sp_bf16_v2_d" #D "_bs" #BS "_w4", D, BS, 4, U, true, true)

Variant g_sparse_bf16_variants[] = {
#include "pa_table_sparse.inc"
};
const int g_sparse_bf16_nvariants = (int)(sizeof(g_sparse_bf16_variants) / sizeof(g_sparse_bf16_variants[0]));

}  // namespace vmi
