// pa_variants_sparse.hip — block-sparse paged attention (the operator with blocksparse_vert_stride > 1:
// attention_kernels.cu:209-254, 385-393, dispatch :778-787 / :939-948).  The reference's callers never enable it
// (gpt2.py:109-112 passes 0, 1, 1, 0); it is part of the operator's surface, so the drop-in carries it: for every
// head size x block size of the dispatch set, one wave and four waves per head, for paged_attention_v1 and for the
// partitions of paged_attention_v2 (fp16 here, bf16 in pa_variants_sparse_bf16.hip).  Skipped blocks are never loaded.
#include "pa_kernel.hpp"

namespace vmi {

#define VMI_SP(D, BS, U)                                                   \
  VMI_ROW_SP("sp_d" #D "_bs" #BS "_w1", D, BS, 1, U, false, false)         \
  VMI_ROW_SP("sp_d" #D "_bs" #BS "_w4", D, BS, 4, U, false, false)         \
  VMI_ROW_SP("sp_v2_d" #D "_bs" #BS "_w1", D, BS, 1, U, false, true)       \
  VMI_ROW_SP("sp_v2_d" #D "_bs" #BS "_w4", D, BS, 4, U, false, true)

Variant g_sparse_variants[] = {
#include "pa_table_sparse.inc"
};
const int g_sparse_nvariants = (int)(sizeof(g_sparse_variants) / sizeof(g_sparse_variants[0]));

}  // namespace vmi
