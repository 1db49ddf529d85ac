// pa_extras_abi.hip — the C-ABI entries of the OUT-OF-SCOPE corners of the reference's dispatch (SURVEY.md §2 rows 8-10:
// bfloat16 tensors, fp8-E5M2 pages, block-sparse attention), declared in include/vmi_paged_attention_extras.h.  Linked into
// libvmi_paged_attention_extras.so (and the diagnostic library) only: the product library neither declares nor exports
// them (tests/test_abi.py checks each library's dynamic symbol table against its own header).  They are thin wrappers over
// the launchers of paged_attention.hip (pa_host.hpp), whose kernel menus for these cases are filled by the extras units.
#include "vmi_paged_attention_extras.h"
#include "pa_host.hpp"

extern "C" {

int vmi_paged_attention_v1_blocksparse(void* out, const void* query, const void* key_cache,
                                       const void* value_cache, int32_t num_seqs, int32_t num_heads,
                                       int32_t head_size, int32_t num_kv_heads, float scale,
                                       const int32_t* block_tables, const int32_t* seq_lens,
                                       int32_t block_size, int32_t max_seq_len,
                                       int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                                       int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                       int32_t device, void* stream, int32_t is_bf16, int32_t tp_rank,
                                       int32_t blocksparse_local_blocks, int32_t blocksparse_vert_stride,
                                       int32_t blocksparse_block_size, int32_t blocksparse_head_sliding_step) {
  const int32_t bsp[5] = {tp_rank, blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                          blocksparse_head_sliding_step};
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, 0, is_bf16 != 0, false, nullptr, nullptr, 0, 0, false,
                           1.0f, bsp);
}

int vmi_paged_attention_v2_blocksparse(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                                       const void* query, const void* key_cache, const void* value_cache,
                                       int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                       int32_t num_kv_heads, float scale, const int32_t* block_tables,
                                       const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
                                       int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                                       int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                       int32_t device, void* stream, int32_t is_bf16, int32_t tp_rank,
                                       int32_t blocksparse_local_blocks, int32_t blocksparse_vert_stride,
                                       int32_t blocksparse_block_size, int32_t blocksparse_head_sliding_step) {
  const int32_t bsp[5] = {tp_rank, blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                          blocksparse_head_sliding_step};
  return vmi::launch_pa_v2(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
                           head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride, kv_head_stride,
                           device, stream, 0, is_bf16 != 0, false, 1.0f, bsp);
}

int vmi_paged_attention_v1_bf16(void* out, const void* query, const void* key_cache,
                                const void* value_cache, int32_t num_seqs, int32_t num_heads,
                                int32_t head_size, int32_t num_kv_heads, float scale,
                                const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
                                int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                                const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                                int64_t kv_head_stride, int32_t device, void* stream, int32_t variant) {
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant, true);
}

int vmi_paged_attention_v1_append_bf16(void* out, const void* query, void* key_cache, void* value_cache,
                                       int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                       int32_t num_kv_heads, float scale, const int32_t* block_tables,
                                       const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
                                       int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                                       int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                       int32_t device, void* stream, const void* key, const void* value,
                                       int64_t key_stride, int64_t value_stride, int32_t variant) {
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant, true, true, key, value, key_stride,
                           value_stride);
}

int vmi_paged_attention_v1_fp8_bf16(void* out, const void* query, const void* key_cache, const void* value_cache,
                                    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
                                    float scale, const int32_t* block_tables, const int32_t* seq_lens,
                                    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                                    const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                                    int64_t kv_head_stride, int32_t device, void* stream, float kv_scale,
                                    int32_t variant) {
  if (!(kv_scale > 0.f))
    return vmi::fail(VMI_E_SHAPE, "paged_attention_v1 (fp8 cache): kv_scale must be positive, got %g", (double)kv_scale);
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant, true, false, nullptr, nullptr, 0, 0,
                           true, kv_scale);
}

int vmi_paged_attention_v1_fp8_e5m2(void* out, const void* query, const void* key_cache, const void* value_cache,
                                    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
                                    float scale, const int32_t* block_tables, const int32_t* seq_lens,
                                    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                                    const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                                    int64_t kv_head_stride, int32_t device, void* stream, float kv_scale,
                                    int32_t variant, int32_t is_bf16) {
  if (!(kv_scale > 0.f))
    return vmi::fail(VMI_E_SHAPE, "paged_attention_v1 (fp8 cache): kv_scale must be positive, got %g", (double)kv_scale);
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant, is_bf16 != 0, false, nullptr, nullptr, 0, 0,
                           2, kv_scale);
}

int vmi_paged_attention_v2_fp8_e5m2(void* out, void* exp_sums, void* max_logits, void* tmp_out, const void* query,
                                    const void* key_cache, const void* value_cache, int32_t num_seqs,
                                    int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
                                    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
                                    int32_t max_seq_len, int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                                    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                    int32_t device, void* stream, float kv_scale, int32_t variant) {
  if (!(kv_scale > 0.f))
    return vmi::fail(VMI_E_SHAPE, "paged_attention_v2 (fp8 cache): kv_scale must be positive, got %g", (double)kv_scale);
  return vmi::launch_pa_v2(out, static_cast<float*>(exp_sums), static_cast<float*>(max_logits), tmp_out, query,
                           key_cache, value_cache, num_seqs, num_heads, head_size, num_kv_heads, scale,
                           block_tables, seq_lens, block_size, max_seq_len, max_num_blocks_per_seq, alibi_slopes,
                           q_stride, kv_block_stride, kv_head_stride, device, stream, variant, false, 2,
                           kv_scale);
}

int vmi_paged_attention_v2_fp8_bf16(void* out, void* exp_sums, void* max_logits, void* tmp_out, const void* query,
                                    const void* key_cache, const void* value_cache, int32_t num_seqs,
                                    int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
                                    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
                                    int32_t max_seq_len, int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                                    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                    int32_t device, void* stream, float kv_scale, int32_t variant, int32_t is_e5m2) {
  if (!(kv_scale > 0.f))
    return vmi::fail(VMI_E_SHAPE, "paged_attention_v2 (fp8 cache): kv_scale must be positive, got %g", (double)kv_scale);
  return vmi::launch_pa_v2(out, static_cast<float*>(exp_sums), static_cast<float*>(max_logits), tmp_out, query,
                           key_cache, value_cache, num_seqs, num_heads, head_size, num_kv_heads, scale,
                           block_tables, seq_lens, block_size, max_seq_len, max_num_blocks_per_seq, alibi_slopes,
                           q_stride, kv_block_stride, kv_head_stride, device, stream, variant, true, is_e5m2 ? 2 : 1,
                           kv_scale);
}

int vmi_paged_attention_v1_pick_variant_fp8_e5m2(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                                 int32_t block_size, int32_t max_seq_len, int32_t mean_seq_len,
                                                 int32_t is_bf16) {
  if (!vmi::head_size_supported(head_size) || (block_size != 16 && block_size != 32)) return 0;
  return vmi::pick_variant_fp8(num_seqs, num_heads, head_size, block_size, max_seq_len, mean_seq_len, is_bf16 != 0, 2);
}

int vmi_paged_attention_v1_pick_variant_fp8_bf16(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                                 int32_t block_size, int32_t max_seq_len, int32_t mean_seq_len) {
  if (!vmi::head_size_supported(head_size) || (block_size != 16 && block_size != 32)) return 0;
  return vmi::pick_variant_fp8(num_seqs, num_heads, head_size, block_size, max_seq_len, mean_seq_len, true);
}

int vmi_paged_attention_v2_bf16(void* out, void* exp_sums, void* max_logits, void* tmp_out,
                                const void* query, const void* key_cache, const void* value_cache,
                                int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                int32_t num_kv_heads, float scale, const int32_t* block_tables,
                                const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
                                int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                                int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                int32_t device, void* stream, int32_t variant) {
  return vmi::launch_pa_v2(out, static_cast<float*>(exp_sums), static_cast<float*>(max_logits), tmp_out,
                           query, key_cache, value_cache, num_seqs, num_heads, head_size, num_kv_heads,
                           scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant, true);
}

int vmi_reshape_and_cache_fp8_bf16(const void* key, const void* value, void* key_cache, void* value_cache,
                                   const int64_t* slot_mapping, int32_t num_tokens, int32_t num_heads,
                                   int32_t head_size, int32_t block_size, int32_t x, int64_t key_stride,
                                   int64_t value_stride, float kv_scale, int32_t device, void* stream) {
  return vmi::reshape_and_cache_fp8_impl(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads, head_size,
                                    block_size, x, key_stride, value_stride, kv_scale, device, stream, true);
}

int vmi_reshape_and_cache_fp8_e5m2(const void* key, const void* value, void* key_cache, void* value_cache,
                                   const int64_t* slot_mapping, int32_t num_tokens, int32_t num_heads,
                                   int32_t head_size, int32_t block_size, int32_t x, int64_t key_stride,
                                   int64_t value_stride, float kv_scale, int32_t device, void* stream, int32_t is_bf16) {
  return vmi::reshape_and_cache_fp8_impl(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads, head_size,
                                    block_size, x, key_stride, value_stride, kv_scale, device, stream, is_bf16 != 0, true);
}

}  // extern "C"
