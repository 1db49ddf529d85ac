/*
 * vmi_paged_attention_extras.h — C-ABI entries that exist ONLY in libvmi_paged_attention_extras.so (`python -m
 * vllmini_amd.build --extras`; the diagnostic library is built on it): the corners of the reference's dispatch surface that
 * lie OUTSIDE the hot path of SURVEY.md §8 (§2 rows 8-10) — bfloat16 and float32 tensors, fp8-E5M2 pages, block-sparse
 * attention, reshape_and_cache_flash, convert_fp8.  The product library (libvmi_paged_attention.so) neither declares nor
 * exports any of them (tests/test_abi.py checks each library's dynamic symbol table against its own header); the Python
 * operators raise RuntimeError("... not in this build ...") for these cases unless the process opted in with
 * vllmini_amd._lib.use_extras().  Conventions, layouts and return codes: vmi_paged_attention.h.
 */
#ifndef VMI_PAGED_ATTENTION_EXTRAS_H
#define VMI_PAGED_ATTENTION_EXTRAS_H

#include "vmi_paged_attention.h"

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: the entries declared here are all it exports */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/*
 * Block-sparse attention: paged_attention_v1 / paged_attention_v2 called with blocksparse_vert_stride > 1
 * (is_block_sparse, attention_kernels.cu:822 / :987; kernel :209-254, :385-393).  fp16 or bf16 tensors
 * (is_bf16), "auto" cache.  A cache block is read when the sparse block (blocksparse_block_size tokens) holding its
 * first token is
 *   "remote": (sparse_block + offset) % blocksparse_vert_stride == 0, with
 *             offset = (tp_rank * num_heads + head) * head_sliding_step + 1             (head_sliding_step >= 0)
 *                    = (tp_rank * num_kv_heads + kv_head) * (-head_sliding_step) + 1    (head_sliding_step <  0), or
 *   "local":  sparse_block > (seq_len - 1) / blocksparse_block_size - blocksparse_local_blocks;
 * every other block is skipped: not loaded, logits -FLT_MAX, no P.V contribution.  Same results as the dense
 * operator's arithmetic restricted to the attended blocks (oracle/pa_kernel_model.c, checked against a masked fp64
 * attention).  blocksparse_vert_stride <= 1 is an error here (call the dense entry).  No tuning variants.
 * The reference's own callers never enable this (gpt2.py:109-112 passes 0, 1, 1, 0).
 */
int vmi_paged_attention_v1_blocksparse(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream, int32_t is_bf16, int32_t tp_rank,
    int32_t blocksparse_local_blocks, int32_t blocksparse_vert_stride,
    int32_t blocksparse_block_size, int32_t blocksparse_head_sliding_step);
int vmi_paged_attention_v2_blocksparse(
    void* out, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream, int32_t is_bf16, int32_t tp_rank,
    int32_t blocksparse_local_blocks, int32_t blocksparse_vert_stride,
    int32_t blocksparse_block_size, int32_t blocksparse_head_sliding_step);

/*
 * bfloat16 forms of the two attention operators (the reference dispatches on the element type,
 * quant_utils.cuh:529-566; arithmetic dtype_bfloat16.cuh).  Same arguments as the _f16 entries plus an
 * explicit variant (0 = heuristic; bf16 variant names start with "bf16_").  query/out/caches hold bfloat16.
 * reshape_and_cache / reshape_and_cache_flash / copy_blocks / swap_blocks are byte copies and serve both types.
 */
int vmi_paged_attention_v1_bf16(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream, int32_t variant);
int vmi_paged_attention_v2_bf16(
    void* out, void* exp_sums, void* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream, int32_t variant);

/* fused decode step over bfloat16 tensors: vmi_paged_attention_v1_append_f16's arguments */
int vmi_paged_attention_v1_append_bf16(
    void* out, const void* query, void* key_cache, void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream,
    const void* key, const void* value, int64_t key_stride, int64_t value_stride,
    int32_t variant);

/* bfloat16 query / key / value over the same fp8 caches (the reference dispatches bf16 x uint8 as well):
 * load __float2bfloat16(float(fp8) * kv_scale) (quant_utils.cuh:350-359), store fp8(float(bf16) / kv_scale) (:468-478). */
int vmi_paged_attention_v1_fp8_bf16(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream,
    float kv_scale, int32_t variant);
int vmi_reshape_and_cache_fp8_bf16(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping,
    int32_t num_tokens, int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
    int64_t key_stride, int64_t value_stride, float kv_scale,
    int32_t device, void* stream);
int vmi_paged_attention_v1_pick_variant_fp8_bf16(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                                 int32_t block_size, int32_t max_seq_len, int32_t mean_seq_len);

/*
 * kv_cache_dtype "fp8_e5m2" (Fp8KVCacheDataType::kFp8E5M2, __NV_E5M2: quant_utils.cuh:552-558): the same operators
 * over fp8 E5M2 bytes — an E5M2 byte is the upper byte of an IEEE half, infinities and NaNs included.  Element seen
 * by the attention arithmetic = half(float(fp8) * kv_scale) (bfloat16 query: bf16(float(fp8) * kv_scale)), cache
 * byte written by reshape_and_cache = fp8(float(x) / kv_scale), round to nearest even, saturating at +-57344
 * (__NV_SATFINITE).  Same layouts and limits as the E4M3 entries above (x = 16; block sizes 16 and 32);
 * is_bf16 selects bfloat16 query / rows.  Variant ids: the "fp8e5m2_" / "bf16_fp8e5m2_" names.
 */
int vmi_paged_attention_v1_fp8_e5m2(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream,
    float kv_scale, int32_t variant, int32_t is_bf16);
int vmi_paged_attention_v2_fp8_e5m2(   /* float16 query */
    void* out, void* exp_sums, void* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream,
    float kv_scale, int32_t variant);
/* paged_attention_v2, bfloat16 query / out / tmp_out over fp8 pages of either format (is_e5m2) */
int vmi_paged_attention_v2_fp8_bf16(
    void* out, void* exp_sums, void* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream,
    float kv_scale, int32_t variant, int32_t is_e5m2);
int vmi_reshape_and_cache_fp8_e5m2(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping,
    int32_t num_tokens, int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
    int64_t key_stride, int64_t value_stride, float kv_scale,
    int32_t device, void* stream, int32_t is_bf16);
int vmi_paged_attention_v1_pick_variant_fp8_e5m2(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                                 int32_t block_size, int32_t max_seq_len, int32_t mean_seq_len,
                                                 int32_t is_bf16);

/*
 * cache_ops.reshape_and_cache_flash — cache_kernels.cu:283-317 (kernel :209-240): scatter rows into the
 * flash layout k_cache / v_cache [num_blocks, block_size, num_heads, head_size]; any 2-byte element type.
 * block_stride = k_cache.stride(0) (must equal v_cache.stride(0), :302).
 */
int vmi_reshape_and_cache_flash_16(const void* key, const void* value, void* k_cache, void* v_cache,
                                   const int64_t* slot_mapping, int32_t num_tokens, int32_t num_heads,
                                   int32_t head_size, int32_t block_size, int64_t block_stride,
                                   int64_t key_stride, int64_t value_stride, int32_t device, void* stream);

/*
 * float32 tensors — the (float, float) branch of the reference's dispatch (quant_utils.cuh:529-535): query / out /
 * caches float32, x = 16 / sizeof(float) = 4: key_cache [NB, H, D/4, BS, 4], value_cache [NB, H, D, BS]; strides in
 * elements; every operation in fp32 (dtype_float32.cuh).  The reference's callers never use it (scheduler.py:13 runs the
 * model in half): one plain kernel per (head size, block size), no tuning variants, paged_attention_v1 and
 * reshape_and_cache only.
 */
int vmi_paged_attention_v1_f32(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream);
int vmi_reshape_and_cache_f32(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping,
    int32_t num_tokens, int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
    int64_t key_stride, int64_t value_stride,
    int32_t device, void* stream);

/*
 * cache_ops.convert_fp8(dst_cache, src_cache, kv_scale, kv_cache_dtype) — cache_kernels.cu:320-392 ("only for
 * testing" in the reference; compiled to assert(false) in its shipped build).  Elementwise over num_elements contiguous
 * elements: to_fp8 != 0: dst (uint8 E4M3) = fp8(float(src) / kv_scale), RNE, saturating; to_fp8 == 0: dst =
 * half / bfloat16 / float of (float(fp8) * kv_scale).  kind: 0 = half, 1 = bfloat16, 2 = float (the non-fp8 side).
 */
int vmi_convert_fp8(void* dst, const void* src, int64_t num_elements, float kv_scale, int32_t kind, int32_t to_fp8,
                    int32_t device, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VMI_PAGED_ATTENTION_EXTRAS_H */
