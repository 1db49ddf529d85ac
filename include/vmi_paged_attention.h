/*
 * vmi_paged_attention.h — C-ABI of the MI355X (gfx950) paged-attention decode path.
 *
 * This is the drop-in boundary for the two operators the reference's Python stack
 * calls through its `paged_attention_cuda` extension module:
 *
 *   paged_attention_v1(...)            reference binding: paged_attention_ext/paged_attention_cuda/
 *                                      paged_attention_cuda.cpp:7-25, :52 ; host entry
 *                                      attention_kernels.cu:805-826 ; launcher :690-767
 *   cache_ops.reshape_and_cache(...)   reference binding: paged_attention_cuda.cpp:58 ,
 *                                      cache_kernels.h:11-14 ; host entry cache_kernels.cu:256-281
 *
 * The reference host entries take torch::Tensor and read sizes/strides from them
 * (attention_kernels.cu:701-707, cache_kernels.cu:265-272).  Here every one of
 * those values is an explicit argument: plain device pointers, sizes and element
 * strides — no torch / pybind types.  The Python mirror (vllmini_amd/ops.py, re-exported
 * under the reference's import name `paged_attention_cuda`) extracts them from the
 * tensors exactly where the reference launcher does.
 *
 * Conventions
 *   - all pointers are DEVICE pointers on HIP device `device`;
 *   - strides are in ELEMENTS (fp16 halves), as `tensor.stride(i)` reports them;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - calls are asynchronous: they enqueue kernels on `stream` and never synchronise;
 *   - nothing is retained after return (no hidden workspace, no cached pointers);
 *   - return value: 0 = launched; >0 = VMI_E_* validation code; <0 = -(hipError_t).
 *     `vmi_last_error_string()` describes the last non-zero return on the calling thread.
 *
 * KV-cache layout (reference: vllmini/kv_cache.py:13-14, cache_kernels.cu:187-194):
 *   key_cache   [num_blocks, num_kv_heads, head_size/x, block_size, x]   x = 8 halves (16 B)
 *   value_cache [num_blocks, num_kv_heads, head_size,   block_size]
 *   kv_block_stride = key_cache.stride(0), kv_head_stride = key_cache.stride(1) are applied
 *   to BOTH caches, as the reference launcher does (attention_kernels.cu:706-707, 273-275, 402-403).
 */
#ifndef VMI_PAGED_ATTENTION_H
#define VMI_PAGED_ATTENTION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: the entries declared here are all it exports */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define VMI_ABI_VERSION 22

/* validation codes (positive); HIP runtime errors are returned negated */
enum {
  VMI_OK = 0,
  VMI_E_NULL_POINTER = 1,        /* a required pointer is NULL                                        */
  VMI_E_HEAD_SIZE = 2,           /* "Unsupported head size"   (ref: attention_kernels.cu:763-765)     */
  VMI_E_BLOCK_SIZE = 3,          /* "Unsupported block size"  (ref: attention_kernels.cu:800-802)     */
  VMI_E_KV_HEADS = 4,            /* num_heads % num_kv_heads != 0 or non-positive                     */
  VMI_E_ALIGNMENT = 5,           /* pointer/stride not 16-byte aligned where the kernel needs it      */
  VMI_E_SHAPE = 6,               /* negative or inconsistent size                                     */
  VMI_E_MAX_SEQ_LEN = 7,         /* logits for max_seq_len do not fit in the 160 KiB LDS of one CU    */
  VMI_E_VARIANT = 8,             /* unknown tuning variant id                                         */
  VMI_E_X = 9,                   /* key_cache innermost dimension is not 8 halves                     */
  VMI_E_WORKSPACE = 11,          /* a split kernel was asked for by id without a (large enough, 16-byte aligned) workspace */
  VMI_E_NOT_BUILT = 10           /* internal guard: a kernel menu of this library is empty for the case asked; no entry of
                                    THIS header can return it (ABI 20: the out-of-scope entries left for
                                    vmi_paged_attention_extras.h and are not exported by the product library) */
};

/* Library identity / diagnostics. */
int vmi_abi_version(void);
const char* vmi_last_error_string(void);
/* Compiled-for architecture string, e.g. "gfx950". */
const char* vmi_target_arch(void);

/*
 * paged_attention_v1, fp16 query/out, fp16 ("auto") KV cache.
 *
 * Replaces: paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale,
 *           block_tables, seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype,
 *           kv_scale, tp_rank, blocksparse_*)          — attention_kernels.cu:805-826.
 * kv_cache_dtype / kv_scale select the _fp8 entries below; tp_rank and the blocksparse arguments only matter with
 * blocksparse_vert_stride > 1, which is out of this path's scope (vmi_paged_attention_extras.h; the reference callers pass
 * "auto" and block-sparse disabled; vllmini/model/gpt2.py:94-113).
 *
 *   out            [num_seqs, num_heads, head_size] fp16, contiguous          (written)
 *   query          [num_seqs, num_heads, head_size] fp16, row stride q_stride (may be 3*hidden)
 *   block_tables   [num_seqs, max_num_blocks_per_seq] int32; entries >= ceil(seq_len/block_size)
 *                  are never read (the reference pads them with -1)
 *   seq_lens       [num_seqs] int32; 0 => that sequence's output rows are zero
 *   max_seq_len    upper bound on seq_lens[] (sizes the per-workgroup logits buffer in LDS,
 *                  like the reference's dynamic shared memory, attention_kernels.cu:725-732)
 *   alibi_slopes   [num_heads] fp32 or NULL
 *
 * Supported: the reference's dispatch set — head_size in {64, 80, 96, 112, 128, 192, 256}
 * (attention_kernels.cu:738-766) x block_size in {8, 16, 32} (:789-803); x == 8.
 */
int vmi_paged_attention_v1_f16(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream);

/*
 * Same operator with an explicit tuning variant (work decomposition only — results are
 * identical across variants up to fp32 summation order).  variant = 0 selects the
 * built-in heuristic, i.e. exactly what vmi_paged_attention_v1_f16 runs.
 * Used by bench.py / tests to sweep decompositions; not part of the reference surface.
 */
int vmi_paged_attention_v1_f16_variant(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream, int32_t variant);

/*
 * The same operator with a CALLER-OWNED WORKSPACE (ABI 21).  The reference's launcher gives every (head, sequence) one
 * workgroup (attention_kernels.cu:734-735); with fewer such items than the chip has CUs — the batch sizes the reference's
 * scheduler runs (vllmini/scheduler.py:60) — most of the chip idles, and the reference's own remedy is the two-kernel
 * paged_attention_v2 (:529-669, 828-990).  With a workspace, an under-filled launch spreads each item over several
 * workgroups of ONE launch: they exchange (max, sum of exp) through the workspace between the K and the V pass, so every
 * probability is normalised with the item's global values and rounded to fp16 exactly where the reference rounds it
 * (:334-346, 398-400), and the item's last workgroup adds the fp32 partial rows in a fixed order.
 *
 *   workspace        device memory of >= vmi_paged_attention_v1_workspace_bytes(...) bytes, 16-byte aligned, whose control
 *                    words are ZERO (vmi_paged_attention_v1_workspace_reset, or any memset) before its first use; a launch
 *                    leaves them zero again.  One workspace serves one stream at a time (launches on one stream are
 *                    ordered; concurrent launches need a workspace each).  Nothing is retained: the pointer is used by the
 *                    kernels of this call only.
 *   NULL / too small with variant 0: exactly vmi_paged_attention_v1_f16 (bit-identical — the same kernels).
 *   variant          0 = heuristic (may pick a split kernel, names "d<head>_x<waves>_u<U>_nt<NT>"), else as _f16_variant;
 *                    a split kernel asked for by id without a workspace is VMI_E_WORKSPACE.
 * A launch that was killed in flight (device reset, aborted process) can leave control words non-zero: reset the workspace
 * before using it again.  Word 0 of the workspace counts polls that gave up (bounded spins; stays 0 on a healthy launch).
 */
int vmi_paged_attention_v1_f16_ws(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream, void* workspace, int64_t workspace_bytes, int32_t variant);
/* Bytes a workspace must have for these sizes (0: no kernel uses one for this head size). */
int64_t vmi_paged_attention_v1_workspace_bytes(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                               int32_t max_seq_len);
/* Zero the workspace's control words on `stream` (hipMemsetAsync): once after allocation, and after a launch that died. */
int vmi_paged_attention_v1_workspace_reset(void* workspace, int64_t workspace_bytes, int32_t device, void* stream);
/* What variant 0 of the _ws entry runs when a workspace is at hand (fp16 pages; num_kv_heads <= 0 = num_heads). */
int vmi_paged_attention_v1_pick_variant_ws(int32_t num_seqs, int32_t num_heads, int32_t num_kv_heads, int32_t head_size,
                                           int32_t block_size, int32_t max_seq_len);

/*
 * Fused decode step: cache_ops.reshape_and_cache + paged_attention_v1 in ONE launch (extension; the reference
 * issues the two ops back to back for every layer, vllmini/model/gpt2.py:87-112).
 *
 * key/value [num_seqs, num_kv_heads, head_size] (row strides key_stride/value_stride, elements) hold this step's
 * token of every sequence.  Row i is stored at position seq_lens[i]-1 of sequence i, i.e. into slot
 *   block_tables[i][(seq_lens[i]-1)/block_size]*block_size + (seq_lens[i]-1)%block_size
 * — the slot the reference's block manager hands to reshape_and_cache for a decode step
 * (vllmini/block_manager.py decode_step) — and the attention over positions 0..seq_lens[i]-1 takes that token
 * from the rows themselves.  Caches and `out` end up bit-identical to
 *   vmi_reshape_and_cache_f16(key, value, ..., slot_mapping = those slots) ; vmi_paged_attention_v1_f16(...)
 * (the plain entry: the _ws entry with a workspace may run a split kernel there — the same up to fp32 summation order).
 * A sequence must own its last block (no two sequences append into one block; the reference never shares blocks).
 * Rows with seq_lens[i] <= 0 append nothing.  key rows must be 16-byte aligned.  variant: 0 = heuristic.
 * The first 20 arguments are those of vmi_paged_attention_v1_f16 (caches mutable here).
 */
int vmi_paged_attention_v1_append_f16(
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

/*
 * fp8 KV cache — kv_cache_dtype "fp8" / "fp8_e4m3" of the reference surface (quantization/fp8/nvidia/quant_utils.cuh:
 * 529-566; the reference build never defines ENABLE_FP8, so its own fp8 path is assert(false) — these entries follow
 * its SOURCE: attention_kernels.cu:283-289, 410-418, cache_kernels.cu:200-205).
 *   caches: E4M3 ("fn": no infinities, max 448) bytes, key_cache [NB, H, D/16, BS, 16], value_cache [NB, H, D, BS];
 *           kv_block_stride / kv_head_stride in elements (= bytes), multiples of 16
 *   store:  fp8(float(x) / kv_scale), round to nearest even, saturating (__NV_SATFINITE), x = 16
 *   load:   float_to_half(float(fp8) * kv_scale), then the fp16 arithmetic of paged_attention_v1 unchanged
 * query/out float16; head sizes as above; block sizes 16 and 32.  variant: 0 = heuristic ("fp8_" names).
 */
int vmi_paged_attention_v1_fp8(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream,
    float kv_scale, int32_t variant);
/* ... with a caller-owned workspace (see vmi_paged_attention_v1_f16_ws: the same contract; split kernels "fp8_d<head>_x<waves>_..."). */
int vmi_paged_attention_v1_fp8_ws(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream,
    float kv_scale,
    void* workspace, int64_t workspace_bytes, int32_t variant);
int vmi_paged_attention_v2_fp8(
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
int vmi_reshape_and_cache_fp8(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping,
    int32_t num_tokens, int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
    int64_t key_stride, int64_t value_stride, float kv_scale,
    int32_t device, void* stream);
int vmi_paged_attention_v1_pick_variant_fp8(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                            int32_t block_size, int32_t max_seq_len, int32_t mean_seq_len);

/* Number of tuning variants (valid ids are 1..count) and a short name for each. */
int vmi_paged_attention_v1_variant_count(void);
const char* vmi_paged_attention_v1_variant_name(int32_t variant);
/* Variant id the heuristic would choose for this shape (>=1), 0 for an unsupported head/block size. */
int vmi_paged_attention_v1_pick_variant(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                        int32_t block_size, int32_t max_seq_len);
/*
 * The heuristic as the operators apply it, i.e. knowing num_kv_heads and the element / cache types: with grouped-query
 * attention (num_heads / num_kv_heads > 1) it selects a kernel that loads each K / V tile once for all the query
 * heads of a KV head ("gq" variants) — the other pick functions describe the multi-head-attention menus only.
 */
int vmi_paged_attention_v1_pick_variant_gqa(int32_t num_seqs, int32_t num_heads, int32_t num_kv_heads,
                                            int32_t head_size, int32_t block_size, int32_t max_seq_len,
                                            int32_t is_bf16 /* extras library only; the product answers 0 */,
                                            int32_t is_fp8 /* 0, 1 = E4M3; 2 = E5M2: extras library only */);

/*
 * Opt-in accuracy/speed switch for grouped-query attention (process-wide, default 0; returns the previous value).
 * on != 0 lets the picks above select the "_pvm" kernels, which run the probabilities x V contraction on the matrix
 * cores too (exact fp16 products, fp32 sums) instead of reproducing the reference kernel's fp16 rounding of every
 * product and pair sum (dtype_float16.cuh:118-124, 399-404).  Results then agree with the reference kernel to the
 * north-star 1e-3 (measured <= 4.9e-4, and closer to an fp64 attention than the reference kernel itself) rather than
 * to 1-2 fp16 ulp; with 8 query heads per KV head the launch is 1.2x faster (profiles/r01k_pv_on_matrix_cores.md).
 * fp8 pages (head size 128): 1.45x (cfg4 shape with 8 KV heads, 135 -> 93 us).
 * No effect on multi-head attention (num_kv_heads == num_heads) or on an explicit `variant`.
 */
int vmi_set_pv_mfma(int32_t on);

/*
 * The variant id the calling thread's last paged_attention_v1 launch ran (what `variant == 0` resolved to, or the
 * explicit one; 0 before the first launch and after a block-sparse one).  For benchmarks and tests that label a
 * measurement with the kernel that produced it; the pick functions above answer without the launch's kv_scale.
 */
int vmi_paged_attention_v1_last_variant(void);
/*
 * Head size 128 on a full chip goes out as a GATED DOUBLE LAUNCH: the kernel vmi_paged_attention_v1_last_variant names
 * (built for equal lengths) and, behind it, a balanced kernel; each reads seq_lens on the device and leaves at once
 * unless the batch is its kind, so the host cannot know which of the two did the work.  This returns the id of that
 * second kernel for the calling thread's last launch, 0 when the launch was a single kernel.
 */
int vmi_paged_attention_v1_last_partner(void);

/*
 * 1 when `variant` can serve a launch with this max_seq_len (its logits rows fit the 160 KiB of LDS) and, with
 * for_append = 1, has a fused-append twin (for_append = 2: a form for vmi_paged_attention_v1_newest_f16, the fused append
 * without the cache write — the fused-append twins and the balanced kernels' append-read form); 0 otherwise.  For callers that pick a variant from what they know about
 * the batch (the pick functions above take the batch's longest length) but launch with a larger max_seq_len — the
 * capacity of the pool, as the reference's scheduler does (vllmini/scheduler.py:97): an explicit `variant` that does not
 * fit is an error (VMI_E_MAX_SEQ_LEN), variant 0 falls back by itself.
 */
int vmi_paged_attention_v1_variant_fits(int32_t variant, int32_t max_seq_len, int32_t for_append);

/*
 * The same heuristic with what a caller may know on the host: the batch's mean sequence length (0 = unknown) and
 * the element type.  mean_seq_len well below max_seq_len marks a ragged batch, for which a many-waves-per-head
 * decomposition is chosen (the hardware dispatcher then balances the chip).  Pass the result as `variant`.
 */
int vmi_paged_attention_v1_pick_variant_hint(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                             int32_t block_size, int32_t max_seq_len,
                                             int32_t mean_seq_len, int32_t is_bf16);

/*
 * paged_attention_v2 (split-KV), fp16 — the operator the reference exports next to v1
 * (paged_attention_cuda.cpp:27-47, :53; host entry attention_kernels.cu:966-990, launcher :845-928).
 * The context is cut into 512-token partitions (PARTITION_SIZE, :847); every (seq, head, partition)
 * is attended independently and a second kernel merges the partitions (:564-669).
 *
 *   exp_sums, max_logits [num_seqs, num_heads, P] fp32   P = ceil(max_seq_len / 512)   (written)
 *   tmp_out              [num_seqs, num_heads, P, head_size] fp16                       (written)
 *   out                  [num_seqs, num_heads, head_size] fp16                          (written)
 * Partitions at or past a sequence's context are left untouched, as in the reference (:116-119).
 * variant: 0 = heuristic, 1..vmi_paged_attention_v2_variant_count() forces a decomposition.
 * All other arguments as vmi_paged_attention_v1_f16.  Limits: num_seqs, num_heads, P <= 65535.
 */
int vmi_paged_attention_v2_f16(
    void* out, void* exp_sums, void* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
    float scale,
    const int32_t* block_tables, const int32_t* seq_lens,
    int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
    const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream, int32_t variant);
int vmi_paged_attention_v2_variant_count(void);
const char* vmi_paged_attention_v2_variant_name(int32_t variant);

/*
 * cache_ops.reshape_and_cache, fp16 key/value into fp16 ("auto") caches.
 *
 * Replaces: reshape_and_cache(key, value, key_cache, value_cache, slot_mapping,
 *           kv_cache_dtype, kv_scale)                   — cache_kernels.cu:256-281.
 *
 *   key, value     [num_tokens, num_heads, head_size] fp16, row strides key_stride / value_stride
 *   slot_mapping   [num_tokens] int64; slot < 0 => token skipped (cache_kernels.cu:165-169);
 *                  block = slot / block_size, offset = slot % block_size (cache_kernels.cu:172-173)
 *   x              key_cache.size(4); must be 8
 * Caches are dense in the layout above (the reference computes dense offsets from
 * num_heads/head_size/block_size/x, cache_kernels.cu:187-194).
 */
int vmi_reshape_and_cache_f16(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping,
    int32_t num_tokens, int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
    int64_t key_stride, int64_t value_stride,
    int32_t device, void* stream);

/*
 * cache_ops.copy_blocks — cache_kernels.cu:96-148 (kernel :68-94).  For every layer l and pair p:
 * K_l[dst_p] = K_l[src_p], V_l[dst_p] = V_l[src_p].
 *   key_cache_ptrs / value_cache_ptrs   HOST arrays of num_layers DEVICE pointers (one cache per layer)
 *   block_mapping                        DEVICE int64 [num_pairs, 2] = (src block, dst block)
 *   block_bytes                          bytes of one block of one cache (element_size * cache[0].numel())
 * Unlike the reference (blocking pointer-table upload, :119-126) the call does not synchronise.
 */
int vmi_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs, int32_t num_layers,
                    const int64_t* block_mapping, int32_t num_pairs, int64_t block_bytes,
                    int32_t device, void* stream);

/*
 * cache_ops.swap_blocks — cache_kernels.cu:24-63: one async memcpy per (src, dst) pair on `stream`.
 *   block_mapping_host   HOST int64 [num_pairs, 2] (the reference requires a CPU tensor, :45)
 *   kind                 0 = device->device (same GPU), 1 = device->host, 2 = host->device
 */
int vmi_swap_blocks(const void* src, void* dst, const int64_t* block_mapping_host, int32_t num_pairs,
                    int64_t block_bytes, int32_t kind, int32_t device, void* stream);

/*
 * paged_attention_v1 over the cache PLUS this step's rows (extension): the arguments and the attention of
 * vmi_paged_attention_v1_append_f16 — the token at position seq_lens[i]-1 is taken from key / value, whatever the cache
 * holds in its slot — WITHOUT the cache write.  The caller stores the rows itself, any time before the next token's
 * attention of the same layer: the 12 layers of a GPT-2 token need ONE vmi_reshape_and_cache_f16 over [layers * num_seqs]
 * rows (the reference's pool is shared by all layers, vllmini/kv_cache.py:13-14) instead of one per layer in front of each
 * attention (vllmini/model/gpt2.py:44).  `out` is bit-identical to the call pair's.  Caches are not touched.
 */
int vmi_paged_attention_v1_newest_f16(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int32_t device, void* stream,
    const void* key, const void* value, int64_t key_stride, int64_t value_stride, int32_t variant);

/*
 * The pool's preemption move (extension; counterpart of BlockManager.swap_to_cpu / swap_from_cpu,
 * vllmini/block_manager.py:70-87, which the reference runs as one copy per block, cache_kernels.cu:56-62): ONE launch on
 * `stream` moves block_mapping[i] = (src block, dst block) of BOTH caches for every i.
 *   src_key / src_value, dst_key / dst_value   DEVICE-ACCESSIBLE pointers: device memory or pinned (mapped) host memory —
 *                                              the GPU reads / writes host pages directly, no staging copy
 *   block_mapping                              device-accessible int64 [num_pairs, 2]
 *   block_bytes                                bytes of one block of one cache; a multiple of 16
 * The bytes moved are exactly vmi_swap_blocks' (tests/test_parity_gpu.py); asynchronous, never synchronises.
 */
int vmi_swap_blocks_batched(const void* src_key, const void* src_value, void* dst_key, void* dst_value,
                            const int64_t* block_mapping, int32_t num_pairs, int64_t block_bytes,
                            int32_t device, void* stream);

/*
 * 0 for the product library, 1 for the diagnostic build (-DVMI_DIAG: adds the entries of vmi_paged_attention_diag.h, the
 * "loads only" and LDS-staging experiment kernels; same sources otherwise).
 */
int vmi_is_diag_build(void);

/*
 * 0 for the product library (libvmi_paged_attention.so): the hot path of SURVEY.md §8 — float16 tensors over float16 or
 * fp8-E4M3 pages, every head / block size of the reference's dispatch, grouped-query heads, ALiBi, paged_attention_v2, the
 * fused append, copy/swap_blocks — i.e. exactly the entries of THIS header.  1 for libvmi_paged_attention_extras.so
 * (`build.py --extras`), which exports, in addition, the entries of vmi_paged_attention_extras.h: the rest of the
 * reference's dispatch surface — bfloat16 and float32 tensors, fp8-E5M2 pages, block-sparse attention,
 * reshape_and_cache_flash, convert_fp8 (out of the path's scope).
 */
int vmi_has_extras(void);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VMI_PAGED_ATTENTION_H */
