/* C-ABI of libvmi_gpt2_layer.so — the GPT-2 block's linear layers around the paged-attention call pair, for the decode
 * harness (the callers of the hot path).  NOT part of the drop-in boundary of the two operators
 * (include/vmi_paged_attention.h): the reference has no counterpart of this file — its block is torch modules
 * (vllmini/model/gpt2.py:14-15 c_attn / c_proj, :117-128 GPT2MLP, :130-135 ln_1 / ln_2, residual adds in GPT2Block.forward) —
 * and the harness is this build's own (vllmini_amd/gpt2_decode.py), so this library is the harness's, loaded by
 * vllmini_amd/gpt2_layer.py only.
 *
 * One entry: a skinny fp16 linear layer y = epilogue(prologue(x) . W^T + bias) for a decode step's M <= a few hundred rows,
 * written for gfx950 (v_mfma_f32_16x16x32_f16, the row tile resident in LDS, the weight rows streamed once into registers):
 *   prologue   none | LayerNorm over K (fp32 statistics, rounded to half as torch.nn.functional.layer_norm does on half input)
 *   epilogue   bias | bias + GELU (erf form, nn.GELU) | bias + residual add
 * Rounding points are the torch module chain's: fp32 accumulation, + bias in fp32, ONE rounding to half, then GELU / the
 * residual add on that half value in fp32, rounded to half again.
 *
 * Plain pointers and sizes, no torch types, nothing retained, never synchronises; returns 0 or a VMI_LAYER_E_* code and
 * vmi_gpt2_layer_last_error() names the reason (thread-local). */
#ifndef VMI_GPT2_LAYER_H
#define VMI_GPT2_LAYER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define VMI_LAYER_API __attribute__((visibility("default")))
#else
#define VMI_LAYER_API
#endif

#define VMI_GPT2_LAYER_ABI_VERSION 1

enum {
  VMI_LAYER_OK = 0,
  VMI_LAYER_E_ARG = 1,         /* null pointer, non-positive size, unknown epilogue */
  VMI_LAYER_E_SHAPE = 2,       /* K % 32 != 0, N % 16 != 0, or a K whose row tile does not fit a CU's LDS */
  VMI_LAYER_E_HIP = 3          /* a HIP call failed (text in vmi_gpt2_layer_last_error) */
};

enum { VMI_LAYER_EPI_BIAS = 0, VMI_LAYER_EPI_BIAS_GELU = 1, VMI_LAYER_EPI_BIAS_RESIDUAL = 2,
       VMI_LAYER_EPI_BIAS_KV_CACHE = 3 /* vmi_gpt2_linear_qkv_cache_f16 only */ };

/* y[M, N] = epilogue( LN?(x)[M, K] . w[N, K]^T + bias[N] ).
 *   x          half [M, K], row stride ldx elements (rows need 16-byte alignment: ldx % 8 == 0)
 *   w          w_layout 0: half [N, K] contiguous — nn.Linear's weight as stored (gpt2.py:14-15, :121-122);
 *              w_layout 1: the same values as MFMA tiles, half [N / 16][K / 32][64][8] with
 *              tile[s][t][kc * 16 + r][e] = w[16 s + r][32 t + 8 kc + e] — a wave's load of one k-step of its 16 columns is one
 *              contiguous KiB (weights are static: the harness packs them once, vllmini_amd/gpt2_layer.py pack_weight)
 *   bias       half [N] or NULL
 *   ln_gamma, ln_beta, ln_eps   LayerNorm over K applied to x first (both NULL: no LayerNorm)
 *   residual   half [M, N], row stride ldr — read only with VMI_LAYER_EPI_BIAS_RESIDUAL; may alias y (same strides)
 *   y          half [M, N], row stride ldy (ldy % 4 == 0)
 *   stream     a hipStream_t (NULL = the default stream)
 * K % 32 == 0, N % 16 == 0, K <= 4608 (<= 2048 behind a LayerNorm). */
VMI_LAYER_API int vmi_gpt2_linear_f16(const void* x, int64_t ldx, const void* w, const void* bias, const void* ln_gamma,
                                      const void* ln_beta, float ln_eps, const void* residual, int64_t ldr, void* y,
                                      int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t w_layout,
                                      int32_t device, void* stream);

/* The q / k / v projection with reshape_and_cache's copy done by the producer: qkv[M, 3E] = LN?(x) . w[3E, K]^T + bias as
 * above (E = num_heads * head_size), and for every row m with slot_mapping[m] >= 0 its k and v columns ALSO go to the paged
 * cache at that slot — key_cache[blk, h, d / 8, off, d % 8], value_cache[blk, h, d, off], blk = slot / block_size,
 * off = slot % block_size: the bytes cache_ops.reshape_and_cache (ext/cache_kernels.cu:152-207) would write from the k / v
 * views of qkv, one launch earlier.  float16 caches only (kv_cache_dtype "auto", x = 8); kv_block_stride / kv_head_stride in
 * elements, applied to both caches as the reference does (attention_kernels.cu:706-707).  head_size % 16 == 0. */
VMI_LAYER_API int vmi_gpt2_linear_qkv_cache_f16(const void* x, int64_t ldx, const void* w, const void* bias, const void* ln_gamma,
                                                const void* ln_beta, float ln_eps, void* qkv, int64_t ldy, int32_t M, int32_t K,
                                                int32_t w_layout, void* key_cache, void* value_cache,
                                                const int64_t* slot_mapping, int32_t num_heads, int32_t head_size,
                                                int32_t block_size, int64_t kv_block_stride, int64_t kv_head_stride,
                                                int32_t device, void* stream);

/* The two ends of a decode step.  out[t, :] = wte[input_ids[t], :] + wpe[position_ids[t], :] (one launch for torch's two gathers
 * and an add; hidden % 8 == 0) — and greedy sampling: out[r] = the index of the first maximum of logits[r, 0:vocab]
 * (torch.argmax's tie rule; row stride ld elements, any alignment). */
VMI_LAYER_API int vmi_gpt2_embed_f16(const int64_t* input_ids, const int64_t* position_ids, const void* wte, const void* wpe,
                                     void* out, int32_t num_tokens, int32_t hidden, int32_t device, void* stream);
VMI_LAYER_API int vmi_gpt2_argmax_f16(const void* logits, int64_t ld, int32_t num_rows, int32_t vocab, int64_t* out,
                                      int32_t device, void* stream);

/* Scheduler.sample_next_token (vllmini/scheduler.py:144-153: logits / temperature, top-k, softmax, one multinomial draw) for a
 * batch of rows in one launch: out[r] = the index drawn from softmax(top_k largest of logits[r, :] / temperature), the draw made by
 * inverse CDF over those top_k in descending order (ties: smaller index first) at uniform[r] in [0, 1) — the caller's random
 * numbers, e.g. torch.rand(num_rows, generator=...), so the generator and its state stay the caller's.  vocab <= 65536, top_k <= 64. */
VMI_LAYER_API int vmi_gpt2_sample_top_k_f16(const void* logits, int64_t ld, int32_t num_rows, int32_t vocab, int32_t top_k,
                                            float temperature, const float* uniform, int64_t* out, int32_t device, void* stream);

/* The kernel vmi_gpt2_linear_f16 would launch for this shape ("bm32_nw4_ks1_r2_ln_gelu"), for records; NULL if refused. */
VMI_LAYER_API const char* vmi_gpt2_linear_kernel_name(int32_t M, int32_t N, int32_t K, int32_t has_ln, int32_t epilogue);

VMI_LAYER_API const char* vmi_gpt2_layer_last_error(void);
VMI_LAYER_API int32_t vmi_gpt2_layer_abi_version(void);
VMI_LAYER_API const char* vmi_gpt2_layer_target_arch(void);

#ifdef __cplusplus
}
#endif
#endif
