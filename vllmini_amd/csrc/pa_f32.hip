// pa_f32.hip — paged_attention_v1 and reshape_and_cache for float32 tensors: the (float, float) branch of the
// reference's dispatch (quant_utils.cuh:529-535; attention_kernels.cu:86-496 with scalar_t = cache_t = float).
// x = 16 / sizeof(float) = 4: key_cache [NB, H, D/4, BS, 4], value_cache [NB, H, D, BS], all arithmetic fp32.
// Nobody serves from an fp32 KV cache — the reference's callers use half (scheduler.py:13) — so this is one
// straightforward kernel per (head size, block size), not a tuned menu: one workgroup of four waves per (sequence, head),
// the blocks dealt round-robin to the waves, a (block, head) tile fetched as 1-KiB wave loads in the layout's own order.
// Out of the hot-path scope: linked into libvmi_paged_attention_extras.so only (kernels AND their two C-ABI entries; the
// product library has neither — pa_extras_absent.hip holds empty menus and no entry).
#include "vmi_paged_attention_extras.h"
#include "pa_kernel.hpp"
#include "pa_host.hpp"

namespace vmi {

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int D, int BS>
__global__ void __launch_bounds__(256) pa_v1_f32_kernel(const PAF32Params p) {
  constexpr int WAVES = 4;
  constexpr int UNITS = D * BS / 4;        // 16-byte units (4 floats) in one (block, head) tile of K — and of V
  constexpr int NL = (UNITS + 63) / 64;    // wave loads per tile
  constexpr int UPR = BS / 4;              // V: units per dim row
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* logits = reinterpret_cast<float*>(smem_raw);   // [lpad]
  float* red = logits + p.lpad;                          // [2 * WAVES]
  float* osm = red + 2 * WAVES;                          // [WAVES][D]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int head = blockIdx.x, seq = blockIdx.y;
  const int kvh = head / (p.num_heads / p.num_kv_heads);                    // :153
  int L = p.seq_lens[seq];
  L = L > p.lpad ? p.lpad : L;
  float* out = p.out + ((int64_t)seq * p.num_heads + head) * D;
  if (L <= 0) {
    for (int d = threadIdx.x; d < D; d += 256) out[d] = 0.f;
    return;
  }
  const int nblk = (L + BS - 1) / BS;
  const int32_t* bt = p.block_tables + (int64_t)seq * p.max_blocks_per_seq;
  const float* q = p.q + (int64_t)seq * p.q_stride + (int64_t)head * D;
  const float slope = p.alibi ? p.alibi[head] : 0.f;
  const int64_t hoff = (int64_t)kvh * p.kv_head_stride;

  // K tile [D/4][BS][4]: unit u = chunk * BS + token; lane takes units 64*i + lane, so its token is lane % BS for every i
  const int tok = lane % BS;
  f32x4v qreg[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int u = 64 * i + lane;
    qreg[i] = u < UNITS ? *reinterpret_cast<const f32x4v*>(q + (u / BS) * 4) : f32x4v{0.f, 0.f, 0.f, 0.f};
  }

  // ---- K pass ----
  float qk_max = -FLT_MAX;
  for (int b = wave; b < nblk; b += WAVES) {
    const float* tile = p.kc + (int64_t)bt[b] * p.kv_block_stride + hoff;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int u = 64 * i + lane;
      if (u < UNITS) {
        const f32x4v k = *reinterpret_cast<const f32x4v*>(tile + (int64_t)u * 4);
        acc = __builtin_fmaf(qreg[i][0], k[0], acc);
        acc = __builtin_fmaf(qreg[i][1], k[1], acc);
        acc = __builtin_fmaf(qreg[i][2], k[2], acc);
        acc = __builtin_fmaf(qreg[i][3], k[3], acc);
      }
    }
#pragma unroll
    for (int m = BS; m < 64; m <<= 1) acc += __shfl_xor(acc, m);   // the lanes that hold the same token
    const int token = b * BS + tok;
    float qk = p.scale * acc;
    qk += (slope != 0.f) ? slope * (float)(token - L + 1) : 0.f;    // :297
    const bool masked = token >= L;                                  // :302-305
    if (lane < BS) logits[token] = masked ? 0.f : qk;
    qk_max = masked ? qk_max : fmaxf(qk_max, qk);
  }
  // ---- softmax (:310-346) ----
  qk_max = wave_max(qk_max);
  if (lane == 0) red[wave] = qk_max;
  __syncthreads();
  float m = red[0];
#pragma unroll
  for (int w = 1; w < WAVES; ++w) m = fmaxf(m, red[w]);
  float e_sum = 0.f;
  for (int i = threadIdx.x; i < L; i += 256) {
    const float e = __expf(logits[i] - m);
    logits[i] = e;
    e_sum += e;
  }
  e_sum = wave_sum(e_sum);
  if (lane == 0) red[WAVES + wave] = e_sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) tot += red[WAVES + w];
  const float inv = __builtin_amdgcn_rcpf(tot + 1e-6f);              // :342
  // probabilities past the context inside the last block are never written: the V side masks them

  // ---- V pass: tile [D][BS]: unit u = row * UPR + part; lane takes units 64*i + lane ----
  float acc[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i] = 0.f;
  const int part = lane % UPR;
  for (int b = wave; b < nblk; b += WAVES) {
    const float* tile = p.vc + (int64_t)bt[b] * p.kv_block_stride + hoff;
    const int token0 = b * BS + part * 4;
    f32x4v pr;
#pragma unroll
    for (int e = 0; e < 4; ++e) pr[e] = (token0 + e < L) ? logits[token0 + e] * inv : 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int u = 64 * i + lane;
      if (u < UNITS) {
        f32x4v v = *reinterpret_cast<const f32x4v*>(tile + (int64_t)u * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (token0 + e < L) ? v[e] : 0.f;   // :420-430 (stale / NaN bytes past the context)
        acc[i] += ((pr[0] * v[0] + pr[1] * v[1]) + pr[2] * v[2]) + pr[3] * v[3];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) {
#pragma unroll
    for (int s = 1; s < UPR; s <<= 1) acc[i] += __shfl_xor(acc[i], s);  // the lanes of one dim row
    const int u = 64 * i + lane;
    if (part == 0 && u < UNITS) osm[wave * D + u / UPR] = acc[i];
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) out[d] = ((osm[d] + osm[D + d]) + osm[2 * D + d]) + osm[3 * D + d];
}

template <int D>
static pa_f32_kernel_t pick_bs(int BS) {
  switch (BS) {
    case 8: return (pa_f32_kernel_t)pa_v1_f32_kernel<D, 8>;
    case 16: return (pa_f32_kernel_t)pa_v1_f32_kernel<D, 16>;
    case 32: return (pa_f32_kernel_t)pa_v1_f32_kernel<D, 32>;
  }
  return nullptr;
}
pa_f32_kernel_t pa_v1_f32_kernel_for(int D, int BS) {
  switch (D) {
    case 64: return pick_bs<64>(BS);
    case 80: return pick_bs<80>(BS);
    case 96: return pick_bs<96>(BS);
    case 112: return pick_bs<112>(BS);
    case 128: return pick_bs<128>(BS);
    case 192: return pick_bs<192>(BS);
    case 256: return pick_bs<256>(BS);
  }
  return nullptr;
}

// reshape_and_cache for float32 rows (cache_kernels.cu:152-207 with x = 4): one workgroup per token, a thread per
// 16-byte chunk of the row: K[blk, h, d/4, off, 0..3] is that chunk as is, V[blk, h, d..d+3, off] its four scalars.
__global__ void __launch_bounds__(256)
    reshape_and_cache_f32_kernel(const float* __restrict__ key, const float* __restrict__ value, float* __restrict__ kc,
                                 float* __restrict__ vc, const int64_t* __restrict__ slot_mapping, int64_t key_stride,
                                 int64_t value_stride, int H, int D, int BS) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;                                                  // :165-169
  const int64_t blk = slot / BS, off = slot % BS;
  const int n4 = (H * D) >> 2;
  for (int c = threadIdx.x; c < n4; c += blockDim.x) {
    const int i = c << 2, h = i / D, d = i - h * D;
    const float* ks = key + token * key_stride + i;
    const float* vs = value + token * value_stride + i;
    float* kd = kc + (((blk * H + h) * (D >> 2) + (d >> 2)) * BS + off) * 4;
    float* vd = vc + ((blk * H + h) * D + d) * BS + off;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kd[e] = ks[e];
      vd[(int64_t)e * BS] = vs[e];
    }
  }
}
void reshape_and_cache_f32_launch(const float* key, const float* value, float* kc, float* vc, const int64_t* slots,
                                  int64_t key_stride, int64_t value_stride, int T, int H, int D, int BS,
                                  hipStream_t stream) {
  int threads = (((H * D) >> 2) + 63) / 64 * 64;
  if (threads > 256) threads = 256;
  hipLaunchKernelGGL(reshape_and_cache_f32_kernel, dim3(T), dim3(threads), 0, stream, key, value, kc, vc, slots,
                     key_stride, value_stride, H, D, BS);
}

}  // namespace vmi

extern "C" {

// ---- float32 tensors: the (float, float) branch of the dispatch (pa_f32.hip) ----
int vmi_paged_attention_v1_f32(void* out, const void* query, const void* key_cache, const void* value_cache,
                               int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
                               float scale, const int32_t* block_tables, const int32_t* seq_lens,
                               int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                               const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                               int64_t kv_head_stride, int32_t device, void* stream) {
  using namespace vmi;
  if (!out || !query || !key_cache || !value_cache || !block_tables || !seq_lens)
    return fail(VMI_E_NULL_POINTER, "paged_attention_v1 (float32): NULL tensor pointer");
  if (!head_size_supported(head_size)) return fail(VMI_E_HEAD_SIZE, "Unsupported head size: %d", head_size);
  if (!block_size_supported(block_size)) return fail(VMI_E_BLOCK_SIZE, "Unsupported block size: %d", block_size);
  if (num_seqs < 0 || num_heads <= 0 || max_seq_len < 0 || max_num_blocks_per_seq < 0)
    return fail(VMI_E_SHAPE, "paged_attention_v1 (float32): negative size");
  if (num_heads > 65535) return fail(VMI_E_SHAPE, "paged_attention_v1 (float32): num_heads above the 65535 grid limit");
  if (num_kv_heads <= 0 || num_heads % num_kv_heads != 0)
    return fail(VMI_E_KV_HEADS, "paged_attention_v1: num_heads=%d not divisible by num_kv_heads=%d", num_heads, num_kv_heads);
  if (!aligned16(query) || !aligned16(key_cache) || !aligned16(value_cache) || (q_stride & 3) || (kv_block_stride & 3) ||
      (kv_head_stride & 3))
    return fail(VMI_E_ALIGNMENT, "paged_attention_v1 (float32): pointers and strides must be 16-byte aligned");
  if (num_seqs == 0) return VMI_OK;
  const int lpad = ((max_seq_len + 31) / 32) * 32;
  const size_t lds = ((size_t)lpad + 8 + 4 * (size_t)head_size) * sizeof(float);
  if (lds > 160 * 1024)
    return fail(VMI_E_MAX_SEQ_LEN, "paged_attention_v1 (float32): max_seq_len=%d needs %zu B of LDS, limit 163840", max_seq_len, lds);
  pa_f32_kernel_t fn = pa_v1_f32_kernel_for(head_size, block_size);
  if (!fn) return fail(VMI_E_HEAD_SIZE, "Unsupported head size: %d", head_size);
  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  if (lds > 48 * 1024) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  }
  PAF32Params p;
  p.alibi = alibi_slopes;
  p.num_heads = num_heads;
  p.num_kv_heads = num_kv_heads;
  p.scale = scale;
  p.max_blocks_per_seq = max_num_blocks_per_seq;
  p.q_stride = q_stride;
  p.kv_block_stride = kv_block_stride;
  p.kv_head_stride = kv_head_stride;
  p.lpad = lpad;
  p.kc = static_cast<const float*>(key_cache);
  p.vc = static_cast<const float*>(value_cache);
  for (int32_t s0 = 0; s0 < num_seqs; s0 += 65535) {
    const int32_t ns = (num_seqs - s0) < 65535 ? (num_seqs - s0) : 65535;
    p.out = static_cast<float*>(out) + (int64_t)s0 * num_heads * head_size;
    p.q = static_cast<const float*>(query) + (int64_t)s0 * q_stride;
    p.block_tables = block_tables + (int64_t)s0 * max_num_blocks_per_seq;
    p.seq_lens = seq_lens + s0;
    hipLaunchKernelGGL(fn, dim3(num_heads, ns), dim3(256), lds, static_cast<hipStream_t>(stream), p);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "paged_attention_v1 (float32) launch");
  }
  return VMI_OK;
}

int vmi_reshape_and_cache_f32(const void* key, const void* value, void* key_cache, void* value_cache,
                              const int64_t* slot_mapping, int32_t num_tokens, int32_t num_heads,
                              int32_t head_size, int32_t block_size, int32_t x, int64_t key_stride,
                              int64_t value_stride, int32_t device, void* stream) {
  using namespace vmi;
  if (!key || !value || !key_cache || !value_cache || !slot_mapping)
    return fail(VMI_E_NULL_POINTER, "reshape_and_cache (float32): NULL tensor pointer");
  if (x != 4) return fail(VMI_E_X, "reshape_and_cache (float32): key_cache.size(4) must be 4, got %d", x);
  if (num_tokens < 0 || num_heads <= 0 || head_size <= 0 || (head_size & 3))
    return fail(VMI_E_SHAPE, "reshape_and_cache (float32): bad sizes");
  if (block_size <= 0) return fail(VMI_E_BLOCK_SIZE, "Unsupported block size: %d", block_size);
  if (num_tokens == 0) return VMI_OK;
  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  reshape_and_cache_f32_launch(static_cast<const float*>(key), static_cast<const float*>(value),
                               static_cast<float*>(key_cache), static_cast<float*>(value_cache), slot_mapping, key_stride,
                               value_stride, num_tokens, num_heads, head_size, block_size, static_cast<hipStream_t>(stream));
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "reshape_and_cache (float32) launch");
  return VMI_OK;
}

}  // extern "C"
