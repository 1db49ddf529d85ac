// paged_attention.hip — MI355X (gfx950 / CDNA4) paged-attention decode path.
//
// Hand-written HIP for wave64; compiled ONLY for gfx950 (no CUDA path, no compat shims).
// Implements the C-ABI declared in include/vmi_paged_attention.h:
//   vmi_paged_attention_v1_f16   <- reference attention_kernels.cu:805-826 (+ launcher :690-767,
//                                   kernel :86-496)
//   vmi_reshape_and_cache_f16    <- reference cache_kernels.cu:256-281 (+ kernel :152-207)
//
// This is NOT a translation of the reference kernel's 32-lane "thread group" scheme.  The
// work decomposition is built around what one 64-lane wavefront reads in one instruction:
//
//   * a K tile (one physical block, one kv head) is D*16*2 bytes, contiguous
//     ([D/8][16 tok][8 halves]); a wave reads it as D/32 fully coalesced 1-KiB
//     global_load_dwordx4 (16 B per lane); lane = (chunk&3)*16 + tok holds 8 consecutive
//     dims of one token, so q.k is 8 v_fma_mix_f32 per load (fp16 operands, fp32 FMA — the
//     reference's arithmetic) followed by a 2-step butterfly over the 4 chunk lanes;
//   * a V tile is [D][16 tok] halves, also D/32 coalesced 1-KiB loads; lane = row*2 + half
//     holds 8 consecutive tokens of one dim row, so p.v is 4 v_pk_mul_f16 + 3 v_pk_add_f16;
//   * block-table entries are loaded once per 64 blocks into a VGPR (lane j = j-th block)
//     and broadcast with v_readlane_b32, so no dependent scalar-memory latency sits between
//     consecutive page loads;
//   * pages are register double-buffered U blocks deep (2*U KiB..4*U KiB per wave in
//     flight), and the first V group is issued before the softmax so HBM stays busy across
//     the K -> V phase change.
//
// Rounding points follow the reference kernel exactly (see oracle/pa_kernel_model.c):
// fp32 q.k with exact products, logits scaled in fp32, fp32 softmax with +1e-6 in the
// denominator, probabilities rounded to fp16 (RNE), fp16 products p*v, fp16 pair sums
// ((p0v0+p2v2)+p4v4)+p6v6 / odd likewise, fp32 accumulation across 8-token groups,
// fp16 (RNE) store.  Only fp32 summation ORDER differs from the reference.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC  (vllmini_amd/build.py)
//        -ffp-contract=off matters: the fp16 p*v products must round before they are added.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <float.h>

#include "vmi_paged_attention.h"

namespace vmi {

typedef _Float16 h16;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// LDS logits are written as float and re-read 4 at a time: the vector view must alias float
typedef float f32x4_alias __attribute__((ext_vector_type(4), may_alias));

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static int hip_fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return -(int)e;
}

// ----------------------------------------------------------------------------------------
// device helpers
// ----------------------------------------------------------------------------------------

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const h16* p) {
  if constexpr (NT) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  } else {
    return *reinterpret_cast<const u32x4*>(p);
  }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

struct PAParams {
  h16* out;
  const h16* q;
  const h16* kc;
  const h16* vc;
  const int32_t* block_tables;
  const int32_t* seq_lens;
  const float* alibi;
  int32_t num_heads;
  int32_t num_kv_heads;
  float scale;
  int32_t max_blocks_per_seq;
  int64_t q_stride;
  int64_t kv_block_stride;
  int64_t kv_head_stride;
  int32_t lpad;  // logits floats reserved per head in LDS (max_seq_len padded to 16)
  // split-KV (paged_attention_v2) only: per-partition softmax statistics, `out` is tmp_out
  float* exp_sums;             // [num_seqs, num_heads, max_num_partitions]
  float* max_logits;           // [num_seqs, num_heads, max_num_partitions]
  int32_t max_num_partitions;  // ceil(max_seq_len / 512)
};

// ----------------------------------------------------------------------------------------
// paged_attention_v1
//
//   D    head size (64 | 128)
//   HPW  heads per workgroup   (each head owns WPH waves)
//   WPH  waves per head        (blocks of one (seq, head) are dealt round-robin to them)
//   U    blocks per register group (software-pipeline depth = 2 groups)
//   NT   non-temporal page loads
//
// grid = (ceil(num_heads / HPW), num_seqs), block = HPW*WPH*64.
// LDS  = HPW*lpad*4 (logits)  +  HPW*2*WPH*4 (max/sum exchange)  +  HPW*WPH*D*4 (partial out)
// ----------------------------------------------------------------------------------------
//
// PART = true is the split-KV form behind paged_attention_v2 (reference attention_kernels.cu:529-562:
// the same kernel body with PARTITION_SIZE = 512): blockIdx.z selects a 512-token partition, the
// partition's normalised output goes to tmp_out and its (max, exp_sum) to max_logits / exp_sums.
template <int D, int HPW, int WPH, int U, bool NT, bool LOADS_ONLY = false, bool PART = false>
__global__ void __launch_bounds__(HPW* WPH * 64)
    pa_v1_kernel(const PAParams p) {
  constexpr int BS = 16;
  constexpr int PBLK = 512 / BS;  // blocks per partition (PARTITION_SIZE = 512, :847)
  constexpr int NL = D / 32;  // 1-KiB loads per K tile == per V tile
  static_assert(D % 32 == 0, "head size must be a multiple of 32");
  static_assert(64 % U == 0, "U must divide 64");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = wave / WPH;
  const int sub = wave % WPH;
  const int seq = blockIdx.y;
  const int head = blockIdx.x * HPW + hl;
  if (WPH == 1 && head >= p.num_heads) return;  // host guarantees H % HPW == 0 when WPH > 1

  // The first 64 block-table entries of this wave are requested BEFORE seq_len is known (any entry
  // inside the row is readable; entries past the context are simply never used), so the table,
  // seq_len and q loads overlap instead of forming a chain in front of the first page load.
  const int32_t* bt = p.block_tables + (int64_t)seq * p.max_blocks_per_seq;
  const int part = PART ? blockIdx.z : 0;
  const int blk_lo = PART ? part * PBLK : 0;  // first block of my range (:126-127)
  int bt_sg = 0;  // which 64-entry slice of my blocks is in bt_reg
  int32_t bt_reg = (blk_lo + sub + lane * WPH < p.max_blocks_per_seq) ? bt[blk_lo + sub + lane * WPH] : 0;

  // seq_len > max_seq_len overflows the logits buffer in the reference (undefined behaviour,
  // attention_kernels.cu:725-732); here the context is truncated to the LDS that was reserved.
  int L = p.seq_lens[seq];
  if constexpr (!PART) L = L > p.lpad ? p.lpad : L;
  const int nblk_seq = (L + BS - 1) / BS;                                     // :121
  const int blk_hi = PART ? (blk_lo + PBLK < nblk_seq ? blk_lo + PBLK : nblk_seq) : nblk_seq;  // :128-129
  if (PART && blk_lo * BS >= L) return;  // nothing in this partition (:116-119); uniform per workgroup
  const int nblk = blk_hi - blk_lo;      // blocks in my range
  const int tok_lo = blk_lo * BS;        // logits in LDS are indexed relative to the range start (:133)
  const int Lloc = (L < blk_hi * BS ? L : blk_hi * BS) - tok_lo;              // tokens in range (:134-136)

  float* logits = reinterpret_cast<float*>(smem) + (size_t)hl * p.lpad;
  float* red = reinterpret_cast<float*>(smem) + (size_t)HPW * p.lpad + hl * 2 * WPH;
  float* osm = reinterpret_cast<float*>(smem) + (size_t)HPW * p.lpad + HPW * 2 * WPH +
               (size_t)hl * WPH * D;

  h16* outp = PART ? p.out + (((int64_t)seq * p.num_heads + head) * p.max_num_partitions + part) * D
                   : p.out + ((int64_t)seq * p.num_heads + head) * D;

  if (L <= 0) {  // uniform over the workgroup (same seq): reference yields exp_sum = 0 -> out = 0
    if (sub == 0) {
      for (int d = lane; d < D; d += 64) outp[d] = (h16)0.f;
    }
    return;
  }

  const int kvh = head / (p.num_heads / p.num_kv_heads);
  const float slope = p.alibi ? p.alibi[head] : 0.f;

  // ---- q: this lane's 8-dim chunks, one per K load -------------------------------------
  const h16* qp = p.q + (int64_t)seq * p.q_stride + (int64_t)head * D;
  const int c4 = lane >> 4;  // chunk-within-load 0..3
  const int tk = lane & 15;  // token-within-block
  u32x4 qreg[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) qreg[i] = *reinterpret_cast<const u32x4*>(qp + (4 * i + c4) * 8);

  const h16* kbase = p.kc + (int64_t)kvh * p.kv_head_stride + lane * 8;
  const h16* vbase = p.vc + (int64_t)kvh * p.kv_head_stride + lane * 8;

  // ---- my share of the blocks: b = sub + idx*WPH, idx in [0, nmy) -----------------------
  const int nmy = nblk > sub ? (nblk - sub + WPH - 1) / WPH : 0;
  const int ngroups = (nmy + U - 1) / U;
  auto table_for = [&](int g) {  // lane j: physical id of my block (bt_sg*64 + j)
    const int sg = (g * U) >> 6;
    if (sg != bt_sg) {
      const int b = blk_lo + sub + (sg * 64 + lane) * WPH;
      bt_reg = (b < p.max_blocks_per_seq) ? bt[b] : 0;
      bt_sg = sg;
    }
  };

  auto load_group = [&](u32x4(&r)[U][NL], const h16* base, int g) {
    table_for(g);
#pragma unroll
    for (int j = 0; j < U; ++j) {
      int idx = g * U + j;
      idx = idx < nmy ? idx : nmy - 1;  // padding slots re-read my last block (never OOB)
      const int64_t phys = __builtin_amdgcn_readlane(bt_reg, idx & 63);
      // (sc0/sc1 cache-policy bits and buffer- vs flat-addressed loads were measured neutral on
      //  this stream; only `nt` pays: profiles/r01_cfg3_sweep_cache_policy_bits.json)
      {
        const h16* ptr = base + phys * p.kv_block_stride;
#pragma unroll
        for (int i = 0; i < NL; ++i) r[j][i] = ld16<NT>(ptr + i * 512);
      }
    }
  };

  // =========================== K pass: logits -> LDS, running max ========================
  float qk_max = -FLT_MAX;

  uint32_t fold = 0;  // LOADS_ONLY diagnostic: xor of everything loaded
  auto compute_k = [&](u32x4(&r)[U][NL], int g) {
    if constexpr (LOADS_ONLY) {
#pragma unroll
      for (int j = 0; j < U; ++j)
#pragma unroll
        for (int i = 0; i < NL; ++i) fold ^= r[j][i][0] ^ r[j][i][1] ^ r[j][i][2] ^ r[j][i][3];
      return;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int idx = g * U + j;
      if (idx < nmy) {  // wave-uniform
        const int b = blk_lo + sub + idx * WPH;
        // q.k over this lane's 8*NL dims: fp16 operands converted to fp32, fp32 FMA chain
        // (v_fma_mix_f32) — the reference's arithmetic (dtype_float16.cuh:292-298, 399-404).
        // One accumulator per load keeps NL independent dependency chains in flight.
        float accv[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const h16x8 qh = __builtin_bit_cast(h16x8, qreg[i]);
          const h16x8 kh = __builtin_bit_cast(h16x8, r[j][i]);
          float a = (float)qh[0] * (float)kh[0];
#pragma unroll
          for (int e = 1; e < 8; ++e) a = __builtin_fmaf((float)qh[e], (float)kh[e], a);
          accv[i] = a;
        }
        float acc = accv[0];
#pragma unroll
        for (int i = 1; i < NL; ++i) acc += accv[i];
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        const int token = b * BS + tk;
        float qk = p.scale * acc;
        qk += (slope != 0.f) ? slope * (float)(token - L + 1) : 0.f;
        const bool masked = token >= L;
        if (lane < 16) logits[token - tok_lo] = masked ? 0.f : qk;
        qk_max = masked ? qk_max : fmaxf(qk_max, qk);
      }
    }
  };

  u32x4 ra[U][NL], rb[U][NL];
  {
    if (ngroups > 0) load_group(ra, kbase, 0);
    int g = 0;
    for (; g + 2 <= ngroups; g += 2) {
      load_group(rb, kbase, g + 1);
      compute_k(ra, g);
      if (g + 2 < ngroups) load_group(ra, kbase, g + 2);
      compute_k(rb, g + 1);
    }
    if (g < ngroups) compute_k(ra, g);
  }

  // first V group goes out now: HBM stays busy while the softmax runs
  if (ngroups > 0) load_group(ra, vbase, 0);

  // =========================== softmax over the logits in LDS ============================
  qk_max = wave_max(qk_max);
  if constexpr (WPH > 1) {
    if (lane == 0) red[sub] = qk_max;
    __syncthreads();
    float m = -FLT_MAX;
#pragma unroll
    for (int w = 0; w < WPH; ++w) m = fmaxf(m, red[w]);
    qk_max = m;
  }

  float exp_sum = 0.f;
  for (int i = sub * 64 + lane; i < Lloc; i += WPH * 64) {
    const float e = __expf(logits[i] - qk_max);
    logits[i] = e;
    exp_sum += e;
  }
  exp_sum = wave_sum(exp_sum);
  if constexpr (WPH > 1) {
    if (lane == 0) red[WPH + sub] = exp_sum;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WPH; ++w) s += red[WPH + w];
    exp_sum = s;
  }
  const float inv_sum = __builtin_amdgcn_rcpf(exp_sum + 1e-6f);
  if constexpr (PART) {  // partition statistics for the reduce kernel (:349-357)
    if (sub == 0 && lane == 0) {
      const int64_t o = ((int64_t)seq * p.num_heads + head) * p.max_num_partitions + part;
      p.max_logits[o] = qk_max;
      p.exp_sums[o] = exp_sum;
    }
  }

  // =========================== V pass ====================================================
  float acc[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i] = 0.f;
  const int hf = lane & 1;  // which 8-token half of the block this lane owns

  auto compute_v = [&](u32x4(&r)[U][NL], int g) {
    if constexpr (LOADS_ONLY) {
#pragma unroll
      for (int j = 0; j < U; ++j)
#pragma unroll
        for (int i = 0; i < NL; ++i) fold ^= r[j][i][0] ^ r[j][i][1] ^ r[j][i][2] ^ r[j][i][3];
      return;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int idx = g * U + j;
      if (idx < nmy) {  // wave-uniform
        const int b = blk_lo + sub + idx * WPH;
        const int token0 = b * BS + hf * 8;
        const f32x4 e0 = *reinterpret_cast<const f32x4_alias*>(logits + token0 - tok_lo);
        const f32x4 e1 = *reinterpret_cast<const f32x4_alias*>(logits + token0 - tok_lo + 4);
        h16x8 pv;
        pv[0] = (h16)(e0[0] * inv_sum);
        pv[1] = (h16)(e0[1] * inv_sum);
        pv[2] = (h16)(e0[2] * inv_sum);
        pv[3] = (h16)(e0[3] * inv_sum);
        pv[4] = (h16)(e1[0] * inv_sum);
        pv[5] = (h16)(e1[1] * inv_sum);
        pv[6] = (h16)(e1[2] * inv_sum);
        pv[7] = (h16)(e1[3] * inv_sum);
        const bool last = (b == nblk_seq - 1);  // last block of the SEQUENCE (:420); wave-uniform
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          h16x8 v = __builtin_bit_cast(h16x8, r[j][i]);
          if (last) {
            // tokens past the context may hold stale/NaN bytes: zero them (ref :420-430)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (token0 + e < L) ? v[e] : (h16)0.f;
          }
          const h16x8 pr = pv * v;  // 4 x v_pk_mul_f16, each product rounded to fp16
          h16x2 c = h16x2{pr[0], pr[1]} + h16x2{pr[2], pr[3]};
          c = c + h16x2{pr[4], pr[5]};
          c = c + h16x2{pr[6], pr[7]};
          acc[i] += ((float)c[0] + (float)c[1]);
        }
      }
    }
  };

  {
    int g = 0;
    for (; g + 2 <= ngroups; g += 2) {
      load_group(rb, vbase, g + 1);
      compute_v(ra, g);
      if (g + 2 < ngroups) load_group(ra, vbase, g + 2);
      compute_v(rb, g + 1);
    }
    if (g < ngroups) compute_v(ra, g);
  }

  if constexpr (LOADS_ONLY) {
    if (fold == 0x9e3779b9u) outp[lane] = (h16)1.f;  // practically never; keeps the loads live
    return;
  }

  // the two lanes of a row hold the two 8-token halves
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i] += __shfl_xor(acc[i], 1);

  if constexpr (WPH > 1) {
    if (hf == 0) {
#pragma unroll
      for (int i = 0; i < NL; ++i) osm[sub * D + 32 * i + (lane >> 1)] = acc[i];
    }
    __syncthreads();
    if (sub == 0) {
      for (int d = lane; d < D; d += 64) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WPH; ++w) s += osm[w * D + d];
        outp[d] = (h16)s;
      }
    }
  } else {
    if (hf == 0) {
#pragma unroll
      for (int i = 0; i < NL; ++i) outp[32 * i + (lane >> 1)] = (h16)acc[i];
    }
  }
}

// ----------------------------------------------------------------------------------------
// paged_attention_v2 reduce: merge the partitions of one (seq, head) — reference
// attention_kernels.cu:564-669.  grid = (num_heads, num_seqs), block = 128.
//   1 partition  -> copy tmp_out to out (:582-594)
//   otherwise    -> m = max_j max_logits[j]; s_j = exp_sums[j]*exp(max_logits[j]-m);
//                   out[d] = sum_j float(tmp_out[j][d]) * s_j * 1/(sum_j s_j + 1e-6)   (fp32, j ascending)
// LDS: 2*max_num_partitions floats + 2 reduction slots per wave.
// ----------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(128)
    pa_v2_reduce_kernel(h16* __restrict__ out, const float* __restrict__ exp_sums,
                        const float* __restrict__ max_logits, const h16* __restrict__ tmp_out,
                        const int32_t* __restrict__ seq_lens, int max_num_partitions) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int num_heads = gridDim.x;
  const int head = blockIdx.x;
  const int seq = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int L = seq_lens[seq];
  const int np = (L + 511) / 512;  // :581
  const int64_t sh = ((int64_t)seq * num_heads + head) * max_num_partitions;
  h16* outp = out + ((int64_t)seq * num_heads + head) * D;
  const h16* tmp = tmp_out + sh * D;
  if (np == 1) {  // :582-594
    for (int i = tid; i < D; i += 128) outp[i] = tmp[i];
    return;
  }
  float* smax = reinterpret_cast<float*>(smem);
  float* ssum = smax + max_num_partitions;
  float* red = ssum + max_num_partitions;  // [4]

  float m = -FLT_MAX;
  for (int i = tid; i < np; i += 128) {  // :611-615
    const float l = max_logits[sh + i];
    smax[i] = l;
    m = fmaxf(m, l);
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(red[0], red[1]);

  float g = 0.f;
  for (int i = tid; i < np; i += 128) {  // :644-649
    const float r = exp_sums[sh + i] * __expf(smax[i] - m);
    g += r;
    ssum[i] = r;
  }
  g = wave_sum(g);
  if (lane == 0) red[2 + wave] = g;
  __syncthreads();
  g = red[2] + red[3];
  const float inv = __builtin_amdgcn_rcpf(g + 1e-6f);  // :652

  for (int i = tid; i < D; i += 128) {  // :661-668
    float acc = 0.f;
    for (int j = 0; j < np; ++j)
      acc = __builtin_fmaf((float)tmp[(int64_t)j * D + i] * ssum[j], inv, acc);
    outp[i] = (h16)acc;
  }
}

// ----------------------------------------------------------------------------------------
// reshape_and_cache: scatter new-token K/V rows into the paged caches (pure copy).
// One workgroup per token (reference grid, cache_kernels.cu:274).  Each lane moves 8-dim
// chunks: K as one 16-B store into [blk][h][d/8][off][0..8), V as 8 two-byte stores at
// stride block_size into [blk][h][d..d+8)[off].
// ----------------------------------------------------------------------------------------
template <bool VEC>
__global__ void __launch_bounds__(256)
    reshape_and_cache_kernel(const h16* __restrict__ key, const h16* __restrict__ value,
                             h16* __restrict__ kc, h16* __restrict__ vc,
                             const int64_t* __restrict__ slot_mapping, int64_t key_stride,
                             int64_t value_stride, int H, int D, int BS) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // padding token (ref cache_kernels.cu:165-169)
  const int64_t blk = slot / BS;
  const int64_t off = slot % BS;
  const int n8 = (H * D) >> 3;
  const h16* ksrc = key + token * key_stride;
  const h16* vsrc = value + token * value_stride;
  for (int c = threadIdx.x; c < n8; c += blockDim.x) {
    const int i = c << 3;
    const int h = i / D;
    const int d = i - h * D;
    h16x8 kv, vv;
    if constexpr (VEC) {
      kv = __builtin_bit_cast(h16x8, *reinterpret_cast<const u32x4*>(ksrc + i));
      vv = __builtin_bit_cast(h16x8, *reinterpret_cast<const u32x4*>(vsrc + i));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        kv[e] = ksrc[i + e];
        vv[e] = vsrc[i + e];
      }
    }
    h16* kdst = kc + (((blk * H + h) * (D >> 3) + (d >> 3)) * BS + off) * 8;
    *reinterpret_cast<u32x4*>(kdst) = __builtin_bit_cast(u32x4, kv);
    h16* vdst = vc + ((blk * H + h) * (int64_t)D + d) * BS + off;
#pragma unroll
    for (int e = 0; e < 8; ++e) vdst[(int64_t)e * BS] = vv[e];
  }
}


// ----------------------------------------------------------------------------------------
// copy_blocks: for every layer and every (src, dst) pair copy one K block and one V block
// inside that layer's caches — reference cache_kernels.cu:68-94 (kernel), :96-148 (host).
// The reference uploads two pointer tables with a blocking .to(device) (:119-126); here up to
// 64 layers' pointers ride in the kernel arguments, so the call never synchronises.
// grid = (layers, pairs), 256 threads, 16-B moves.
// ----------------------------------------------------------------------------------------
struct CopyBlocksArgs {
  uint8_t* key[64];
  uint8_t* value[64];
};

__global__ void __launch_bounds__(256)
    copy_blocks_kernel(const CopyBlocksArgs a, const int64_t* __restrict__ block_mapping,
                       int64_t block_bytes) {
  const int layer = blockIdx.x;
  const int pair = blockIdx.y;
  const int64_t src = block_mapping[2 * pair] * block_bytes;      // :77-78
  const int64_t dst = block_mapping[2 * pair + 1] * block_bytes;
  const int64_t n16 = block_bytes >> 4;
  const u32x4* ks = reinterpret_cast<const u32x4*>(a.key[layer] + src);
  u32x4* kd = reinterpret_cast<u32x4*>(a.key[layer] + dst);
  const u32x4* vs = reinterpret_cast<const u32x4*>(a.value[layer] + src);
  u32x4* vd = reinterpret_cast<u32x4*>(a.value[layer] + dst);
  for (int64_t i = threadIdx.x; i < n16; i += 256) kd[i] = ks[i];  // :82-86
  for (int64_t i = threadIdx.x; i < n16; i += 256) vd[i] = vs[i];  // :87-91
}

// ----------------------------------------------------------------------------------------
// diagnostics (not part of the reference surface): what read bandwidth does this box give a
// plain coalesced 16-B/lane stream?  Used by bench.py --diag to state the achievable ceiling
// next to the attention kernel's number.
// ----------------------------------------------------------------------------------------
template <bool NT>
__global__ void __launch_bounds__(256) stream_read_kernel(const u32x4* __restrict__ src, size_t n16,
                                                          uint32_t* __restrict__ sink) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    u32x4 a, b, c, d;
    if constexpr (NT) {
      a = __builtin_nontemporal_load(src + i);
      b = __builtin_nontemporal_load(src + i + stride);
      c = __builtin_nontemporal_load(src + i + 2 * stride);
      d = __builtin_nontemporal_load(src + i + 3 * stride);
    } else {
      a = src[i]; b = src[i + stride]; c = src[i + 2 * stride]; d = src[i + 3 * stride];
    }
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n16; i += stride) acc ^= src[i];
  const uint32_t x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (x == 0x9e3779b9u) sink[0] = x;  // practically never: keeps the loads live
}

// ----------------------------------------------------------------------------------------
// host side: variant table, validation, launch
// ----------------------------------------------------------------------------------------
typedef void (*pa_kernel_t)(const PAParams);

struct Variant {
  const char* name;
  int D, HPW, WPH, U;
  bool NT;
  pa_kernel_t fn;
  int lds_attr_set;  // largest dynamic-LDS size already granted through hipFuncSetAttribute
};

#define VMI_VARIANT(D, HPW, WPH, U, NT)                                          \
  {                                                                              \
    "d" #D "_h" #HPW "_w" #WPH "_u" #U "_nt" #NT, D, HPW, WPH, U, (bool)NT,      \
        (pa_kernel_t)pa_v1_kernel<D, HPW, WPH, U, (bool)NT>, 0                   \
  }

static Variant g_variants[] = {
    // ---- head size 64 ----
    VMI_VARIANT(64, 1, 1, 4, 0),   // 1
    VMI_VARIANT(64, 4, 1, 4, 0),   // 2
    VMI_VARIANT(64, 1, 4, 4, 0),   // 3
    VMI_VARIANT(64, 2, 2, 4, 0),   // 4
    VMI_VARIANT(64, 4, 1, 8, 0),   // 5
    VMI_VARIANT(64, 4, 1, 2, 0),   // 6
    VMI_VARIANT(64, 1, 1, 4, 1),   // 7
    VMI_VARIANT(64, 4, 1, 4, 1),   // 8
    VMI_VARIANT(64, 1, 4, 4, 1),   // 9
    VMI_VARIANT(64, 1, 8, 2, 0),   // 10
    VMI_VARIANT(64, 1, 16, 1, 0),  // 11
    VMI_VARIANT(64, 4, 1, 8, 1),   // 12
    VMI_VARIANT(64, 2, 1, 4, 1),   // 13
    VMI_VARIANT(64, 1, 2, 4, 1),   // 14
    // ---- head size 128 ----
    VMI_VARIANT(128, 1, 1, 2, 0),  // 15
    VMI_VARIANT(128, 4, 1, 2, 0),  // 16
    VMI_VARIANT(128, 1, 4, 2, 0),  // 17
    VMI_VARIANT(128, 4, 1, 4, 0),  // 18
    VMI_VARIANT(128, 1, 1, 2, 1),  // 19
    VMI_VARIANT(128, 4, 1, 2, 1),  // 20
    VMI_VARIANT(128, 1, 4, 2, 1),  // 21
    VMI_VARIANT(128, 1, 8, 2, 0),  // 22
    VMI_VARIANT(128, 1, 16, 1, 0), // 23
    VMI_VARIANT(128, 4, 1, 4, 1),  // 24
    // ---- many waves per head, non-temporal (appended: earlier ids stay stable) ----
    VMI_VARIANT(64, 1, 8, 2, 1),    // 25
    VMI_VARIANT(64, 1, 16, 1, 1),   // 26
    VMI_VARIANT(128, 1, 8, 2, 1),   // 27
    VMI_VARIANT(128, 1, 16, 1, 1),  // 28
    // ---- diagnostics: same gather pattern, no math ("loads only"); wrong results by design ----
    {"d64_h4_w1_u4_nt1_LOADSONLY", 64, 4, 1, 4, true,
     (pa_kernel_t)pa_v1_kernel<64, 4, 1, 4, true, true>, 0},   // 29
    {"d64_h1_w1_u4_nt1_LOADSONLY", 64, 1, 1, 4, true,
     (pa_kernel_t)pa_v1_kernel<64, 1, 1, 4, true, true>, 0},   // 30
};
static const int g_nvariants = (int)(sizeof(g_variants) / sizeof(g_variants[0]));

static int find_variant(int D, int HPW, int WPH, int U, bool NT) {
  for (int i = 0; i < g_nvariants; ++i) {
    const Variant& v = g_variants[i];
    if (v.D == D && v.HPW == HPW && v.WPH == WPH && v.U == U && v.NT == NT &&
        !strstr(v.name, "LOADSONLY"))
      return i + 1;
  }
  return 0;
}

// Heuristic: with >= ~2 waves per SIMD worth of (seq, head) units one wave per head keeps
// every CU streaming with no barriers; below that, deal each head's blocks to more waves.
// Non-temporal page loads pay once the KV working set no longer fits the 256 MiB Infinity
// Cache (cfg3 146 -> 133 us, "long" 195 -> 186 us) and are neutral below it.
static int pick_variant(int num_seqs, int num_heads, int head_size, int max_seq_len) {
  const long units = (long)num_seqs * num_heads;
  const int nblk = (max_seq_len + 15) / 16;
  int wph = 1;
  while (wph < 16 && units * wph < 2048 && wph * 2 <= (nblk > 0 ? nblk : 1)) wph *= 2;
  const double kv_bytes = 4.0 * (double)units * (double)max_seq_len * head_size;
  const bool nt = kv_bytes > 128e6;
  const int hpw = (wph == 1 && num_heads % 4 == 0) ? 4 : 1;
  const int u = (head_size == 64) ? (wph <= 4 ? 4 : (wph == 8 ? 2 : 1)) : (wph <= 8 ? 2 : 1);
  int v = find_variant(head_size, hpw, wph, u, nt);
  if (!v) v = find_variant(head_size, hpw, wph, u, !nt);
  if (!v) v = find_variant(head_size, 1, 1, head_size == 64 ? 4 : 2, true);
  return v;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int launch_pa_v1(void* out, const void* query, const void* key_cache,
                        const void* value_cache, int32_t num_seqs, int32_t num_heads,
                        int32_t head_size, int32_t num_kv_heads, float scale,
                        const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
                        int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                        const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                        int64_t kv_head_stride, int32_t device, void* stream, int32_t variant) {
  if (!out || !query || !key_cache || !value_cache || !block_tables || !seq_lens)
    return fail(VMI_E_NULL_POINTER, "paged_attention_v1: NULL tensor pointer");
  if (head_size != 64 && head_size != 128)
    return fail(VMI_E_HEAD_SIZE, "Unsupported head size: %d", head_size);
  if (block_size != 16) return fail(VMI_E_BLOCK_SIZE, "Unsupported block size: %d", block_size);
  if (num_seqs < 0 || num_heads <= 0 || max_seq_len < 0 || max_num_blocks_per_seq < 0)
    return fail(VMI_E_SHAPE, "paged_attention_v1: negative size (num_seqs=%d num_heads=%d "
                "max_seq_len=%d max_num_blocks_per_seq=%d)", num_seqs, num_heads, max_seq_len,
                max_num_blocks_per_seq);
  if (num_kv_heads <= 0 || num_heads % num_kv_heads != 0)
    return fail(VMI_E_KV_HEADS, "paged_attention_v1: num_heads=%d not divisible by num_kv_heads=%d",
                num_heads, num_kv_heads);
  if (!aligned16(query) || !aligned16(key_cache) || !aligned16(value_cache) || (q_stride & 7) ||
      (kv_block_stride & 7) || (kv_head_stride & 7))
    return fail(VMI_E_ALIGNMENT, "paged_attention_v1: query/key_cache/value_cache and their "
                "strides must be 16-byte aligned (q_stride=%lld kv_block_stride=%lld "
                "kv_head_stride=%lld)", (long long)q_stride, (long long)kv_block_stride,
                (long long)kv_head_stride);
  if (num_seqs == 0) return VMI_OK;

  if (variant == 0) variant = pick_variant(num_seqs, num_heads, head_size, max_seq_len);
  if (variant < 1 || variant > g_nvariants)
    return fail(VMI_E_VARIANT, "paged_attention_v1: unknown variant %d", variant);
  Variant& v = g_variants[variant - 1];
  if (v.D != head_size)
    return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s is for head size %d, got %d", v.name,
                v.D, head_size);
  if (v.WPH > 1 && num_heads % v.HPW != 0)
    return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s needs num_heads %% %d == 0", v.name,
                v.HPW);

  const int lpad = ((max_seq_len + 15) / 16) * 16;
  const size_t lds = (size_t)v.HPW * lpad * 4 + (size_t)v.HPW * 2 * v.WPH * 4 +
                     (size_t)v.HPW * v.WPH * v.D * 4;
  if (lds > 160 * 1024)
    return fail(VMI_E_MAX_SEQ_LEN, "paged_attention_v1: max_seq_len=%d needs %zu B of LDS per "
                "workgroup (variant %s), limit 163840", max_seq_len, lds, v.name);

  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  if (lds > 48 * 1024 && (int)lds > v.lds_attr_set) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(v.fn),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    v.lds_attr_set = (int)lds;
  }

  PAParams p;
  p.out = static_cast<h16*>(out);
  p.q = static_cast<const h16*>(query);
  p.kc = static_cast<const h16*>(key_cache);
  p.vc = static_cast<const h16*>(value_cache);
  p.block_tables = block_tables;
  p.seq_lens = seq_lens;
  p.alibi = alibi_slopes;
  p.num_heads = num_heads;
  p.num_kv_heads = num_kv_heads;
  p.scale = scale;
  p.max_blocks_per_seq = max_num_blocks_per_seq;
  p.q_stride = q_stride;
  p.kv_block_stride = kv_block_stride;
  p.kv_head_stride = kv_head_stride;
  p.lpad = lpad;
  p.exp_sums = nullptr;
  p.max_logits = nullptr;
  p.max_num_partitions = 1;

  dim3 block(v.HPW * v.WPH * 64);
  // gridDim.y is limited to 65535: longer batches go out as consecutive launches over slices
  for (int32_t s0 = 0; s0 < num_seqs; s0 += 65535) {
    const int32_t ns = (num_seqs - s0) < 65535 ? (num_seqs - s0) : 65535;
    PAParams ps = p;
    ps.out = p.out + (int64_t)s0 * num_heads * head_size;
    ps.q = p.q + (int64_t)s0 * q_stride;
    ps.block_tables = p.block_tables + (int64_t)s0 * max_num_blocks_per_seq;
    ps.seq_lens = p.seq_lens + s0;
    dim3 grid((num_heads + v.HPW - 1) / v.HPW, ns, 1);
    hipLaunchKernelGGL(v.fn, grid, block, lds, static_cast<hipStream_t>(stream), ps);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "paged_attention_v1 launch");
  }
  return VMI_OK;
}

// ---- split-KV (paged_attention_v2) variants: same kernel body, PART = true -----------------
#define VMI_VARIANT_V2(D, HPW, WPH, U, NT)                                          \
  {                                                                                 \
    "v2_d" #D "_h" #HPW "_w" #WPH "_u" #U "_nt" #NT, D, HPW, WPH, U, (bool)NT,      \
        (pa_kernel_t)pa_v1_kernel<D, HPW, WPH, U, (bool)NT, false, true>, 0         \
  }
static Variant g_variants_v2[] = {
    VMI_VARIANT_V2(64, 4, 1, 4, 1),   // 1
    VMI_VARIANT_V2(64, 1, 1, 4, 1),   // 2
    VMI_VARIANT_V2(64, 1, 2, 4, 1),   // 3
    VMI_VARIANT_V2(64, 1, 4, 4, 1),   // 4
    VMI_VARIANT_V2(64, 1, 8, 2, 1),   // 5
    VMI_VARIANT_V2(128, 4, 1, 2, 1),  // 6
    VMI_VARIANT_V2(128, 1, 1, 2, 1),  // 7
    VMI_VARIANT_V2(128, 1, 2, 2, 1),  // 8
    VMI_VARIANT_V2(128, 1, 4, 2, 1),  // 9
    VMI_VARIANT_V2(128, 1, 8, 2, 1),  // 10
};
static const int g_nvariants_v2 = (int)(sizeof(g_variants_v2) / sizeof(g_variants_v2[0]));

static int find_variant_v2(int D, int HPW, int WPH) {
  for (int i = 0; i < g_nvariants_v2; ++i) {
    const Variant& v = g_variants_v2[i];
    if (v.D == D && v.HPW == HPW && v.WPH == WPH) return i + 1;
  }
  return 0;
}

// a partition holds at most 32 blocks; give each (seq, head, partition) 1..8 waves so that the
// launch has >= ~2048 waves when the batch allows it
static int pick_variant_v2(int num_seqs, int num_heads, int head_size, int max_seq_len) {
  const int parts = (max_seq_len + 511) / 512;
  const long units = (long)num_seqs * num_heads * (parts > 0 ? parts : 1);
  int wph = 1;
  while (wph < 8 && units * wph < 2048) wph *= 2;
  int v = (wph == 1) ? find_variant_v2(head_size, (num_heads % 4 == 0) ? 4 : 1, 1)
                     : find_variant_v2(head_size, 1, wph);
  return v ? v : find_variant_v2(head_size, 1, 1);
}

static int launch_pa_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
                        const void* key_cache, const void* value_cache, int32_t num_seqs,
                        int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
                        const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
                        int32_t max_seq_len, int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                        int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                        int32_t device, void* stream, int32_t variant) {
  if (!out || !exp_sums || !max_logits || !tmp_out || !query || !key_cache || !value_cache ||
      !block_tables || !seq_lens)
    return fail(VMI_E_NULL_POINTER, "paged_attention_v2: NULL tensor pointer");
  if (head_size != 64 && head_size != 128)
    return fail(VMI_E_HEAD_SIZE, "Unsupported head size: %d", head_size);
  if (block_size != 16) return fail(VMI_E_BLOCK_SIZE, "Unsupported block size: %d", block_size);
  if (num_seqs < 0 || num_heads <= 0 || max_seq_len < 0 || max_num_blocks_per_seq < 0)
    return fail(VMI_E_SHAPE, "paged_attention_v2: negative size");
  if (num_seqs > 65535 || num_heads > 65535)
    return fail(VMI_E_SHAPE, "paged_attention_v2: num_seqs/num_heads above the 65535 grid limit");
  if (num_kv_heads <= 0 || num_heads % num_kv_heads != 0)
    return fail(VMI_E_KV_HEADS, "paged_attention_v2: num_heads=%d not divisible by num_kv_heads=%d",
                num_heads, num_kv_heads);
  if (!aligned16(query) || !aligned16(key_cache) || !aligned16(value_cache) || (q_stride & 7) ||
      (kv_block_stride & 7) || (kv_head_stride & 7))
    return fail(VMI_E_ALIGNMENT, "paged_attention_v2: pointers/strides must be 16-byte aligned");
  const int parts = (max_seq_len + 511) / 512;  // attention_kernels.cu:885
  if (num_seqs == 0 || parts == 0) return VMI_OK;
  if (parts > 65535) return fail(VMI_E_MAX_SEQ_LEN, "paged_attention_v2: too many partitions");
  if (variant == 0) variant = pick_variant_v2(num_seqs, num_heads, head_size, max_seq_len);
  if (variant < 1 || variant > g_nvariants_v2)
    return fail(VMI_E_VARIANT, "paged_attention_v2: unknown variant %d", variant);
  Variant& v = g_variants_v2[variant - 1];
  if (v.D != head_size)
    return fail(VMI_E_VARIANT, "paged_attention_v2: variant %s is for head size %d", v.name, v.D);
  if (v.WPH > 1 && num_heads % v.HPW != 0)
    return fail(VMI_E_VARIANT, "paged_attention_v2: variant %s needs num_heads %% %d == 0", v.name, v.HPW);
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");

  const int lpad = 512;  // one partition of logits (:886)
  const size_t lds = (size_t)v.HPW * lpad * 4 + (size_t)v.HPW * 2 * v.WPH * 4 +
                     (size_t)v.HPW * v.WPH * v.D * 4;
  PAParams p;
  p.out = static_cast<h16*>(tmp_out);
  p.q = static_cast<const h16*>(query);
  p.kc = static_cast<const h16*>(key_cache);
  p.vc = static_cast<const h16*>(value_cache);
  p.block_tables = block_tables;
  p.seq_lens = seq_lens;
  p.alibi = alibi_slopes;
  p.num_heads = num_heads;
  p.num_kv_heads = num_kv_heads;
  p.scale = scale;
  p.max_blocks_per_seq = max_num_blocks_per_seq;
  p.q_stride = q_stride;
  p.kv_block_stride = kv_block_stride;
  p.kv_head_stride = kv_head_stride;
  p.lpad = lpad;
  p.exp_sums = exp_sums;
  p.max_logits = max_logits;
  p.max_num_partitions = parts;
  dim3 grid((num_heads + v.HPW - 1) / v.HPW, num_seqs, parts);  // :890
  hipLaunchKernelGGL(v.fn, grid, dim3(v.HPW * v.WPH * 64), lds, static_cast<hipStream_t>(stream), p);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "paged_attention_v2 launch");

  const size_t rlds = (size_t)(2 * parts + 4) * sizeof(float);  // :894
  dim3 rgrid(num_heads, num_seqs);                               // :893
  if (head_size == 64)
    hipLaunchKernelGGL(pa_v2_reduce_kernel<64>, rgrid, dim3(128), rlds, static_cast<hipStream_t>(stream),
                       static_cast<h16*>(out), exp_sums, max_logits, static_cast<const h16*>(tmp_out),
                       seq_lens, parts);
  else
    hipLaunchKernelGGL(pa_v2_reduce_kernel<128>, rgrid, dim3(128), rlds, static_cast<hipStream_t>(stream),
                       static_cast<h16*>(out), exp_sums, max_logits, static_cast<const h16*>(tmp_out),
                       seq_lens, parts);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "paged_attention_v2 reduce launch");
  return VMI_OK;
}

}  // namespace vmi

// ----------------------------------------------------------------------------------------
// C-ABI
// ----------------------------------------------------------------------------------------
extern "C" {

int vmi_abi_version(void) { return VMI_ABI_VERSION; }
const char* vmi_last_error_string(void) { return vmi::g_err; }
const char* vmi_target_arch(void) { return "gfx950"; }

int vmi_paged_attention_v1_f16(void* out, const void* query, const void* key_cache,
                               const void* value_cache, int32_t num_seqs, int32_t num_heads,
                               int32_t head_size, int32_t num_kv_heads, float scale,
                               const int32_t* block_tables, const int32_t* seq_lens,
                               int32_t block_size, int32_t max_seq_len,
                               int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                               int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                               int32_t device, void* stream) {
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, 0);
}

int vmi_paged_attention_v1_f16_variant(void* out, const void* query, const void* key_cache,
                                       const void* value_cache, int32_t num_seqs,
                                       int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
                                       float scale, const int32_t* block_tables,
                                       const int32_t* seq_lens, int32_t block_size,
                                       int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                                       const float* alibi_slopes, int64_t q_stride,
                                       int64_t kv_block_stride, int64_t kv_head_stride,
                                       int32_t device, void* stream, int32_t variant) {
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant);
}

int vmi_paged_attention_v1_variant_count(void) { return vmi::g_nvariants; }

const char* vmi_paged_attention_v1_variant_name(int32_t variant) {
  if (variant < 1 || variant > vmi::g_nvariants) return "";
  return vmi::g_variants[variant - 1].name;
}

int vmi_paged_attention_v1_pick_variant(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                        int32_t max_seq_len) {
  if (head_size != 64 && head_size != 128) return 0;
  return vmi::pick_variant(num_seqs, num_heads, head_size, max_seq_len);
}

int vmi_reshape_and_cache_f16(const void* key, const void* value, void* key_cache,
                              void* value_cache, const int64_t* slot_mapping, int32_t num_tokens,
                              int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
                              int64_t key_stride, int64_t value_stride, int32_t device,
                              void* stream) {
  using namespace vmi;
  if (!key || !value || !key_cache || !value_cache || !slot_mapping)
    return fail(VMI_E_NULL_POINTER, "reshape_and_cache: NULL tensor pointer");
  if (x != 8) return fail(VMI_E_X, "reshape_and_cache: key_cache.size(4) must be 8, got %d", x);
  if (num_tokens < 0 || num_heads <= 0 || head_size <= 0 || (head_size & 7))
    return fail(VMI_E_SHAPE, "reshape_and_cache: bad sizes (num_tokens=%d num_heads=%d "
                "head_size=%d)", num_tokens, num_heads, head_size);
  if (block_size <= 0)
    return fail(VMI_E_BLOCK_SIZE, "Unsupported block size: %d", block_size);
  if (!aligned16(key_cache))
    return fail(VMI_E_ALIGNMENT, "reshape_and_cache: key_cache must be 16-byte aligned");
  if (num_tokens == 0) return VMI_OK;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  const bool vec = aligned16(key) && aligned16(value) && !(key_stride & 7) && !(value_stride & 7);
  const int n8 = (num_heads * head_size) >> 3;
  int threads = ((n8 + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  dim3 grid(num_tokens), block(threads);
  if (vec) {
    hipLaunchKernelGGL(reshape_and_cache_kernel<true>, grid, block, 0,
                       static_cast<hipStream_t>(stream), static_cast<const h16*>(key),
                       static_cast<const h16*>(value), static_cast<h16*>(key_cache),
                       static_cast<h16*>(value_cache), slot_mapping, key_stride, value_stride,
                       num_heads, head_size, block_size);
  } else {
    hipLaunchKernelGGL(reshape_and_cache_kernel<false>, grid, block, 0,
                       static_cast<hipStream_t>(stream), static_cast<const h16*>(key),
                       static_cast<const h16*>(value), static_cast<h16*>(key_cache),
                       static_cast<h16*>(value_cache), slot_mapping, key_stride, value_stride,
                       num_heads, head_size, block_size);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "reshape_and_cache launch");
  return VMI_OK;
}

int vmi_paged_attention_v2_f16(void* out, void* exp_sums, void* max_logits, void* tmp_out,
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
                           kv_head_stride, device, stream, variant);
}

int vmi_paged_attention_v2_variant_count(void) { return vmi::g_nvariants_v2; }

const char* vmi_paged_attention_v2_variant_name(int32_t variant) {
  if (variant < 1 || variant > vmi::g_nvariants_v2) return "";
  return vmi::g_variants_v2[variant - 1].name;
}

int vmi_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs, int32_t num_layers,
                    const int64_t* block_mapping, int32_t num_pairs, int64_t block_bytes,
                    int32_t device, void* stream) {
  using namespace vmi;
  if (num_layers < 0 || num_pairs < 0 || block_bytes <= 0 || (block_bytes & 15))
    return fail(VMI_E_SHAPE, "copy_blocks: bad sizes (layers=%d pairs=%d block_bytes=%lld)", num_layers,
                num_pairs, (long long)block_bytes);
  if (num_layers == 0 || num_pairs == 0) return VMI_OK;   // cache_kernels.cu:101-103
  if (!key_cache_ptrs || !value_cache_ptrs || !block_mapping)
    return fail(VMI_E_NULL_POINTER, "copy_blocks: NULL pointer");
  if (num_pairs > 65535) return fail(VMI_E_SHAPE, "copy_blocks: more than 65535 pairs in one call");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  for (int l0 = 0; l0 < num_layers; l0 += 64) {
    const int nl = (num_layers - l0) < 64 ? (num_layers - l0) : 64;
    CopyBlocksArgs a;
    memset(&a, 0, sizeof(a));
    for (int l = 0; l < nl; ++l) {
      if (!key_cache_ptrs[l0 + l] || !value_cache_ptrs[l0 + l] || !aligned16(key_cache_ptrs[l0 + l]) ||
          !aligned16(value_cache_ptrs[l0 + l]))
        return fail(VMI_E_ALIGNMENT, "copy_blocks: layer %d cache pointer NULL or not 16-byte aligned", l0 + l);
      a.key[l] = static_cast<uint8_t*>(key_cache_ptrs[l0 + l]);
      a.value[l] = static_cast<uint8_t*>(value_cache_ptrs[l0 + l]);
    }
    hipLaunchKernelGGL(copy_blocks_kernel, dim3(nl, num_pairs), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a, block_mapping, block_bytes);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "copy_blocks launch");
  }
  return VMI_OK;
}

int vmi_swap_blocks(const void* src, void* dst, const int64_t* block_mapping_host, int32_t num_pairs,
                    int64_t block_bytes, int32_t kind, int32_t device, void* stream) {
  using namespace vmi;
  if (num_pairs < 0 || block_bytes <= 0) return fail(VMI_E_SHAPE, "swap_blocks: bad sizes");
  if (num_pairs == 0) return VMI_OK;
  if (!src || !dst || !block_mapping_host) return fail(VMI_E_NULL_POINTER, "swap_blocks: NULL pointer");
  hipMemcpyKind k;
  switch (kind) {  // cache_kernels.cu:28-40
    case 0: k = hipMemcpyDeviceToDevice; break;
    case 1: k = hipMemcpyDeviceToHost; break;
    case 2: k = hipMemcpyHostToDevice; break;
    default: return fail(VMI_E_SHAPE, "Invalid device combination");
  }
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  const char* s = static_cast<const char*>(src);
  char* d = static_cast<char*>(dst);
  for (int i = 0; i < num_pairs; ++i) {  // :56-62
    const int64_t so = block_mapping_host[2 * i] * block_bytes;
    const int64_t doff = block_mapping_host[2 * i + 1] * block_bytes;
    e = hipMemcpyAsync(d + doff, s + so, (size_t)block_bytes, k, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "swap_blocks hipMemcpyAsync");
  }
  return VMI_OK;
}

int vmi_diag_stream_read(const void* src, int64_t bytes, void* sink, int32_t blocks, int32_t nt,
                         int32_t device, void* stream) {
  using namespace vmi;
  if (!src || !sink || bytes < 16 || blocks <= 0) return fail(VMI_E_SHAPE, "diag_stream_read: bad args");
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  const size_t n16 = (size_t)bytes / 16;
  if (nt)
    hipLaunchKernelGGL(stream_read_kernel<true>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const u32x4*>(src), n16, static_cast<uint32_t*>(sink));
  else
    hipLaunchKernelGGL(stream_read_kernel<false>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const u32x4*>(src), n16, static_cast<uint32_t*>(sink));
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "diag_stream_read launch");
  return VMI_OK;
}

}  // extern "C"
