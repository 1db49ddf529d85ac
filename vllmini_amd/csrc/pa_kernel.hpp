// pa_kernel.hpp — kernel templates of the MI355X paged-attention decode path (gfx950 only).
//
//   device helpers        16-B loads, bf16 helpers, dot8, fp8 E4M3 / E5M2 decode (deq8, dot16_f8*), PV8, wave reductions
//   PAParams              kernel argument block (v1, v2 partitions, fused append, fp8 scale, block-sparse pattern)
//   pa_v1_kernel          THE attention kernel: paged_attention_v1, the partition pass of paged_attention_v2 (PART),
//                         the fused append (APP), fp16 / bf16 query (BF), 16-bit or fp8 pages (F8 = 1 E4M3, 2 E5M2),
//                         grouped-query tile sharing with q.K^T on MFMA (GQS), opt-in P.V on MFMA (FPV),
//                         block-sparse attention (SPARSE)
//   pa_v2_reduce_kernel   merge of the 512-token partitions
//   Variant, VMI_ROW*     one row of a kernel menu (pa_table_*.inc) and the tables' extern declarations
//   PAF32Params           float32 tensors have kernels of their own (pa_f32.hip)
//
// Instantiated by the translation units listed in vllmini_amd/build.py (they compile concurrently):
// paged_attention.hip (core menu + host code + C-ABI), pa_variants_{extra,bf16}.hip, pa_append_{core,extra,bf16}.hip,
// pa_variants_fp8{,_bf16,_e5m2,_e5m2_bf16}.hip, pa_variants_sparse{,_bf16}.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <type_traits>

namespace vmi {

typedef _Float16 h16;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));  // MFMA operand type only
// LDS holds the probabilities as 16-bit patterns and they are read 8 at a time: the vector view must alias them
typedef uint32_t u32x4_alias __attribute__((ext_vector_type(4), may_alias));
typedef float f32x4_alias __attribute__((ext_vector_type(4), may_alias));  // four logits written at once (MFMA path)

// ----------------------------------------------------------------------------------------
// device helpers
// ----------------------------------------------------------------------------------------

template <bool NT>
__device__ __forceinline__ u32x4 ld16(const void* p) {
  if constexpr (NT) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  } else {
    return *reinterpret_cast<const u32x4*>(p);
  }
}

// ---- bfloat16 helpers (element type selected by the BF template flag) ---------------------------
// Reference arithmetic (dtype_bfloat16.cuh): operands widen to fp32 by a 16-bit shift, __hmul2 rounds each
// product to bf16 (RNE), sums are fp32, stores round to bf16 (RNE).  fp32 holds bf16 products exactly, so
// "fp32 multiply, then round the upper 16 bits" reproduces __hmul2.
__device__ __forceinline__ float bf_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ float bf_round(float x) {  // x rounded to bf16 precision, kept in an fp32
  uint32_t u = __builtin_bit_cast(uint32_t, x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return __builtin_bit_cast(float, u & 0xffff0000u);
}
template <bool BF>
__device__ __forceinline__ uint16_t to_elem(float x) {  // fp32 -> fp16 / bf16 bit pattern, RNE
  if constexpr (BF) return (uint16_t)(__builtin_bit_cast(uint32_t, bf_round(x)) >> 16);
  else return __builtin_bit_cast(uint16_t, (h16)x);
}
template <bool BF>
__device__ __forceinline__ float from_elem(uint16_t b) {
  if constexpr (BF) return __builtin_bit_cast(float, (uint32_t)b << 16);
  else return (float)__builtin_bit_cast(h16, b);
}

// q.k over one 16-byte chunk pair (8 dims): fp32 FMA chain on widened operands
template <bool BF>
__device__ __forceinline__ float dot8(const u32x4 q, const u32x4 k) {
  if constexpr (BF) {
    float a = bf_lo(q[0]) * bf_lo(k[0]);
    a = __builtin_fmaf(bf_hi(q[0]), bf_hi(k[0]), a);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      a = __builtin_fmaf(bf_lo(q[w]), bf_lo(k[w]), a);
      a = __builtin_fmaf(bf_hi(q[w]), bf_hi(k[w]), a);
    }
    return a;
  } else {
    const h16x8 qh = __builtin_bit_cast(h16x8, q);
    const h16x8 kh = __builtin_bit_cast(h16x8, k);
    float a = (float)qh[0] * (float)kh[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) a = __builtin_fmaf((float)qh[e], (float)kh[e], a);
    return a;
  }
}

// ---- fp8 E4M3 KV cache (kv_cache_dtype "fp8"): every cache element becomes float_to_half(float(fp8) * kv_scale)
// first (reference quant_utils.cuh:295-300), then the fp16 arithmetic applies unchanged.  v_cvt_pk_f32_fp8 decodes
// two OCP E4M3 bytes exactly on gfx950; the multiply and the RNE conversion to half are separate instructions
// (-ffp-contract=off). ----
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// E5 = the cache bytes are fp8 E5M2 (kv_cache_dtype "fp8_e5m2": the upper byte of an IEEE half, "bf8" in the ISA)
// instead of E4M3; same element definition half(float(fp8) * kv_scale), same instruction count.
template <bool E5, bool HI>
__device__ __forceinline__ f32x2_t cvt2_f32_f8(uint32_t w) {  // bytes (0,1) or (2,3) of w -> two floats, exact
  if constexpr (E5) return __builtin_amdgcn_cvt_pk_f32_bf8((int)w, HI);
  else return __builtin_amdgcn_cvt_pk_f32_fp8((int)w, HI);
}
template <bool E5, bool HI>
__device__ __forceinline__ uint32_t cvt2_f16_f8(uint32_t w) {  // ... -> two halves (every fp8 value is a half value)
  if constexpr (E5) {  // an E5M2 byte IS the upper byte of its half
    return HI ? (((w >> 8) & 0x0000ff00u) | (w & 0xff000000u)) : (((w & 0xffu) << 8) | ((w & 0xff00u) << 16));
  } else {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)w, 1.0f, HI));
  }
}
// 8 fp8 bytes (two dwords) -> 8 halves packed like a 16-byte fp16 unit.
// S1 (kv_scale == 1): every E4M3 value is a float16 value, so half(float(fp8) * 1) is the byte's own value and
// v_cvt_scalef32_pk_f16_fp8 with scale 1.0 produces it directly, two per instruction.
template <bool S1, bool BF = false, bool E5 = false>
__device__ __forceinline__ u32x4 deq8(uint32_t w0, uint32_t w1, float s) {
  if constexpr (BF) {
    // bfloat16 query: element = __float2bfloat16(float(fp8) * kv_scale) (quant_utils.cuh:350-359); with kv_scale == 1
    // the multiply is skipped (the fp32 decode has at most 4 significant bits: it already is a bfloat16 value)
    u32x4 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t w = h ? w1 : w0;
      const f32x2_t lo = cvt2_f32_f8<E5, false>(w);
      const f32x2_t hi = cvt2_f32_f8<E5, true>(w);
      if constexpr (S1) {
        // (the upper 16 bits of each decode would do — the value is exact in bfloat16 — but hipcc 7.2 turns
        //  `(bits(lo[0]) >> 16) | (bits(lo[1]) & 0xffff0000)` into a pack that reads lo[0] twice; measured on gfx950,
        //  odd tokens received their even neighbour's value.  The rounding form below is immune and also exact.)
        o[2 * h] = (uint32_t)to_elem<true>(lo[0]) | ((uint32_t)to_elem<true>(lo[1]) << 16);
        o[2 * h + 1] = (uint32_t)to_elem<true>(hi[0]) | ((uint32_t)to_elem<true>(hi[1]) << 16);
      } else {
        o[2 * h] = (uint32_t)to_elem<true>(lo[0] * s) | ((uint32_t)to_elem<true>(lo[1] * s) << 16);
        o[2 * h + 1] = (uint32_t)to_elem<true>(hi[0] * s) | ((uint32_t)to_elem<true>(hi[1] * s) << 16);
      }
    }
    return o;
  } else if constexpr (S1) {
    u32x4 o;
    o[0] = cvt2_f16_f8<E5, false>(w0);
    o[1] = cvt2_f16_f8<E5, true>(w0);
    o[2] = cvt2_f16_f8<E5, false>(w1);
    o[3] = cvt2_f16_f8<E5, true>(w1);
    return o;
  } else {
    h16x8 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t w = h ? w1 : w0;
      const f32x2_t lo = cvt2_f32_f8<E5, false>(w);  // bytes 0, 1
      const f32x2_t hi = cvt2_f32_f8<E5, true>(w);   // bytes 2, 3
      o[4 * h + 0] = (h16)(lo[0] * s);
      o[4 * h + 1] = (h16)(lo[1] * s);
      o[4 * h + 2] = (h16)(hi[0] * s);
      o[4 * h + 3] = (h16)(hi[1] * s);
    }
    return __builtin_bit_cast(u32x4, o);
  }
}
// q.k over one 16-byte fp8 chunk (16 dims): fp32 FMA chain on widened operands, like dot8.
// S1: the fp32 decode IS the operand (half(fp8) widens back to the same fp32), so q (f16) x k (f32) goes straight
// into v_fma_mix_f32 — 8 decodes + 16 FMAs for 16 dims.
template <bool S1, bool BF = false, bool E5 = false>
__device__ __forceinline__ float dot16_f8(const u32x4 q0, const u32x4 q1, const u32x4 k, float s) {
  static_assert(!S1, "kv_scale == 1 uses dot16_f8_s1");
  return dot8<BF>(q0, deq8<false, BF, E5>(k[0], k[1], s)) + dot8<BF>(q1, deq8<false, BF, E5>(k[2], k[3], s));
}
// kv_scale == 1: half(float(fp8)) widens back to the same fp32, so the fp32 decode IS the operand.  q is held as
// fp32 pairs (converted once per wave), v_cvt_pk_f32_fp8 yields k as fp32 pairs, and v_pk_fma_f32 does two exact
// fp32 FMAs per instruction: 8 decodes + 8 packed FMAs for 16 dims (even and odd dims are separate chains).
template <bool E5 = false>
__device__ __forceinline__ float dot16_f8_s1(const f32x2_t (&qf)[8], const u32x4 k) {
  f32x2_t acc = {0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    acc = __builtin_elementwise_fma(qf[2 * w], cvt2_f32_f8<E5, false>(k[w]), acc);
    acc = __builtin_elementwise_fma(qf[2 * w + 1], cvt2_f32_f8<E5, true>(k[w]), acc);
  }
  return acc[0] + acc[1];
}

// p.v over 8 tokens of one dim row; the probabilities come from LDS already rounded to fp16 / bf16.
//   fp16: p -> fp16, 4 packed fp16 products, packed fp16 adds ((p0v0+p2v2)+p4v4)+p6v6 / odd, fp32 add of halves
//   bf16: p -> bf16, products rounded to bf16, fp32 sums ((s01+s23)+s45)+s67
template <bool BF>
struct PV8 {
  h16x8 ph;     // fp16 probabilities
  float pf[8];  // bf16-rounded probabilities held in fp32
  // the 8 probabilities as the softmax left them in LDS: fp16 (bf16) bit patterns, already rounded
  __device__ __forceinline__ void load(const u32x4 raw) {
    if constexpr (BF) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        pf[2 * w] = bf_lo(raw[w]);
        pf[2 * w + 1] = bf_hi(raw[w]);
      }
    } else {
      ph = __builtin_bit_cast(h16x8, raw);
    }
  }
  // MASK = false is the steady state: no tail handling is compiled in at all (the compiler would
  // otherwise if-convert the wave-uniform `last` test into v_cndmask on every block).
  template <bool MASK>
  __device__ __forceinline__ float dot(const u32x4 vraw, bool last, int token0, int L) const {
    if constexpr (BF) {
      float s[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float v0 = bf_lo(vraw[w]), v1 = bf_hi(vraw[w]);
        if constexpr (MASK) {
          if (last) {  // tokens past the context may hold stale/NaN bytes: zero them (ref :420-430)
            v0 = (token0 + 2 * w < L) ? v0 : 0.f;
            v1 = (token0 + 2 * w + 1 < L) ? v1 : 0.f;
          }
        }
        s[w] = bf_round(pf[2 * w] * v0) + bf_round(pf[2 * w + 1] * v1);
      }
      return ((s[0] + s[1]) + s[2]) + s[3];
    } else {
      h16x8 v = __builtin_bit_cast(h16x8, vraw);
      if constexpr (MASK) {
        if (last) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (token0 + e < L) ? v[e] : (h16)0.f;
        }
      }
      const h16x8 pr = ph * v;  // 4 x v_pk_mul_f16, each product rounded to fp16
      h16x2 c = h16x2{pr[0], pr[1]} + h16x2{pr[2], pr[3]};
      c = c + h16x2{pr[4], pr[5]};
      c = c + h16x2{pr[6], pr[7]};
      // (fp32 add of the two halves, dtype_float16.cuh:439-443 — written as fma(c0, 1, c1): the same single rounding.  The
      //  compiler folds it back into two conversions and an add; forcing ONE v_fma_mix_f32 with inline asm — a third of the
      //  instructions, bit-identical — was measured in round 4: fp8 pages 66.3 -> 65.3 us on equal lengths, level on ragged
      //  ones, and the fp16 headline kernels LOST 1 - 3 % (122.2 -> 123.6, ragged 70.8 -> 73.1: the asm pins the schedule of a
      //  kernel that sits exactly on its 168-VGPR budget), so it is not used: profiles/r04_fp8_ragged_accounting.md)
      return __builtin_fmaf((float)c[0], 1.0f, (float)c[1]);
    }
  }
};

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

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(v, m);
    v = v > o ? v : o;
  }
  return v;
}

// Batch statistics that every wave of a launch computes for itself from the same seq_lens (clamped to [0, lpad]), in
// the same order of operations — so all waves, and the two kernels of a gated double launch (launch_pa_v1), reach
// the same verdict without talking to each other.  8 loads in flight per trip; each(i, clamped_len) sees every
// sequence once per wave.
template <typename F>
__device__ __forceinline__ void batch_stats(const int32_t* __restrict__ seq_lens, int B, int lpad, int lane, int& maxL,
                                            float& sum, F&& each) {
  maxL = 0;
  sum = 0.f;
  for (int base = 0; base < B; base += 512) {
    int l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * 64 + lane;
      l[k] = i < B ? seq_lens[i] : -1;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * 64 + lane;
      const int c = l[k] < 0 ? 0 : (l[k] > lpad ? lpad : l[k]);
      if (i < B) {
        maxL = maxL > c ? maxL : c;
        sum += (float)c;
        each(i, c);
      }
    }
  }
  maxL = wave_max_i(maxL);
  sum = wave_sum(sum);
}
// a batch is RAGGED when its mean length is below 0.8 of its longest
__device__ __forceinline__ bool batch_is_ragged(int maxL, float sum, int B) { return sum < 0.8f * (float)maxL * (float)B; }
// q_flags bits shared by pa_v1_kernel and pa_q_kernel: a gated double launch runs the kernel built for equal lengths
// and the balanced kernel back to back, and each leaves at once unless the batch is its kind
constexpr int QF_GATE_UNIFORM = 1 << 16;  // pa_v1_kernel: leave unless the batch has (nearly) equal lengths
constexpr int QF_GATE_RAGGED = 1 << 17;   // pa_q_kernel: leave unless the batch is ragged

// Workgroup barrier for waves that exchange data through LDS only.  __syncthreads() is a release/acquire fence at
// workgroup scope plus s_barrier, and the fence makes every wave wait for ALL its outstanding global loads
// (s_waitcnt vmcnt(0)) — here that would be the V pages requested just before the softmax, i.e. the prefetch would
// overlap nothing.  Only this wave's LDS writes have to have landed (lgkmcnt(0)) before the others read them.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Diagnostic library only (-DVMI_DIAG): pa_v1_kernel writes ten stage stamps per wave — ticks of the constant 100 MHz
// clock at entry, lengths known, first pages requested, first K group consumed, K pass done, max exchanged, probabilities
// written, V pass done, partial outputs exchanged, out stored — to g_stage_stamps (vmi_diag_set_stage_stamps;
// scripts/stage_timeline_probe.py turns them into the account of a launch's latency chain).  One copy per translation
// unit (no relocatable device code): the setter lives beside the core menu in paged_attention.hip and serves its kernels.
// With VMI_DIAG undefined VMI_STAMP expands to nothing and the kernel text is what it was.
#ifdef VMI_DIAG
static __device__ uint64_t* g_stage_stamps = nullptr;  // [waves of the launch][12]: 10 stamps, HW_ID, XCC_ID | nmy << 8
#define VMI_STAMP(k) do { if (tl_) ts_[k] = wall_clock64(); } while (0)
#else
#define VMI_STAMP(k) ((void)0)
#endif

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
  // fused append (vmi_paged_attention_v1_append_*), v1 only; key == nullptr -> plain paged_attention_v1.
  // This step's rows [num_seqs, num_kv_heads, D] are stored into slot seq_len-1 of each sequence's last block
  // (what cache_ops.reshape_and_cache does with slot = table[(L-1)/BS]*BS + (L-1)%BS, cache_kernels.cu:219-260)
  // and the attention takes that token from the rows themselves, so the two ops need no launch boundary.
  const h16* key;
  const h16* value;
  int64_t key_stride;
  int64_t value_stride;
  float kv_scale;  // fp8 cache only (F8): cache element = fp8(x / kv_scale)
  // block-sparse attention (SPARSE kernels only; the operator's tp_rank + four blocksparse_* arguments,
  // attention_kernels.cu:108-110): a cache block is read when its sparse block is "remote" or "local" (:232-254)
  int32_t bs_tp_rank, bs_local_blocks, bs_vert_stride, bs_block_size, bs_head_sliding_step;
  // balanced kernels (pa_queue.hpp) only: the grid is not (heads, seqs), so the batch size rides here;
  // q_flags = test / experiment knobs, 0 = automatic
  int32_t num_seqs;
  int32_t q_flags;
  // fused append: bit 0 = do NOT store the rows (vmi_paged_attention_v1_newest_*: the attention takes the newest token from
  // key / value, the caller writes the cache itself — e.g. one reshape_and_cache for all the layers of a token)
  int32_t app_flags;
};

// ----------------------------------------------------------------------------------------
// paged_attention_v1 / the partition kernel of paged_attention_v2
//
//   D     head size (any multiple of 8; the reference set is 64..256)
//   HPW   head SLOTS per workgroup        (each slot owns WPH waves)
//   WPH   waves per slot                  (the blocks of a slot are dealt round-robin to them)
//   HPT   ADJACENT heads per slot/wave    (a wave reads the HPT tiles of a block as one contiguous
//                                          HPT*D*BS*2-byte chunk: bigger chunks are served faster by HBM)
//   GQS   grouped-query attention: the HPT query heads of a slot share ONE KV head (num_heads / num_kv_heads is a
//         multiple of HPT), so the wave loads each K / V tile ONCE and uses it for all HPT heads — the reference
//         (one workgroup per query head, attention_kernels.cu:153) re-reads it per query head and leans on L2
//   U     blocks per register group       (group g+1 is in flight while group g is consumed)
//   UMAX  adaptive queue depth: a wave whose sequence is >= 1.4x / 2.8x the launch's mean length runs the
//         2U / 4U-deep form of the same code (0 = fixed U) — see the selection code at the end of the kernel
//   NT    non-temporal page loads
//   PART  split-KV form behind paged_attention_v2 (reference attention_kernels.cu:529-562: the same
//         kernel body with PARTITION_SIZE = 512): blockIdx.z selects a 512-token partition, the
//         partition's normalised output goes to tmp_out and its (max, exp_sum) to max_logits / exp_sums
//   BS    block size 8 | 16 | 32;  BF  bfloat16 instead of float16 elements
//   LOCK  the waves of a workgroup issue each page group together (one s_barrier per group): with
//         WPH = 1 they own adjacent head slots, so HBM sees one HPW*HPT-tile burst per block
//
// Lane maps (BS and D general):
//   K tile = D/8 chunks x BS tokens of 16 B; a load covers 64/BS chunks; lane = chunk*BS + token
//   V tile = D rows x BS/8 units of 16 B;   a load covers 512/BS rows;  lane = row*(BS/8) + unit
//   when D*BS/8 is not a multiple of 64 (head 80/112, ...) the last load of a tile is predicated.
//
// grid = (ceil(H / (HPW*HPT)), num_seqs[, partitions]), block = HPW*WPH*64.
// LDS  = HPW*HPT * ( lpad*4 (logits) + 2*WPH*4 (max/sum exchange) + WPH*D*4 (partial out)
//                    + (WPH > 1 ? lpad*2 : 0) (fp16 probabilities; in place over the logits when WPH = 1) ).
// ----------------------------------------------------------------------------------------
template <int D, int HPW, int WPH, int U, bool NT, bool LOADS_ONLY = false, bool PART = false, int BS = 16,
          bool LOCK = false, bool BF = false, int HPT = 1, bool APP = false, int UMAX = 0, int F8 = 0,
          bool GQS = false, bool FPV = false, bool SPARSE = false>
// (second launch bound = minimum waves per SIMD.  The adaptive-depth kernels are the full-chip defaults: 12 waves
//  per CU = 3 per SIMD that must ALL be resident, i.e. stay under 170 VGPRs — the fused-append form had drifted to 180
//  and ran 173 us instead of 125.  Not applied elsewhere: on the big-tile kernels it only forces spills.)
__global__ void __launch_bounds__(HPW* WPH * 64, (UMAX > 0 && WPH == 1) ? 3 : 1)
    pa_v1_kernel(const PAParams p) {
  constexpr int PBLK = 512 / BS;          // blocks per partition (PARTITION_SIZE = 512, :847)
  // F8: the caches hold fp8 E4M3 bytes — key_cache [NB, H, D/16, BS, 16], value_cache [NB, H, D, BS]
  // (x = 16 / sizeof(cache_t), attention_kernels.cu:200): a 16-byte unit carries 16 elements instead of 8
  constexpr bool E5 = F8 == 2;            // F8: 0 = 16-bit caches, 1 = fp8 E4M3 bytes, 2 = fp8 E5M2 bytes
  constexpr int EPU = F8 ? 16 : 8;        // cache elements per 16-B unit
  constexpr int ES = F8 ? 1 : 2;          // bytes per cache element
  constexpr int UNITS = D * BS / EPU;     // 16-B units in one (block, head) tile of K — and of V
  constexpr int NL = (UNITS + 63) / 64;   // 1-KiB loads per tile
  constexpr int TAIL = UNITS - 64 * (NL - 1);  // active lanes of the last load (64 = full)
  constexpr int CPL = 64 / BS;            // K: chunks per load
  constexpr int UPR = BS / EPU;           // V: 16-B units per dim row
  constexpr int RPL = 64 / UPR;           // V: rows per load
  static_assert(D % 8 == 0, "head size must be a multiple of 8");
  static_assert(BS == 8 || BS == 16 || BS == 32, "block size 8, 16 or 32");
  static_assert(64 % U == 0, "U must divide 64");
  static_assert(!(LOCK && WPH > 1), "lockstep needs every wave to run the same number of page groups");
  static_assert(!(APP && (PART || LOADS_ONLY)), "the fused append exists for paged_attention_v1 only");
  // (GQS with HPT == 1 — multi-head attention through the matrix-core code paths — was measured over fp8 pages:
  //  cfg3 67.1 -> 63.9 us, cfg4 338 -> 339 us; those kernels are bound by the 2-KiB tile request pattern, not the VALU)
  static_assert(!GQS || HPT > 1, "GQS shares one KV tile between HPT > 1 query heads");
  constexpr int RH = GQS ? 1 : HPT;  // K / V register tiles per block: one per KV head this wave reads
  // FPV ("fast P.V", opt-in, never picked automatically): grouped-query kernels that ALSO run the probabilities x V
  // contraction on the matrix cores.  Exact fp16 products summed in fp32 — it drops the reference's fp16 rounding of
  // every product and pair sum (dtype_float16.cuh:118-124, 399-404), so results sit within the north-star 1e-3 of the
  // reference but not within an ulp of it.
  static_assert(!FPV || (GQS && BS == 16 && D % 32 == 0 && U % (F8 ? 4 : 2) == 0 && !LOADS_ONLY && UMAX == 0),
                "FPV: grouped-query kernels, block size 16, pairs (fp8 pages: quads) of blocks per register group");
  static_assert(!SPARSE || (HPT == 1 && !APP && !F8 && !LOADS_ONLY && UMAX == 0 && !LOCK),
                "block-sparse kernels: one head per wave (the stripe pattern slides per head), 16-bit caches");
  constexpr int VG = F8 ? 4 : 2;  // FPV: blocks that share one MFMA's K = 32 tokens (8 per block with fp8 pages, 16 else)
  static_assert(!F8 || (BS >= 16 && D % 16 == 0 && !APP && !LOADS_ONLY),
                "fp8 cache: block size 16 or 32 (a V row must fill whole 16-byte units), no fused append");

  extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef VMI_DIAG
  uint64_t* const tl_ = g_stage_stamps;
  uint64_t ts_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  VMI_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = wave / WPH;   // head slot
  const int sub = wave % WPH;  // wave within the slot
  const int seq = blockIdx.y;

  const int head0 = (blockIdx.x * HPW + hl) * HPT;
  // valid heads of this slot (wave-uniform).  With WPH > 1 the host guarantees H % (HPW*HPT) == 0, so every
  // wave reaches every barrier; with WPH == 1 there are no barriers except LOCK's, which ignores ended waves.
  const int nh = (p.num_heads - head0) < HPT ? (p.num_heads - head0) : HPT;
  if (WPH == 1 && nh <= 0) return;
  auto valid = [&](int hh) { return HPT == 1 || hh < nh; };

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
  const int Lfull = L;
  // UMAX: 64 sequence lengths sampled evenly across this launch (requested now, used when the queue depth is chosen)
  int samp = 0;
  if constexpr (UMAX >= 2 * U && !PART) samp = p.seq_lens[(int)(((int64_t)lane * gridDim.y) >> 6)];
  if constexpr (!PART) L = L > p.lpad ? p.lpad : L;
  if constexpr (!PART && !APP && !SPARSE) {
    if (p.q_flags & QF_GATE_UNIFORM) {  // gated double launch: a ragged batch belongs to the balanced kernel behind me
      int mx;
      float sm;
      batch_stats(p.seq_lens, p.num_seqs, p.lpad, lane, mx, sm, [](int, int) {});
      if (batch_is_ragged(mx, sm, p.num_seqs)) return;  // the same verdict in every wave of the launch
    }
  }
  const int nblk_seq = (L + BS - 1) / BS;                                     // :121
  const int blk_hi = PART ? (blk_lo + PBLK < nblk_seq ? blk_lo + PBLK : nblk_seq) : nblk_seq;  // :128-129
  if (PART && blk_lo * BS >= L) return;  // nothing in this partition (:116-119); uniform per workgroup
  const int nblk = blk_hi - blk_lo;      // blocks in my range
  const int tok_lo = blk_lo * BS;        // logits in LDS are indexed relative to the range start (:133)
  const int Lloc = (L < blk_hi * BS ? L : blk_hi * BS) - tok_lo;              // tokens in range (:134-136)

  float* smem_f = reinterpret_cast<float*>(smem);
  float* logits0 = smem_f + (size_t)(hl * HPT) * p.lpad;                                    // + hh*lpad
  float* red0 = smem_f + (size_t)HPW * HPT * p.lpad + (hl * HPT) * 2 * WPH;                 // + hh*2*WPH
  float* osm0 = smem_f + (size_t)HPW * HPT * p.lpad + HPW * HPT * 2 * WPH + (size_t)(hl * HPT) * WPH * D;
  // Probabilities as fp16 / bf16 bit patterns, written ONCE per token after the softmax (the V pass used to convert
  // exp * inv_sum again in every lane that met the token: D/8-fold redundant VALU work).  One wave per head: in
  // place over the fp32 values (a value is read before any lane overwrites it — see the normalise loop); WPH waves
  // per head: a region of their own behind the exchange buffers, because another wave may not have read yet.
  const int ph_stride = WPH == 1 ? 2 * p.lpad : p.lpad;  // uint16 units between the heads of a slot
  uint16_t* ph0 =
      WPH == 1 ? reinterpret_cast<uint16_t*>(logits0)
               : reinterpret_cast<uint16_t*>(smem_f + (size_t)HPW * HPT * p.lpad + HPW * HPT * 2 * WPH +
                                             (size_t)HPW * HPT * WPH * D) + (size_t)(hl * HPT) * p.lpad;

  const int64_t ostride = PART ? (int64_t)p.max_num_partitions * D : D;  // out elements between heads
  uint16_t* out0 = reinterpret_cast<uint16_t*>(p.out) +
                   (PART ? (((int64_t)seq * p.num_heads + head0) * p.max_num_partitions + part) * D
                         : ((int64_t)seq * p.num_heads + head0) * D);

  if (L <= 0) {  // uniform over the workgroup (same seq): reference yields exp_sum = 0 -> out = 0
    if (sub == 0) {
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh)
        if (valid(hh))
          for (int d = lane; d < D; d += 64) out0[hh * ostride + d] = 0;  // +0.0 in fp16 and in bf16
    }
    return;
  }
  VMI_STAMP(1);

  // ---- per head: tile offset inside a block, ALiBi slope, this lane's q chunks (one per K load) ----
  const int c4 = lane / BS;  // chunk-within-load
  const int tk = lane % BS;  // token-within-block
  const bool tail_ok = (TAIL == 64) || lane < TAIL;  // this lane takes part in the last load of a tile
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int qpk = p.num_heads / p.num_kv_heads;
  // SPARSE (attention_kernels.cu:209-254, 385-393): cache block b is attended when the sparse block holding its first
  // token is "remote" — on this head's vertical stripe — or "local" — within local_blocks of the query's sparse block.
  // The attended blocks are listed up front (ablk, below) and the passes walk that list, so the loads in flight are
  // all useful ones; skipped blocks get logits -FLT_MAX (exp -> 0) and no P.V contribution.
  int bs_off = 0, q_bs = 0;
  if constexpr (SPARSE) {
    q_bs = (L - 1) / p.bs_block_size;  // :215
    bs_off = p.bs_head_sliding_step >= 0
                 ? (p.bs_tp_rank * p.num_heads + head0) * p.bs_head_sliding_step + 1          // :216-219
                 : (p.bs_tp_rank * p.num_kv_heads + head0 / qpk) * (-p.bs_head_sliding_step) + 1;  // :220-224
  }
  auto attended = [&](int b) -> bool {
    if constexpr (!SPARSE) return true;
    const int kb = b * BS / p.bs_block_size;  // :235
    return ((kb + bs_off) % p.bs_vert_stride == 0) || (kb > q_bs - p.bs_local_blocks);
  };
  int64_t hoff[HPT];
  float slope[HPT];
  u32x4 qreg[HPT][NL][F8 ? 2 : 1];  // the EPU dims of q that face this lane's chunk of each K load
#pragma unroll
  for (int hh = 0; hh < HPT; ++hh) {
    const int head = head0 + (valid(hh) ? hh : 0);
    hoff[hh] = (int64_t)(head / qpk) * p.kv_head_stride + lane * EPU;
    slope[hh] = p.alibi ? p.alibi[head] : 0.f;
    const h16* qp = p.q + (int64_t)seq * p.q_stride + (int64_t)head * D;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
      for (int w = 0; w < (F8 ? 2 : 1); ++w)
        qreg[hh][i][w] = (i < NL - 1 || tail_ok)
                             ? *reinterpret_cast<const u32x4*>(qp + (CPL * i + c4) * EPU + 8 * w)
                             : zero4;
    }
  }

  // ---- fused append (APP): this step's key/value rows, in the lane map of the K / V tiles.  They are patched
  //      into the page registers where the attention meets position Lfull-1; whatever bytes the cache holds there
  //      at that moment (old, or already the writer workgroup's) are never used. ----
  // APP_TILE: write the patched last-block tiles back whole (full 128-B lines, non-temporal) instead of the
  // token's 16-B / 2-B pieces; costs NL*4 registers per head to hold the K tile until the end, so only for HPT = 1.
  constexpr bool APP_TILE = APP && HPT == 1;
  u32x4 klast[APP_TILE ? NL : 1], vlast[APP_TILE ? NL : 1];
  bool own_last = false;              // wave-uniform: this wave met block lbA
  int32_t phys_last = 0;              // ... and its physical block id (the table slice in bt_reg moves on afterwards)
  const int lbA = (Lfull - 1) / BS;   // block and in-block offset of the appended token
  const int offA = (Lfull - 1) % BS;
  // The step's rows in the lane map of the K / V tiles: knew_at(hh, i) = this lane's 16-B chunk of load i,
  // vnew_at(hh, i) = the element of this lane's dim row.  One head per wave (APP_TILE): preloaded next to q.
  // Several heads per wave: fetched where they are used (once per wave, at its last block and in the epilogue) —
  // holding them costs HPT*NL*5 registers, which pushed the 4-heads-per-wave kernel into spills.
  auto knew_at = [&](int hh, int i) -> u32x4 {
    const int kvh = (head0 + (valid(hh) ? hh : 0)) / qpk;
    const h16* kr = p.key + (int64_t)seq * p.key_stride + (int64_t)kvh * D;
    return (i < NL - 1 || tail_ok) ? *reinterpret_cast<const u32x4*>(kr + (CPL * i + c4) * 8) : zero4;
  };
  auto vnew_at = [&](int hh, int i) -> uint32_t {
    const int kvh = (head0 + (valid(hh) ? hh : 0)) / qpk;
    const h16* vr = p.value + (int64_t)seq * p.value_stride + (int64_t)kvh * D;
    const int row = (64 / (BS / 8)) * i + lane / (BS / 8);
    return row < D ? (uint32_t)__builtin_bit_cast(uint16_t, vr[row]) : 0u;
  };
  u32x4 knew[APP_TILE ? NL : 1];
  uint32_t vnew[APP_TILE ? NL : 1];
  if constexpr (APP_TILE) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      knew[i] = knew_at(0, i);
      vnew[i] = vnew_at(0, i);
    }
  }

  // ---- grouped-query kernels compute q.K^T on the matrix cores: with HPT query heads per tile it is a (16 tokens) x
  //      (HPT heads, padded to 16) x (D dims) product, one v_mfma_f32_16x16x32 per 1-KiB K load.  The K tile is ALREADY
  //      in the A-operand layout (lane = chunk*16 + token holds 8 dims of one token); B is q with lane & 15 = head.
  //      Products of 16-bit operands are exact in fp32 and the accumulation is fp32 — the reference's arithmetic up to
  //      summation order (dtype_float16.cuh:292-298).  The V pass keeps its fp16 rounding points on the VALU. ----
  //      (fp8 pages: the 16-byte chunk is decoded ONCE into two 8-dim operands, whatever the number of heads)
  //      (Tried for the one-head-per-wave fp8 kernels as well: no gain — 73 vs 67-71 us on cfg3 — a single useful MFMA
  //       column does not pay for the different logits write pattern.)
  constexpr bool QK_MFMA = GQS && BS == 16 && TAIL == 64 && !LOADS_ONLY;
  u32x4 qB[QK_MFMA ? NL : 1][F8 ? 2 : 1];
  float slopeB = 0.f;
  if constexpr (QK_MFMA) {
    const int n = lane & 15;
    const bool has = n < HPT;
    const h16* qp = p.q + (int64_t)seq * p.q_stride + (int64_t)(head0 + (has ? n : 0)) * D;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
      for (int w = 0; w < (F8 ? 2 : 1); ++w)
        qB[i][w] = has ? *reinterpret_cast<const u32x4*>(qp + (CPL * i + c4) * EPU + 8 * w) : zero4;
    }
    slopeB = (has && p.alibi) ? p.alibi[head0 + n] : 0.f;
  }
  float qmaxB = -FLT_MAX;  // QK_MFMA: running max of head (lane & 15) over this lane's token rows

  // ---- my share of the blocks: b = blk_lo + sub + idx*WPH, idx in [0, nmy) ----------------
  //      (SPARSE: the idx-th block of this wave is entry sub + idx*WPH of the list of attended blocks)
  int natt = nblk;
  int32_t* ablk = nullptr;
  if constexpr (SPARSE) {
    // every wave builds the same ascending list (same values to the same LDS words: no barrier needed before a wave
    // reads what it wrote itself); the -FLT_MAX fill of skipped blocks is shared out by 64-block chunk and is
    // complete at the barrier in front of the softmax
    ablk = reinterpret_cast<int32_t*>(smem_f + (size_t)HPW * HPT * p.lpad + HPW * HPT * 2 * WPH +
                                      (size_t)HPW * HPT * WPH * D + (WPH > 1 ? (size_t)HPW * HPT * p.lpad / 2 : 0)) +
           (size_t)hl * (p.lpad / 8);
    natt = 0;
    for (int c = 0; c * 64 < nblk; ++c) {
      const int b = blk_lo + c * 64 + lane;
      const bool in = c * 64 + lane < nblk;
      const bool att = in && attended(b);
      const uint64_t m = __ballot(att);
      if (att) ablk[natt + __popcll(m & ((1ull << lane) - 1ull))] = b;
      if (in && !att && (c % WPH) == sub) {  // :240-252
        for (int t = 0; t < BS; ++t) logits0[(b - blk_lo) * BS + t] = -FLT_MAX;
      }
      natt += __popcll(m);
    }
    bt_sg = -1;  // the table prefetch above was for the dense order
  }
  const int nmy = natt > sub ? (natt - sub + WPH - 1) / WPH : 0;
  auto block_of = [&](int idx) -> int {  // wave-uniform
    if constexpr (SPARSE) return __builtin_amdgcn_readfirstlane(ablk[sub + idx * WPH]);
    else return blk_lo + sub + idx * WPH;
  };

  float qk_max[HPT];
#pragma unroll
  for (int hh = 0; hh < HPT; ++hh) qk_max[hh] = -FLT_MAX;
  uint32_t fold = 0;  // LOADS_ONLY diagnostic: xor of everything loaded
  float inv_sum[HPT];
  float acc[HPT][NL];
#pragma unroll
  for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[hh][i] = 0.f;
  // FPV: out[head][dim] accumulators in the MFMA C/D layout — tile t covers dims 16t..16t+15; lane holds column
  // (lane & 15) = dim, rows 4*(lane >> 4) + reg = head
  f32x4 accM[FPV ? D / 16 : 1];
#pragma unroll
  for (int t = 0; t < (FPV ? D / 16 : 1); ++t) accM[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int hf = lane % UPR;   // which 8-token group of the block this lane owns
  const int rowl = lane / UPR;  // dim row within a load

  auto store_tile = [&](h16* cache, u32x4(&t)[NL]) {  // APP_TILE (HPT = 1)
    if ((head0 % qpk) == 0) {  // one writer per KV head
      const int64_t phys = phys_last;
      h16* dst = cache + phys * p.kv_block_stride + hoff[0];
#pragma unroll
      for (int i = 0; i < NL; ++i)
        if (i < NL - 1 || tail_ok) __builtin_nontemporal_store(t[i], reinterpret_cast<u32x4*>(dst + i * 512));
    }
  };

  // The K pass, the softmax and the V pass for a compile-time group size UU (blocks per register group).
  auto run = [&](auto utag, auto s1tag) {
    constexpr int UU = decltype(utag)::value;
    constexpr bool S1 = decltype(s1tag)::value;  // fp8 cache with kv_scale == 1: cheaper, bit-identical dequantisation
    f32x2_t qf[S1 ? HPT : 1][S1 ? NL : 1][8];    // S1: q as fp32 pairs for v_pk_fma_f32
    if constexpr (S1) {
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const u32x4 qraw = qreg[hh][i][F8 ? w : 0];
            const h16x8 qh = __builtin_bit_cast(h16x8, qraw);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              qf[hh][i][4 * w + e] = BF ? f32x2_t{bf_lo(qraw[e]), bf_hi(qraw[e])}
                                        : f32x2_t{(float)qh[2 * e], (float)qh[2 * e + 1]};
          }
    }
    const int ngroups = (nmy + UU - 1) / UU;
    auto table_for = [&](int g) {  // lane j: physical id of my block (bt_sg*64 + j)
      const int sg = (g * UU) >> 6;
      if (sg != bt_sg) {
        int b = blk_lo + sub + (sg * 64 + lane) * WPH;
        if constexpr (SPARSE) {
          const int a = sub + (sg * 64 + lane) * WPH;
          b = a < natt ? ablk[a] : 0;
        }
        bt_reg = (b < p.max_blocks_per_seq) ? bt[b] : 0;
        bt_sg = sg;
        // The wait for this load belongs INSIDE the branch.  Left to the compiler it lands at the join in front of
        // the v_readlane that follows — as s_waitcnt vmcnt(0) on EVERY page group: the group in flight was drained
        // before the next one was requested, i.e. the "register double buffer" of round 1 never had two groups in
        // flight (found in the ISA this round; profiles/r02b_call_pair_aftermath.md, r02f_fp8.md).
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bt_reg));
      }
    };

    auto load_group = [&](u32x4(&r)[UU][RH][NL], const h16* cache, int g) {
      if constexpr (LOCK) __builtin_amdgcn_s_barrier();
      table_for(g);
  #pragma unroll
      for (int j = 0; j < UU; ++j) {
        int idx = g * UU + j;
        idx = idx < nmy ? idx : nmy - 1;  // padding slots re-read my last block (never OOB)
        const int64_t phys = __builtin_amdgcn_readlane(bt_reg, idx & 63);
        // (sc0/sc1 cache-policy bits and buffer- vs flat-addressed loads were measured neutral on
        //  this stream; only `nt` pays: profiles/r01_cfg3_sweep_cache_policy_bits.json)
        const char* blk = reinterpret_cast<const char*>(cache) + phys * p.kv_block_stride * ES;
  #pragma unroll
        for (int hh = 0; hh < RH; ++hh) {
  #pragma unroll
          for (int i = 0; i < NL; ++i)  // masked lanes / absent heads contribute zeros
            r[j][GQS ? 0 : hh][i] = (valid(hh) && (i < NL - 1 || tail_ok)) ? ld16<NT>(blk + (hoff[hh] + i * 64 * EPU) * ES) : zero4;
        }
      }
    };

    // FPV: the V pages in the B-operand layout of v_mfma_f32_16x16x32 — N = 16 dims per instruction, K = 32 tokens.
    // 16-bit pages: lane (n = lane & 15, kg = lane >> 4) takes dim row 16t+n, tokens 8*(kg & 1).. of block (kg >> 1)
    // of a PAIR of my blocks.  fp8 pages: a 16-byte unit is a whole 16-token row, so lane (n, kg) takes row 16t+n of
    // block kg of a QUAD and feeds two instructions (tokens 0-7 and 8-15 of the four blocks).  Either way one 1-KiB
    // request per load instruction (512-B / 256-B runs), D/16 = VG*NL of them per pair / quad, kept in the register
    // slots of those VG tiles.
    const int vm_n = lane & 15, vm_mem = F8 ? (lane >> 4) : (lane >> 5), vm_hf = F8 ? 0 : ((lane >> 4) & 1);
    auto load_group_vm = [&](u32x4(&r)[UU][RH][NL], int g) {
      if constexpr (LOCK) __builtin_amdgcn_s_barrier();
      table_for(g);
  #pragma unroll
      for (int q = 0; q < UU / VG; ++q) {
        int64_t phys = 0;
  #pragma unroll
        for (int k = 0; k < VG; ++k) {
          int ik = g * UU + VG * q + k;
          ik = ik < nmy ? ik : nmy - 1;
          const int64_t pk = __builtin_amdgcn_readlane(bt_reg, ik & 63);
          phys = (vm_mem == k) ? pk : phys;
        }
        const char* blk = reinterpret_cast<const char*>(p.vc) + phys * p.kv_block_stride * ES;
        const int64_t base = (hoff[0] - lane * EPU) * ES;  // the KV head's tile, bytes
  #pragma unroll
        for (int t = 0; t < VG * NL; ++t)
          r[VG * q + t / NL][0][t % NL] = ld16<NT>(blk + base + ((16 * t + vm_n) * UPR + vm_hf) * 16);
      }
    };
    auto load_v = [&](u32x4(&r)[UU][RH][NL], int g) {
      if constexpr (FPV) load_group_vm(r, g);
      else load_group(r, p.vc, g);
    };

    // =========================== K pass: logits -> LDS, running max ========================
    auto fold_all = [&](u32x4(&r)[UU][RH][NL]) {
  #pragma unroll
      for (int j = 0; j < UU; ++j)
  #pragma unroll
        for (int hh = 0; hh < RH; ++hh)
  #pragma unroll
          for (int i = 0; i < NL; ++i) fold ^= r[j][GQS ? 0 : hh][i][0] ^ r[j][GQS ? 0 : hh][i][1] ^ r[j][GQS ? 0 : hh][i][2] ^ r[j][GQS ? 0 : hh][i][3];
    };
    // `final_tag` is a compile-time tag: true only at the call sites that handle a wave's LAST page group — the
    // only place the appended token can be met — so the steady-state loop body carries no append code.
    auto compute_k = [&](auto final_tag, u32x4(&r)[UU][RH][NL], int g) {
      constexpr bool FINAL = decltype(final_tag)::value;
      if constexpr (LOADS_ONLY) {
        fold_all(r);
        return;
      }
  #pragma unroll
      for (int j = 0; j < UU; ++j) {
        const int idx = g * UU + j;
        if (idx < nmy) {  // wave-uniform
          const int b = block_of(idx);
          const int token = b * BS + tk;
          const bool masked = token >= L;
          if constexpr (QK_MFMA) {
            if constexpr (APP && FINAL) {
              if (b == lbA) {  // the appended token's K row goes into the tile before it is multiplied
  #pragma unroll
                for (int i = 0; i < NL; ++i) r[j][0][i] = (tk == offA) ? knew_at(0, i) : r[j][0][i];
                own_last = true;
              }
            }
            f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
  #pragma unroll
            for (int i = 0; i < NL; ++i) {
  #pragma unroll
              for (int w = 0; w < (F8 ? 2 : 1); ++w) {
                u32x4 a = r[j][0][i];
                if constexpr (F8) a = deq8<S1, BF, E5>(r[j][0][i][2 * w], r[j][0][i][2 * w + 1], p.kv_scale);
                if constexpr (BF)
                  d4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                               __builtin_bit_cast(bf16x8, qB[i][w]), d4, 0, 0, 0);
                else
                  d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a),
                                                              __builtin_bit_cast(h16x8, qB[i][w]), d4, 0, 0, 0);
              }
            }
            // C/D layout: column = lane & 15 (head), rows = 4*(lane >> 4) + reg (tokens of this block)
            const int tok4 = b * BS + 4 * c4;
            f32x4 lg4;
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              float qk = p.scale * d4[e];
              qk += (slopeB != 0.f) ? slopeB * (float)(tok4 + e - L + 1) : 0.f;
              const bool m = tok4 + e >= L;
              lg4[e] = m ? 0.f : qk;
              qmaxB = m ? qmaxB : fmaxf(qmaxB, qk);
            }
            if ((lane & 15) < HPT)
              *reinterpret_cast<f32x4_alias*>(logits0 + (lane & 15) * p.lpad + tok4 - tok_lo) = lg4;
            continue;
          }
  #pragma unroll
          for (int hh = 0; hh < HPT; ++hh) {
            if (valid(hh)) {
              // q.k over this lane's 8*NL dims: operands widened to fp32, fp32 FMA chain (v_fma_mix_f32 for
              // fp16) — the reference's arithmetic (dtype_float16.cuh:292-298, 399-404).  One accumulator
              // per load keeps NL independent dependency chains in flight.
              if constexpr (APP && FINAL) {
                if (b == lbA && (!GQS || hh == 0)) {  // wave-uniform, once per wave (and per shared tile) at most
  #pragma unroll
                  for (int i = 0; i < NL; ++i) {
                    const u32x4 kn = APP_TILE ? knew[APP_TILE ? i : 0] : knew_at(hh, i);
                    r[j][GQS ? 0 : hh][i] = (tk == offA) ? kn : r[j][GQS ? 0 : hh][i];
                    if constexpr (APP_TILE) klast[i] = r[j][GQS ? 0 : hh][i];
                  }
                  own_last = true;
                  phys_last = __builtin_amdgcn_readlane(bt_reg, idx & 63);  // this (final) group's slice is still in bt_reg
                }
              }
              float accv[NL];
  #pragma unroll
              for (int i = 0; i < NL; ++i) {
                if constexpr (F8 && S1) accv[i] = dot16_f8_s1<E5>(qf[hh][i], r[j][GQS ? 0 : hh][i]);
                else if constexpr (F8) accv[i] = dot16_f8<false, BF, E5>(qreg[hh][i][0], qreg[hh][i][1], r[j][GQS ? 0 : hh][i], p.kv_scale);
                else accv[i] = dot8<BF>(qreg[hh][i][0], r[j][GQS ? 0 : hh][i]);
              }
              float acc = accv[0];
  #pragma unroll
              for (int i = 1; i < NL; ++i) acc += accv[i];
  #pragma unroll
              for (int m = BS; m < 64; m <<= 1) acc += __shfl_xor(acc, m);  // lanes holding the same token
              float qk = p.scale * acc;
              qk += (slope[hh] != 0.f) ? slope[hh] * (float)(token - L + 1) : 0.f;
              if (lane < BS) logits0[hh * p.lpad + token - tok_lo] = masked ? 0.f : qk;
              qk_max[hh] = masked ? qk_max[hh] : fmaxf(qk_max[hh], qk);
            }
          }
        }
      }
    };

    // Register double buffer over page groups: group g+1 is in flight while group g is consumed.
    // (A third stage was measured and changed nothing; profiles/r01c_cfg3_variant_sweep.json.)
    u32x4 ra[UU][RH][NL], rb[UU][RH][NL];
    // Latency regime (small batches): when everything this wave owns fits ONE register group, the V pages are
    // requested together with the K pages (into the idle second buffer), so K and V cost one memory round trip
    // between them instead of two in a row.
    const bool single = (ngroups == 1);  // wave-uniform (and workgroup-uniform under LOCK)
    if (single) {
      load_group(ra, p.kc, 0);
      load_v(rb, 0);
      VMI_STAMP(2);
      compute_k(std::integral_constant<bool, APP>{}, ra, 0);
      VMI_STAMP(3);
    } else {
      if (ngroups > 0) load_group(ra, p.kc, 0);
      VMI_STAMP(2);
      int g = 0;
      for (; g + 2 <= ngroups; g += 2) {
        load_group(rb, p.kc, g + 1);
        compute_k(std::false_type{}, ra, g);
#ifdef VMI_DIAG
        if (g == 0) VMI_STAMP(3);
#endif
        if (g + 2 < ngroups) {
          load_group(ra, p.kc, g + 2);
          compute_k(std::false_type{}, rb, g + 1);
        } else {
          compute_k(std::integral_constant<bool, APP>{}, rb, g + 1);  // final group of an even count
        }
      }
      if (g < ngroups) compute_k(std::integral_constant<bool, APP>{}, ra, g);  // final group of an odd count
    }

    // first V group goes out now: HBM stays busy while the softmax runs.
    // VREV: the V pass walks the page groups from the LAST one back to the first.  In the reference's call pair the
    // sequence's last block has just been written by reshape_and_cache with scattered partial stores, and reading those
    // lines is slow (profiles/r01o_call_pair_gap.md): requested here, the wait hides behind the softmax instead of
    // standing at the very end of the wave.  (Only the order of the fp32 accumulation over blocks changes; the fused
    // append walks the same order, so it stays bit-identical to the call pair.)
    constexpr bool VREV = true;
    // VAHEAD (round 4, profiles/r04_underfilled_chip.md): the kernels with TEMPORAL page loads and several waves per head are
    // the ones picked when the launch's working set fits the Infinity Cache and the chip is under-filled — there the launch is a
    // chain, not a stream, and the 1.8 us between the end of a wave's K pass and the start of its V pass (barrier, softmax,
    // barrier: every wave of the chip is in the same phase) left the memory system idle with ONE V group requested.  Both
    // register buffers are free here, so these kernels request the first TWO V groups in front of the softmax.  The
    // full-chip kernels (non-temporal loads, or one wave per head) keep one: there, fewer bytes in flight are faster.
    constexpr bool VAHEAD = VREV && WPH > 1 && !NT;
    VMI_STAMP(4);
    if (ngroups > 1) load_v(ra, VREV ? ngroups - 1 : 0);
    if constexpr (VAHEAD) {
      if (ngroups > 1) load_v(rb, ngroups - 2);
    }

    if constexpr (QK_MFMA) {  // per-head maxima live in lanes (lane & 15) = head: fold the 4 row groups, then hand out
      qmaxB = fmaxf(qmaxB, __shfl_xor(qmaxB, 16));
      qmaxB = fmaxf(qmaxB, __shfl_xor(qmaxB, 32));
  #pragma unroll
      for (int hh = 0; hh < HPT; ++hh) qk_max[hh] = __shfl(qmaxB, hh);
    }

    // =========================== softmax over the logits in LDS ============================
    {
      float m[HPT], es[HPT];
  #pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        m[hh] = wave_max(qk_max[hh]);
        if constexpr (WPH > 1) {
          if (lane == 0) red0[hh * 2 * WPH + sub] = m[hh];
        }
      }
      if constexpr (WPH > 1) {
        lds_barrier();  // also: every wave's logits are in LDS
        VMI_STAMP(5);
  #pragma unroll
        for (int hh = 0; hh < HPT; ++hh) {
          float mm = -FLT_MAX;
  #pragma unroll
          for (int w = 0; w < WPH; ++w) mm = fmaxf(mm, red0[hh * 2 * WPH + w]);
          m[hh] = mm;
        }
      }
  #pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        es[hh] = 0.f;
        if (valid(hh)) {
          float* lg = logits0 + hh * p.lpad;
          float e_sum = 0.f;
          for (int i = sub * 64 + lane; i < Lloc; i += WPH * 64) {
            const float e = __expf(lg[i] - m[hh]);
            lg[i] = e;
            e_sum += e;
          }
          es[hh] = wave_sum(e_sum);
        }
        if constexpr (WPH > 1) {
          if (lane == 0) red0[hh * 2 * WPH + WPH + sub] = es[hh];
        }
      }
      if constexpr (WPH > 1) {
        lds_barrier();
  #pragma unroll
        for (int hh = 0; hh < HPT; ++hh) {
          float ssum = 0.f;
  #pragma unroll
          for (int w = 0; w < WPH; ++w) ssum += red0[hh * 2 * WPH + WPH + w];
          es[hh] = ssum;
        }
      }
  #pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        inv_sum[hh] = __builtin_amdgcn_rcpf(es[hh] + 1e-6f);
        if constexpr (PART) {  // partition statistics for the reduce kernel (:349-357)
          if (valid(hh) && sub == 0 && lane == 0) {
            const int64_t o = ((int64_t)seq * p.num_heads + head0 + hh) * p.max_num_partitions + part;
            p.max_logits[o] = m[hh];
            p.exp_sums[o] = es[hh];
          }
        }
      }
    }

    // ---- normalise: p = exp * inv_sum -> fp16 (bf16), once per token, for the blocks THIS wave consumes below
    //      (so no barrier is needed); positions past the context inside my last block become 0 ----
    if constexpr (!LOADS_ONLY) {
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        if (valid(hh)) {
          const float* lg = logits0 + hh * p.lpad;
          uint16_t* ph = ph0 + hh * ph_stride;
          for (int t = lane; t < nmy * BS; t += 64) {
            int i = (sub + (t / BS) * WPH) * BS + (t % BS);  // relative to tok_lo
            if constexpr (SPARSE) i = (ablk[sub + (t / BS) * WPH] - blk_lo) * BS + (t % BS);
            const float e = lg[i];
            // in place (WPH == 1, i == t): the 64 lanes read fp32 values [t0, t0+64) and then write bytes
            // [2*t0, 2*t0+128), i.e. fp32 slots [t0/2, t0/2+32) — already consumed, or being read by this very access
            ph[i] = i < Lloc ? to_elem<BF>(e * inv_sum[hh]) : (uint16_t)0;
          }
        }
      }
    }

    VMI_STAMP(6);
    // =========================== V pass ====================================================
    // `masked` is a compile-time tag: only the LAST page group of a wave can contain the sequence's last
    // block, so only that call site compiles the tail masking in.
    auto compute_v = [&](auto masked, u32x4(&r)[UU][RH][NL], int g) {
      constexpr bool MASK = decltype(masked)::value;
      if constexpr (LOADS_ONLY) {
        fold_all(r);
        return;
      }
      if constexpr (FPV) {
  #pragma unroll
        for (int q = 0; q < UU / VG; ++q) {
          const int i0 = g * UU + VG * q;
          if (i0 < nmy) {  // wave-uniform
            const bool live = i0 + vm_mem < nmy;  // the last pair / quad may be partly empty
            const int b = blk_lo + sub + (i0 + vm_mem) * WPH;
            const int tokb = b * BS + 8 * vm_hf;
            // A operand: probabilities of head (lane & 15) for this lane's 8 tokens (fp8 pages: 8 + 8)
            u32x4 pa[F8 ? 2 : 1];
  #pragma unroll
            for (int hs = 0; hs < (F8 ? 2 : 1); ++hs)
              pa[hs] = (live && vm_n < HPT)
                           ? *reinterpret_cast<const u32x4_alias*>(ph0 + vm_n * ph_stride + (tokb + 8 * hs - tok_lo))
                           : zero4;
            uint16_t vnew16[VG * NL];
            bool patch = false;
            if constexpr (APP && MASK) {
              patch = live && b == lbA && vm_hf == (offA >> 3);
              if (patch) {  // per lane: the half-wave that holds block lbA's tokens
                const int kvh = head0 / qpk;
                const h16* vr = p.value + (int64_t)seq * p.value_stride + (int64_t)kvh * D;
  #pragma unroll
                for (int t = 0; t < VG * NL; ++t) vnew16[t] = __builtin_bit_cast(uint16_t, vr[16 * t + vm_n]);
              }
            }
  #pragma unroll
            for (int t = 0; t < VG * NL; ++t) {
              const u32x4 raw = r[VG * q + t / NL][0][t % NL];
  #pragma unroll
              for (int hs = 0; hs < (F8 ? 2 : 1); ++hs) {
                u32x4 v = raw;
                if constexpr (F8) v = deq8<S1, BF, E5>(raw[2 * hs], raw[2 * hs + 1], p.kv_scale);
                v = live ? v : zero4;
                const int token0 = tokb + 8 * hs;
                if constexpr (APP && MASK) {
                  if (patch) {
                    const int e = offA & 7;
  #pragma unroll
                    for (int w = 0; w < 4; ++w) {
                      const uint32_t old = v[w], vb = vnew16[t];
                      const uint32_t patched = (e & 1) ? ((old & 0x0000ffffu) | (vb << 16)) : ((old & 0xffff0000u) | vb);
                      v[w] = ((e >> 1) == w) ? patched : old;
                    }
                  }
                }
                if constexpr (MASK) {  // elements past the context are zeroed, as the reference does (:420-430): 0 * NaN
  #pragma unroll
                  for (int w = 0; w < 4; ++w) {
                    const uint32_t keep = (token0 + 2 * w < L ? 0x0000ffffu : 0u) | (token0 + 2 * w + 1 < L ? 0xffff0000u : 0u);
                    v[w] &= keep;
                  }
                }
                if constexpr (BF)
                  accM[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pa[hs]),
                                                                    __builtin_bit_cast(bf16x8, v), accM[t], 0, 0, 0);
                else
                  accM[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, pa[hs]),
                                                                   __builtin_bit_cast(h16x8, v), accM[t], 0, 0, 0);
              }
            }
          }
        }
        return;
      }
  #pragma unroll
      for (int j = 0; j < UU; ++j) {
        const int idx = g * UU + j;
        if (idx < nmy) {  // wave-uniform
          const int b = block_of(idx);
          const int token0 = b * BS + hf * EPU;
          const bool last = (b == nblk_seq - 1);  // last block of the SEQUENCE (:420); wave-uniform
          // grouped-query kernels over fp8 pages: the shared tile is decoded ONCE here, not once per query head (the
          // compiler merged only part of the per-head decodes: 40 conversions per block instead of 16)
          constexpr bool DQ1 = GQS && F8;
          u32x4 vdq[DQ1 ? NL : 1][2];
          if constexpr (DQ1) {
  #pragma unroll
            for (int i = 0; i < NL; ++i) {
              vdq[i][0] = deq8<S1, BF, E5>(r[j][0][i][0], r[j][0][i][1], p.kv_scale);
              vdq[i][1] = deq8<S1, BF, E5>(r[j][0][i][2], r[j][0][i][3], p.kv_scale);
            }
          }
  #pragma unroll
          for (int hh = 0; hh < HPT; ++hh) {
            if (valid(hh)) {
              const uint16_t* php = ph0 + hh * ph_stride + (token0 - tok_lo);
              if constexpr (APP && MASK) {  // the appended token lives in the sequence's last block -> final group only
                if (b == lbA && hf == (offA >> 3) && (!GQS || hh == 0)) {
                  const int e = offA & 7;
  #pragma unroll
                  for (int i = 0; i < NL; ++i) {
  #pragma unroll
                    for (int w = 0; w < 4; ++w) {
                      const uint32_t old = r[j][GQS ? 0 : hh][i][w];
                      const uint32_t vb = APP_TILE ? vnew[APP_TILE ? i : 0] : vnew_at(hh, i);
                      const uint32_t patched = (e & 1) ? ((old & 0x0000ffffu) | (vb << 16)) : ((old & 0xffff0000u) | vb);
                      r[j][GQS ? 0 : hh][i][w] = ((e >> 1) == w) ? patched : old;
                    }
                  }
                }
                // held until the epilogue: with WPH > 1 a barrier follows; keeping the store behind it keeps the epilogue short
                if constexpr (APP_TILE) {
                  if (b == lbA) {
#pragma unroll
                    for (int i = 0; i < NL; ++i) vlast[i] = r[j][GQS ? 0 : hh][i];
                  }
                }
              }
              PV8<BF> pv;
              pv.load(*reinterpret_cast<const u32x4_alias*>(php));
  #pragma unroll
              for (int i = 0; i < NL; ++i) {
                if constexpr (DQ1)
                  acc[hh][i] += pv.template dot<MASK>(vdq[DQ1 ? i : 0][0], last, token0, L);
                else if constexpr (F8)  // first 8 of the unit's 16 tokens
                  acc[hh][i] += pv.template dot<MASK>(deq8<S1, BF, E5>(r[j][GQS ? 0 : hh][i][0], r[j][GQS ? 0 : hh][i][1], p.kv_scale), last, token0, L);
                else
                  acc[hh][i] += pv.template dot<MASK>(r[j][GQS ? 0 : hh][i], last, token0, L);
              }
              if constexpr (F8) {  // the other 8 tokens: their own fp16 probability vector, fp32 accumulation
                PV8<BF> pw;
                pw.load(*reinterpret_cast<const u32x4_alias*>(php + 8));
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                  if constexpr (DQ1)
                    acc[hh][i] += pw.template dot<MASK>(vdq[DQ1 ? i : 0][1], last, token0 + 8, L);
                  else
                    acc[hh][i] += pw.template dot<MASK>(deq8<S1, BF, E5>(r[j][GQS ? 0 : hh][i][2], r[j][GQS ? 0 : hh][i][3], p.kv_scale), last, token0 + 8, L);
                }
              }
            }
          }
        }
      }
    };

    if (single) {
      compute_v(std::true_type{}, rb, 0);
    } else if constexpr (VREV) {
      // processing step s handles group ngroups-1-s; step 0 (the masked "final" group) is in ra, odd steps in rb
      const int last = ngroups - 1;
      if (ngroups > 1) {  // (a wave without blocks — more waves than blocks — has ngroups == 0; one group is `single`)
        if constexpr (!VAHEAD) load_v(rb, last - 1);
        compute_v(std::true_type{}, ra, last);
        int s = 1;
        for (; s + 1 < ngroups; s += 2) {
          load_v(ra, last - (s + 1));
          compute_v(std::false_type{}, rb, last - s);
          if (s + 2 < ngroups) load_v(rb, last - (s + 2));
          compute_v(std::false_type{}, ra, last - (s + 1));
        }
        if (s < ngroups) compute_v(std::false_type{}, rb, last - s);
      }
    } else {
      int g = 0;
      for (; g + 2 <= ngroups; g += 2) {
        load_v(rb, g + 1);
        compute_v(std::false_type{}, ra, g);
        if (g + 2 < ngroups) {
          load_v(ra, g + 2);
          compute_v(std::false_type{}, rb, g + 1);
        } else {
          compute_v(std::true_type{}, rb, g + 1);  // final group of an even count
        }
      }
      if (g < ngroups) compute_v(std::true_type{}, ra, g);  // final group of an odd count
    }
    VMI_STAMP(7);
  };

  // Queue depth per wave.  On a full chip the shallowest queue is the fastest when every sequence has the same
  // length, but a wave moves at most its bytes in flight per memory round trip: in a RAGGED batch
  // the long sequences are left running alone at that rate after the short ones have finished (cfg3 with
  // seq_lens ~ U{1..1024}: 101 us for 50 % of the bytes).  Giving every wave a queue proportional to its share of
  // the work, U_i ~ L_i / mean(L), makes them finish together while the total bytes in flight stay what the
  // uniform case uses.  mean(L) is estimated from 64 sequence lengths sampled evenly across the launch.
  if constexpr (UMAX >= 2 * U && !PART) {
    const float mean_len = wave_sum((float)(samp > 0 ? samp : 0)) * (1.f / 64.f);
    const float ratio = (float)Lfull / fmaxf(mean_len, 1.f);
    const int level = __builtin_amdgcn_readfirstlane(ratio >= 2.8f ? 2 : (ratio >= 1.4f ? 1 : 0));
    if (UMAX >= 4 * U && level == 2)
      run(std::integral_constant<int, (UMAX >= 4 * U ? 4 * U : U)>{}, std::false_type{});
    else if (level >= 1)
      run(std::integral_constant<int, 2 * U>{}, std::false_type{});
    else
      run(std::integral_constant<int, U>{}, std::false_type{});
  } else if constexpr (F8) {
    if (p.kv_scale == 1.0f) run(std::integral_constant<int, U>{}, std::true_type{});
    else run(std::integral_constant<int, U>{}, std::false_type{});
  } else {
    run(std::integral_constant<int, U>{}, std::false_type{});
  }

  if constexpr (LOADS_ONLY) {
    if (fold == 0x9e3779b9u) out0[lane] = 1;  // practically never; keeps the loads live
    return;
  }

  // the UPR lanes of a row hold its 8-token groups
  if constexpr (!FPV) {
#pragma unroll
    for (int hh = 0; hh < HPT; ++hh) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
#pragma unroll
        for (int m = 1; m < UPR; m <<= 1) acc[hh][i] += __shfl_xor(acc[hh][i], m);
      }
    }
  }

  if constexpr (FPV) {  // accM: column (lane & 15) = dim 16t + n, rows 4*(lane >> 4) + reg = head
    const int n = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int t = 0; t < D / 16; ++t) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int hh = 4 * g4 + e;
        if (hh < HPT && valid(hh)) {
          if constexpr (WPH > 1) osm0[(hh * WPH + sub) * D + 16 * t + n] = accM[t][e];
          else out0[hh * ostride + 16 * t + n] = to_elem<BF>(accM[t][e]);
        }
      }
    }
  }
  if constexpr (WPH > 1) {
    if (hf == 0 && !FPV) {
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const int row = RPL * i + rowl;
          if (row < D) osm0[(hh * WPH + sub) * D + row] = acc[hh][i];
        }
      }
    }
    lds_barrier();
    VMI_STAMP(8);
    if (sub == 0) {
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        if (valid(hh)) {
          for (int d = lane; d < D; d += 64) {
            float ssum = 0.f;
#pragma unroll
            for (int w = 0; w < WPH; ++w) ssum += osm0[(hh * WPH + w) * D + d];
            out0[hh * ostride + d] = to_elem<BF>(ssum);
          }
        }
      }
    }
  } else {
    if (hf == 0 && !FPV) {
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        if (valid(hh)) {
#pragma unroll
          for (int i = 0; i < NL; ++i) {
            const int row = RPL * i + rowl;
            if (row < D) out0[hh * ostride + row] = to_elem<BF>(acc[hh][i]);
          }
        }
      }
    }
  }

#ifdef VMI_DIAG
  if (tl_) {  // (the wave's stores have left: the last stamp is the end of its work)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ts_[9] = wall_clock64();
    if (lane == 0) {
      const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      uint64_t* rec = tl_ + (wg * (HPW * WPH) + wave) * 12;
#pragma unroll
      for (int k = 0; k < 10; ++k) rec[k] = ts_[k];
      rec[10] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);                               // HW_REG_HW_ID
      rec[11] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((uint64_t)nmy << 8);       // HW_REG_XCC_ID
    }
  }
#endif
  // APP: the cache write — what cache_ops.reshape_and_cache does for slot (table[lbA], offA),
  // cache_kernels.cu:219-260 — is the wave's LAST act.  Measured on the 124-us cfg3 launch
  // (profiles/r01f_fused_append_experiments.md): the same stores issued between the K and V passes +14 us (loads
  // and stores share vmcnt on gfx9-family ISAs, so the next wait on a page load also waits for the write
  // acknowledgement); a writer wave per workgroup +18 us and writer workgroups +10 us (writes into the middle of
  // the read stream); here +5 us (pieces) / +3 us (whole tiles, non-temporal).
  if constexpr (APP_TILE) {
    if (own_last && !(p.app_flags & 1)) {
      store_tile(const_cast<h16*>(p.kc), klast);
      store_tile(const_cast<h16*>(p.vc), vlast);
    }
  } else if constexpr (APP) {
    if (sub == 0 && lbA < p.max_blocks_per_seq && !(p.app_flags & 1)) {
      const int64_t phys = bt[lbA];
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        if (valid(hh) && (head0 + hh) % qpk == 0) {  // one writer per KV head
          h16* kdst = const_cast<h16*>(p.kc) + phys * p.kv_block_stride + hoff[hh];
          h16* vtile = const_cast<h16*>(p.vc) + phys * p.kv_block_stride + hoff[hh] - lane * 8;
#pragma unroll
          for (int i = 0; i < NL; ++i) {
            // K: the lane holding (chunk, token offA) of load i owns exactly those 16 bytes of the tile
            if (tk == offA && (i < NL - 1 || tail_ok)) *reinterpret_cast<u32x4*>(kdst + i * 512) = knew_at(hh, i);
            // V: element offA of dim row `row`
            const int row = RPL * i + rowl;
            if (hf == (offA >> 3) && row < D)
              reinterpret_cast<uint16_t*>(vtile)[row * BS + offA] = (uint16_t)vnew_at(hh, i);
          }
        }
      }
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
template <int D, bool BF = false>
__global__ void __launch_bounds__(128)
    pa_v2_reduce_kernel(h16* __restrict__ out_, const float* __restrict__ exp_sums,
                        const float* __restrict__ max_logits, const h16* __restrict__ tmp_out_,
                        const int32_t* __restrict__ seq_lens, int max_num_partitions) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int num_heads = gridDim.x;
  const int head = blockIdx.x;
  const int seq = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // seq_len > max_seq_len is undefined behaviour in the reference (its logits buffer overflows); the partition kernels
  // here truncate the context to the partitions that exist, so the merge does too
  const int Lraw = seq_lens[seq];
  const int L = Lraw > max_num_partitions * 512 ? max_num_partitions * 512 : Lraw;
  const int np = (L + 511) / 512;  // :581
  const int64_t sh = ((int64_t)seq * num_heads + head) * max_num_partitions;
  uint16_t* outp = reinterpret_cast<uint16_t*>(out_) + ((int64_t)seq * num_heads + head) * D;
  const uint16_t* tmp = reinterpret_cast<const uint16_t*>(tmp_out_) + sh * D;
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
      acc = __builtin_fmaf(from_elem<BF>(tmp[(int64_t)j * D + i]) * ssum[j], inv, acc);
    outp[i] = to_elem<BF>(acc);
  }
}

// ----------------------------------------------------------------------------------------
// host-side variant descriptor (shared by the translation units that instantiate kernels)
// ----------------------------------------------------------------------------------------
typedef void (*pa_kernel_t)(const PAParams);

struct Variant {
  const char* name;
  int D, BS, HPW, WPH, U;
  bool NT;
  int HPT;  // heads per wave (1 except for the multi-head kernel)
  bool BF;  // element type: false = fp16, true = bfloat16
  pa_kernel_t fn;
  int reserved0;     // (was a cache of the granted dynamic-LDS size: process-global mutable state, removed)
  int UMAX;          // adaptive queue depth limit (0 = fixed U)
  int reserved1;
  int F8;            // caches hold fp8 bytes: 1 = E4M3 (kv_cache_dtype "fp8" / "fp8_e4m3"), 2 = E5M2 ("fp8_e5m2"); 0 = 16-bit
  bool GQS;          // the HPT query heads of a wave share one KV head: num_heads / num_kv_heads % HPT == 0 required
  bool FPV;          // opt-in: probabilities x V on the matrix cores too (vmi_set_pv_mfma); north-star bound, not 1 ulp
  bool SPARSE;       // block-sparse attention (blocksparse_vert_stride > 1); menus of their own (pa_variants_sparse.hip)
  bool QUEUE;        // balanced kernel (pa_queue.hpp): persistent grid of 3 workgroups per CU, mode chosen on the device
  bool STAGE;        // experiment (pa_stage.hip): pages staged through an LDS ring of U slots by global_load_lds
  bool KM;           // balanced kernels: q.K^T of the K pass on the matrix cores (pa_queue.hpp); "m" names
  int XW;            // split kernels (pa_split.hpp): waves per (sequence, head), spread over XW / WPH workgroups that meet in a
                     //   caller-owned workspace; 0 = not a split kernel.  fn is a pa_split_kernel_t there
  pa_kernel_t fn_rounds;  // split kernels: the same kernel serving more items than are resident in rounds (nullptr: none)
  pa_kernel_t fn_app;     // balanced kernels: the append-read form of the same kernel (pa_queue.hpp APP; nullptr: none)
};

typedef void (*pa_reduce_t)(h16*, const float*, const float*, const h16*, const int32_t*, int);

// One row of a paged_attention_v1 menu (pa_table_*.inc).  The including unit defines VMI_APP: false for the plain
// kernels, true for the fused-append kernels ("loads only" diagnostics stay plain).
#define VMI_ROW_A(NAME, D, BS, HPW, WPH, U, NT, LO, LOCK, BF, HPT, UMAX)                                            \
  {NAME, D, BS, HPW, WPH, U, (bool)(NT), HPT, BF,                                                                  \
   (pa_kernel_t)pa_v1_kernel<D, HPW, WPH, U, (bool)(NT), LO, false, BS, LOCK, BF, HPT, (VMI_APP) && !(LO), UMAX>, 0, \
   UMAX},
// fp8-cache rows (pa_table_fp8.inc): fp16 query, no fused-append twin
// (the including unit may define VMI_F8_FMT = 2 and VMI_F8_PFX = "fp8e5m2_" for the E5M2 menus)
#ifndef VMI_F8_FMT
#define VMI_F8_FMT 1
#define VMI_F8_PFX "fp8_"
#endif
#define VMI_ROW_F8B(NAME, D, BS, HPW, WPH, U, NT, LOCK, HPT, UMAX, BF)                                             \
  {NAME, D, BS, HPW, WPH, U, (bool)(NT), HPT, BF,                                                                  \
   (pa_kernel_t)pa_v1_kernel<D, HPW, WPH, U, (bool)(NT), false, false, BS, LOCK, BF, HPT, false, UMAX, VMI_F8_FMT>, 0, \
   UMAX, 0, VMI_F8_FMT, false},
#define VMI_ROW_F8G(NAME, D, BS, HPW, WPH, U, HPT)  /* fp8 pages, grouped-query sharing, fp16 query */              \
  {NAME, D, BS, HPW, WPH, U, true, HPT, false,                                                                     \
   (pa_kernel_t)pa_v1_kernel<D, HPW, WPH, U, true, false, false, BS, false, false, HPT, false, 0, VMI_F8_FMT, true>, 0, \
   0, 0, VMI_F8_FMT, true},
#define VMI_ROW_F8GP(NAME, D, BS, HPW, WPH, U, HPT)  /* ... and P.V on the matrix cores too (opt-in "_pvm") */      \
  {NAME, D, BS, HPW, WPH, U, true, HPT, false,                                                                           \
   (pa_kernel_t)pa_v1_kernel<D, HPW, WPH, U, true, false, false, BS, false, false, HPT, false, 0, VMI_F8_FMT, true, true>, 0, \
   0, 0, VMI_F8_FMT, true, true},
#define VMI_ROW_F8(NAME, D, BS, HPW, WPH, U, NT, LOCK, HPT, UMAX) \
  VMI_ROW_F8B(NAME, D, BS, HPW, WPH, U, NT, LOCK, HPT, UMAX, false)
// grouped-query rows: HPT query heads of one KV head per wave, each tile loaded once
#define VMI_ROW_G(NAME, D, BS, HPW, WPH, U, NT, LOCK, BF, HPT)                                                     \
  {NAME, D, BS, HPW, WPH, U, (bool)(NT), HPT, BF,                                                                  \
   (pa_kernel_t)pa_v1_kernel<D, HPW, WPH, U, (bool)(NT), false, false, BS, LOCK, BF, HPT, VMI_APP, 0, false, true>, \
   0, 0, 0, false, true},
// ... and with the probabilities x V contraction on the matrix cores as well (opt-in, "_pvm" names)
#define VMI_ROW_GP(NAME, D, BS, HPW, WPH, U, BF, HPT)                                                              \
  {NAME, D, BS, HPW, WPH, U, true, HPT, BF,                                                                        \
   (pa_kernel_t)pa_v1_kernel<D, HPW, WPH, U, true, false, false, BS, false, BF, HPT, VMI_APP, 0, false, true, true>, \
   0, 0, 0, false, true, true},
#define VMI_ROW(NAME, D, BS, HPW, WPH, U, NT, LO, LOCK, BF, HPT) \
  VMI_ROW_A(NAME, D, BS, HPW, WPH, U, NT, LO, LOCK, BF, HPT, 0)

// block-sparse rows (pa_variants_sparse.hip): one head per wave, v1 (PART = false) and v2 partitions (PART = true)
#define VMI_ROW_SP(NAME, D, BS, WPH, U, BF, PART)                                                                  \
  {NAME, D, BS, 1, WPH, U, true, 1, BF,                                                                            \
   (pa_kernel_t)pa_v1_kernel<D, 1, WPH, U, true, false, PART, BS, false, BF, 1, false, 0, false, false, false, true>, \
   0, 0, 0, false, false, false, true},
// false in the product library (pa_extras_absent.hip: the out-of-scope menus below are EMPTY there), true in
// libvmi_paged_attention_extras.so (pa_extras_cache.hip)
extern const bool g_has_extras;
extern Variant g_sparse_variants[];
extern const int g_sparse_nvariants;
extern Variant g_sparse_bf16_variants[];
extern const int g_sparse_bf16_nvariants;

// float32 tensors (pa_f32.hip): the (float, float) branch of the reference's dispatch, x = 4
struct PAF32Params {
  float* out;
  const float* q;
  const float* kc;
  const float* vc;
  const int32_t* block_tables;
  const int32_t* seq_lens;
  const float* alibi;
  int32_t num_heads, num_kv_heads;
  float scale;
  int32_t max_blocks_per_seq;
  int64_t q_stride, kv_block_stride, kv_head_stride;
  int32_t lpad;
};
typedef void (*pa_f32_kernel_t)(const PAF32Params);
pa_f32_kernel_t pa_v1_f32_kernel_for(int D, int BS);
void reshape_and_cache_f32_launch(const float* key, const float* value, float* kc, float* vc, const int64_t* slots,
                                  int64_t key_stride, int64_t value_stride, int T, int H, int D, int BS,
                                  hipStream_t stream);

// kernels for the non-core (head size, block size) combinations live in pa_variants_extra.hip
extern Variant g_extra_variants_v1[];
extern const int g_extra_nvariants_v1;
extern Variant g_extra_variants_v2[];
extern const int g_extra_nvariants_v2;
pa_reduce_t extra_reduce_kernel(int head_size, bool bf16);  // nullptr if that head size is not built there
// bfloat16 instantiations of the same (head size, block size) set live in pa_variants_bf16.hip
extern Variant g_bf16_variants_v1[];
extern const int g_bf16_nvariants_v1;
extern Variant g_bf16_variants_v2[];
extern const int g_bf16_nvariants_v2;
// fused-append twins of the three v1 menus (pa_append_*.hip), row for row
extern Variant g_app_core_variants[];
extern const int g_app_core_nvariants;
extern Variant g_app_extra_variants[];
extern const int g_app_extra_nvariants;
extern Variant g_app_bf16_variants[];
extern const int g_app_bf16_nvariants;
// fp8 E4M3 cache (pa_variants_fp8.hip): ids continue after the bf16 menu; no append twins
extern Variant g_fp8_variants_v1[];
extern const int g_fp8_nvariants_v1;
extern Variant g_fp8_variants_v2[];
extern const int g_fp8_nvariants_v2;
// bfloat16 query over the fp8 cache (pa_variants_fp8_bf16.hip): v1 ids continue after the fp16-query fp8 menu
extern Variant g_fp8bf_variants_v1[];
extern const int g_fp8bf_nvariants_v1;
// the same menus over fp8 E5M2 bytes (kv_cache_dtype "fp8_e5m2")
extern Variant g_fp8_variants_v1_e5m2[];
extern const int g_fp8_nvariants_v1_e5m2;
extern Variant g_fp8_variants_v2_e5m2[];
extern const int g_fp8_nvariants_v2_e5m2;
extern Variant g_fp8bf_variants_v1_e5m2[];
extern const int g_fp8bf_nvariants_v1_e5m2;
extern Variant g_fp8bf_variants_v2[];
extern const int g_fp8bf_nvariants_v2;
extern Variant g_fp8bf_variants_v2_e5m2[];
extern const int g_fp8bf_nvariants_v2_e5m2;
pa_reduce_t bf16_reduce_kernel(int head_size);
// balanced (work-queue) kernels, pa_queue.hip: v1 ids continue after every other menu
extern Variant g_queue_variants[];
extern const int g_queue_nvariants;
// split kernels, pa_split.hip: behind the balanced ones; launched only through an entry that carries a workspace
extern Variant g_split_variants[];
extern const int g_split_nvariants;
#ifdef VMI_DIAG
// LDS-staged experiment kernels, pa_stage.hip (diagnostic library only): the last ids of all; never picked by a heuristic
extern Variant g_stage_variants[];
extern const int g_stage_nvariants;
#endif

}  // namespace vmi
