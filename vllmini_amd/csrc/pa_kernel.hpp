// pa_kernel.hpp — kernel templates of the MI355X paged-attention decode path (gfx950 only).
// Included by paged_attention.hip (core instantiations + host code + C-ABI) and by
// pa_variants_extra.hip (the remaining head-size / block-size combinations).
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
// LDS logits are written as float and re-read 4 at a time: the vector view must alias float
typedef float f32x4_alias __attribute__((ext_vector_type(4), may_alias));

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

// p.v over 8 tokens of one dim row.  e0/e1: the 8 exp values, is: 1/(sum+1e-6); `keep` = bit mask of the
// tokens inside the context (only consulted when `last`).
//   fp16: p -> fp16, 4 packed fp16 products, packed fp16 adds ((p0v0+p2v2)+p4v4)+p6v6 / odd, fp32 add of halves
//   bf16: p -> bf16, products rounded to bf16, fp32 sums ((s01+s23)+s45)+s67
template <bool BF>
struct PV8 {
  h16x8 ph;     // fp16 probabilities
  float pf[8];  // bf16-rounded probabilities held in fp32
  __device__ __forceinline__ void set(const f32x4 e0, const f32x4 e1, float is) {
    if constexpr (BF) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pf[k] = bf_round(e0[k] * is);
        pf[4 + k] = bf_round(e1[k] * is);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ph[k] = (h16)(e0[k] * is);
        ph[4 + k] = (h16)(e1[k] * is);
      }
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
      return (float)c[0] + (float)c[1];
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
//
// BS (block size 8 | 16 | 32) and D (any multiple of 8) generalise the lane maps:
//   K tile = D/8 chunks x BS tokens of 16 B; a load covers 64/BS chunks; lane = chunk*BS + token
//   V tile = D rows x BS/8 units of 16 B;   a load covers 512/BS rows;  lane = row*(BS/8) + unit
// When D*BS/8 is not a multiple of 64 (head 80/112, ...) the last load of a tile is predicated.
template <int D, int HPW, int WPH, int U, bool NT, bool LOADS_ONLY = false, bool PART = false, int BS = 16,
          bool LOCK = false, bool BF = false>
__global__ void __launch_bounds__(HPW* WPH * 64)
    pa_v1_kernel(const PAParams p) {
  constexpr int PBLK = 512 / BS;          // blocks per partition (PARTITION_SIZE = 512, :847)
  constexpr int UNITS = D * BS / 8;       // 16-B units in one (block, head) tile of K — and of V
  constexpr int NL = (UNITS + 63) / 64;   // 1-KiB loads per tile
  constexpr int TAIL = UNITS - 64 * (NL - 1);  // active lanes of the last load (64 = full)
  constexpr int CPL = 64 / BS;            // K: chunks per load
  constexpr int UPR = BS / 8;             // V: 16-B units per dim row
  constexpr int RPL = 64 / UPR;           // V: rows per load
  static_assert(D % 8 == 0, "head size must be a multiple of 8");
  static_assert(BS == 8 || BS == 16 || BS == 32, "block size 8, 16 or 32");
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

  uint16_t* outp = reinterpret_cast<uint16_t*>(p.out) +
                   (PART ? (((int64_t)seq * p.num_heads + head) * p.max_num_partitions + part) * D
                         : ((int64_t)seq * p.num_heads + head) * D);

  if (L <= 0) {  // uniform over the workgroup (same seq): reference yields exp_sum = 0 -> out = 0
    if (sub == 0) {
      for (int d = lane; d < D; d += 64) outp[d] = 0;  // +0.0 in fp16 and in bf16
    }
    return;
  }

  const int kvh = head / (p.num_heads / p.num_kv_heads);
  const float slope = p.alibi ? p.alibi[head] : 0.f;

  // ---- q: this lane's 8-dim chunks, one per K load -------------------------------------
  const h16* qp = p.q + (int64_t)seq * p.q_stride + (int64_t)head * D;
  const int c4 = lane / BS;  // chunk-within-load
  const int tk = lane % BS;  // token-within-block
  const bool tail_ok = (TAIL == 64) || lane < TAIL;  // this lane takes part in the last load of a tile
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 qreg[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i)
    qreg[i] = (i < NL - 1 || tail_ok) ? *reinterpret_cast<const u32x4*>(qp + (CPL * i + c4) * 8) : zero4;

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
    // LOCK: the waves of this workgroup own ADJACENT heads of one sequence, whose tiles are contiguous
    // in the cache (HPW * tile bytes per block).  Issuing their page loads in lockstep turns HPW
    // separate 2-KiB reads into one HPW*2-KiB burst per block, which HBM serves measurably faster
    // (gather microbenchmark: 2 KiB chunks 6.36 TB/s, 8 KiB 6.67, 16 KiB 6.8).  All waves of the
    // workgroup run the same number of groups (same sequence), so the barrier count matches.
    if constexpr (LOCK) __builtin_amdgcn_s_barrier();
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
        for (int i = 0; i < NL; ++i)
          r[j][i] = (i < NL - 1 || tail_ok) ? ld16<NT>(ptr + i * 512) : zero4;  // masked lanes add 0
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
        for (int i = 0; i < NL; ++i) accv[i] = dot8<BF>(qreg[i], r[j][i]);
        float acc = accv[0];
#pragma unroll
        for (int i = 1; i < NL; ++i) acc += accv[i];
#pragma unroll
        for (int m = BS; m < 64; m <<= 1) acc += __shfl_xor(acc, m);  // lanes holding the same token
        const int token = b * BS + tk;
        float qk = p.scale * acc;
        qk += (slope != 0.f) ? slope * (float)(token - L + 1) : 0.f;
        const bool masked = token >= L;
        if (lane < BS) logits[token - tok_lo] = masked ? 0.f : qk;
        qk_max = masked ? qk_max : fmaxf(qk_max, qk);
      }
    }
  };

  // Register double buffer over page groups: group g+1 is in flight while group g is consumed.
  // (A third stage was measured and changed nothing; profiles/r01c_cfg3_variant_sweep.json.)
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
  const int hf = lane % UPR;   // which 8-token group of the block this lane owns
  const int rowl = lane / UPR;  // dim row within a load

  // `masked` is a compile-time tag: only the LAST page group of a wave can contain the sequence's last
  // block, so only that call site compiles the tail masking in.
  auto compute_v = [&](auto masked, u32x4(&r)[U][NL], int g) {
    constexpr bool MASK = decltype(masked)::value;
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
        PV8<BF> pv;
        pv.set(e0, e1, inv_sum);
        const bool last = (b == nblk_seq - 1);  // last block of the SEQUENCE (:420); wave-uniform
#pragma unroll
        for (int i = 0; i < NL; ++i) acc[i] += pv.template dot<MASK>(r[j][i], last, token0, L);
      }
    }
  };

  {
    int g = 0;
    for (; g + 2 <= ngroups; g += 2) {
      load_group(rb, vbase, g + 1);
      compute_v(std::false_type{}, ra, g);
      if (g + 2 < ngroups) {
        load_group(ra, vbase, g + 2);
        compute_v(std::false_type{}, rb, g + 1);
      } else {
        compute_v(std::true_type{}, rb, g + 1);  // final group of an even count
      }
    }
    if (g < ngroups) compute_v(std::true_type{}, ra, g);  // final group of an odd count
  }

  if constexpr (LOADS_ONLY) {
    if (fold == 0x9e3779b9u) outp[lane] = 1;  // practically never; keeps the loads live
    return;
  }

  // the UPR lanes of a row hold its 8-token groups
#pragma unroll
  for (int i = 0; i < NL; ++i) {
#pragma unroll
    for (int m = 1; m < UPR; m <<= 1) acc[i] += __shfl_xor(acc[i], m);
  }

  if constexpr (WPH > 1) {
    if (hf == 0) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int row = RPL * i + rowl;
        if (row < D) osm[sub * D + row] = acc[i];
      }
    }
    __syncthreads();
    if (sub == 0) {
      for (int d = lane; d < D; d += 64) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WPH; ++w) s += osm[w * D + d];
        outp[d] = to_elem<BF>(s);
      }
    }
  } else {
    if (hf == 0) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int row = RPL * i + rowl;
        if (row < D) outp[row] = to_elem<BF>(acc[i]);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// paged_attention_v1, "multi-head wave" form for large batches (block size 16, D % 32 == 0):
// ONE wavefront owns HPT ADJACENT heads of one sequence.  In the reference layout the tiles of
// adjacent heads of a block are contiguous, so the wave's page reads become HPT*D*32-byte
// contiguous chunks (8 KiB at D=64, HPT=4) instead of 2-KiB ones; HBM serves bigger chunks
// faster (profiles/: gather microbenchmark 2 KiB 6.36 TB/s, 8 KiB 6.67, 16 KiB 6.8).  With LOCK
// the HPW waves of a workgroup (HPW*HPT adjacent heads) additionally issue in lockstep.
// Arithmetic per head is exactly that of pa_v1_kernel (same rounding points).
// grid = (ceil(H / (HPW*HPT)), num_seqs), block = HPW*64, LDS = HPW*HPT*lpad*4.
// ----------------------------------------------------------------------------------------
template <int D, int HPW, int HPT, int U, bool NT, bool LOCK, bool BF = false>
__global__ void __launch_bounds__(HPW * 64)
    pa_v1_mh_kernel(const PAParams p) {
  constexpr int BS = 16;
  constexpr int NL = D / 32;
  static_assert(D % 32 == 0 && 64 % U == 0, "multi-head kernel: D multiple of 32, U divides 64");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int hl = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.y;
  const int head0 = (blockIdx.x * HPW + hl) * HPT;
  const int nh = (p.num_heads - head0) < HPT ? (p.num_heads - head0) : HPT;  // wave-uniform
  if (nh <= 0) return;  // a terminated wave does not take part in s_barrier

  const int32_t* bt = p.block_tables + (int64_t)seq * p.max_blocks_per_seq;
  int bt_sg = 0;
  int32_t bt_reg = (lane < p.max_blocks_per_seq) ? bt[lane] : 0;

  int L = p.seq_lens[seq];
  L = L > p.lpad ? p.lpad : L;
  const int nblk = (L + BS - 1) / BS;
  const int ngroups = (nblk + U - 1) / U;
  float* logits0 = reinterpret_cast<float*>(smem) + (size_t)hl * HPT * p.lpad;
  uint16_t* out0 = reinterpret_cast<uint16_t*>(p.out) + ((int64_t)seq * p.num_heads + head0) * D;

  if (L <= 0) {
    for (int hh = 0; hh < nh; ++hh)
      for (int d = lane; d < D; d += 64) out0[hh * D + d] = 0;
    return;
  }

  const int c4 = lane >> 4;
  const int tk = lane & 15;
  const int qpk = p.num_heads / p.num_kv_heads;
  int64_t hoff[HPT];  // element offset of each head's tile inside a block
  float slope[HPT];
  u32x4 qreg[HPT][NL];
#pragma unroll
  for (int hh = 0; hh < HPT; ++hh) {
    const int head = head0 + (hh < nh ? hh : 0);
    hoff[hh] = (int64_t)(head / qpk) * p.kv_head_stride + lane * 8;
    slope[hh] = p.alibi ? p.alibi[head] : 0.f;
    const h16* qp = p.q + (int64_t)seq * p.q_stride + (int64_t)head * D;
#pragma unroll
    for (int i = 0; i < NL; ++i) qreg[hh][i] = *reinterpret_cast<const u32x4*>(qp + (4 * i + c4) * 8);
  }

  auto table_for = [&](int g) {
    const int sg = (g * U) >> 6;
    if (sg != bt_sg) {
      const int b = sg * 64 + lane;
      bt_reg = (b < p.max_blocks_per_seq) ? bt[b] : 0;
      bt_sg = sg;
    }
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_group = [&](u32x4(&r)[U][HPT][NL], const h16* cache, int g) {
    if constexpr (LOCK) __builtin_amdgcn_s_barrier();
    table_for(g);
#pragma unroll
    for (int j = 0; j < U; ++j) {
      int idx = g * U + j;
      idx = idx < nblk ? idx : nblk - 1;
      const int64_t phys = __builtin_amdgcn_readlane(bt_reg, idx & 63);
      const h16* blk = cache + phys * p.kv_block_stride;
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
#pragma unroll
        for (int i = 0; i < NL; ++i)
          r[j][hh][i] = (hh < nh) ? ld16<NT>(blk + hoff[hh] + i * 512) : zero4;
      }
    }
  };

  float qk_max[HPT];
#pragma unroll
  for (int hh = 0; hh < HPT; ++hh) qk_max[hh] = -FLT_MAX;

  auto compute_k = [&](u32x4(&r)[U][HPT][NL], int g) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int b = g * U + j;
      if (b < nblk) {
        const int token = b * BS + tk;
        const bool masked = token >= L;
#pragma unroll
        for (int hh = 0; hh < HPT; ++hh) {
          if (hh < nh) {
            float accv[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) accv[i] = dot8<BF>(qreg[hh][i], r[j][hh][i]);
            float acc = accv[0];
#pragma unroll
            for (int i = 1; i < NL; ++i) acc += accv[i];
            acc += __shfl_xor(acc, 16);
            acc += __shfl_xor(acc, 32);
            float qk = p.scale * acc;
            qk += (slope[hh] != 0.f) ? slope[hh] * (float)(token - L + 1) : 0.f;
            if (lane < 16) logits0[hh * p.lpad + token] = masked ? 0.f : qk;
            qk_max[hh] = masked ? qk_max[hh] : fmaxf(qk_max[hh], qk);
          }
        }
      }
    }
  };

  u32x4 ra[U][HPT][NL], rb[U][HPT][NL];
  {
    if (ngroups > 0) load_group(ra, p.kc, 0);
    int g = 0;
    for (; g + 2 <= ngroups; g += 2) {
      load_group(rb, p.kc, g + 1);
      compute_k(ra, g);
      if (g + 2 < ngroups) load_group(ra, p.kc, g + 2);
      compute_k(rb, g + 1);
    }
    if (g < ngroups) compute_k(ra, g);
  }
  if (ngroups > 0) load_group(ra, p.vc, 0);

  float inv_sum[HPT];
#pragma unroll
  for (int hh = 0; hh < HPT; ++hh) {
    inv_sum[hh] = 0.f;
    if (hh < nh) {
      const float m = wave_max(qk_max[hh]);
      float* lg = logits0 + hh * p.lpad;
      float es = 0.f;
      for (int i = lane; i < L; i += 64) {
        const float e = __expf(lg[i] - m);
        lg[i] = e;
        es += e;
      }
      inv_sum[hh] = __builtin_amdgcn_rcpf(wave_sum(es) + 1e-6f);
    }
  }

  float acc[HPT][NL];
#pragma unroll
  for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[hh][i] = 0.f;
  const int hf = lane & 1;

  auto compute_v = [&](auto masked, u32x4(&r)[U][HPT][NL], int g) {
    constexpr bool MASK = decltype(masked)::value;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int b = g * U + j;
      if (b < nblk) {
        const int token0 = b * BS + hf * 8;
        const bool last = (b == nblk - 1);
#pragma unroll
        for (int hh = 0; hh < HPT; ++hh) {
          if (hh < nh) {
            const float* lg = logits0 + hh * p.lpad + token0;
            const f32x4 e0 = *reinterpret_cast<const f32x4_alias*>(lg);
            const f32x4 e1 = *reinterpret_cast<const f32x4_alias*>(lg + 4);
            const float is = inv_sum[hh];
            PV8<BF> pv;
            pv.set(e0, e1, is);
#pragma unroll
            for (int i = 0; i < NL; ++i) acc[hh][i] += pv.template dot<MASK>(r[j][hh][i], last, token0, L);
          }
        }
      }
    }
  };
  {
    int g = 0;
    for (; g + 2 <= ngroups; g += 2) {
      load_group(rb, p.vc, g + 1);
      compute_v(std::false_type{}, ra, g);
      if (g + 2 < ngroups) {
        load_group(ra, p.vc, g + 2);
        compute_v(std::false_type{}, rb, g + 1);
      } else {
        compute_v(std::true_type{}, rb, g + 1);
      }
    }
    if (g < ngroups) compute_v(std::true_type{}, ra, g);
  }

#pragma unroll
  for (int hh = 0; hh < HPT; ++hh) {
    if (hh < nh) {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const float a = acc[hh][i] + __shfl_xor(acc[hh][i], 1);
        if (hf == 0) out0[hh * D + 32 * i + (lane >> 1)] = to_elem<BF>(a);
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
  const int L = seq_lens[seq];
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
  int lds_attr_set;  // largest dynamic-LDS size already granted through hipFuncSetAttribute
};

typedef void (*pa_reduce_t)(h16*, const float*, const float*, const h16*, const int32_t*, int);

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
pa_reduce_t bf16_reduce_kernel(int head_size);

}  // namespace vmi
