// libvmi_gpt2_layer.so — the GPT-2 block's linear layers for the decode harness (include/vmi_gpt2_layer.h).
//
// What the reference does here: torch modules — ln_1 -> c_attn, c_proj + residual, ln_2 -> c_fc -> GELU -> c_proj + residual
// (vllmini/model/gpt2.py:14-15, 117-128, 130-135 and GPT2Block.forward).  Through torch on this GPU that is, per layer of a
// 256-row decode step, four hipBLASLt launches of 8 - 9 us each for 0.3 - 1.2 GFLOP, two LayerNorm, two add and one GELU
// launch of ~5 us each: 67 us of launches whose arithmetic and bytes are worth about 3 (profiles/r05u_e2e_native_layers.md).
//
// One kernel family, four launches per layer, each one round trip to memory:
//   * y = epi(LN?(x) . W^T + b), M <= a few hundred rows (the decode batch), W = nn.Linear's [N, K] as stored;
//   * a workgroup owns BM rows x (16 * NW) columns.  Its BM x K row tile is loaded ONCE into LDS (rows padded by 16 bytes: the
//     16-lane groups of a ds_read_b128 then cover all 64 banks), LayerNorm runs on it in place (fp32 statistics, two passes,
//     one wave per row, rounded to half like torch's layer_norm on half input);
//   * every wave owns 16 output columns: its W rows are the A operand of v_mfma_f32_16x16x32_f16 exactly as they lie in
//     memory (lane = kgroup * 16 + row holds 8 consecutive k of one row = one 16-byte load), streamed straight into
//     registers, two chunks of 12 k-steps in flight — for K = 768 the whole weight slab is requested before the row tile
//     arrives; the row tile's fragments (B operand, same 16-byte shape) come from LDS;
//   * KS > 1: the K range of a column slab is split over KS waves, partial tiles meet in LDS and are added in a fixed order;
//   * D[row = column n][col = row m]: a lane ends with 4 consecutive columns of one output row -> bias, GELU or the residual
//     add on 8-byte vectors, one 8-byte store.
// The weights of a GPT-2 small layer are 14 MB and stay in the 256 MiB Infinity Cache between tokens; the row tiles are L2 hits.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "vmi_gpt2_layer.h"

namespace vmi_layer {

using h16 = _Float16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct LinearParams {
  const h16* x;
  int64_t ldx;
  const h16* w;
  const h16* bias;
  const h16* gamma;
  const h16* beta;
  float eps;
  const h16* res;
  int64_t ldr;
  h16* y;
  int64_t ldy;
  int M, N, K;
  // EPI 3 (the q / k / v projection): columns [E, 2E) and [2E, 3E) of a row also go to the paged cache at the row's slot
  const int64_t* slots;
  h16* kcache;
  h16* vcache;
  int64_t kv_bstride, kv_hstride;
  int qdim, hsize, bs, bs_shift;
  int w_packed;   // 0: nn.Linear rows [N, K]; 1: MFMA tiles [N / 16][K / 32][64 lanes][8] (vmi_gpt2_layer.h)
};

constexpr int CH = 12;    // k-steps (of 32) per weight chunk: 48 VGPRs
constexpr int XU = 12;    // 16-byte units of the row tile a thread requests at once (GPT-2 small: all of its share)

__device__ __forceinline__ void load_w(u32x4 (&dst)[CH], const h16* wrow, int wstep, int ksb, int ks1) {
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int k = ksb + j;   // wave-uniform
    if (k < ks1) dst[j] = *reinterpret_cast<const u32x4*>(wrow + (int64_t)k * wstep);
  }
}

template <int MT>
__device__ __forceinline__ void mma_chunk(f32x4 (&acc)[MT], const u32x4 (&wf)[CH], const h16* xa, int ldxs, int ksb, int ks1) {
  if (ksb + CH <= ks1) {   // a whole chunk: straight-line code, the row tile's fragments read PD k-steps ahead of the matrix pipe
    constexpr int PD = 4;  // (left to itself the scheduler keeps ONE step of distance: an LDS round trip per MFMA pair)
    h16x8 xf[PD][MT];
#pragma unroll
    for (int j = 0; j < PD; ++j)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) xf[j][mi] = *reinterpret_cast<const h16x8*>(xa + (int64_t)mi * 16 * ldxs + (ksb + j) * 32);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wf[j]), xf[j % PD][mi], acc[mi], 0, 0, 0);
        if (j + PD < CH) xf[j % PD][mi] = *reinterpret_cast<const h16x8*>(xa + (int64_t)mi * 16 * ldxs + (ksb + j + PD) * 32);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    return;
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int k = ksb + j;
    if (k < ks1) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const h16x8 xf = *reinterpret_cast<const h16x8*>(xa + (int64_t)mi * 16 * ldxs + k * 32);
        acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wf[j]), xf, acc[mi], 0, 0, 0);
      }
    }
  }
}

// two elements of a row through LayerNorm's affine form: t = x * rstd - mean * rstd in fp32 straight from the half (v_fma_mix_f32),
// y = t * gamma + beta in fp32 written as a half (v_fma_mixlo / mixhi_f16: one rounding, to nearest even) — 2 instructions per
// element where conversions + fp32 math take 7
__device__ __forceinline__ uint32_t ln_pair(uint32_t x2, uint32_t g2, uint32_t b2, float rstd, float nmr) {
  float t0, t1;
  uint32_t y;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(x2), "v"(rstd), "v"(nmr));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(x2), "v"(rstd), "v"(nmr));
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(y) : "v"(t0), "v"(g2), "v"(b2));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[0,1,1]" : "+v"(y) : "v"(t1), "v"(g2), "v"(b2));
  return y;
}

// s += x, q += x * x for the two halves of a dword, fp32 sums (v_dot2_f32_f16 would take half the instructions, but its sums
// came out ~10 % off the fp32 ones on this data — measured, not used)
__device__ __forceinline__ void ln_stats_pair(uint32_t x2, float& s, float& q) {
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(s) : "v"(x2));
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(s) : "v"(x2));
  asm("v_fma_mix_f32 %0, %1, %1, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(q) : "v"(x2));
  asm("v_fma_mix_f32 %0, %1, %1, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(q) : "v"(x2));
}

// sum over the LPR consecutive lanes that share a row: inside a row of 16 lanes on the DPP cross-lane paths (quad permutes, then
// the half-row and row mirrors: each a VALU operand modifier, no LDS round trip), beyond that through ds_bpermute
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (LPR >= 2) v += dpp_get<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (LPR >= 4) v += dpp_get<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (LPR >= 8) v += dpp_get<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of each 8
  if constexpr (LPR >= 16) v += dpp_get<0x140>(v);  // row_mirror: lane i <-> 15 - i of each 16
#pragma unroll
  for (int o = 16; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Order of the memory requests (they complete in order, so what is needed first is asked for first): the row tile, weight
// chunk 0, [tile to LDS], weight chunks 1 .. NBUF - 1, [LayerNorm], the ring of chunks.  For K = 768 (NBUF = 2) and for
// mlp.c_proj's K = 3072 over two waves (NBUF = 4) everything a wave will ever read is in flight before its first MFMA.
template <int BM, int NW, int KS, int NBUF, bool LN, int EPI>
__global__ __launch_bounds__(NW* KS * 64) void linear_kernel(const LinearParams p) {
  constexpr int T = NW * KS * 64, MT = BM / 16, LPR = T / BM;   // LPR lanes share a row of the tile (load, LayerNorm)
  static_assert(LPR >= 1 && LPR <= 64 && (LPR & (LPR - 1)) == 0, "a row's lanes lie inside one wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h16* xs = reinterpret_cast<h16*>(smem);
  const int ldxs = p.K + 8, upr = p.K >> 3;
  h16* gb = xs + (size_t)BM * ldxs;                                      // LN: gamma [K], beta [K]
  float* red = reinterpret_cast<float*>(gb + (LN ? 2 * p.K : 0));       // [(KS - 1)][NW][MT][64][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nw = wave % NW, ks = wave / NW;
  const int NB = (p.N + 16 * NW - 1) / (16 * NW);
  const int nb = blockIdx.x % NB, mb = blockIdx.x / NB;
  // (the cache epilogue deals a workgroup's waves to slabs N / NW apart, one each of q | q,k | k,v | v: the value cache's
  //  scattered two-byte pieces then leave through every CU's store path, not through the third of the workgroups that would
  //  hold the v columns — the projection with the cache write 18.2 -> see profiles/r05u_e2e_native_layers.md §3b)
  const int m0 = mb * BM, n0 = (EPI == VMI_LAYER_EPI_BIAS_KV_CACHE ? nw * NB + nb : nb * NW + nw) * 16;
  const int nks = p.K >> 5;
  const int ks_per = (nks + KS - 1) / KS, ks0 = ks * ks_per, ks1 = min(nks, ks0 + ks_per);
  const int kc = lane >> 4, r16 = lane & 15;
  const bool nvalid = n0 < p.N;   // N % 16 == 0: a slab is wholly inside or wholly outside

  // 1. LayerNorm's gamma and beta (on their way into LDS: 2 * K / 8 units <= 2 * T, checked on the host), then the row
  //    tile: BM x K halves, rows past M as zeros; LPR consecutive lanes walk one row
  u32x4 gv[2];
  if constexpr (LN) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + i * T;
      gv[i] = u32x4{0u, 0u, 0u, 0u};
      if (u < 2 * upr) gv[i] = *reinterpret_cast<const u32x4*>((u < upr ? p.gamma + u * 8 : p.beta + (u - upr) * 8));
    }
  }
  const int xrow = tid / LPR, xj = tid % LPR;
  const bool xin = m0 + xrow < p.M;
  const h16* xsrc = p.x + (int64_t)(m0 + (xin ? xrow : 0)) * p.ldx;
  h16* xdst = xs + (int64_t)xrow * ldxs;
  u32x4 xv[XU];
#pragma unroll
  for (int i = 0; i < XU; ++i) {
    const int u = xj + i * LPR;
    xv[i] = u32x4{0u, 0u, 0u, 0u};
    if (xin && u < upr) xv[i] = *reinterpret_cast<const u32x4*>(xsrc + u * 8);
  }
  // 2. the weight slab of this wave, chunk 0
  // (rows as stored: lane (kc, r) reads 16 bytes of row n0 + r, a k-step is 32 elements on; packed tiles: a k-step of a
  //  16-column slab is 1 KiB in lane order — one contiguous KiB per wave load instead of sixteen 64-byte pieces)
  const int wstep = p.w_packed ? 512 : 32;
  const h16* wrow = !nvalid ? p.w : p.w_packed ? p.w + ((int64_t)(n0 >> 4) * nks * 64 + lane) * 8
                                               : p.w + (int64_t)(n0 + r16) * p.K + 8 * kc;
  u32x4 wb[NBUF][CH];
#pragma unroll
  for (int b = 0; b < NBUF; ++b)
#pragma unroll
    for (int j = 0; j < CH; ++j) wb[b][j] = u32x4{0u, 0u, 0u, 0u};
  if (nvalid) load_w(wb[0], wrow, wstep, ks0, ks1);
  // residual values of this lane's outputs: asked for now, used in the epilogue
  const int n = n0 + 4 * kc;
  h16x4 rr[MT];
  if constexpr (EPI == VMI_LAYER_EPI_BIAS_RESIDUAL) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int m = m0 + mi * 16 + r16;
      rr[mi] = h16x4{0, 0, 0, 0};
      if (nvalid && m < p.M) rr[mi] = *reinterpret_cast<const h16x4*>(p.res + (int64_t)m * p.ldr + n);
    }
  }
  // EPI 3: the rows' cache slots too — a load in the epilogue would wait for the stores in front of it (loads and stores share
  // vmcnt), one acknowledged round trip per row block
  int64_t slot[MT];
  if constexpr (EPI == VMI_LAYER_EPI_BIAS_KV_CACHE) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int m = m0 + mi * 16 + r16;
      slot[mi] = -1;
      if (nvalid && n0 >= p.qdim && m < p.M) slot[mi] = p.slots[m];
    }
  }
  if constexpr (!LN) {
    // 3. tile to LDS (the few shapes whose rows are longer than XU * LPR units finish them here, one more round trip each)
#pragma unroll
    for (int i = 0; i < XU; ++i) {
      const int u = xj + i * LPR;
      if (u < upr) *reinterpret_cast<u32x4*>(xdst + u * 8) = xv[i];
    }
    for (int ub = XU * LPR; ub < upr; ub += LPR) {
      const int u = ub + xj;
      if (u < upr) *reinterpret_cast<u32x4*>(xdst + u * 8) = xin ? *reinterpret_cast<const u32x4*>(xsrc + u * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    // 4. the other chunks of the ring
    if (nvalid) {
#pragma unroll
      for (int b = 1; b < NBUF; ++b) load_w(wb[b], wrow, wstep, ks0 + b * CH, ks1);
    }
    __syncthreads();
  } else {
    // 3. LayerNorm on the values as they sit in registers (a row's K <= XU * LPR * 8, checked on the host): the LPR lanes of a
    //    row add up the sum and the sum of squares (fp32; var = E[x^2] - mean^2), gamma / beta come back from LDS, and the tile is written ONCE,
    //    normalised: y = (x - mean) * rstd * gamma + beta rounded to half — torch's layer_norm on half input
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + i * T;
      if (u < 2 * upr) *reinterpret_cast<u32x4*>(gb + u * 8) = gv[i];
    }
    const float inv_k = 1.f / (float)p.K;
    float s = 0.f, q = 0.f;   // sum and sum of squares in fp32, each element straight from its half: v_fma_mix_f32
#pragma unroll
    for (int i = 0; i < XU; ++i) {   // (units past the row are zeros)
#pragma unroll
      for (int e = 0; e < 4; ++e) ln_stats_pair(xv[i][e], s, q);
    }
    const float mean = group_sum<LPR>(s) * inv_k;
    const float var = fmaxf(group_sum<LPR>(q) * inv_k - mean * mean, 0.f);
    const float rstd = rsqrtf(var + p.eps), nmr = -mean * rstd;
    // 4. the other chunks of the ring
    if (nvalid) {
#pragma unroll
      for (int b = 1; b < NBUF; ++b) load_w(wb[b], wrow, wstep, ks0 + b * CH, ks1);
    }
    __syncthreads();   // gamma / beta are in LDS
#pragma unroll
    for (int i = 0; i < XU; ++i) {
      const int u = xj + i * LPR;
      if (u < upr) {
        const u32x4 g = *reinterpret_cast<const u32x4*>(gb + u * 8);
        const u32x4 b = *reinterpret_cast<const u32x4*>(gb + p.K + u * 8);
        u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ln_pair(xv[i][e], g[e], b[e], rstd, nmr);
        *reinterpret_cast<u32x4*>(xdst + u * 8) = v;
      }
    }
    __syncthreads();
  }

  f32x4 acc[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nvalid) {
    const h16* xa = xs + (int64_t)r16 * ldxs + 8 * kc;
    for (int ksb = ks0; ksb < ks1; ksb += NBUF * CH) {
#pragma unroll
      for (int b = 0; b < NBUF; ++b) {
        mma_chunk<MT>(acc, wb[b], xa, ldxs, ksb + b * CH, ks1);
        if (ksb + (b + NBUF) * CH < ks1) load_w(wb[b], wrow, wstep, ksb + (b + NBUF) * CH, ks1);
      }
    }
  }

  if constexpr (KS > 1) {
    if (ks > 0) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        *reinterpret_cast<f32x4*>(red + ((((int64_t)(ks - 1) * NW + nw) * MT + mi) * 64 + lane) * 4) = acc[mi];
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll
    for (int s = 1; s < KS; ++s)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(red + ((((int64_t)(s - 1) * NW + nw) * MT + mi) * 64 + lane) * 4);
        acc[mi] += o;
      }
  }
  if (!nvalid) return;

  // D[row = 4 * kc + r -> column n][col = r16 -> row m]
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr) {
    const h16x4 b = *reinterpret_cast<const h16x4*>(p.bias + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = (float)b[r];
  }
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int m = m0 + mi * 16 + r16;
    if (m >= p.M) continue;
    h16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (h16)(acc[mi][r] + bias[r]);   // the linear layer's own half output
    if constexpr (EPI == VMI_LAYER_EPI_BIAS_GELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (float)o[r];
        o[r] = (h16)(0.5f * v * (1.f + erff(v * 0.70710678118654752440f)));
      }
    } else if constexpr (EPI == VMI_LAYER_EPI_BIAS_RESIDUAL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (h16)((float)rr[mi][r] + (float)o[r]);
    }
    *reinterpret_cast<h16x4*>(p.y + (int64_t)m * p.ldy + n) = o;
    if constexpr (EPI == VMI_LAYER_EPI_BIAS_KV_CACHE) {
      // reshape_and_cache's copy (cache_kernels.cu:172-199) done by the producer: K[blk, h, d / 8, off, d % 8], V[blk, h, d, off];
      // a slab of 16 columns lies inside one of q | k | v and inside one head (E and head_size are multiples of 16)
      const int part = n0 / p.qdim;   // wave-uniform
      if (part > 0) {
        if (slot[mi] >= 0) {          // (a negative slot is a padded token: skipped, cache_kernels.cu:165-169)
          const int64_t blk = slot[mi] >> p.bs_shift;     // block sizes are powers of two (8 / 16 / 32)
          const int off = (int)(slot[mi] & (p.bs - 1));
          const int c0 = n0 - part * p.qdim, hh = c0 / p.hsize, d = c0 - hh * p.hsize + 4 * kc;   // (hh: wave-uniform)
          const int64_t base = blk * p.kv_bstride + (int64_t)hh * p.kv_hstride;
          if (part == 1) {
            // (non-temporal, like the scatter kernel's stores: a token's 2-byte V pieces dirty one 128-byte line each, and lines
            //  left dirty in L2 are written back under the attention launch that follows — profiles/r02b_call_pair_aftermath.md)
            __builtin_nontemporal_store(o, reinterpret_cast<h16x4*>(p.kcache + base + (int64_t)(d >> 3) * p.bs * 8 + off * 8 + (d & 7)));
          } else {
            h16* dst = p.vcache + base + (int64_t)d * p.bs + off;
#pragma unroll
            for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(o[r], dst + (int64_t)r * p.bs);
          }
        }
      }
    }
  }
}

// ---- the two ends of a decode step: token + position embedding, greedy argmax --------------------------------------------------

// out[b, :] = wte[ids[b], :] + wpe[pos[b], :] (gpt2.py:183-233: inputs_embeds + position_embeds; half + half rounded once)
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos,
                                                    const h16* __restrict__ wte, const h16* __restrict__ wpe, h16* __restrict__ out,
                                                    int E) {
  const int b = blockIdx.x;
  const h16* t = wte + ids[b] * (int64_t)E;
  const h16* q = wpe + pos[b] * (int64_t)E;
  h16* o = out + (int64_t)b * E;
  for (int u = threadIdx.x; u < (E >> 3); u += 256) {
    const h16x8 a = *reinterpret_cast<const h16x8*>(t + u * 8), c = *reinterpret_cast<const h16x8*>(q + u * 8);
    h16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (h16)((float)a[e] + (float)c[e]);
    *reinterpret_cast<h16x8*>(o + u * 8) = r;
  }
}

// out[b] = index of the first maximum of logits[b, 0:V] (torch.argmax's tie rule); one workgroup per row, 16-byte loads from the
// first aligned element on (rows of 50 257 halves start at odd addresses)
__device__ __forceinline__ void amax_take(float v, int i, float& best, int& at) {
  if (v > best || (v == best && i < at)) {
    best = v;
    at = i;
  }
}
__global__ __launch_bounds__(1024) void argmax_kernel(const h16* __restrict__ logits, int64_t ld, int V, int64_t* __restrict__ out) {
  __shared__ float sb[16];
  __shared__ int si[16];
  const h16* row = logits + (int64_t)blockIdx.x * ld;
  const int tid = threadIdx.x;
  float best = -INFINITY;
  int at = 0x7fffffff;
  const int head = (int)(((16 - ((uintptr_t)row & 15)) & 15) >> 1);   // elements in front of the first 16-byte boundary
  const int a0 = head < V ? head : V;
  if (tid < a0) amax_take((float)row[tid], tid, best, at);
  const int nvec = (V - a0) >> 3;
  for (int u = tid; u < nvec; u += 1024) {
    const h16x8 v = *reinterpret_cast<const h16x8*>(row + a0 + u * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) amax_take((float)v[e], a0 + u * 8 + e, best, at);
  }
  const int t0 = a0 + nvec * 8;
  if (t0 + tid < V) amax_take((float)row[t0 + tid], t0 + tid, best, at);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(at, o, 64);
    amax_take(ob, oi, best, at);
  }
  if ((tid & 63) == 0) {
    sb[tid >> 6] = best;
    si[tid >> 6] = at;
  }
  __syncthreads();
  if (tid < 64) {
    best = tid < 16 ? sb[tid] : -INFINITY;
    at = tid < 16 ? si[tid] : 0x7fffffff;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(at, o, 64);
      amax_take(ob, oi, best, at);
    }
    if (tid == 0) out[blockIdx.x] = at == 0x7fffffff ? 0 : at;
  }
}

// ---- top-k sampling (Scheduler.sample_next_token, scheduler.py:144-153: logits / temperature -> top-k 50 -> softmax -> one draw) ---
//
// torch does this as topk (a radix select over 50 257 values) + softmax + multinomial + gather: 127 us for one row, 309 us for
// 256 — a quarter of a one-sequence token.  Here one 1024-thread workgroup per row keeps the row in registers as unique 32-bit
// keys (order-preserving half bits << 16 | 65535 - index: larger key = larger logit, ties to the smaller index) and never sorts
// it: (1) every thread's maximum goes to LDS and ONE wave finds the k-th largest of those 1024 — a lower bound of the row's k-th
// largest value; (2) the few hundred elements at or above it are compacted into LDS; (3) one wave finds the exact k-th largest
// key among them by bisection on the key, ranks the k survivors, and (4) draws from their softmax by inverse CDF with a uniform
// number the caller supplies (so the generator stays torch's).  Rows whose candidates overflow the LDS list (thousands of equal
// logits) take a block-wide bisection instead: slower, same answer.
constexpr int SK_T = 1024, SK_UPT = 8, SK_CAP = 512, SK_KMAX = 64;   // (a row of N(0, s) logits leaves ~50 - 60 candidates)
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

// two halves of a dword -> two order-preserving 16-bit keys (negative: all bits flipped, else the sign bit set)
__device__ __forceinline__ uint32_t sk_keys2(uint32_t x) {
  const uint32_t s = (x >> 15) & 0x00010001u;
  return x ^ ((s * 0xffffu) | 0x80008000u);
}
__device__ __forceinline__ uint32_t sk_max2(uint32_t a, uint32_t b) {   // v_pk_max_u16
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t sk_key16(h16 v) {
  const uint16_t b = __builtin_bit_cast(uint16_t, v);
  return (b & 0x8000u) ? (uint16_t)~b : (uint16_t)(b | 0x8000u);
}
__device__ __forceinline__ float sk_value(uint32_t key) {
  const uint16_t k = (uint16_t)(key >> 16);
  const uint16_t b = (k & 0x8000u) ? (uint16_t)(k & 0x7fffu) : (uint16_t)~k;
  return (float)__builtin_bit_cast(h16, b);
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// The row sits in registers as packed 16-bit value keys, eight consecutive elements (one 16-byte load from the first aligned
// element on) per unit, SK_UPT units per thread; the up to 7 + 7 elements in front of the first and behind the last unit are one
// extra element of threads 0 .. 13.  A candidate's unique 32-bit key (value key << 16 | 65535 - index: ties to the smaller index)
// is built only when it is appended.
__global__ __launch_bounds__(SK_T) void sample_top_k_kernel(const h16* __restrict__ logits, int64_t ld, int V, int top_k,
                                                            float inv_temperature, const float* __restrict__ uniform,
                                                            int64_t* __restrict__ out) {
  __shared__ uint32_t tmax_s[SK_T];
  __shared__ uint32_t cand[SK_CAP];
  __shared__ uint32_t sel[SK_KMAX], sorted[SK_KMAX];
  __shared__ int cnt_s, wcount[SK_T / 64];
  __shared__ uint32_t bound_s;
  const h16* row = logits + (int64_t)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = min(min(top_k, SK_KMAX), V);
  const int head = (int)(((16 - ((uintptr_t)row & 15)) & 15) >> 1);
  const int a0 = head < V ? head : V, nvec = (V - a0) >> 3, t0 = a0 + nvec * 8, ntail = V - t0;
  u32x4 kv[SK_UPT];      // packed value keys, 0 where there is no element
  uint32_t umax[SK_UPT];  // the largest value key of each unit
#pragma unroll
  for (int j = 0; j < SK_UPT; ++j) {
    const int u = tid + j * SK_T;
    kv[j] = u32x4{0u, 0u, 0u, 0u};
    if (u < nvec) kv[j] = *reinterpret_cast<const u32x4*>(row + a0 + u * 8);
  }
  int xi = -1;            // the extra element of this thread (index), if any
  if (tid < a0) xi = tid;
  else if (tid - a0 < ntail) xi = t0 + tid - a0;
  uint32_t xk = xi >= 0 ? sk_key16(row[xi]) : 0u;
  uint32_t tmax = xk;
#pragma unroll
  for (int j = 0; j < SK_UPT; ++j) {
    const bool has = tid + j * SK_T < nvec;
#pragma unroll
    for (int e = 0; e < 4; ++e) kv[j][e] = has ? sk_keys2(kv[j][e]) : 0u;
    const uint32_t m2 = sk_max2(sk_max2(kv[j][0], kv[j][1]), sk_max2(kv[j][2], kv[j][3]));
    umax[j] = max(m2 & 0xffffu, m2 >> 16);
    tmax = max(tmax, umax[j]);
  }
  tmax_s[tid] = tmax;
  if (tid == 0) cnt_s = 0;
  __syncthreads();
  // (1) wave 0: the k-th largest of the 1024 thread maxima — a lower bound of the row's k-th largest value (counts by ballot +
  //     scalar popcount: no cross-lane traffic)
  if (wave == 0) {
    uint32_t m[SK_T / 64];
#pragma unroll
    for (int j = 0; j < SK_T / 64; ++j) m[j] = tmax_s[lane + j * 64];
    uint32_t lo = 0, hi = 0xffffu;
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      int c = 0;
#pragma unroll
      for (int j = 0; j < SK_T / 64; ++j) c += __builtin_popcountll(__ballot(m[j] >= mid));
      if (c >= k) lo = mid; else hi = mid - 1;
    }
    if (lane == 0) bound_s = lo;
  }
  __syncthreads();
  const uint32_t bound = bound_s;   // a VALUE key; 0 only when fewer than k thread maxima are real (V < 1024 or so)
  // (2) everything at or above the bound into the LDS list, as unique 32-bit keys
  auto append = [&](uint32_t vkey, int idx) {
    const int pos = atomicAdd(&cnt_s, 1);
    if (pos < SK_CAP) cand[pos] = (vkey << 16) | (uint32_t)(65535 - idx);
  };
  if (xi >= 0 && xk >= bound) append(xk, xi);
#pragma unroll
  for (int j = 0; j < SK_UPT; ++j) {
    const int u = tid + j * SK_T;
    if (u < nvec && umax[j] >= bound) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t vk = (kv[j][e >> 1] >> ((e & 1) * 16)) & 0xffffu;
        if (vk >= bound) append(vk, a0 + u * 8 + e);
      }
    }
  }
  __syncthreads();
  const int n = cnt_s;
  uint32_t thr;
  if (n <= SK_CAP) {
    // (3) wave 0: the exact k-th largest key of the list by bisection (keys are unique: exactly k keys are >= it)
    if (wave != 0) return;
    uint32_t c[SK_CAP / 64];
#pragma unroll
    for (int j = 0; j < SK_CAP / 64; ++j) c[j] = lane + j * 64 < n ? cand[lane + j * 64] : 0u;
    const int nl = (n + 63) >> 6;
    uint32_t lo = bound << 16, hi = 0xffffffffu;
    while (lo < hi) {
      const uint32_t mid = lo + (uint32_t)(((uint64_t)hi - lo + 1) >> 1);
      int cc = 0;
#pragma unroll
      for (int j = 0; j < SK_CAP / 64; ++j)
        if (j < nl) cc += __builtin_popcountll(__ballot(c[j] >= mid));
      if (cc >= k) lo = mid; else hi = mid - 1;
    }
    thr = lo;
    // the k survivors into sel[] (any order)
    if (lane == 0) cnt_s = 0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < SK_CAP / 64; ++j)
      if (j < nl && c[j] >= thr && c[j] != 0u) sel[atomicAdd(&cnt_s, 1)] = c[j];
  } else {
    // thousands of candidates (long runs of equal logits): block-wide bisection over the unique keys of everything in registers
    auto count_ge = [&](uint32_t mid) {
      int cc = (xi >= 0 && ((xk << 16) | (uint32_t)(65535 - xi)) >= mid) ? 1 : 0;
#pragma unroll
      for (int j = 0; j < SK_UPT; ++j) {
        const int u = tid + j * SK_T;
        if (u < nvec) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t vk = (kv[j][e >> 1] >> ((e & 1) * 16)) & 0xffffu;
            cc += ((vk << 16) | (uint32_t)(65535 - (a0 + u * 8 + e))) >= mid;
          }
        }
      }
      return cc;
    };
    uint32_t lo = bound << 16, hi = 0xffffffffu;
    while (lo < hi) {
      const uint32_t mid = lo + (uint32_t)(((uint64_t)hi - lo + 1) >> 1);
      const int cc = wave_sum_i(count_ge(mid));
      __syncthreads();
      if (lane == 0) wcount[wave] = cc;
      __syncthreads();
      int tot = 0;
#pragma unroll
      for (int w = 0; w < SK_T / 64; ++w) tot += wcount[w];
      if (tot >= k) lo = mid; else hi = mid - 1;
    }
    thr = lo;
    __syncthreads();
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    if (xi >= 0 && ((xk << 16) | (uint32_t)(65535 - xi)) >= thr) sel[atomicAdd(&cnt_s, 1)] = (xk << 16) | (uint32_t)(65535 - xi);
#pragma unroll
    for (int j = 0; j < SK_UPT; ++j) {
      const int u = tid + j * SK_T;
      if (u < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t vk = (kv[j][e >> 1] >> ((e & 1) * 16)) & 0xffffu;
          const uint32_t key = (vk << 16) | (uint32_t)(65535 - (a0 + u * 8 + e));
          if (key >= thr) sel[atomicAdd(&cnt_s, 1)] = key;
        }
      }
    }
    __syncthreads();
    if (wave != 0) return;
  }
  // (4) wave 0: rank the k survivors (descending), softmax over their values, inverse CDF
  __builtin_amdgcn_wave_barrier();
  const uint32_t mine = lane < k ? sel[lane] : 0u;
  int rank = 0;
  for (int j = 0; j < k; ++j) rank += (uint32_t)__builtin_amdgcn_readlane((int)mine, j) > mine;   // (j is uniform: v_readlane)
  if (lane < k) sorted[rank] = mine;
  __builtin_amdgcn_wave_barrier();
  const uint32_t skey = lane < k ? sorted[lane] : 0u;
  const float top = sk_value(sorted[0]);
  const float e = lane < k ? __expf((sk_value(skey) - top) * inv_temperature) : 0.f;
  float cum = e;   // inclusive scan over the lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float up = __shfl_up(cum, o, 64);
    if (lane >= o) cum += up;
  }
  const float total = __shfl(cum, k - 1, 64);
  const float target = uniform[blockIdx.x] * total;
  const uint64_t hit = __ballot(lane < k && cum >= target);
  const int pick = hit ? __builtin_ctzll(hit) : k - 1;
  if (lane == pick) out[blockIdx.x] = 65535 - (int)(skey & 0xffffu);
}

// ---- host side ------------------------------------------------------------------------------------------------------------

static thread_local std::string g_err;
// (assign + append: an operator+ on std::string would leave a weak template symbol in the library's dynamic table)
static void set_err(const char* a, const char* b) {
  g_err.assign(a);
  g_err.append(b);
}

struct Shape {
  int bm, nw, ks, nbuf;
};

constexpr int LDS_PER_CU = 160 * 1024;
constexpr int MAX_K = 4608;
constexpr int MAX_K_LN = 2048;   // gamma and beta ride into LDS with two 16-byte units per thread of a 256-thread workgroup

// BM = 32 while three such tiles fit a CU's LDS side by side AND the launch still has workgroups for most of the chip;
// otherwise BM = 16 (twice the workgroups); the K range of a slab is split over two waves from 48 k-steps on.
static bool pick(int M, int N, int K, bool ln, Shape* s) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 31) || (N & 15) || K > (ln ? MAX_K_LN : MAX_K)) return false;
  const int nb = (N + 63) / 64;
  // (behind a LayerNorm a row's K must sit in the registers of its lanes: 12 units x T / BM lanes -> K <= 768 at BM = 32,
  //  1536 at BM = 16, 3072 with the K range over two waves)
  const bool fits32 = 3 * 32 * (K + 8) * 2 <= LDS_PER_CU && !(ln && K > 768);
  const int wgs32 = ((M + 31) / 32) * nb;
  s->bm = (M > 16 && fits32 && wgs32 >= 200) ? 32 : 16;
  s->nw = 4;
  s->ks = (s->bm == 16 && K >= 1536) ? 2 : 1;
  // the ring: two weight chunks (24 k-steps) per wave, four where a wave has more to read and the registers to hold them
  s->nbuf = (!ln && s->ks == 2 && (K / 32 + 1) / 2 > 2 * CH) ? 4 : 2;   // (behind a LayerNorm the row's values hold those registers)
  // few rows (one sequence per step is the reference scheduler's regime) x a long K: 64-column workgroups would be N / 64 per
  // row block — 12 for mlp.c_proj — and a dozen CUs would stream the whole 4.7 MB.  There a workgroup owns 16 columns and its
  // four waves a quarter of K each: four times the workgroups, a quarter of the bytes per CU (batch 1: 7.4 -> 6.5 us).  At
  // K = 768 the same split was measured behind (the extra meeting in LDS costs more than the thinner stream saves: c_attn
  // 5.5 -> 6.0, c_proj 4.3 -> 5.0 us) and is not used.
  if (s->bm == 16 && ((M + 15) / 16) * nb < 96 && K >= 1536 && !ln) {
    s->nw = 1;
    s->ks = 4;
    s->nbuf = 2;
  }
  return true;
}

static size_t lds_bytes(const Shape& s, int K, bool ln) {
  return (size_t)s.bm * (K + 8) * 2 + (ln ? (size_t)4 * K : 0) + (size_t)(s.ks - 1) * s.nw * (s.bm / 16) * 64 * 16;
}

template <int BM, int NW, int KS, int NBUF, bool LN, int EPI>
static int launch(const LinearParams& p, const Shape& s, hipStream_t stream) {
  const size_t lds = lds_bytes(s, p.K, LN);
  auto* fn = linear_kernel<BM, NW, KS, NBUF, LN, EPI>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_err("hipFuncSetAttribute: ", hipGetErrorString(e));
      return VMI_LAYER_E_HIP;
    }
  }
  const int nb = (p.N + 16 * NW - 1) / (16 * NW), mbs = (p.M + BM - 1) / BM;
  hipLaunchKernelGGL(fn, dim3(nb * mbs), dim3(NW * KS * 64), lds, stream, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_err("launch: ", hipGetErrorString(e));
    return VMI_LAYER_E_HIP;
  }
  return VMI_LAYER_OK;
}

template <int BM, int NW, int KS, int NBUF>
static int launch_le(const LinearParams& p, const Shape& s, bool ln, int epi, hipStream_t stream) {
  if constexpr (NBUF > 2) {   // (the four-chunk ring is picked for shapes without a LayerNorm only: no such kernels are built)
    if (ln || epi > 2) return VMI_LAYER_E_SHAPE;
    switch (epi) {
      case 0: return launch<BM, NW, KS, NBUF, false, 0>(p, s, stream);
      case 1: return launch<BM, NW, KS, NBUF, false, 1>(p, s, stream);
      default: return launch<BM, NW, KS, NBUF, false, 2>(p, s, stream);
    }
  } else
  switch ((ln ? 4 : 0) + epi) {
    case 0: return launch<BM, NW, KS, NBUF, false, 0>(p, s, stream);
    case 1: return launch<BM, NW, KS, NBUF, false, 1>(p, s, stream);
    case 2: return launch<BM, NW, KS, NBUF, false, 2>(p, s, stream);
    case 3: return launch<BM, NW, KS, NBUF, false, 3>(p, s, stream);
    case 4: return launch<BM, NW, KS, NBUF, true, 0>(p, s, stream);
    case 5: return launch<BM, NW, KS, NBUF, true, 1>(p, s, stream);
    case 6: return launch<BM, NW, KS, NBUF, true, 2>(p, s, stream);
    default: return launch<BM, NW, KS, NBUF, true, 3>(p, s, stream);
  }
}

}  // namespace vmi_layer

extern "C" {

static int linear_common(const char* who, const void* x, int64_t ldx, const void* w, const void* bias, const void* ln_gamma,
                         const void* ln_beta, float ln_eps, const void* residual, int64_t ldr, void* y, int64_t ldy, int32_t M,
                         int32_t N, int32_t K, int32_t epilogue, int32_t w_layout, const int64_t* slots, void* kcache,
                         void* vcache, int64_t kv_bstride, int64_t kv_hstride, int qdim, int hsize, int bs, int32_t device,
                         void* stream) {
  using namespace vmi_layer;
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || epilogue < 0 || epilogue > 3 || w_layout < 0 || w_layout > 1 || (!ln_gamma) != (!ln_beta) ||
      (epilogue == VMI_LAYER_EPI_BIAS_RESIDUAL && !residual)) {
    set_err(who, ": null pointer, non-positive size, unknown epilogue or a residual epilogue without a residual");
    return VMI_LAYER_E_ARG;
  }
  const bool ln = ln_gamma != nullptr;
  Shape s;
  if (!pick(M, N, K, ln, &s) || (ldx & 7) || (ldy & 3) || (epilogue == VMI_LAYER_EPI_BIAS_RESIDUAL && (ldr & 3)) ||
      ((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 7)) {
    set_err(who, ": needs K % 32 == 0, N % 16 == 0, K <= 4608 (2048 behind a LayerNorm), 16-byte aligned rows of x and w, 8-byte aligned rows of y");
    return VMI_LAYER_E_SHAPE;
  }
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess || (prev != device && hipSetDevice(device) != hipSuccess)) {
    set_err(who, ": hipSetDevice failed");
    return VMI_LAYER_E_HIP;
  }
  LinearParams p{static_cast<const h16*>(x), ldx, static_cast<const h16*>(w), static_cast<const h16*>(bias),
                 static_cast<const h16*>(ln_gamma), static_cast<const h16*>(ln_beta), ln_eps,
                 static_cast<const h16*>(residual), ldr, static_cast<h16*>(y), ldy, M, N, K,
                 slots, static_cast<h16*>(kcache), static_cast<h16*>(vcache), kv_bstride, kv_hstride, qdim, hsize, bs, bs > 0 ? __builtin_ctz(bs) : 0, w_layout};
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (epilogue == VMI_LAYER_EPI_BIAS_KV_CACHE) s.nbuf = 2;   // (the four-chunk ring is built for bias / GELU / residual only)
  int rc;
  if (s.bm == 32)   // (three 32-row tiles side by side in LDS means K <= 845: never with a split K range)
    rc = launch_le<32, 4, 1, 2>(p, s, ln, epilogue, st);
  else if (s.nw == 1)
    rc = launch_le<16, 1, 4, 2>(p, s, ln, epilogue, st);
  else if (s.ks == 1)
    rc = launch_le<16, 4, 1, 2>(p, s, ln, epilogue, st);
  else
    rc = s.nbuf == 2 ? launch_le<16, 4, 2, 2>(p, s, ln, epilogue, st) : launch_le<16, 4, 2, 4>(p, s, ln, epilogue, st);
  if (prev != device) (void)hipSetDevice(prev);
  return rc;
}

int vmi_gpt2_linear_f16(const void* x, int64_t ldx, const void* w, const void* bias, const void* ln_gamma, const void* ln_beta,
                        float ln_eps, const void* residual, int64_t ldr, void* y, int64_t ldy, int32_t M, int32_t N, int32_t K,
                        int32_t epilogue, int32_t w_layout, int32_t device, void* stream) {
  if (epilogue == VMI_LAYER_EPI_BIAS_KV_CACHE) {
    vmi_layer::g_err = "vmi_gpt2_linear_f16: the cache epilogue has its own entry, vmi_gpt2_linear_qkv_cache_f16";
    return VMI_LAYER_E_ARG;
  }
  return linear_common("vmi_gpt2_linear_f16", x, ldx, w, bias, ln_gamma, ln_beta, ln_eps, residual, ldr, y, ldy, M, N, K, epilogue,
                       w_layout, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, device, stream);
}

int vmi_gpt2_linear_qkv_cache_f16(const void* x, int64_t ldx, const void* w, const void* bias, const void* ln_gamma,
                                  const void* ln_beta, float ln_eps, void* qkv, int64_t ldy, int32_t M, int32_t K,
                                  int32_t w_layout, void* key_cache, void* value_cache, const int64_t* slot_mapping,
                                  int32_t num_heads, int32_t head_size, int32_t block_size, int64_t kv_block_stride,
                                  int64_t kv_head_stride, int32_t device, void* stream) {
  const int64_t E = (int64_t)num_heads * head_size;
  if (!key_cache || !value_cache || !slot_mapping || num_heads <= 0 || head_size <= 0 || block_size <= 0 || (head_size & 15) ||
      (block_size & (block_size - 1)) ||
      kv_block_stride <= 0 || kv_head_stride <= 0 || E > (1 << 20)) {
    vmi_layer::g_err = "vmi_gpt2_linear_qkv_cache_f16: null cache / slot pointer, a head size that is not a multiple of 16 or a block size that is not a power of two";
    return VMI_LAYER_E_ARG;
  }
  return linear_common("vmi_gpt2_linear_qkv_cache_f16", x, ldx, w, bias, ln_gamma, ln_beta, ln_eps, nullptr, 0, qkv, ldy, M,
                       (int32_t)(3 * E), K, VMI_LAYER_EPI_BIAS_KV_CACHE, w_layout, slot_mapping, key_cache, value_cache,
                       kv_block_stride, kv_head_stride, (int)E, head_size, block_size, device, stream);
}

static int with_device(const char* who, int32_t device, int* prev) {
  if (hipGetDevice(prev) != hipSuccess || (*prev != device && hipSetDevice(device) != hipSuccess)) {
    vmi_layer::set_err(who, ": hipSetDevice failed");
    return VMI_LAYER_E_HIP;
  }
  return VMI_LAYER_OK;
}

static int after_launch(const char* who, int prev, int32_t device) {
  const hipError_t e = hipGetLastError();
  if (prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    vmi_layer::set_err(who, ": ");
    vmi_layer::g_err.append(hipGetErrorString(e));
    return VMI_LAYER_E_HIP;
  }
  return VMI_LAYER_OK;
}

int vmi_gpt2_embed_f16(const int64_t* input_ids, const int64_t* position_ids, const void* wte, const void* wpe, void* out,
                       int32_t num_tokens, int32_t hidden, int32_t device, void* stream) {
  using namespace vmi_layer;
  if (!input_ids || !position_ids || !wte || !wpe || !out || num_tokens <= 0 || hidden <= 0 || (hidden & 7) ||
      ((uintptr_t)wte & 15) || ((uintptr_t)wpe & 15) || ((uintptr_t)out & 15)) {
    g_err = "vmi_gpt2_embed_f16: null pointer, or a hidden size / table address that is not a multiple of 8 halves / 16 bytes";
    return VMI_LAYER_E_ARG;
  }
  int prev = -1;
  if (int rc = with_device("vmi_gpt2_embed_f16", device, &prev)) return rc;
  hipLaunchKernelGGL(embed_kernel, dim3(num_tokens), dim3(256), 0, static_cast<hipStream_t>(stream), input_ids, position_ids,
                     static_cast<const h16*>(wte), static_cast<const h16*>(wpe), static_cast<h16*>(out), hidden);
  return after_launch("vmi_gpt2_embed_f16", prev, device);
}

int vmi_gpt2_argmax_f16(const void* logits, int64_t ld, int32_t num_rows, int32_t vocab, int64_t* out, int32_t device,
                        void* stream) {
  using namespace vmi_layer;
  if (!logits || !out || num_rows <= 0 || vocab <= 0 || ld < vocab) {
    g_err = "vmi_gpt2_argmax_f16: null pointer or non-positive size";
    return VMI_LAYER_E_ARG;
  }
  int prev = -1;
  if (int rc = with_device("vmi_gpt2_argmax_f16", device, &prev)) return rc;
  hipLaunchKernelGGL(argmax_kernel, dim3(num_rows), dim3(1024), 0, static_cast<hipStream_t>(stream),
                     static_cast<const h16*>(logits), ld, vocab, out);
  return after_launch("vmi_gpt2_argmax_f16", prev, device);
}

int vmi_gpt2_sample_top_k_f16(const void* logits, int64_t ld, int32_t num_rows, int32_t vocab, int32_t top_k, float temperature,
                              const float* uniform, int64_t* out, int32_t device, void* stream) {
  using namespace vmi_layer;
  if (!logits || !uniform || !out || num_rows <= 0 || vocab <= 0 || ld < vocab || top_k <= 0 || !(temperature > 0.f)) {
    g_err = "vmi_gpt2_sample_top_k_f16: null pointer, non-positive size / top_k / temperature";
    return VMI_LAYER_E_ARG;
  }
  if (vocab > 65536 || top_k > SK_KMAX) {
    g_err = "vmi_gpt2_sample_top_k_f16: vocab <= 65536 and top_k <= 64";
    return VMI_LAYER_E_SHAPE;
  }
  int prev = -1;
  if (int rc = with_device("vmi_gpt2_sample_top_k_f16", device, &prev)) return rc;
  hipLaunchKernelGGL(sample_top_k_kernel, dim3(num_rows), dim3(SK_T), 0, static_cast<hipStream_t>(stream),
                     static_cast<const h16*>(logits), ld, vocab, top_k, 1.f / temperature, uniform, out);
  return after_launch("vmi_gpt2_sample_top_k_f16", prev, device);
}

const char* vmi_gpt2_linear_kernel_name(int32_t M, int32_t N, int32_t K, int32_t has_ln, int32_t epilogue) {
  using namespace vmi_layer;
  static thread_local std::string name;
  Shape s;
  if (!pick(M, N, K, has_ln != 0, &s) || epilogue < 0 || epilogue > 3) return nullptr;
  static const char* epi[] = {"bias", "gelu", "residual", "kvcache"};
  name = "bm" + std::to_string(s.bm) + "_nw" + std::to_string(s.nw) + "_ks" + std::to_string(s.ks) + "_r" + std::to_string(s.nbuf) + (has_ln ? "_ln_" : "_") +
         epi[epilogue];
  return name.c_str();
}

const char* vmi_gpt2_layer_last_error(void) { return vmi_layer::g_err.c_str(); }
int32_t vmi_gpt2_layer_abi_version(void) { return VMI_GPT2_LAYER_ABI_VERSION; }
const char* vmi_gpt2_layer_target_arch(void) { return "gfx950"; }

}  // extern "C"
