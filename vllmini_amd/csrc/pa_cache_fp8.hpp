// pa_cache_fp8.hpp — fp8 encoders and the fp8 form of reshape_and_cache (device code shared by paged_attention.hip, which
// instantiates the float16 -> E4M3 kernels of the product library, and pa_extras_cache.hip, which instantiates the
// bfloat16 / E5M2 ones and convert_fp8).
#pragma once

#include "pa_kernel.hpp"

namespace vmi {

// ----------------------------------------------------------------------------------------
// reshape_and_cache with kv_cache_dtype "fp8": cache element = fp8_e4m3(float(x) / kv_scale), round to nearest
// even, saturating at +-448, NaN kept (reference cache_kernels.cu:200-205 -> quant_utils.cuh:458-464,
// __nv_cvt_float_to_fp8(..., __NV_SATFINITE, __NV_E4M3)).  Layout x = 16: K[blk, h, d/16, off, d%16], V[blk, h, d, off].
// The conversion is integer arithmetic on the fp32 bit pattern (no dependence on a hardware rounding mode).
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_to_fp8e4m3_satfinite(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  const uint32_t sign = (u >> 24) & 0x80u;
  u &= 0x7fffffffu;
  if (u > 0x7f800000u) return sign | 0x7fu;   // NaN
  if (u >= 0x43e80000u) return sign | 0x7eu;  // |x| >= 464 (the midpoint past 448) and infinity: saturate
  if (u < 0x3c800000u) {                      // |x| < 2^-6: subnormal range, step 2^-9; 8 * 2^-9 encodes as the smallest normal
    return sign | (uint32_t)__builtin_rintf(__builtin_bit_cast(float, u) * 512.f);
  }
  u += 0x7ffffu + ((u >> 20) & 1u);           // RNE to 3 mantissa bits; a carry moves into the exponent
  return sign | ((u >> 20) - ((127u - 7u) << 3));
}

// The same conversion on the gfx950 converter (round 5): v_cvt_pk_fp8_f32 rounds two floats to OCP E4M3 (RNE, subnormals
// included); what it does past the largest finite value is not what __NV_SATFINITE asks for, so magnitudes are clamped to
// 448 first (448 < |x| < 464 rounds to 448 either way) and a NaN keeps its sign with code 0x7f.  Checked bit for bit against
// the integer form above over all 65 536 half values x two scales (tests/test_parity_gpu.py::test_reshape_and_cache_fp8_
// every_half_value_bit_exact) — the integer form stays the definition, this is the fast path of the scatter.
__device__ __forceinline__ uint32_t f32x2_to_fp8e4m3_satfinite_hw(float a, float b) {
  const float ca = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f), cb = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);
  uint32_t r = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(ca, cb, 0, false) & 0xffffu;
  if (a != a) r = (r & 0xff00u) | ((__builtin_bit_cast(uint32_t, a) >> 24) & 0x80u) | 0x7fu;
  if (b != b) r = (r & 0x00ffu) | ((((__builtin_bit_cast(uint32_t, b) >> 24) & 0x80u) | 0x7fu) << 8);
  return r;
}

// fp8 E5M2 (kv_cache_dtype "fp8_e5m2"): the upper byte of an IEEE half.  RNE on 2 mantissa bits, saturating at +-57344
// (__NV_SATFINITE: infinities saturate too), NaN kept as a NaN code.  Integer arithmetic on the fp32 bit pattern.
__device__ __forceinline__ uint32_t f32_to_fp8e5m2_satfinite(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  const uint32_t sign = (u >> 24) & 0x80u;
  u &= 0x7fffffffu;
  if (u > 0x7f800000u) return sign | 0x7fu;   // NaN
  if (u >= 0x47700000u) return sign | 0x7bu;  // |x| >= 61440 (the midpoint past 57344) and infinity: saturate
  if (u < 0x38800000u) {                      // |x| < 2^-14: subnormal range, step 2^-16; 4 * 2^-16 encodes as the smallest normal
    return sign | (uint32_t)__builtin_rintf(__builtin_bit_cast(float, u) * 65536.f);
  }
  u += 0xfffffu + ((u >> 21) & 1u);           // RNE to 2 mantissa bits; a carry moves into the exponent
  return sign | ((u >> 21) - ((127u - 15u) << 2));
}

template <bool VEC, bool BF = false, bool E5 = false>
__global__ void __launch_bounds__(256)
    reshape_and_cache_fp8_kernel(const h16* __restrict__ key, const h16* __restrict__ value,
                                 uint8_t* __restrict__ kc, uint8_t* __restrict__ vc,
                                 const int64_t* __restrict__ slot_mapping, int64_t key_stride,
                                 int64_t value_stride, int H, int D, int BS, float kv_scale) {
  const int64_t token = blockIdx.x;
  const h16* ksrc = key + token * key_stride;
  const h16* vsrc = value + token * value_stride;
  const int n16 = (H * D) >> 4;
  const int cph = D >> 4;  // 16-dim chunks (lanes) per head
  // K: the lane's 16 consecutive dims are one 16-byte unit of the tile.
  // V: the lanes of a head take the rows e*cph + cc (not 16*cc + e), so that store instruction e writes cph CONSECUTIVE
  // rows of the tile — one 64- or 128-byte piece of a line per token instead of cph pieces of cph lines; and both are
  // stored NON-TEMPORALLY: dirty partial lines left in L2 are paid for by the attention launch behind this one
  // (profiles/r02b_call_pair_aftermath.md).
  auto load_rows = [&](int c, h16(&kv)[16], h16(&vv)[16]) {
    const int i = c << 4, h = i / D, d = i - h * D, cc = d >> 4;
    if constexpr (VEC) {
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const h16x8 a = __builtin_bit_cast(h16x8, *reinterpret_cast<const u32x4*>(ksrc + i + 8 * w));
#pragma unroll
        for (int e = 0; e < 8; ++e) kv[8 * w + e] = a[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) kv[e] = ksrc[i + e];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) vv[e] = vsrc[h * D + e * cph + cc];
  };
  // (round 5) this lane's first chunk of the rows is requested BEFORE the slot is consumed: the rows do not depend on the
  // slot, only the stores do — one memory round trip instead of two in a row (rows of padding tokens are valid memory too)
  h16 kv0[16], vv0[16];
  const int c0 = threadIdx.x;
  if (c0 < n16) load_rows(c0, kv0, vv0);
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // padding token (ref cache_kernels.cu:165-169)
  const int64_t blk = slot / BS, off = slot % BS;
  for (int c = c0; c < n16; c += blockDim.x) {
    const int i = c << 4, h = i / D, d = i - h * D;
    const int cc = d >> 4;  // this lane's chunk within its head
    h16 kv[16], vv[16];
    if (c == c0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        kv[e] = kv0[e];
        vv[e] = vv0[e];
      }
    } else {
      load_rows(c, kv, vv);
    }
    u32x4 kq = {0u, 0u, 0u, 0u};
    uint8_t* vdst = vc + ((blk * H + h) * (int64_t)D + cc) * BS + off;
    // kv_scale == 1 (the reference's callers): x / 1.0f is x, and 32 IEEE divisions per lane were half of this kernel's time —
    // one wave per token is instruction-bound, not memory-bound (round 5: 5.2 us for a one-token call)
    auto quantise = [&](auto unit_scale) {
      constexpr bool S1 = decltype(unit_scale)::value;
      auto widen = [&](h16 x) -> float {   // bfloat16 rows (quant_utils.cuh:468-478) widen by a 16-bit shift; float16 rows by v_cvt_f32_f16
        float f = BF ? __builtin_bit_cast(float, (uint32_t)__builtin_bit_cast(uint16_t, x) << 16) : (float)x;
        if constexpr (!S1) f = f / kv_scale;
        return f;
      };
      if constexpr (!E5) {   // E4M3: two values per v_cvt_pk_fp8_f32
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const uint32_t k2 = f32x2_to_fp8e4m3_satfinite_hw(widen(kv[e]), widen(kv[e + 1]));
          const uint32_t v2 = f32x2_to_fp8e4m3_satfinite_hw(widen(vv[e]), widen(vv[e + 1]));
          kq[e >> 2] |= k2 << (8 * (e & 3));
          __builtin_nontemporal_store((uint8_t)(v2 & 0xffu), vdst + (int64_t)(e * cph) * BS);
          __builtin_nontemporal_store((uint8_t)(v2 >> 8), vdst + (int64_t)((e + 1) * cph) * BS);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          kq[e >> 2] |= f32_to_fp8e5m2_satfinite(widen(kv[e])) << (8 * (e & 3));
          __builtin_nontemporal_store((uint8_t)f32_to_fp8e5m2_satfinite(widen(vv[e])), vdst + (int64_t)(e * cph) * BS);
        }
      }
    };
    if (kv_scale == 1.0f) quantise(std::true_type{});
    else quantise(std::false_type{});
    __builtin_nontemporal_store(kq, reinterpret_cast<u32x4*>(kc + (((blk * H + h) * (D >> 4) + (d >> 4)) * BS + off) * 16));
  }
}

typedef void (*fp8_scatter_fn)(const h16*, const h16*, uint8_t*, uint8_t*, const int64_t*, int64_t, int64_t, int, int, int, float);
// the bfloat16-row and E5M2 instantiations (pa_extras_cache.hip); nullptr in the product library
fp8_scatter_fn fp8_scatter_extra_kernel(bool vec, bool bf, bool e5);

}  // namespace vmi
