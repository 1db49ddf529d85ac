// pa_extras_cache.hip — cache operators OUTSIDE the hot-path scope, linked into
// libvmi_paged_attention_extras.so only: convert_fp8, reshape_and_cache_flash, and the bfloat16-row / E5M2 instantiations
// of the fp8 reshape_and_cache kernel.  The product library links pa_extras_absent.hip instead: empty kernel menus, and none of
// these entries.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vmi_paged_attention_extras.h"
#include "pa_kernel.hpp"
#include "pa_host.hpp"
#include "pa_cache_fp8.hpp"

namespace vmi {

const bool g_has_extras = true;

fp8_scatter_fn fp8_scatter_extra_kernel(bool vec, bool bf, bool e5) {
  if (!bf && !e5) return nullptr;  // float16 rows -> E4M3: the product library's own kernels
  const fp8_scatter_fn fns[8] = {nullptr, nullptr,
                                 (fp8_scatter_fn)reshape_and_cache_fp8_kernel<false, true, false>,  (fp8_scatter_fn)reshape_and_cache_fp8_kernel<true, true, false>,
                                 (fp8_scatter_fn)reshape_and_cache_fp8_kernel<false, false, true>,  (fp8_scatter_fn)reshape_and_cache_fp8_kernel<true, false, true>,
                                 (fp8_scatter_fn)reshape_and_cache_fp8_kernel<false, true, true>,   (fp8_scatter_fn)reshape_and_cache_fp8_kernel<true, true, true>};
  return fns[(e5 ? 4 : 0) + (bf ? 2 : 0) + (vec ? 1 : 0)];
}

// ----------------------------------------------------------------------------------------
// convert_fp8 (cache_kernels.cu:320-392, "only for testing" there): elementwise conversion of a whole cache between
// fp8 E4M3 bytes and float / half / bfloat16 — dst = scaled_convert(src, kv_scale) (quant_utils.cuh):
//   to fp8:   fp8(float(x) / kv_scale), RNE, saturating      from fp8:  half(float(fp8) * kv_scale), bf16(...), float(...)
// KIND: 0 half, 1 bfloat16, 2 float.  Pure streaming: 16 elements per thread.
// ----------------------------------------------------------------------------------------
template <int KIND, bool TO_FP8>
__global__ void __launch_bounds__(256)
    convert_fp8_kernel(void* __restrict__ dst, const void* __restrict__ src, int64_t n, float kv_scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 16;
  for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i0 < n; i0 += stride) {
    const int cnt = (n - i0) < 16 ? (int)(n - i0) : 16;
    for (int e = 0; e < cnt; ++e) {
      const int64_t i = i0 + e;
      if constexpr (TO_FP8) {
        float x;
        if constexpr (KIND == 0) x = (float)static_cast<const h16*>(src)[i];
        else if constexpr (KIND == 1) x = __builtin_bit_cast(float, (uint32_t)static_cast<const uint16_t*>(src)[i] << 16);
        else x = static_cast<const float*>(src)[i];
        static_cast<uint8_t*>(dst)[i] = (uint8_t)f32_to_fp8e4m3_satfinite(x / kv_scale);
      } else {
        const uint32_t b = static_cast<const uint8_t*>(src)[i];
        const float f = __builtin_amdgcn_cvt_pk_f32_fp8((int)b, false)[0];  // exact
        // the two zero codes are written as signed zeros directly: hipcc fuses half(f * s) into v_fma_mixlo_f16(s, f, +0),
        // and (-0 * s) + (+0) is +0 — invisible inside the attention sums, visible in a bit-exact conversion
        const bool zero = (b & 0x7fu) == 0;
        if constexpr (KIND == 0)
          static_cast<uint16_t*>(dst)[i] = zero ? (uint16_t)(b << 8) : __builtin_bit_cast(uint16_t, (h16)(f * kv_scale));
        else if constexpr (KIND == 1)
          static_cast<uint16_t*>(dst)[i] = zero ? (uint16_t)(b << 8) : to_elem<true>(f * kv_scale);
        else
          static_cast<float*>(dst)[i] = zero ? __builtin_bit_cast(float, b << 24) : f * kv_scale;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// reshape_and_cache_flash: scatter new-token rows into the flash layout
// [num_blocks, block_size, num_heads, head_size] — reference cache_kernels.cu:209-240 (kernel),
// :283-317 (host).  A token's H*D row stays contiguous, so this is one 16-B copy per lane per chunk.
// Works for any 2-byte element type (pure copy).
// ----------------------------------------------------------------------------------------
template <bool VEC>
__global__ void __launch_bounds__(256)
    reshape_and_cache_flash_kernel(const h16* __restrict__ key, const h16* __restrict__ value,
                                   h16* __restrict__ kc, h16* __restrict__ vc,
                                   const int64_t* __restrict__ slot_mapping, int64_t block_stride,
                                   int64_t key_stride, int64_t value_stride, int n, int BS) {
  const int64_t token = blockIdx.x;
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // :218-221
  const int64_t dst = (slot / BS) * block_stride + (slot % BS) * (int64_t)n;  // :226-228
  const h16* ks = key + token * key_stride;
  const h16* vs = value + token * value_stride;
  if constexpr (VEC) {
    for (int c = threadIdx.x; c < (n >> 3); c += blockDim.x) {
      *reinterpret_cast<u32x4*>(kc + dst + (c << 3)) = *reinterpret_cast<const u32x4*>(ks + (c << 3));
      *reinterpret_cast<u32x4*>(vc + dst + (c << 3)) = *reinterpret_cast<const u32x4*>(vs + (c << 3));
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      kc[dst + i] = ks[i];
      vc[dst + i] = vs[i];
    }
  }
}

}  // namespace vmi

extern "C" {

int vmi_convert_fp8(void* dst, const void* src, int64_t num_elements, float kv_scale, int32_t kind, int32_t to_fp8,
                    int32_t device, void* stream) {
  using namespace vmi;
  if (num_elements < 0 || kind < 0 || kind > 2) return fail(VMI_E_SHAPE, "convert_fp8: bad arguments (n=%lld kind=%d)", (long long)num_elements, kind);
  if (!(kv_scale > 0.f)) return fail(VMI_E_SHAPE, "convert_fp8: kv_scale must be positive, got %g", (double)kv_scale);
  if (num_elements == 0) return VMI_OK;
  if (!dst || !src) return fail(VMI_E_NULL_POINTER, "convert_fp8: NULL tensor pointer");
  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  typedef void (*cv_fn)(void*, const void*, int64_t, float);
  const cv_fn fns[6] = {(cv_fn)convert_fp8_kernel<0, false>, (cv_fn)convert_fp8_kernel<0, true>,
                        (cv_fn)convert_fp8_kernel<1, false>, (cv_fn)convert_fp8_kernel<1, true>,
                        (cv_fn)convert_fp8_kernel<2, false>, (cv_fn)convert_fp8_kernel<2, true>};
  int64_t blocks = (num_elements + 256 * 16 - 1) / (256 * 16);
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(fns[2 * kind + (to_fp8 ? 1 : 0)], dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dst, src, num_elements, kv_scale);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "convert_fp8 launch");
  return VMI_OK;
}

int vmi_reshape_and_cache_flash_16(const void* key, const void* value, void* k_cache, void* v_cache,
                                   const int64_t* slot_mapping, int32_t num_tokens, int32_t num_heads,
                                   int32_t head_size, int32_t block_size, int64_t block_stride,
                                   int64_t key_stride, int64_t value_stride, int32_t device, void* stream) {
  using namespace vmi;
  if (!key || !value || !k_cache || !v_cache || !slot_mapping)
    return fail(VMI_E_NULL_POINTER, "reshape_and_cache_flash: NULL tensor pointer");
  if (num_tokens < 0 || num_heads <= 0 || head_size <= 0 || block_size <= 0)
    return fail(VMI_E_SHAPE, "reshape_and_cache_flash: bad sizes");
  if (num_tokens == 0) return VMI_OK;
  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  const int n = num_heads * head_size;
  const bool vec = aligned16(key) && aligned16(value) && aligned16(k_cache) && aligned16(v_cache) &&
                   !(key_stride & 7) && !(value_stride & 7) && !(block_stride & 7) && !(n & 7);
  int threads = (((vec ? n >> 3 : n) + 63) / 64) * 64;
  threads = threads > 256 ? 256 : threads;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (vec)
    hipLaunchKernelGGL(reshape_and_cache_flash_kernel<true>, dim3(num_tokens), dim3(threads), 0, st,
                       static_cast<const h16*>(key), static_cast<const h16*>(value), static_cast<h16*>(k_cache),
                       static_cast<h16*>(v_cache), slot_mapping, block_stride, key_stride, value_stride, n,
                       block_size);
  else
    hipLaunchKernelGGL(reshape_and_cache_flash_kernel<false>, dim3(num_tokens), dim3(threads), 0, st,
                       static_cast<const h16*>(key), static_cast<const h16*>(value), static_cast<h16*>(k_cache),
                       static_cast<h16*>(v_cache), slot_mapping, block_stride, key_stride, value_stride, n,
                       block_size);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "reshape_and_cache_flash launch");
  return VMI_OK;
}


}  // extern "C"
