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
#include <atomic>
#include <mutex>
#include <stdint.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <float.h>

#include "vmi_paged_attention.h"
#ifdef VMI_DIAG
#include "vmi_paged_attention_diag.h"
#endif
#include "pa_kernel.hpp"
#include "pa_split.hpp"
#include "pa_host.hpp"
#include "pa_cache_fp8.hpp"

namespace vmi {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return -(int)e;
}

int not_built(const char* what) {
  return fail(VMI_E_NOT_BUILT, "%s: not in this build of the library (the product library holds the float16 / fp8-E4M3 "
              "path; `python -m vllmini_amd.build --extras` builds libvmi_paged_attention_extras.so)", what);
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
  const int n8 = (H * D) >> 3;
  const h16* ksrc = key + token * key_stride;
  const h16* vsrc = value + token * value_stride;
  // The row loads do not depend on the slot: the first chunk of every lane is requested BEFORE the
  // slot lookup is consumed, so the two memory round trips overlap (rows of skipped tokens are valid
  // memory too; they are simply not stored).
  h16x8 kv0, vv0;
  const int c0 = threadIdx.x;
  if (c0 < n8) {
    if constexpr (VEC) {
      kv0 = __builtin_bit_cast(h16x8, *reinterpret_cast<const u32x4*>(ksrc + (c0 << 3)));
      vv0 = __builtin_bit_cast(h16x8, *reinterpret_cast<const u32x4*>(vsrc + (c0 << 3)));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        kv0[e] = ksrc[(c0 << 3) + e];
        vv0[e] = vsrc[(c0 << 3) + e];
      }
    }
  }
  const int64_t slot = slot_mapping[token];
  if (slot < 0) return;  // padding token (ref cache_kernels.cu:165-169)
  const int64_t blk = slot / BS;
  const int64_t off = slot % BS;
  for (int c = c0; c < n8; c += blockDim.x) {
    const int i = c << 3;
    const int h = i / D;
    const int d = i - h * D;
    h16x8 kv, vv;
    if (c == c0) {
      kv = kv0;
      vv = vv0;
    } else if constexpr (VEC) {
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
// reshape_and_cache, run form (calls of >= 2*block_size tokens).  A prompt's tokens arrive with consecutive slots, so
// BS consecutive tokens usually ARE one cache block.  The per-token kernel above then writes every 32-B V row in BS
// separate 2-byte pieces (1.0-1.2 TB/s read+write on MI355X).  Here a workgroup takes (a run of BS tokens) x (4 wave
// slices of 32 dims of a head; whole heads when the head size is not a multiple of 32).  Each wave checks that the run's slots are one aligned block in order and then writes the whole
// (block, head) tiles: the K tile needs no transposition at all (lane chunk*BS + tok loads 16 B of row tok and owns
// exactly that 16-B unit of the tile), the V tile is transposed through the wave's LDS slice.  A run that is not a whole
// aligned block (prompt tails, decode batches, padding) is written token by token by the same waves.  No workgroup
// is idle in either case, and none of the work is keyed to "every BS-th workgroup" (workgroups are placed round-robin
// over the XCDs by index: that pattern put all the work on one XCD and ran 8x slower).
// Same bytes as the kernel above (cache_kernels.cu:152-207).
// ----------------------------------------------------------------------------------------
template <int BS>
__global__ void __launch_bounds__(256)
    reshape_and_cache_blocks_kernel(const h16* __restrict__ key, const h16* __restrict__ value,
                                    h16* __restrict__ kc, h16* __restrict__ vc,
                                    const int64_t* __restrict__ slot_mapping, int64_t key_stride,
                                    int64_t value_stride, int T, int H, int D, int DW, int spec) {
  // A wave owns DW dims of one head for a run of BS tokens (DW = 32 when D % 32 == 0 — with BS = 16 that is one
  // 16-byte unit per lane, the same parallelism as the per-token kernel — else the whole head).  A dim range is
  // self-contained in both layouts: chunks d/8 of the K tile, rows d of the V tile.
  constexpr int UPR = BS / 8;  // 16-B units per V dim row
  constexpr int PAD = 8;       // halves; keeps LDS rows 16-B aligned and the two 8-token halves on different banks
  extern __shared__ __attribute__((aligned(16))) char blk_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wph = D / DW;                      // waves per head
  const int gw = blockIdx.y * (blockDim.x >> 6) + wave;   // (decode-sized calls come as one-wave workgroups: more CUs share the scatter)
  const int h = gw / wph, d0 = (gw % wph) * DW;
  if (h >= H) return;  // waves are independent: no workgroup barrier below
  const int t0 = blockIdx.x * BS;
  const int nt = (T - t0) < BS ? (T - t0) : BS;
  const long long mine = (lane < nt) ? (long long)slot_mapping[t0 + lane] : -1;
  // spec (decode-sized calls, 32-dim slices): this lane's piece of the token-by-token form below — 16 bytes of a key row,
  // eight halves of a value row — is requested NOW, next to the slots, instead of behind them: the rows do not depend on
  // the slots, only the stores do (one memory round trip instead of two in a row; rows of padding tokens are valid memory)
  u32x4 kv_pre = {0u, 0u, 0u, 0u};
  h16 ve_pre[8];
  const bool pre = spec && DW == 32 && lane < nt * 4;
  if (pre) {
    const int tok = lane >> 2, c = lane & 3;
    kv_pre = *reinterpret_cast<const u32x4*>(key + (int64_t)(t0 + tok) * key_stride + h * D + d0 + c * 8);
    const h16* vsrc = value + (int64_t)(t0 + tok) * value_stride + h * D + d0 + c;
#pragma unroll
    for (int e = 0; e < 8; ++e) ve_pre[e] = vsrc[4 * e];
  }
  const long long s0 = __shfl(mine, 0);
  const bool in_order = lane >= BS || mine == s0 + lane;
  const bool whole = nt == BS && s0 >= 0 && (s0 % BS) == 0 && __all(in_order);
  const int units = DW * BS / 8;  // 16-B units of this wave's slice of one (block, head) tile, K and V alike
  if (whole) {
    const int64_t blk = s0 / BS;
    h16* lds = reinterpret_cast<h16*>(blk_smem) + (size_t)wave * BS * (DW + PAD);
    h16* ktile = kc + ((blk * H + h) * (int64_t)D + d0) * BS;   // chunks d0/8.. of the K tile: a contiguous run
    h16* vtile = vc + ((blk * H + h) * (int64_t)D + d0) * BS;   // rows d0.. of the V tile: a contiguous run
    for (int u = lane; u < units; u += 64) {
      const int c = u / BS, tok = u % BS;
      const u32x4 kv = *reinterpret_cast<const u32x4*>(key + (int64_t)(t0 + tok) * key_stride + h * D + d0 + c * 8);
      const u32x4 vv = *reinterpret_cast<const u32x4*>(value + (int64_t)(t0 + tok) * value_stride + h * D + d0 + c * 8);
      *reinterpret_cast<u32x4*>(ktile + (int64_t)u * 8) = kv;               // K[blk,h,d0/8+c,tok,0..8)
      *reinterpret_cast<u32x4*>(lds + tok * (DW + PAD) + c * 8) = vv;       // V rows, token-major, for the transpose
    }
    // the lanes now read what OTHER lanes of this wave wrote: make the order explicit instead of leaning on in-order
    // LDS execution and on the compiler keeping the two loops apart
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int u = lane; u < units; u += 64) {
      const int row = u / UPR, unit = u % UPR;
      h16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = lds[(unit * 8 + e) * (DW + PAD) + row];
      *reinterpret_cast<u32x4*>(vtile + (int64_t)u * 8) = __builtin_bit_cast(u32x4, o);  // V[blk,h,d0+row,unit*8..+8)
    }
    return;
  }
  // not a whole aligned block: token by token, as reshape_and_cache_kernel does (this wave: dims d0.. of head h of the
  // run's tokens).  This is the DECODE case (every token of the batch in a different block), so it is written for
  // latency: a uniform trip count — every lane stays in the loop, so the slots come from the `mine` registers by
  // shuffle instead of a dependent global load per trip.
  const int c8 = DW >> 3;
  const int total = nt * c8;
  for (int u0 = 0; u0 < total; u0 += 64) {  // wave-uniform
    const int u = u0 + lane;
    const bool act = u < total;
    const int tok = act ? u / c8 : 0, c = u - tok * c8;
    long long slot = __shfl(mine, tok);
    if (!act) slot = -1;
    if (slot >= 0) {  // padding tokens (slot < 0) are skipped (ref cache_kernels.cu:165-169)
      const int64_t blk = slot / BS, off = slot % BS;
      const int i = h * D + d0 + c * 8;
      const bool use_pre = pre && u0 == 0;   // (wave-uniform: with 32-dim slices a run of <= 16 tokens is one trip)
      const u32x4 kv = use_pre ? kv_pre : *reinterpret_cast<const u32x4*>(key + (int64_t)(t0 + tok) * key_stride + i);
      h16* kdst = kc + (((blk * H + h) * (int64_t)(D >> 3) + (d0 >> 3) + c) * BS + off) * 8;
      // NON-TEMPORAL stores.  A decode batch writes 16-byte and 2-byte pieces into 24 different cache lines per
      // (token, head); left dirty in L2 by plain stores they are evicted piecemeal by the attention launch that follows
      // and cost THAT kernel 4-7 us (cfg3: 75.9 -> 68.6 us on ragged lengths, 123.1 -> 118.8 us on equal ones; a no-op
      // kernel in between costs nothing, so it is these lines, not the launch).  Streamed out here they cost this
      // kernel some of that back — the pair is 4 us faster on ragged batches and no slower on equal ones.  Write-
      // through (sc1) stores and touching the lines first were measured too: profiles/r02b_call_pair_aftermath.md.
      __builtin_nontemporal_store(kv, reinterpret_cast<u32x4*>(kdst));
      if (c8 == 4) {
        // V: the 4 lanes of a token take the rows 4e + c (not 8c + e): in store instruction e they write 4 CONSECUTIVE
        // rows of the tile = one 128-byte line, which leaves the wave as one request instead of four
        const h16* vsrc = value + (int64_t)(t0 + tok) * value_stride + h * D + d0 + c;
        h16 ve[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ve[e] = use_pre ? ve_pre[e] : vsrc[4 * e];
        h16* vdst = vc + ((blk * H + h) * (int64_t)D + d0 + c) * BS + off;
#pragma unroll
        for (int e = 0; e < 8; ++e) __builtin_nontemporal_store(ve[e], vdst + (int64_t)(4 * e) * BS);
      } else {
        const h16x8 vv = __builtin_bit_cast(h16x8, *reinterpret_cast<const u32x4*>(value + (int64_t)(t0 + tok) * value_stride + i));
        h16* vdst = vc + ((blk * H + h) * (int64_t)D + d0 + c * 8) * BS + off;
#pragma unroll
        for (int e = 0; e < 8; ++e) __builtin_nontemporal_store(vv[e], vdst + (int64_t)e * BS);
      }
    }
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
// swap_blocks, batched: the pool's preemption move (reference BlockManager.swap_to_cpu / swap_from_cpu,
// vllmini/block_manager.py:70-87, over cache_kernels.cu:24-63's one memcpy per block).  A preempted sequence of
// 1 000 tokens owns 12 x 63 blocks in each cache: 1 512 memcpys of 24 KiB (~6 ms of host calls for 36 MB).  Here ONE
// launch moves every pair of BOTH caches; either side may be pinned host memory, which the GPU addresses directly
// (the stores / loads cross PCIe as full 64-lane x 16-B bursts), so the host's cost is one launch whatever the length.
// grid = (pairs, 2): y = 0 the K cache, 1 the V cache; 256 threads, 16-B moves.
// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    swap_blocks_kernel(const uint8_t* __restrict__ src_k, const uint8_t* __restrict__ src_v, uint8_t* __restrict__ dst_k,
                       uint8_t* __restrict__ dst_v, const int64_t* __restrict__ block_mapping, int64_t block_bytes) {
  const int64_t pair = blockIdx.x;
  const int64_t so = block_mapping[2 * pair] * block_bytes;
  const int64_t doff = block_mapping[2 * pair + 1] * block_bytes;
  const u32x4* s = reinterpret_cast<const u32x4*>((blockIdx.y ? src_v : src_k) + so);
  u32x4* d = reinterpret_cast<u32x4*>((blockIdx.y ? dst_v : dst_k) + doff);
  const int64_t n16 = block_bytes >> 4;
  int64_t i = threadIdx.x;
  for (; i + 768 < n16; i += 1024) {  // four requests in flight per lane: a PCIe round trip is microseconds
    const u32x4 a = s[i], b = s[i + 256], c = s[i + 512], e = s[i + 768];
    d[i] = a;
    d[i + 256] = b;
    d[i + 512] = c;
    d[i + 768] = e;
  }
  for (; i < n16; i += 256) d[i] = s[i];
}

#ifdef VMI_DIAG   // the diagnostic library only (build.py --diag): not in the product .so
// ----------------------------------------------------------------------------------------
// diagnostics (not part of the reference surface): what read bandwidth does this box give a
// plain coalesced 16-B/lane stream?  Used by scripts/bench_diag.py --diag to state the achievable ceiling
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

// diagnostic: read the buffer as pseudo-randomly ordered contiguous chunks of CHUNK_KB KiB, one
// chunk stream per wave, with INFLIGHT_KB KiB requested per wave before anything is consumed (the
// attention kernel's pattern with the math removed; chunk size and queue depth are the variables).
template <int CHUNK_KB, int INFLIGHT_KB, bool NT>
__global__ void __launch_bounds__(256) gather_read_kernel(const u32x4* __restrict__ src, uint32_t nchunks,
                                                          uint32_t stride, uint32_t* __restrict__ sink) {
  constexpr int LPG = INFLIGHT_KB;                                   // 1-KiB loads per group
  constexpr int CPG = INFLIGHT_KB >= CHUNK_KB ? INFLIGHT_KB / CHUNK_KB : 1;  // chunks per group
  constexpr int LPC = CHUNK_KB < INFLIGHT_KB ? CHUNK_KB : INFLIGHT_KB;       // loads per chunk per group
  constexpr int GPC = CHUNK_KB > INFLIGHT_KB ? CHUNK_KB / INFLIGHT_KB : 1;   // groups per chunk
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (uint32_t i = wave * CPG; i < nchunks; i += nwaves * CPG) {
    for (int part = 0; part < GPC; ++part) {
      u32x4 r[LPG];
#pragma unroll
      for (int g = 0; g < CPG; ++g) {
        const uint32_t c = (uint32_t)(((uint64_t)(i + g) * stride) % nchunks);  // stride coprime with nchunks
        const u32x4* base = src + (size_t)c * (CHUNK_KB * 64) + (size_t)part * LPC * 64 + lane;
#pragma unroll
        for (int l = 0; l < LPC; ++l) {
          if constexpr (NT) r[g * LPC + l] = __builtin_nontemporal_load(base + l * 64);
          else r[g * LPC + l] = base[l * 64];
        }
      }
#pragma unroll
      for (int l = 0; l < LPG; ++l) acc ^= r[l];
    }
  }
  const uint32_t x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (x == 0x9e3779b9u) sink[0] = x;
}

#endif  // VMI_DIAG

// ----------------------------------------------------------------------------------------
// host side: variant table, validation, launch
// ----------------------------------------------------------------------------------------
// Core menu: pa_table_core.inc (shared with pa_append_core.hip).
#define VMI_APP false
static Variant g_variants[] = {
#include "pa_table_core.inc"
};
static const int g_ncore = (int)(sizeof(g_variants) / sizeof(g_variants[0]));

#ifndef VMI_DIAG   // the LDS-staging experiment (pa_stage.hip) is linked into the diagnostic library only
static Variant* const g_stage_variants = nullptr;
static const int g_stage_nvariants = 0;
#endif
static int nvariants_v1() {
  return g_ncore + g_extra_nvariants_v1 + g_bf16_nvariants_v1 + g_fp8_nvariants_v1 + g_fp8bf_nvariants_v1 +
         g_fp8_nvariants_v1_e5m2 + g_fp8bf_nvariants_v1_e5m2 + g_queue_nvariants + g_split_nvariants + g_stage_nvariants;
}
static Variant& variant_v1(int id) {  // 1-based over [core fp16][extra fp16][bf16][fp8 cache]
  if (id <= g_ncore) return g_variants[id - 1];
  if (id <= g_ncore + g_extra_nvariants_v1) return g_extra_variants_v1[id - 1 - g_ncore];
  if (id <= g_ncore + g_extra_nvariants_v1 + g_bf16_nvariants_v1)
    return g_bf16_variants_v1[id - 1 - g_ncore - g_extra_nvariants_v1];
  const int f0 = g_ncore + g_extra_nvariants_v1 + g_bf16_nvariants_v1;
  if (id <= f0 + g_fp8_nvariants_v1) return g_fp8_variants_v1[id - 1 - f0];
  const int f1 = f0 + g_fp8_nvariants_v1;
  if (id <= f1 + g_fp8bf_nvariants_v1) return g_fp8bf_variants_v1[id - 1 - f1];  // bf16 query over the fp8 cache
  const int f2 = f1 + g_fp8bf_nvariants_v1;                                        // ... and the E5M2 menus
  if (id <= f2 + g_fp8_nvariants_v1_e5m2) return g_fp8_variants_v1_e5m2[id - 1 - f2];
  const int f3 = f2 + g_fp8_nvariants_v1_e5m2;
  if (id <= f3 + g_fp8bf_nvariants_v1_e5m2) return g_fp8bf_variants_v1_e5m2[id - 1 - f3];
  const int f4 = f3 + g_fp8bf_nvariants_v1_e5m2;
  if (id <= f4 + g_queue_nvariants) return g_queue_variants[id - 1 - f4];  // balanced kernels (pa_queue.hip)
  const int f5 = f4 + g_queue_nvariants;
  if (id <= f5 + g_split_nvariants) return g_split_variants[id - 1 - f5];  // split kernels (pa_split.hip): need a workspace
  return g_stage_variants[id - 1 - f5 - g_split_nvariants];                 // LDS-staged experiment (pa_stage.hip)
}

static bool is_diag(const Variant& v) { return strstr(v.name, "LOADSONLY") != nullptr; }
static bool is_lock(const Variant& v) { return strstr(v.name, "_lock") != nullptr || v.HPT > 1; }

static int find_variant(int D, int BS, int HPW, int WPH, int U, int NT /* -1 = any */, bool bf = false,
                        int f8 = false) {
  for (int id = 1; id <= nvariants_v1(); ++id) {
    const Variant& v = variant_v1(id);
    if (v.F8 == f8 && v.BF == bf && v.D == D && v.BS == BS && v.HPW == HPW && v.WPH == WPH && (U < 0 || v.U == U) &&
        (NT < 0 || v.NT == (bool)NT) && !is_diag(v) && !is_lock(v) && v.UMAX == 0 && !v.GQS && !v.QUEUE && !v.STAGE && !v.XW)
      return id;
  }
  return 0;
}

bool head_size_supported(int d) {  // the reference's switch, attention_kernels.cu:738-766
  return d == 64 || d == 80 || d == 96 || d == 112 || d == 128 || d == 192 || d == 256;
}
bool block_size_supported(int b) { return b == 8 || b == 16 || b == 32; }  // :789-803

// Heuristic (measured on MI355X, profiles/r01c_*):
//  * waves: one wave per (seq, head) once that gives >= 3072 waves (12 per CU); otherwise deal each
//    head's blocks to 2..16 waves so the launch still has ~3072 waves;
//  * queue depth: what matters is BYTES IN FLIGHT PER CU, and more is not better — deeper register
//    rings congest the memory path (cfg3: U=4 133 us, U=2 127 us, U=1 124 us at 12 waves/CU) while too
//    few bytes leave the launch latency-bound (U=1 at 6 waves/CU: 90 us vs 63 us).  U is the smallest
//    of {1,2,4} that keeps >= 24 KiB in flight per CU;
//  * non-temporal page loads once the KV working set exceeds the 256 MiB Infinity Cache;
//  * D = 128 with a full chip: multi-head waves in lockstep give HBM 16-64 KiB bursts (cfg4 660 -> 616 us).
// mean_seq_len: optional hint from a caller that knows the lengths on the host (0 = unknown).  A batch whose mean
// length is well below max_seq_len is RAGGED: with one resident wave per (sequence, head) nothing rebalances the
// chip once the short sequences are done, so the heads are cut into 8 waves each — 8x the workgroups' waves, more
// workgroups than fit at once, and the hardware dispatcher does the balancing
// (profiles/r01g_ragged_batches.md: cfg3 U{1..1024} 98.7 -> 77.1 us, cfg4 446.8 -> 360.5 us).
// Grouped-query attention (num_heads / num_kv_heads = qpk > 1): the largest built group size dividing qpk, one wave
// per group when the launch fills the chip, four or eight waves per group otherwise.  0 = no such kernel.
// Opt-in (vmi_set_pv_mfma): let the grouped-query picks use the "_pvm" kernels, which run probabilities x V on the
// matrix cores as well.  Off by default: those results are within the north-star 1e-3 of the reference kernel, not
// within an ulp of it (pa_kernel.hpp, FPV).
// (thread-local: one host thread opting in must not change another thread's numerics)
static thread_local int g_pv_mfma = 0;
// CU count the heuristics below size their launches for: the launch paths set it from hipDeviceProp (device_cus),
// the pick queries of the C-ABI, which name no device, use the MI355X's 256
static thread_local int g_cus = 256;

static int pick_variant_gqa_of(int num_seqs, int num_heads, int qpk, int head_size, int block_size, int max_seq_len,
                               bool bf, int f8, bool fpv);
static int pick_variant_gqa(int num_seqs, int num_heads, int qpk, int head_size, int block_size, int max_seq_len,
                            bool bf, int f8) {
  if (g_pv_mfma) {
    const int v = pick_variant_gqa_of(num_seqs, num_heads, qpk, head_size, block_size, max_seq_len, bf, f8, true);
    if (v) return v;
  }
  return pick_variant_gqa_of(num_seqs, num_heads, qpk, head_size, block_size, max_seq_len, bf, f8, false);
}
static int pick_variant_gqa_of(int num_seqs, int num_heads, int qpk, int head_size, int block_size, int max_seq_len,
                               bool bf, int f8, bool fpv) {
  if (qpk < 2 || block_size != 16) return 0;
  const int nblk = (max_seq_len + block_size - 1) / block_size;
  const size_t lpad = (size_t)((max_seq_len + 31) / 32) * 32;
  auto fits = [&](const Variant& c) {  // the group's logits (and, with several waves, probabilities) must fit LDS
    return (size_t)c.HPW * c.HPT * (lpad * 4 + 2 * c.WPH * 4 + (size_t)c.WPH * c.D * 4 + (c.WPH > 1 ? lpad * 2 : 0)) <=
           (size_t)160 * 1024;
  };
  for (int g = 8; g >= 2; --g) {
    if (qpk % g) continue;
    const long units = (long)num_seqs * (num_heads / g);
    // g query heads per tile make the wave VALU-bound, so the group's blocks are dealt to several waves even on a
    // full chip: measured best H32/Hkv8 B256 L1024 — D=128: 4 waves per group 189 us (1 wave: 237), D=64: 2 waves
    // 125 us (scripts/gqa_probe.py)
    const long want = head_size >= 128 ? 6144 : 4096;
    int wph = 1;
    while (wph < 8 && units * wph < want && wph * 2 <= (nblk > 0 ? nblk : 1)) wph *= 2;
    int best = 0;
    for (int id = 1; id <= nvariants_v1(); ++id) {
      const Variant& c = variant_v1(id);
      if (!c.GQS || c.XW || c.FPV != fpv || c.BF != bf || c.F8 != f8 || c.D != head_size || c.BS != block_size || c.HPT != g ||
          !fits(c))
        continue;
      if (wph == 1 ? (c.WPH == 1 && (num_heads / g) % c.HPW == 0)
                   : (c.HPW == 1 && c.WPH <= wph && c.WPH * 4 >= wph)) {  // at most 4x fewer waves than wanted
        if (!best || (wph == 1 ? c.HPW > variant_v1(best).HPW || (c.HPW == variant_v1(best).HPW && c.U < variant_v1(best).U)
                               : c.WPH > variant_v1(best).WPH))
          best = id;
      }
    }
    if (best) return best;
    for (int id = 1; id <= nvariants_v1(); ++id) {  // no kernel of the wanted shape: any kernel of this group size
      const Variant& c = variant_v1(id);
      if (c.GQS && !c.XW && c.FPV == fpv && c.BF == bf && c.F8 == f8 && c.D == head_size && c.BS == block_size && c.HPT == g &&
          (num_heads / g) % c.HPW == 0 && fits(c) && units * c.WPH * 4 >= units * wph)  // at most 4x fewer waves than wanted
        return id;
    }
  }
  return 0;
}

// ----------------------------------------------------------------------------------------
// THE MEASURED THRESHOLDS OF THE WORK-DECOMPOSITION HEURISTICS, IN ONE PLACE (round 4).
//
// pick_variant / pick_variant_fp8 / waves_per_head_without_balancing below are control flow over this table: every MEASURED
// threshold is a field here (what remains in the functions are properties of the kernel menu — which U / waves-per-head
// values have rows — and of the hardware: 160 KiB of LDS, 2 / 3 resident workgroups of the balanced kernels).  Thresholds are expressed in RESIDENT WAVES, WORKGROUP SLOTS and
// BYTES, not in batch sizes, so they carry over to other head counts: profiles/r04_pick_generalisation.md compares the
// default pick with the best enumerated variant over H in {8, 16, 20, 25, 40} x D in {64, 128} at the regime edges.
// Each field names the file under profiles/ that holds its measurement.
// ----------------------------------------------------------------------------------------
struct PickRules {
  // ---- the chip ----
  int waves_per_cu_full = 12;        // one wave per (sequence, head) fills the chip at 12 resident waves per CU (3 per SIMD at
                                     //   <= 168 VGPRs): r01c_batch_sweep_auto.csv
  int wave_slots_per_cu = 32;        // hardware wave slots per CU
  size_t lds_per_cu = 160 * 1024;    // bytes
  // ---- how many waves share one head (under-filled chip) ----
  int max_waves_per_head = 16;
  int min_blocks_per_wave = 2;       // a head is cut further only while every wave keeps >= 2 blocks: 256 tokens run 7 - 13 %
                                     //   faster on 8 waves x 2 blocks than on 16 x 1 (r04_underfilled_sweep.md)
  int waves16_max_units_per_cu = 1;  // 16 waves per head = a 1024-thread workgroup = a CU to itself: only while (sequence, head)
                                     //   units <= CUs.  288 / 336 units (batch 24 / 28 at 12 heads) ran 1.33 - 1.59x slower on
                                     //   16 waves per head than on 8 (r04_underfilled_sweep.md)
  // ---- temporal or non-temporal page loads ----
  double nt_kv_bytes = 128e6;        // non-temporal once the launch's K+V bytes pass half the 256 MiB Infinity Cache
                                     //   (146 -> 133 us at cfg3; neutral inside it): r01_cfg3_sweep_cache_policy_bits.json
  double nt_resident_factor = 1.25;  // ... or once the launch's waves exceed 1.25x the resident ones (a second round of
                                     //   workgroups): the temporal kernels keep two V groups in flight across the softmax
                                     //   (VAHEAD, pa_kernel.hpp), which pays only while every workgroup is resident — batch 48 /
                                     //   56 at 12 heads: 5 - 15 % slower than the non-temporal form (r04_underfilled_sweep.md)
  // ---- blocks per register group ----
  double min_kib_in_flight_per_cu = 16.0;   // U = smallest of {1, 2, 4} with >= 16 KiB of register groups per CU.  On a full chip
                                            //   (12 waves per CU) that is U = 1 — more congests: U=4 133, U=2 127, U=1 124 us,
                                            //   r01c_cfg3_variant_sweep.json — and on the under-filled chip U = 2 pays only below
                                            //   ~8 waves per CU (batch 8 x 12 heads on 16 waves each, 2048 tokens: 12.6 against
                                            //   15.0 us) and costs 15 % from 9 waves per CU on (batch 24: 13.6 against 15.9):
                                            //   r04_underfilled_sweep.md (round 1's 24 KiB came from kernels that, as found in
                                            //   round 2, never had two groups in flight)
  // ---- the balanced kernel (pa_queue.hpp): full chip, block 16, head 64 / 128 ----
  size_t q_lds_per_token = 16;       // 4 waves' fp32 logits
  size_t q_lds_fixed = 16 * 1024;    //   + ranking, masks, a team's exchange buffers
  int q_wgs_per_cu_d64 = 3, q_wgs_per_cu_d128 = 2;        // must all be resident (mode S): bounds max_seq_len at ~2400 tokens
  int near_full_num = 7, near_full_den = 8;               // fp16 pages: from 7/8 of the resident waves on (batch 224 at 12
                                                          //   heads) it beats eight waves per head: r03l_nearly_full_chip.md,
                                                          //   r03z_eight_waves_per_head.md
  int near_full_fp8_num = 17, near_full_fp8_den = 20;     // fp8 pages: from 85 % on: r03x_fp8_four_solo_workers.md
  double q_fp8_min_kv_bytes = 128e6;                      // fp8: only past the Infinity Cache (bytes of the fp8 pages)
  // ---- more items than resident waves, or a chip that 2 / 4 waves per head would just fill: MANY waves per head, handed
  //      out by the hardware dispatcher as workgroups finish (r03z_eight_waves_per_head.md) ----
  int many_min_blocks = 32;          // from 512 tokens on (shorter contexts are launch-bound either way)
  int many_waves_long = 8, many_waves_short = 4, many_long_blocks = 64;   // eight from 1024 tokens on, else four
  int many_waves_fp8 = 4, many_fp8_min_blocks = 8;        // fp8 pages: half-size tiles want FOUR waves per head between half a
                                                          //   chip and 85 % of one: r03x_fp8_four_solo_workers.md
  int ragged_hint_num = 3, ragged_hint_den = 4;           // a caller's mean_seq_len below 3/4 of max_seq_len = ragged: up to
  int ragged_hint_waves = 8;                              //   eight waves per head: r01g_ragged_batches.md
  // ---- long contexts on a full chip, the balanced kernel's LDS does not fit (r03m_long_context_full_chip.md) ----
  int long_ctx_tokens = 3400;        // an under-filled chip meets the same LDS limit from here on
  int long_ctx_max_wgs_per_cu = 16;
  double few_waves_base = 0.4, few_waves_slope = 0.075;   // score penalty below 8 resident waves per CU: 0.4 + 0.075 * waves
  int enough_waves_per_cu = 8;
  double two_waves_margin = 1.15;    // two waves per head need not win on paper (lengths are usually ragged there): within 15 %
  double finer_margin = 1.05;        // beyond two, a finer form must be 5 % ahead
  double fine_enough_fill = 0.75;    // eight / sixteen waves per head only where nothing smaller keeps 3/4 of the chip busy
  // ---- head size 128: the lockstep 4-heads-per-wave kernel (one 16-head workgroup per CU: 365 VGPRs) ----
  int lock_heads_per_wg = 16;
  double lock_min_waves_per_cu = 12.0;
  double lock_min_round_fill = 0.97; // only where the launch fills whole rounds of those slots: 129 x 32 heads ran 1120 us
                                     //   against 701: r03n_head_128_round_fit.md
  // ---- split kernels (pa_split.hpp; a workspace is at hand): r05_split_kernels.md ----
  int split_wgs_per_cu = 3;          // workgroups per CU a split launch may have (all resident)
  int split_min_blocks_per_wave = 4; // an item is cut into at most blocks / 4 waves (the kernel trims likewise by seq_len)
  int split_min_blocks_per_wave_loaded = 16;  // ... blocks / 16 once the launch has more workgroups than CUs
  int split_max_units_num = 1, split_max_units_den = 2;   // only while (sequence, head) items <= half the CUs (batch 8 at 12 heads;
                                     //   192 items lost in every cell, 96 won from 8192 tokens on)
  long split_min_item_bytes_per_unit = 20480;   // ... and an item's pages >= 20 KiB x items (head size 64: max_seq_len >= 80 x
                                     //   items — batch 2 from 2048 tokens, batch 4 from 4096, batch 8 from 8192)
  int split_gqa_min_tokens = 1024;   // grouped-query heads: from 1024 tokens on (r05i_split_gqa_sweep_rocprof.json)
  long split_min_item_bytes = 384 * 1024;   // ... and >= 384 KiB (1536 tokens at head size 64: batch 1 at 1024 tokens is level)
  int gate_max_seqs = 2048;          // the gated double launch behind it (launch_pa_v1): every wave of BOTH kernels reads all the
                                     //   lengths for the verdict, bounded by what the balanced kernel ranks in LDS (QSORT_MAX)
  // Head size 128 with MORE items than resident waves and no lockstep fit: eight waves per head as at head size 64 — over
  // H in {8, 16, 20, 25, 40} at 1.0 - 2 x the resident waves the one-wave kernels of the gated double launch ran 3 - 12 %
  // behind on equal lengths and 2 - 8 % on U{1..L} (r04_pick_generalisation.md); level from 3 x on.
};
static constexpr PickRules R{};
static inline long full_chip_waves() { return (long)R.waves_per_cu_full * g_cus; }  // one (sequence, head) per resident wave

static inline size_t lpad32(int max_seq_len) { return (size_t)((max_seq_len + 31) / 32) * 32; }
// LDS of one balanced-kernel workgroup, and whether the CU holds as many as mode S needs
static inline size_t q_lds_bytes(int max_seq_len) { return R.q_lds_per_token * lpad32(max_seq_len) + R.q_lds_fixed; }
static inline bool q_lds_fits(int max_seq_len, int head_size) {
  return head_size == 128 ? R.q_wgs_per_cu_d128 * (q_lds_bytes(max_seq_len) + 4 * 1024) <= R.lds_per_cu
                          : R.q_wgs_per_cu_d64 * q_lds_bytes(max_seq_len) <= R.lds_per_cu;
}
// waves per head that fill the resident waves from `units` (sequence, head) items, every wave keeping min_blocks_per_wave
static int waves_to_fill_the_chip(long units, int nblk) {
  const int nb = nblk > 0 ? nblk : 1;
  int wph = 1;
  while (wph < R.max_waves_per_head && units * wph < full_chip_waves() && wph * 2 * R.min_blocks_per_wave <= nb) wph *= 2;
  if (wph == 1 && units * wph < full_chip_waves() && nb >= 2) wph = 2;   // (two or three blocks: two waves, one of them a block)
  if (wph >= 16 && units > (long)R.waves16_max_units_per_cu * g_cus) wph = 8;
  return wph;
}

// A FULL CHIP WITHOUT THE BALANCED KERNEL (its LDS does not fit: contexts past ~2400 tokens) — round 3,
// profiles/r03m_long_context_full_chip.md.  The one-wave-per-head kernels come as 4-head workgroups whose logits
// (4 bytes per token and head) decide how many fit a CU — three at 3000 tokens, two at 4096, one at 8192 — and a batch
// that is not a multiple of those slots runs a nearly empty last round: 256 x 12 heads at 4096 tokens 548 us (0.74 of
// the roofline), 352: 816 us (0.68); at 8192 tokens one 4-wave workgroup per CU is also too few waves: 1379 us (0.58).
// Workgroups of ONE head cut into 2 / 4 / 8 waves share a logits row (6 bytes per token and head), fit several times
// per CU and quantise finer: 477 us (0.84) / 664 us (0.83) / 947 us (0.85).  Chosen by the fraction of the resident
// slots the launch's rounds keep busy (what equal lengths would see), times a penalty for fewer than 8 waves per CU.
// `wph` = what the launch would use otherwise (1 on a full chip; 2, 4, ... when (sequence, head) units alone do not fill
// it): only that form and finer ones are considered — a chip that is not full and contexts past ~4000 tokens meet the
// same LDS limit (two waves per head: 6 bytes per token, three such workgroups per CU at 8192 tokens = 6 waves; fp8 pages,
// 128 sequences x 12 heads: 380 us = 4.2 TB/s).
static int waves_per_head_without_balancing(int num_seqs, int num_heads, int head_size, int max_seq_len, int nblk,
                                            int wph = 1) {
  const size_t lp = lpad32(max_seq_len);
  auto score = [&](int hpw, int w) -> double {
    const size_t lds = (size_t)hpw * (lp * 4 + 2 * w * 4 + (size_t)w * head_size * 4 + (w > 1 ? lp * 2 : 0));
    long per_cu = (long)(R.lds_per_cu / lds);
    if (per_cu * hpw * w > R.wave_slots_per_cu) per_cu = R.wave_slots_per_cu / (hpw * w);
    if (per_cu > R.long_ctx_max_wgs_per_cu) per_cu = R.long_ctx_max_wgs_per_cu;
    if (per_cu < 1) return 0.0;
    const double slots = (double)g_cus * per_cu;
    const double wgs = (double)num_seqs * ((num_heads + hpw - 1) / hpw);
    const double rounds = (double)(long)((wgs + slots - 1) / slots);
    const double waves = (double)per_cu * hpw * w;
    const double fill = wgs > slots ? wgs / slots : 1.0;  // (a launch that fits one round has no tail)
    return fill / rounds * (waves >= R.enough_waves_per_cu ? 1.0 : R.few_waves_base + R.few_waves_slope * waves);
  };
  double best = wph == 1 ? score(num_heads % 4 == 0 ? 4 : 1, 1) : score(1, wph);
  for (int w = wph * 2; w <= R.max_waves_per_head && w <= (nblk > 0 ? nblk : 1); w *= 2) {
    // eight and sixteen waves per head only where nothing smaller keeps three quarters of the chip busy (a one-head
    // workgroup that fits a CU just once: contexts past ~13 600 tokens)
    if (w >= 8 && best >= R.fine_enough_fill) break;
    const double sc = score(1, w);
    // The score is what EQUAL lengths would see; the lengths are a device tensor and at these contexts batches are
    // usually ragged, where one-head workgroups were ahead in every cell measured (4096 tokens, 320 sequences: 363 against
    // 391 us on U{1..L}, 618 against 599 on equal lengths).  So two waves per head need not win on paper.
    if (w == 2 && wph == 1 ? sc * R.two_waves_margin > best : sc > best * R.finer_margin) {
      best = sc;
      wph = w;
    }
  }
  return wph;
}

// fp8 cache: a (block, head) tile is half the bytes; measured picks in profiles/r01h_fp8_kv.md, r03x_fp8_four_solo_workers.md
// unit_scale: the caller's kv_scale is 1 (the balanced fp8 kernels are built for that case only; the pick queries of
// the C-ABI, which carry no scale, describe the general case)
int pick_variant_fp8(int num_seqs, int num_heads, int head_size, int block_size, int max_seq_len,
                     int mean_seq_len, bool bf, int fmt, bool unit_scale) {
  const long units = (long)num_seqs * num_heads;
  const int nblk = (max_seq_len + block_size - 1) / block_size;
  const bool core = block_size == 16 && (head_size == 64 || head_size == 128);
  int wph = 1;
  // (the tuned menu: the under-filled rules of the fp16 pages — every wave keeps two blocks, sixteen waves per head only while
  //  units <= CUs — hold over fp8 pages too: 256 tokens 10 - 18 % faster on 8 waves x 2 blocks, batch 24 / 28 at 512 tokens
  //  7.6 / 8.1 against 8.9 us; profiles/r04_underfilled_chip.md, r04z_underfilled_sweep_fp8.json)
  if (core) wph = waves_to_fill_the_chip(units, nblk);
  else while (wph < R.max_waves_per_head && units * wph < full_chip_waves() && wph * 2 <= (nblk > 0 ? nblk : 1)) wph *= 2;
  // A nearly full chip goes to the balanced kernel: since its solo workers are four per workgroup over fp8 pages (pa_queue.hpp,
  // WQ_SOLO) it is ahead of several waves per head there on equal lengths (batch 224: 61.4 against 64.4 - 66.7 us) and
  // level on ragged ones (39.7 / 39.9).
  const bool near_full = wph == 2 && units * R.near_full_fp8_den >= full_chip_waves() * R.near_full_fp8_num;
  // Head size 128 (two workgroups per CU): the balanced kernel on a full chip — the lockstep 4-head kernel it replaces is
  // 0.8 % ahead on equal lengths (cfg4 fp8: 330.9 against 333.4 us) and 12 % behind on ragged ones (201.8 against 178.1);
  // fp8 pages have no gated double launch (that pairs two fp16 kernels).
  // (more items than resident waves: FOUR waves per head instead — batch 288 / 384 / 768 at 12 heads, equal lengths 78.4 /
  //  98.6 / 187.3 against the balanced kernel's 77.1 / 105.4 / 195.0 us, U{1..L} 46.3 / 56.7 / 103.2 against 45.4 / 59.3 / 107.7,
  //  "3/4 full" 61.9 / 77.3 / 145.3 against 65.8 / 92.2 / 152.0, exponential 25.5 / 35.2 / 57.3 against 30.5 / 37.6 / 62.8;
  //  behind only on "1/16 full, rest 1/16": 23.3 / 27.0 against 19.7 / 24.7)
  const bool over_full = units > full_chip_waves() && head_size == 64 && block_size == 16 && nblk >= R.many_min_blocks &&
                         q_lds_fits(max_seq_len, 64);
  const bool q64 = head_size == 64 && !over_full && (wph == 1 || near_full) && q_lds_fits(max_seq_len, 64);
  const bool q128 = head_size == 128 && wph == 1 && q_lds_fits(max_seq_len, 128);
  if (unit_scale && (q64 || q128) && !bf && block_size == 16 &&
      2.0 * (double)units * max_seq_len * head_size > R.q_fp8_min_kv_bytes) {  // (1 byte per token and dim, K and V)
    // the balanced kernel over fp8 pages (pa_queue.hpp) — ragged batches without a hint
    const int us = head_size == 64 ? 2 : 1;   // (blocks per group of its mode S: the row's U)
    for (int id = 1; id <= nvariants_v1(); ++id) {
      const Variant& c = variant_v1(id);
      if (c.QUEUE && c.F8 == fmt && c.D == head_size && c.BS == 16 && c.U == us && c.KM) return id;  // K pass on MFMA
    }
    for (int id = 1; id <= nvariants_v1(); ++id) {  // (formats without an "m" kernel: E5M2)
      const Variant& c = variant_v1(id);
      if (c.QUEUE && c.F8 == fmt && c.D == head_size && c.BS == 16 && c.U == us) return id;
    }
  }
  // Half a chip to 85 % of one (batch 128 .. 217 at 12 heads): FOUR waves per head, not the two that would just fill the
  // resident waves — an fp8 tile is half the bytes, two waves per head leave the memory system short of requests, and twice
  // the resident waves let the dispatcher balance a ragged batch: batch 128 / 144 / 176 / 208, equal lengths 42.2 / 42.6 /
  // 50.5 / 61.6 -> 36.9 / 41.3 / 51.5 / 59.5 us, U{1..L} 34.2 / 34.1 / 36.9 / 40.7 -> 26.5 / 27.8 / 32.5 / 37.8 (the balanced
  // kernel there: 42.6 / 43.5 / 52.6 / 61.2 and 31.2 / 32.2 / 34.4 / 39.5).  Below batch 128 the rule above already gives four.
  if (head_size == 64 && block_size == 16 && wph == 2 && nblk >= R.many_fp8_min_blocks) wph = R.many_waves_fp8;
  if (over_full && wph == 1) wph = R.many_waves_fp8;
  if (mean_seq_len > 0 && (long)mean_seq_len * R.ragged_hint_den < (long)max_seq_len * R.ragged_hint_num)
    while (wph < R.ragged_hint_waves && wph * 2 <= (nblk > 0 ? nblk : 1)) wph *= 2;
  // (a full chip past the balanced kernel's LDS: the same choice by round efficiency as over fp16 pages — the logits are
  //  fp32 either way; fp8 pages at 8192 tokens ran 906 us = 3.6 TB/s with one 4-head workgroup per CU)
  if (head_size == 64 && block_size == 16 && 2.0 * (double)units * max_seq_len * head_size > R.q_fp8_min_kv_bytes &&
      !q_lds_fits(max_seq_len, 64) && (wph == 1 || max_seq_len > R.long_ctx_tokens))
    wph = waves_per_head_without_balancing(num_seqs, num_heads, head_size, max_seq_len, nblk, wph);
  int v = 0;
  if (block_size == 16 && (head_size == 64 || head_size == 128)) {
    if (wph == 1) {
      v = find_variant(head_size, 16, (num_heads % 4 == 0) ? 4 : 1, 1, head_size == 64 ? 2 : 1, -1, bf, fmt);
    } else {
      // (sixteen waves per head = a nearly empty chip: the temporal two-blocks-per-group kernel where the pages fit the
      //  Infinity Cache — batch 1 ... 20 at 512 ... 2048 tokens 3 - 10 % ahead of the non-temporal one-block form)
      if (wph == 16 && 2.0 * (double)units * max_seq_len * head_size <= R.nt_kv_bytes) v = find_variant(head_size, 16, 1, 16, 2, 0, bf, fmt);
      for (int ww = wph; ww >= 1 && !v; ww /= 2) v = find_variant(head_size, 16, 1, ww, -1, -1, bf, fmt);
    }
  }
  if (!v && wph > 1) v = find_variant(head_size, block_size, 1, wph >= 16 ? 16 : 4, -1, -1, bf, fmt);
  if (!v) v = find_variant(head_size, block_size, 1, wph == 1 ? 1 : 4, -1, -1, bf, fmt);
  if (!v) v = find_variant(head_size, block_size, 1, 1, -1, -1, bf, fmt);
  return v;
}

static int pick_variant(int num_seqs, int num_heads, int head_size, int block_size, int max_seq_len,
                        bool bf = false, int mean_seq_len = 0, bool allow_balanced = true) {
  const long units = (long)num_seqs * num_heads;
  const int nblk = (max_seq_len + block_size - 1) / block_size;
  const bool core = block_size == 16 && (head_size == 64 || head_size == 128);   // the tuned menu (pa_table_core.inc)
  // (the other head / block sizes have one-wave and four-wave kernels only: the plain fill rule decides between them)
  int wph = 1;
  if (core) wph = waves_to_fill_the_chip(units, nblk);
  else while (wph < R.max_waves_per_head && units * wph < full_chip_waves() && wph * 2 <= (nblk > 0 ? nblk : 1)) wph *= 2;
  const double kv_bytes = 4.0 * (double)units * (double)max_seq_len * head_size;
  // (a batch the caller knows to be ragged: many waves per head, so that the hardware dispatcher balances the chip —
  //  except where the balanced kernel below does that itself, from the lengths it reads on the device)
  // The balanced kernel also serves a chip that is only NEARLY full: its one-item-per-wave mode then beats two waves per
  // head on equal lengths and its ranked modes beat it on ragged ones (r03l_nearly_full_chip.md; since the alternative
  // below the threshold is EIGHT waves per head the crossover sits at 7/8 of the resident waves: batch 208: 103.3 / 58.7 us
  // against the balanced kernel's 104.6 / 61.9 on equal / ragged lengths, 224: 110.5 / 64.7 against 109.8 / 67.1).
  const bool near_full = wph == 2 && units * R.near_full_den >= full_chip_waves() * R.near_full_num;
  const bool lds_fits_q = q_lds_fits(max_seq_len, 64);
  // MORE items than resident waves: several waves per head again, i.e. many times the resident waves, handed out by the
  // hardware dispatcher as workgroups finish.  Measured against the balanced kernel's ranked hand-out over 12 length
  // distributions (scripts/default_vs_waves_probe.py, r03z_eight_waves_per_head.md): equal lengths level, continuous
  // spreads 3 - 13 % faster, bimodal ones 3 - 9 %; behind only where nearly every sequence is very short.  The balanced
  // kernel keeps the chip it was built for: 7/8 ... 1 x the resident waves.
  const bool nt_by_bytes = kv_bytes > R.nt_kv_bytes;
  // (the lockstep 4-heads-per-wave kernel needs the whole register file: ONE 16-head workgroup per CU, so it pays only
  //  where the launch's workgroups fill whole rounds of those slots — 128 x 32 heads: 630 us against 641 for the
  //  4-head workgroups; 129 sequences: 1120 against 701, 160: 1142 against 797, 96: 571 against 483 — and only as the front
  //  half of a gated double launch, which is bounded in sequences)
  const double lock_fill = (double)num_seqs * (num_heads / R.lock_heads_per_wg) / (double)g_cus;
  const double lock_eff = lock_fill / (double)(long)(lock_fill + 0.999999);
  // (... and only while the 16 heads' logits fit the workgroup's LDS — 4 bytes per token and head: up to ~2500 tokens.  Without
  //  this term contexts of 3400 ... 16384 tokens at 16 / 32 heads were handed a kernel that launch_pa_v1 then had to replace by
  //  one wave per head, and the eight-waves-per-head rule for a full chip never saw them: round-4 advisor finding)
  const bool lock_lds_fits = (size_t)R.lock_heads_per_wg * (lpad32(max_seq_len) * 4 + 2 * 4 + (size_t)head_size * 4) <= R.lds_per_cu;
  const bool lock_ok = head_size == 128 && block_size == 16 && nt_by_bytes && num_heads % R.lock_heads_per_wg == 0 &&
                       (double)units / (double)g_cus >= R.lock_min_waves_per_cu && lock_eff >= R.lock_min_round_fill &&
                       num_seqs <= R.gate_max_seqs && lock_lds_fits;
  // (head size 128: from EXACTLY the resident waves on — there the one-wave kernels are level on equal lengths and 3 - 6 % behind on U{1..L})
  const bool over_full = core && nblk >= R.many_min_blocks &&
                         (head_size == 64 ? units > full_chip_waves() && lds_fits_q : units >= full_chip_waves() && !lock_ok);
  const bool balanced = allow_balanced && !over_full && (wph == 1 || near_full) && nt_by_bytes && block_size == 16 && head_size == 64 &&
                        lds_fits_q;
  const bool ragged = !balanced && mean_seq_len > 0 && (long)mean_seq_len * R.ragged_hint_den < (long)max_seq_len * R.ragged_hint_num;
  if (ragged)
    while (wph < R.ragged_hint_waves && wph * 2 <= (nblk > 0 ? nblk : 1)) wph *= 2;
  if (over_full && wph == 1) wph = nblk >= R.many_long_blocks ? R.many_waves_long : R.many_waves_short;
  // A chip that two or four waves per head would just fill (batch 64 .. 223 at 12 heads) gets EIGHT, hint or no hint: several
  // times the resident waves cost equal lengths nothing and let the hardware dispatcher balance a ragged batch (head size 128
  // likewise, by a smaller margin).  Where the long-context scoring below does not apply.
  if (!balanced && !over_full && lds_fits_q && core && (wph == 2 || wph == 4) && nblk >= R.many_min_blocks) wph = R.many_waves_long;
  // (only where the balanced kernel's LDS does not fit: shorter contexts keep the tuned picks — the fused append, which
  //  has no balanced twin, stays bit-identical to the call pair there)
  if (!balanced && !lds_fits_q && nt_by_bytes && head_size == 64 && block_size == 16 && (wph == 1 || max_seq_len > R.long_ctx_tokens))
    wph = waves_per_head_without_balancing(num_seqs, num_heads, head_size, max_seq_len, nblk, wph);
  // temporal loads (and two V groups across the softmax) only while the working set is cache-sized AND every workgroup is resident
  const int nt = (nt_by_bytes || (wph > 1 && (double)units * wph > R.nt_resident_factor * (double)full_chip_waves())) ? 1 : 0;
  if (core) {  // core table: full menu
    const double waves_per_cu = (double)units * wph / (double)g_cus;
    const double tile_kib = head_size * 16 * 2 / 1024.0;
    int u = 1;
    while (u < 4 && waves_per_cu * u * tile_kib < R.min_kib_in_flight_per_cu) u *= 2;
    while (u > 1 && wph > 1 && (long)u * wph > (nblk > 0 ? nblk : 1)) u /= 2;   // (never more register slots than a wave has blocks)
    if (balanced && (u == 1 || near_full)) {
      // full chip: the balanced kernel (pa_queue.hpp) — it reads seq_lens on the device and runs one wave per
      // (sequence, head) on equal lengths, ranked work lists on ragged ones; needs 3 workgroups' LDS per CU
      for (int id = 1; id <= nvariants_v1(); ++id) {
        const Variant& c = variant_v1(id);
        if (c.QUEUE && c.BF == bf && c.D == head_size && c.BS == 16 && !c.F8 && !c.KM) return id;
      }
    }
    if (wph == 1 && u == 1 && nt) {  // full chip, one wave per head: the adaptive-depth form where one is built
      for (int id = 1; id <= nvariants_v1(); ++id) {
        const Variant& c = variant_v1(id);
        if (c.BF == bf && c.D == head_size && c.BS == 16 && c.UMAX > 0 && c.WPH == 1 && c.U == 1 && !c.QUEUE &&
            c.HPW == ((num_heads % 4 == 0) ? 4 : 1))
          return id;
      }
    }
    if (wph == 1 && lock_ok) {
      for (int id = 1; id <= nvariants_v1(); ++id) {  // d128_mh4_h4_u1_nt1_lock
        const Variant& c = variant_v1(id);
        if (c.BF == bf && c.D == 128 && c.HPT == 4 && c.HPW == 4 && c.U == 1 && !c.QUEUE) return id;
      }
    }
    const int hpw = (wph == 1 && num_heads % 4 == 0) ? 4 : 1;
    int v = 0;
    for (int uu = u; uu >= 1 && !v; uu /= 2) {  // nearest available depth at or below the target
      v = find_variant(head_size, 16, hpw, wph, uu, nt, bf);
      if (!v) v = find_variant(head_size, 16, hpw, wph, uu, -1, bf);
    }
    for (int uu = u * 2; uu <= 8 && !v; uu *= 2) v = find_variant(head_size, 16, hpw, wph, uu, -1, bf);
    if (!v && wph > 1) {  // smaller menus (bf16): nearest available waves-per-head
      for (int ww = wph / 2; ww >= 1 && !v; ww /= 2) v = find_variant(head_size, 16, 1, ww, -1, -1, bf);
    }
    if (!v) v = find_variant(head_size, 16, 1, 1, -1, -1, bf);
    return v;
  }
  // extra table: one wave or four waves per head
  int v = find_variant(head_size, block_size, 1, wph == 1 ? 1 : 4, -1, -1, bf);
  if (!v) v = find_variant(head_size, block_size, 1, 1, -1, -1, bf);
  return v;
}

// a long max_seq_len may not leave room for several heads' logits in one workgroup's LDS: fall back to one head per
// workgroup, then to one wave per head (no second copy of the probabilities) before giving up
// logits per wave of a split kernel: its share of max_seq_len's blocks, or the 8 blocks a wave can meet when the kernel uses
// fewer waves for a shorter context (pa_split.hpp, NWe)
static inline int split_wtok(int lpad, int xw) {
  const int b = (lpad / 16 + xw - 1) / xw;
  return 16 * (b > 8 ? b : 8);
}
static size_t variant_lds_bytes(const Variant& c, int lpad);
static int fit_lds(int variant, int head_size, int block_size, int lpad, bool bf, int f8) {
  if (variant >= 1 && variant <= nvariants_v1() && variant_lds_bytes(variant_v1(variant), lpad) > R.lds_per_cu) {
    const int wph0 = variant_v1(variant).WPH;
    int alt = find_variant(head_size, block_size, 1, wph0, -1, -1, bf, f8);
    if (!alt || variant_lds_bytes(variant_v1(alt), lpad) > R.lds_per_cu) alt = find_variant(head_size, block_size, 1, 1, -1, -1, bf, f8);
    if (alt) variant = alt;
  }
  return variant;
}

// ---- split kernels (pa_split.hpp): a (sequence, head) over XW waves = XW / 4 workgroups of one launch ----
// Every workgroup of the launch must be resident (an item's waves wait for one another): the kernels' launch bounds hold
// them to six 4-wave workgroups per CU at head size 64 and three at 128; the picks below ask for at most split_wgs_per_cu.
static inline int split_resident_wgs(int head_size, int hpt = 1) {
  const int n = (hpt > 1 ? (head_size == 64 ? 4 : 2) : (head_size == 64 ? 6 : 3)) * g_cus;   // the kernels' launch bounds
  return n < SPLIT_MAX_WGS ? n : SPLIT_MAX_WGS;
}
static int find_split(int D, int xw, int U, int nt, int f8 = 0, int hpt = 1) {
  for (int i = 0; i < g_split_nvariants; ++i) {
    const Variant& c = g_split_variants[i];
    if (c.D == D && c.XW == xw && c.U == U && c.NT == (bool)nt && c.F8 == f8 && c.HPT == hpt)
      return nvariants_v1() - g_stage_nvariants - g_split_nvariants + i + 1;
  }
  return 0;
}
// Which split kernel, if any, serves a launch the plain heuristic gave `plain` to.  0 = keep the plain pick.
// Measured (profiles/r05_split_kernels.md, rocprofv3, 12 heads x 64, batch 1 ... 16 x 1024 ... 16384 tokens): an item's waves
// pay two trips through memory (exchange of max / exp-sum: 1.5 us on an idle chip, 2.5 - 2.9 us under load; partial rows and
// the last arriver's merge: 1.7 - 2.4 us), so spreading an item pays where its work on ONE CU lasts much longer than that:
// few items, long contexts — batch 1 at 16384 tokens 53.4 -> 19.0 us, batch 4 at 8192 30.2 -> 22.0, batch 8 at 16384
// 98.4 -> 70.5 — and loses below (batch 1 at 1024 tokens 8.5 -> 8.8, BASELINE configs[1] 10.5 -> 17.0).
static int pick_split(int num_seqs, int num_heads, int head_size, int block_size, int max_seq_len, int plain, int qpk = 1,
                      int f8 = 0) {
  if (block_size != 16 || (head_size != 64 && head_size != 128) || g_split_nvariants == 0) return 0;
  const long units = (long)num_seqs * num_heads;
  bool starved = false;
  if (plain >= 1 && plain <= nvariants_v1()) {
    const Variant& pv = variant_v1(plain);
    // (grouped-query picks included: with few items the gq kernels — one tile load for several query heads, but a KV head's
    //  whole context on ONE CU — lose to the split kernels by 2 - 6 x although those read a tile once per QUERY head (the
    //  repeats hit L2): 32 / 8 heads x 128, batch 1 x 4096 tokens 56 -> 17 us, x 16 384 164 -> 26, batch 4 x 8192 159 -> 43;
    //  16 / 4 heads x 64, batch 4 x 8192 130 -> 20: profiles/r05h_split_gqa_rocprof.json)
    if (pv.QUEUE || pv.XW) return 0;
    // ONE wave per head on a chip those waves do not fill: the plain pick fell back there because several waves' logits and
    // probabilities (6 bytes per token) no longer fit a workgroup's LDS — from ~27 000 tokens on; batch 1 at 32768 tokens ran
    // 1487 us that way, 24.5 us split (a wave of a split kernel holds its own blocks' logits only)
    starved = pv.WPH == 1 && units * 2 < full_chip_waves();
    if (pv.WPH == 1 && !starved) return 0;   // a full chip
  }
  const long item_bytes = 4L * max_seq_len * head_size;   // K and V pages of one (sequence, head)
  bool few = units * R.split_max_units_den <= (long)g_cus * R.split_max_units_num;
  bool big = item_bytes >= R.split_min_item_bytes_per_unit * units && item_bytes >= R.split_min_item_bytes;
  int hpt = 1;   // query heads per item
  if (qpk > 1) {
    // Grouped-query heads: the alternative is a gq kernel that keeps a KV head's whole context on one CU, and it loses from
    // 1024 tokens on wherever the split launch stays within the resident workgroups (r05i_split_gqa_sweep_rocprof.json:
    // 32 / 8 heads x 128, batch 1 ... 4 x 1024 ... 8192 tokens 0.60 ... 0.13 of the gq kernel's time; 16 / 4 x 64 up to batch
    // 16: 0.96 ... 0.14).  Two forms (r05l_split_gq4_rocprof.json): ONE query head per item — every query head's waves read
    // their KV head's tiles themselves, the repeats hit L2 — is ahead while query heads < CUs / 2 (32 / 8 x 128, batch 1 x 8192:
    // 19.7 against 22.8 us); FOUR query heads of a KV head per item — every tile loaded once, q.K^T on the matrix cores — from
    // CUs / 2 query heads on (batch 4 x 8192: 43 -> 35 us; batch 8 x 4096 / 8192, where the first form no longer fits: 64 / 164
    // -> 34 / 58), up to CUs / 2 such items.
    big = max_seq_len >= R.split_gqa_min_tokens;
    if (qpk % 4 == 0 && f8 != 2 && units * 2 >= (long)g_cus) {   // (fp8 E4M3 pages too: r05q_split_gq4_fp8_rocprof.json)
      hpt = 4;
      few = (units / 4) * 2 <= (long)g_cus;
      if ((units / 4) * 4 > (long)g_cus) big = max_seq_len >= 2 * R.split_gqa_min_tokens;   // (128 such items at 1024 tokens: 23.9 against 22.9 us)
    } else {
      few = units * (head_size == 128 ? 2 : 1) <= (long)g_cus;
    }
  }
  if (!starved && !(few && big)) return 0;
  const long nitems = units / hpt;
  const int nblk = (max_seq_len + 15) / 16;
  // workgroups the launch may have (all resident; four heads per item: their partial rows bound it too)
  long cap = (long)R.split_wgs_per_cu * g_cus;
  if (hpt > 1) cap = cap < split_resident_wgs(head_size, hpt) ? cap : split_resident_wgs(head_size, hpt);
  if (hpt > 1 && cap > SPLIT_MAX_WGS / hpt) cap = SPLIT_MAX_WGS / hpt;
  // the most waves per item such that the launch's workgroups fit the CUs once with >= 4 blocks per wave, or up to
  // split_wgs_per_cu times with >= 16 (a loaded chip's exchange is slower: batch 4 at 4096 tokens 21.3 us on 768 workgroups of
  // 4-block waves, 16.3 on 384 of 8-block ones, 19.0 unsplit; batch 2 at 16384 tokens 27.3 us on 768 workgroups, 22.8 on 384)
  const int lpad = (int)lpad32(max_seq_len);
  auto lds_fits = [&](int x) { return (size_t)4 * hpt * ((size_t)split_wtok(lpad, x) * 6 + (size_t)head_size * 4) + 16 <= R.lds_per_cu; };
  int xw = 0;
  for (int x = hpt > 1 ? 64 : SPLIT_MAX_WAVES; x >= 8; x /= 2) {
    const long wgs = nitems * (x / 4);
    const int bpw = nblk / x;
    if (lds_fits(x) &&
        ((wgs <= g_cus && bpw >= R.split_min_blocks_per_wave) || (wgs <= cap && bpw >= R.split_min_blocks_per_wave_loaded))) { xw = x; break; }
  }
  if (starved) {
    // Contexts too long for several waves' logits in one workgroup's LDS: at least the narrowest form of which three workgroups
    // fit a CU's LDS — in ROUNDS (Variant::fn_rounds) where that is more workgroups than are resident
    // (r05n_split_rounds_rocprof.json, 12 heads x 64: batch 64 x 32768 tokens 4574 us one wave per head, 1131 at 8 waves per
    // item in one round, 959 at 16 in rounds; batch 32: 8 waves 566, 16 waves 484; batch 80: 6073 -> 1229 (8 / 32 / 64 waves:
    // 1476 / 1240 / 1306); batch 40 x 65536: 16 waves — one workgroup per CU — 1487, 32: 1236, 64: 1268)
    int x3 = 0;
    for (int x = 16; x <= (hpt > 1 ? 64 : 128); x *= 2) {
      const size_t lds = (size_t)4 * hpt * ((size_t)split_wtok(lpad, x) * 6 + (size_t)head_size * 4) + 16;
      if (3 * (lds + 1024) <= R.lds_per_cu) { x3 = x; break; }
    }
    if (x3 > xw) xw = x3;
  }
  if (!xw) return 0;
  const double kv_bytes = (f8 ? 2.0 : 4.0) * (double)units / (double)(qpk > 0 ? qpk : 1) * (double)max_seq_len * head_size;
  // (one query head per item over grouped-query heads: the other query heads' reads of a tile must HIT L2 — never non-temporal:
  //  32 / 8 x 128, batch 4 x 8192 tokens 43.2 us with temporal loads, 62.8 with non-temporal ones)
  const int nt = (kv_bytes > R.nt_kv_bytes && !(qpk > 1 && hpt == 1)) ? 1 : 0;
  if (hpt > 1) {
    int vg = find_split(head_size, xw, (head_size == 64 || f8) ? 2 : 1, nt, f8, hpt);
    if (!vg) vg = find_split(head_size, xw, head_size == 64 ? 1 : 2, nt, f8, hpt);
    return vg;
  }
  int v = f8 ? find_split(head_size, xw, 4, nt, f8) : 0;   // (fp8 pages, half-size tiles: four blocks per group 3 - 5 % ahead of two)
  if (!v) v = find_split(head_size, xw, 2, nt, f8);   // (two blocks per register group: ahead of one in 23 of 25 cells)
  if (!v) v = find_split(head_size, xw, 1, nt, f8);
  if (!v && xw > 128) v = find_split(head_size, 128, 2, nt, f8);   // (the fp8 menu ends at 128 waves per item)
  return v;
}

// ---- per-device facts, guarded by one mutex (several host threads may drive several GPUs through this library) ----
constexpr int MAX_DEVICES = 64;
struct DeviceState {
  int cus = 0;  // CU count (hipDeviceProp), read once per device
};
static std::mutex g_dev_mutex;
static DeviceState g_dev[MAX_DEVICES];
#ifdef VMI_DIAG
static int env_int(const char* name) {
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
}
// q_flags bits 2-4 (QF_WQ, pa_queue.hpp) = solo workers per workgroup: a workgroup has 4 waves, so only 0 (default) .. 4 name workers that
// exist — a larger count would hand items to waves that are not there and leave their outputs unwritten
static int clamp_queue_flags(int flags) { return ((flags >> 2) & 7) > 4 ? (flags & ~(7 << 2)) | (4 << 2) : flags; }
// test / bench knob for the balanced kernels (pa_queue.hpp QF_*), diagnostic library only; initial value from
// VMI_QUEUE_FLAGS.  The product library always passes 0: the kernel decides everything from seq_lens.
static thread_local int g_queue_flags = clamp_queue_flags(env_int("VMI_QUEUE_FLAGS"));
// ... and for the split kernels (pa_split.hpp SPF_*); initial value from VMI_SPLIT_FLAGS
static thread_local int g_split_flags = env_int("VMI_SPLIT_FLAGS");
#else
static constexpr int g_queue_flags = 0;
static constexpr int g_split_flags = 0;
#endif
static thread_local int g_last_variant = 0;  // what this thread's last paged_attention_v1 launch ran (0: none yet / block-sparse)
static thread_local int g_last_partner = 0;  // ... and the balanced kernel launched behind it in a gated double launch (0: none)

static int device_cus(int device) {  // caller holds the device current
  if (device < 0 || device >= MAX_DEVICES) return 256;
  std::lock_guard<std::mutex> lk(g_dev_mutex);
  if (!g_dev[device].cus) {
    hipDeviceProp_t prop;
    g_dev[device].cus = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
                            ? prop.multiProcessorCount : 256;
  }
  return g_dev[device].cus;
}

// dynamic LDS a kernel needs for logits rows of `lpad` floats (max_seq_len padded to 32)
static size_t variant_lds_bytes(const Variant& c, int lpad) {
  if (c.XW)     // per wave: its share of the logits (fp32) and probabilities (fp16) + one partial output row
    return (size_t)c.WPH * c.HPT * ((size_t)split_wtok(lpad, c.XW) * 6 + (size_t)c.D * 4) + 16;   // + the "I am last" flag
  if (c.STAGE)  // per wave: the logits + a ring of U slots, each one (block, head) tile
    return (size_t)4 * ((size_t)lpad * 4 + (size_t)c.U * (c.D * 16 * (c.F8 ? 1 : 2)));
  if (c.QUEUE)  // 4 waves' logits + the ranking, its bucket counts and masks + a team's exchange buffers (pa_queue.hpp)
    return (size_t)4 * lpad * 4 + (size_t)2048 * 6 + 2048 + 32 + (size_t)4 * c.D * 4;
  return (size_t)c.HPW * c.HPT *
         ((size_t)lpad * 4 + 2 * c.WPH * 4 + (size_t)c.WPH * c.D * 4 + (c.WPH > 1 ? (size_t)lpad * 2 : 0) +
          (c.SPARSE ? (size_t)lpad / 2 : 0));  // SPARSE: the list of attended blocks, one int per block (BS >= 8)
}

// fused-append twin of a v1 variant id (same row of the same menu, pa_append_*.hip)
static Variant* app_variant_v1(int id) {
  if (g_app_core_nvariants != g_ncore || g_app_extra_nvariants != g_extra_nvariants_v1 ||
      g_app_bf16_nvariants != g_bf16_nvariants_v1)
    return nullptr;
  if (id <= g_ncore) return &g_app_core_variants[id - 1];
  if (id <= g_ncore + g_extra_nvariants_v1) return &g_app_extra_variants[id - 1 - g_ncore];
  if (id <= g_ncore + g_extra_nvariants_v1 + g_bf16_nvariants_v1)
    return &g_app_bf16_variants[id - 1 - g_ncore - g_extra_nvariants_v1];
  return nullptr;  // fp8-cache variants have no fused-append twin
}
// ... or, for the append-read entry (no cache write), a balanced kernel's APP form (pa_queue.hpp): launched through fn_app
static Variant* app_read_variant_v1(int id) {
  if (Variant* v = app_variant_v1(id)) return v;
  if (id >= 1 && id <= nvariants_v1() && variant_v1(id).QUEUE && variant_v1(id).fn_app) return &variant_v1(id);
  return nullptr;
}
// is there a balanced kernel with a fused-append form for this head size?
static bool balanced_append_built(int head_size, bool bf) {
  for (int i = 0; i < g_queue_nvariants; ++i)
    if (g_queue_variants[i].fn_app && g_queue_variants[i].D == head_size && g_queue_variants[i].BF == bf && !g_queue_variants[i].F8)
      return true;
  return false;
}

// ---- block-sparse attention (blocksparse_vert_stride > 1): kernels of their own (pa_variants_sparse*.hip) ----
// bsp = {tp_rank, local_blocks, vert_stride, blocksparse_block_size, head_sliding_step}
static Variant* find_sparse(int D, int BS, int WPH, bool bf, bool part) {
  Variant* tab = bf ? g_sparse_bf16_variants : g_sparse_variants;
  const int n = bf ? g_sparse_bf16_nvariants : g_sparse_nvariants;
  for (int i = 0; i < n; ++i) {
    const bool is_part = strstr(tab[i].name, "_v2_") != nullptr;
    if (tab[i].D == D && tab[i].BS == BS && tab[i].WPH == WPH && is_part == part) return &tab[i];
  }
  return nullptr;
}
static int check_sparse(const char* op, const int32_t* bsp) {
  if (bsp[2] <= 1) return fail(VMI_E_SHAPE, "%s: blocksparse_vert_stride=%d does not enable block-sparse attention", op, bsp[2]);
  if (bsp[3] <= 0) return fail(VMI_E_SHAPE, "%s: blocksparse_block_size=%d must be positive", op, bsp[3]);
  return VMI_OK;
}
static void fill_sparse(PAParams& p, const int32_t* bsp) {
  p.bs_tp_rank = bsp ? bsp[0] : 0;
  p.bs_local_blocks = bsp ? bsp[1] : 0;
  p.bs_vert_stride = bsp ? bsp[2] : 1;
  p.bs_block_size = bsp ? bsp[3] : 1;
  p.bs_head_sliding_step = bsp ? bsp[4] : 0;
}

// The out-of-scope corners of the reference's dispatch live in libvmi_paged_attention_extras.so;
// in the product library their kernel menus are empty (pa_extras_absent.hip) and the launchers say so by name.
static int extras_gate(const char* op, bool bf, int f8, const int32_t* bsp) {
  if (g_has_extras) return VMI_OK;
  char what[96];
  snprintf(what, sizeof(what), "%s %s", op,
           bsp ? "with block-sparse attention" : bf ? "over bfloat16 tensors" : "over fp8-E5M2 pages");
  return (bsp || bf || f8 == 2) ? not_built(what) : VMI_OK;
}

int launch_pa_v1(void* out, const void* query, const void* key_cache,
                        const void* value_cache, int32_t num_seqs, int32_t num_heads,
                        int32_t head_size, int32_t num_kv_heads, float scale,
                        const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
                        int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                        const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                        int64_t kv_head_stride, int32_t device, void* stream, int32_t variant,
                        bool bf, bool append, const void* key, const void* value, int64_t key_stride,
                        int64_t value_stride, int f8, float kv_scale, const int32_t* bsp, void* workspace,
                        int64_t workspace_bytes, bool append_no_write) {
  // (an EMPTY batch — num_seqs == 0: the per-sequence tensors have no storage, torch hands out null data pointers — is
  //  a no-op below, not an error; the caches must exist either way)
  if (!key_cache || !value_cache || (num_seqs != 0 && (!out || !query || !block_tables || !seq_lens)))
    return fail(VMI_E_NULL_POINTER, "paged_attention_v1: NULL tensor pointer");
  if (int rc = extras_gate(append ? "paged_attention_v1_append" : "paged_attention_v1", bf, f8, bsp)) return rc;
  if (bsp) {
    if (int rc = check_sparse("paged_attention_v1", bsp)) return rc;
    if (append || f8) return fail(VMI_E_VARIANT, "paged_attention_v1: block-sparse attention is built for fp16 / bf16 caches, without the fused append");
  }
  if (append) {
    if (num_seqs != 0 && (!key || !value)) return fail(VMI_E_NULL_POINTER, "paged_attention_v1_append: NULL key/value pointer");
    // the fused kernel moves a key row as 16-B chunks (the stand-alone reshape_and_cache has a scalar path)
    if (!aligned16(key) || (key_stride & 7))
      return fail(VMI_E_ALIGNMENT, "paged_attention_v1_append: key rows must be 16-byte aligned "
                  "(key_stride=%lld)", (long long)key_stride);
  }
  if (!head_size_supported(head_size))
    return fail(VMI_E_HEAD_SIZE, "Unsupported head size: %d", head_size);
  if (!block_size_supported(block_size))
    return fail(VMI_E_BLOCK_SIZE, "Unsupported block size: %d", block_size);
  if (num_seqs < 0 || num_heads <= 0 || max_seq_len < 0 || max_num_blocks_per_seq < 0)
    return fail(VMI_E_SHAPE, "paged_attention_v1: negative size (num_seqs=%d num_heads=%d "
                "max_seq_len=%d max_num_blocks_per_seq=%d)", num_seqs, num_heads, max_seq_len,
                max_num_blocks_per_seq);
  if (num_kv_heads <= 0 || num_heads % num_kv_heads != 0)
    return fail(VMI_E_KV_HEADS, "paged_attention_v1: num_heads=%d not divisible by num_kv_heads=%d",
                num_heads, num_kv_heads);
  const int kv_am = f8 ? 15 : 7;  // cache strides are in elements: 16 bytes = 8 halves or 16 fp8 bytes
  if (!aligned16(query) || !aligned16(key_cache) || !aligned16(value_cache) || (q_stride & 7) ||
      (kv_block_stride & kv_am) || (kv_head_stride & kv_am))
    return fail(VMI_E_ALIGNMENT, "paged_attention_v1: query/key_cache/value_cache and their "
                "strides must be 16-byte aligned (q_stride=%lld kv_block_stride=%lld "
                "kv_head_stride=%lld)", (long long)q_stride, (long long)kv_block_stride,
                (long long)kv_head_stride);
  if (num_seqs == 0) return VMI_OK;

  if (f8 && block_size == 8)
    return fail(VMI_E_BLOCK_SIZE, "Unsupported block size: 8 with an fp8 KV cache (a value row of 8 bytes does not "
                "fill a 16-byte unit; block sizes 16 and 32 are built)");
  const int lpad = ((max_seq_len + 31) / 32) * 32;  // whole blocks for every block size, 16-B aligned rows
  auto lds_of = [&](const Variant& c) { return variant_lds_bytes(c, lpad); };
  const bool gate_ok = variant == 0 && !append && !bsp && !f8;  // an explicit variant is run as asked
  g_cus = device_cus(device);  // the heuristics size the launch for THIS device
  Variant* sparse_v = nullptr;
  bool picked = false;  // the library chose the variant (the caller passed 0)
  if (bsp) {  // one or four waves per head, by how many (seq, head) units there are to fill the chip with
    const int nblk = (max_seq_len + block_size - 1) / block_size;
    const bool many = (long)num_seqs * num_heads >= full_chip_waves() || nblk < 4;
    sparse_v = find_sparse(head_size, block_size, many ? 1 : 4, bf, false);
    if (sparse_v && lds_of(*sparse_v) > 160 * 1024) sparse_v = find_sparse(head_size, block_size, 1, bf, false);
    if (!sparse_v) return fail(VMI_E_VARIANT, "paged_attention_v1: no block-sparse kernel for head size %d / block size %d", head_size, block_size);
  } else if (variant == 0) {
    picked = true;
    variant = pick_variant_gqa(num_seqs, num_heads, num_heads / num_kv_heads, head_size, block_size, max_seq_len, bf, f8);
    if (!variant || (append && !(append_no_write ? app_read_variant_v1(variant) : app_variant_v1(variant))))
      variant = f8 ? pick_variant_fp8(num_seqs, num_heads, head_size, block_size, max_seq_len, 0, bf, f8,
                                      kv_scale == 1.0f && !append)
                   : pick_variant(num_seqs, num_heads, head_size, block_size, max_seq_len, bf, 0,
                                  !append || (append_no_write && balanced_append_built(head_size, bf)));
    variant = fit_lds(variant, head_size, block_size, lpad, bf, f8);
  }
  const bool have_ws = workspace != nullptr && aligned16(workspace) &&
                       workspace_bytes >= (int64_t)pa_split_layout(head_size).bytes;
  if (!sparse_v && picked && have_ws && !append && f8 != 2 && !bf) {
    // a caller-owned workspace lets an under-filled launch spread each (sequence, head) over several workgroups
    if (const int sv = pick_split(num_seqs, num_heads, head_size, block_size, max_seq_len, variant, num_heads / num_kv_heads, f8)) variant = sv;
  }
  if (!sparse_v && (variant < 1 || variant > nvariants_v1()))
    return fail(VMI_E_VARIANT, "paged_attention_v1: unknown variant %d", variant);
  Variant* vp = sparse_v ? sparse_v : (append ? (append_no_write ? app_read_variant_v1(variant) : app_variant_v1(variant)) : &variant_v1(variant));
  if (!vp) return fail(VMI_E_VARIANT, "paged_attention_v1_append: variant %d (%s) has no fused-append twin", variant,
                       variant_v1(variant).name);
  Variant& v = *vp;
  if (append && is_diag(v))
    return fail(VMI_E_VARIANT, "paged_attention_v1_append: %s is a bandwidth diagnostic", v.name);
  if (v.F8 != f8)
    return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s is for an %s KV cache", v.name,
                v.F8 == 2 ? "fp8 E5M2" : (v.F8 ? "fp8 E4M3" : "fp16/bf16"));
  if (v.BF != bf)
    return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s is for %s elements", v.name,
                v.BF ? "bfloat16" : "float16");
  if (v.D != head_size || v.BS != block_size)
    return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s is for head size %d / block size %d, "
                "got %d / %d", v.name, v.D, v.BS, head_size, block_size);
  if (v.WPH > 1 && num_heads % (v.HPW * v.HPT) != 0)
    return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s needs num_heads %% %d == 0", v.name,
                v.HPW * v.HPT);
  if (v.GQS && (num_heads / num_kv_heads) % v.HPT != 0)
    return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s shares a KV head between %d query heads, got "
                "num_heads / num_kv_heads = %d", v.name, v.HPT, num_heads / num_kv_heads);
  if (v.STAGE && (append || bsp || (f8 && kv_scale != 1.0f)))
    return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s (LDS-staging experiment) takes fp16 pages or fp8 E4M3 "
                "pages with kv_scale 1, without the fused append or block-sparse attention", v.name);
  if (v.QUEUE && ((append && !(v.fn_app && append_no_write)) || bsp || (f8 && kv_scale != 1.0f)))
    return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s (balanced kernel) takes fp16 / bf16 pages, or fp8 pages "
                "with kv_scale 1, without block-sparse attention%s", v.name,
                append ? "; of the fused append it has the append-read form only (vmi_paged_attention_v1_newest_f16)" : "");
  if (v.QUEUE && (int64_t)num_seqs * num_heads > 0x7fffffff)
    return fail(VMI_E_SHAPE, "paged_attention_v1: num_seqs * num_heads = %lld items exceed 2^31",
                (long long)num_seqs * num_heads);

  bool split_rounds = false;
  if (v.XW) {
    if (append || bsp || f8 == 2 || bf)
      return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s (split kernel) takes fp16 tensors over fp16 or fp8-E4M3 pages, "
                  "without the fused append or block-sparse attention", v.name);
    if (!have_ws)
      return fail(VMI_E_WORKSPACE, "paged_attention_v1: variant %s spreads a (sequence, head) over several workgroups and needs a "
                  "16-byte aligned workspace of vmi_paged_attention_v1_workspace_bytes() = %zu bytes (got %p, %lld)", v.name,
                  pa_split_layout(head_size).bytes, workspace, (long long)workspace_bytes);
    if (num_heads % v.HPT)
      return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s takes %d query heads of one KV head per item, got num_heads = %d",
                  v.name, v.HPT, num_heads);
    const int64_t wgs = (int64_t)num_seqs * (num_heads / v.HPT) * (v.XW / v.WPH);
    if (wgs > 0x7fffffff)
      return fail(VMI_E_SHAPE, "paged_attention_v1: variant %s would launch %lld workgroups", v.name, (long long)wgs);
    // more workgroups than are resident — by the launch bounds and by LDS — or than the workspace holds words for: the kernel's
    // twin that goes in rounds (waiting for a workgroup that is not on the chip yet works only as long as workgroups are
    // dispatched in index order; the twin does not lean on that)
    const int64_t by_lds = (int64_t)(((size_t)160 * 1024) / (lds_of(v) + 1024)) * g_cus;
    split_rounds = v.XW > v.WPH && (wgs > split_resident_wgs(head_size, v.HPT) || wgs > by_lds || wgs * v.HPT > SPLIT_MAX_WGS);
    if (v.XW == v.WPH && wgs * v.HPT > SPLIT_MAX_WGS)   // (one workgroup per item: nothing waits across workgroups)
      return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s would launch %lld workgroups (the workspace holds the rows of %d)",
                  v.name, (long long)wgs, SPLIT_MAX_WGS / v.HPT);
    if (split_rounds && !v.fn_rounds)
      return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s would launch %lld workgroups; the split kernels need every "
                  "workgroup resident (at most %d on this device)", v.name, (long long)wgs, split_resident_wgs(head_size, v.HPT));
  }
  const size_t lds = lds_of(v);
  if (lds > 160 * 1024)
    return fail(VMI_E_MAX_SEQ_LEN, "paged_attention_v1: max_seq_len=%d needs %zu B of LDS per "
                "workgroup (variant %s), limit 163840 — paged_attention_v2 has no such limit", max_seq_len, lds,
                v.name);

  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  if (lds > 48 * 1024) {
    // (set on every such launch: the attribute is per device and the library may be driven from several host threads —
    //  remembering "already granted" in the variant table was a data race; the call costs about a microsecond)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(split_rounds ? v.fn_rounds : v.fn),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
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
  p.key = append ? static_cast<const h16*>(key) : nullptr;
  p.value = append ? static_cast<const h16*>(value) : nullptr;
  p.key_stride = key_stride;
  p.value_stride = value_stride;
  p.kv_scale = kv_scale;
  fill_sparse(p, bsp);
  p.num_seqs = num_seqs;
  p.q_flags = 0;
  p.app_flags = (append && append_no_write) ? 1 : 0;
  g_last_variant = sparse_v ? 0 : variant;
  g_last_partner = 0;

  // the balanced kernel's launch: persistent geometry — as many 4-wave workgroups as stay resident (3 per CU while their
  // LDS fits, 2 for head size 128), never more than one wave per item; the kernel picks its mode from seq_lens
  auto launch_balanced = [&](Variant& q, int gate) -> int {
    const size_t qlds = variant_lds_bytes(q, lpad);
    if (qlds > 48 * 1024) {
      hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(append ? q.fn_app : q.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)qlds);
      if (ea != hipSuccess) return hip_fail(ea, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    const int cus = device_cus(device);
    int per_cu = qlds * 3 <= 160 * 1024 ? 3 : (qlds * 2 <= 160 * 1024 ? 2 : 1);
    if (q.D > 64 && per_cu > 2) per_cu = 2;  // register budget of the head-size-128 kernels (pa_queue.hpp launch bounds)
    const int64_t items = (int64_t)num_seqs * num_heads;
    int64_t g = (int64_t)cus * per_cu;
    if (g * 4 > items) g = (items + 3) / 4;
    pa_kernel_t fn = append ? q.fn_app : q.fn;
    PAParams pq = p;
    pq.q_flags = g_queue_flags | gate;
    hipLaunchKernelGGL(fn, dim3((unsigned)g), dim3(256), qlds, static_cast<hipStream_t>(stream), pq);
    hipError_t el = hipGetLastError();
    if (el != hipSuccess) return hip_fail(el, "paged_attention_v1 launch");
    return VMI_OK;
  };
  if (v.QUEUE) return launch_balanced(v, 0);
  if (v.XW) {  // split kernel (pa_split.hpp): items x (XW / 4) workgroups, all resident, meeting in the caller's workspace
    const SplitLayout lay = pa_split_layout(head_size);
    char* wsb = static_cast<char*>(workspace);
    PASplit sp;
    sp.status = reinterpret_cast<unsigned int*>(wsb + lay.status_off);
    sp.counters = reinterpret_cast<unsigned int*>(wsb + lay.counters_off);
    sp.slots = reinterpret_cast<unsigned long long*>(wsb + lay.slots_off);
    sp.partials = reinterpret_cast<float*>(wsb + lay.partials_off);
    sp.nw = v.XW;
    sp.wtok = split_wtok(lpad, v.XW);
    sp.flags = g_split_flags;
    unsigned grid = (unsigned)((int64_t)num_seqs * (num_heads / v.HPT) * (v.XW / v.WPH));
    if (split_rounds) {
      // As many whole items as are truly resident — by the twin's launch bounds and by LDS — and as half the workspace holds
      // words for (its two halves alternate by round): a workgroup serves grid-strided items, an item's workgroups always
      // together (pa_split.hpp).
      const int G = v.XW / v.WPH;
      int per_cu = split_wgs_per_cu(head_size, v.HPT, true);
      const int by_lds = (int)(((size_t)160 * 1024) / (lds + 1024));
      per_cu = per_cu < by_lds ? per_cu : by_lds;
      int64_t res = (int64_t)per_cu * device_cus(device);
      const int64_t by_ws = SPLIT_MAX_WGS / (2 * v.HPT);
      res = res < by_ws ? res : by_ws;
      if (res < G)
        return fail(VMI_E_VARIANT, "paged_attention_v1: variant %s: not even one item's %d workgroups are resident at %zu B of "
                    "LDS each", v.name, G, lds);
      grid = (unsigned)(res / G * G);
    }
    hipLaunchKernelGGL(reinterpret_cast<pa_split_kernel_t>(split_rounds ? v.fn_rounds : v.fn), dim3(grid), dim3(v.WPH * 64), lds,
                       static_cast<hipStream_t>(stream), p, sp);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "paged_attention_v1 (split) launch");
    return VMI_OK;
  }

  // Head size 128 on a full chip: the fastest kernel for equal lengths reads 4 adjacent heads per wave in lockstep
  // (d128_mh4_*_lock, 0.87 of the roofline) and is the slowest on ragged ones (cfg4 U{1..2048}: 469 us against 325 us
  // for the balanced kernel, which in turn is 4 % behind on equal lengths).  Only the device knows which batch this
  // is, and the two kernels have different launch geometries — so BOTH are launched, each told to leave at once
  // unless the batch is its kind (same statistics, same arithmetic in both: pa_kernel.hpp batch_stats).  The kernel
  // that leaves costs a launch boundary, about 2 us of a 300-600 us call.
  Variant* partner = nullptr;
  if (gate_ok && v.D == 128 && v.BS == 16 && v.WPH == 1 && !v.GQS && !v.F8 && !v.SPARSE &&
      // (every wave of BOTH kernels reads all the lengths for the verdict — 4*B bytes per wave out of L2: bounded by 2048, the
      //  size the balanced kernel ranks in LDS; pick_variant never hands a larger batch the lockstep kernel — lock_ok — so
      //  beyond that bound whatever kernel was picked runs alone)
      num_seqs <= R.gate_max_seqs && (int64_t)num_seqs * num_heads >= (int64_t)device_cus(device) * 8) {
    for (int i = 0; i < g_queue_nvariants; ++i)
      if (g_queue_variants[i].D == v.D && g_queue_variants[i].BF == v.BF && g_queue_variants[i].BS == v.BS &&
          g_queue_variants[i].F8 == v.F8 && !g_queue_variants[i].KM &&
          2 * variant_lds_bytes(g_queue_variants[i], lpad) <= (size_t)160 * 1024)
        partner = &g_queue_variants[i];
  }
  if (partner) {
    p.q_flags |= QF_GATE_UNIFORM;
    g_last_partner = nvariants_v1() - g_stage_nvariants - g_split_nvariants - g_queue_nvariants + (int)(partner - g_queue_variants) + 1;
  }

  dim3 block(v.HPW * v.WPH * 64);
  // gridDim.y is limited to 65535: longer batches go out as consecutive launches over slices
  for (int32_t s0 = 0; s0 < num_seqs; s0 += 65535) {
    const int32_t ns = (num_seqs - s0) < 65535 ? (num_seqs - s0) : 65535;
    PAParams ps = p;
    ps.out = p.out + (int64_t)s0 * num_heads * head_size;
    ps.q = p.q + (int64_t)s0 * q_stride;
    ps.block_tables = p.block_tables + (int64_t)s0 * max_num_blocks_per_seq;
    ps.seq_lens = p.seq_lens + s0;
    if (append) {
      ps.key = p.key + (int64_t)s0 * key_stride;
      ps.value = p.value + (int64_t)s0 * value_stride;
    }
    const int hpg = v.HPW * v.HPT;  // heads per workgroup
    dim3 grid((num_heads + hpg - 1) / hpg, ns, 1);
    hipLaunchKernelGGL(v.fn, grid, block, lds, static_cast<hipStream_t>(stream), ps);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "paged_attention_v1 launch");
  }
  if (partner) return launch_balanced(*partner, QF_GATE_RAGGED);
  return VMI_OK;
}

// ---- split-KV (paged_attention_v2) variants: same kernel body, PART = true -----------------
#define VMI_VARIANT_V2(D, HPW, WPH, U, NT)                                          \
  {                                                                                 \
    "v2_d" #D "_h" #HPW "_w" #WPH "_u" #U "_nt" #NT, D, 16, HPW, WPH, U, (bool)NT, 1, false, \
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
    // grouped-query attention: gq<N> query heads of one KV head per wave, each tile loaded once, q.K^T on MFMA
#define VMI_V2_GQ(D, WPH, HPT, U)                                                                              \
  {"v2_d" #D "_gq" #HPT "_h1_w" #WPH "_u" #U "_nt1", D, 16, 1, WPH, U, true, HPT, false,                        \
   (pa_kernel_t)pa_v1_kernel<D, 1, WPH, U, true, false, true, 16, false, false, HPT, false, 0, false, true>, 0, \
   0, 0, false, true}
    VMI_V2_GQ(128, 1, 4, 1), VMI_V2_GQ(128, 4, 4, 1), VMI_V2_GQ(128, 1, 8, 1), VMI_V2_GQ(128, 4, 8, 1),
    VMI_V2_GQ(128, 1, 2, 1), VMI_V2_GQ(128, 4, 2, 1), VMI_V2_GQ(128, 4, 7, 1), VMI_V2_GQ(128, 4, 3, 1),
    VMI_V2_GQ(64, 1, 4, 2), VMI_V2_GQ(64, 4, 4, 2), VMI_V2_GQ(64, 4, 8, 2), VMI_V2_GQ(64, 4, 2, 2),
#undef VMI_V2_GQ
};
static const int g_ncore_v2 = (int)(sizeof(g_variants_v2) / sizeof(g_variants_v2[0]));

static int nvariants_v2() {
  return g_ncore_v2 + g_extra_nvariants_v2 + g_bf16_nvariants_v2 + g_fp8_nvariants_v2 + g_fp8_nvariants_v2_e5m2 +
         g_fp8bf_nvariants_v2 + g_fp8bf_nvariants_v2_e5m2;
}
static Variant& variant_v2(int id) {  // [core fp16][extra fp16][bf16][fp8 cache]
  if (id <= g_ncore_v2) return g_variants_v2[id - 1];
  if (id <= g_ncore_v2 + g_extra_nvariants_v2) return g_extra_variants_v2[id - 1 - g_ncore_v2];
  if (id <= g_ncore_v2 + g_extra_nvariants_v2 + g_bf16_nvariants_v2)
    return g_bf16_variants_v2[id - 1 - g_ncore_v2 - g_extra_nvariants_v2];
  const int f0 = g_ncore_v2 + g_extra_nvariants_v2 + g_bf16_nvariants_v2;
  if (id <= f0 + g_fp8_nvariants_v2) return g_fp8_variants_v2[id - 1 - f0];
  const int f1 = f0 + g_fp8_nvariants_v2;
  if (id <= f1 + g_fp8_nvariants_v2_e5m2) return g_fp8_variants_v2_e5m2[id - 1 - f1];
  const int f2 = f1 + g_fp8_nvariants_v2_e5m2;                                     // bfloat16 query over fp8 pages
  if (id <= f2 + g_fp8bf_nvariants_v2) return g_fp8bf_variants_v2[id - 1 - f2];
  return g_fp8bf_variants_v2_e5m2[id - 1 - f2 - g_fp8bf_nvariants_v2];
}

static int find_variant_v2(int D, int BS, int HPW, int WPH, bool bf = false, int f8 = false) {
  for (int id = 1; id <= nvariants_v2(); ++id) {
    const Variant& v = variant_v2(id);
    if (v.F8 == f8 && v.BF == bf && v.D == D && v.BS == BS && v.HPW == HPW && v.WPH == WPH && !v.GQS) return id;
  }
  return 0;
}

// a partition holds 512 / block_size blocks; give each (seq, head, partition) 1..8 waves so that the
// launch has >= ~2048 waves when the batch allows it
static int pick_variant_v2(int num_seqs, int num_heads, int head_size, int block_size, int max_seq_len,
                           bool bf = false, int f8 = false, int qpk = 1) {
  const int parts = (max_seq_len + 511) / 512;
  if (qpk > 1 && !bf && !f8 && block_size == 16) {  // grouped-query: largest built group size dividing qpk
    for (int g = 8; g >= 2; --g) {
      if (qpk % g) continue;
      const long gunits = (long)num_seqs * (num_heads / g) * (parts > 0 ? parts : 1);
      int best = 0;
      for (int id = 1; id <= nvariants_v2(); ++id) {
        const Variant& c = variant_v2(id);
        if (!c.GQS || c.D != head_size || c.HPT != g) continue;
        if (!best || (gunits >= 4096 ? c.WPH < variant_v2(best).WPH : c.WPH > variant_v2(best).WPH)) best = id;
      }
      if (best) return best;
    }
  }
  const long units = (long)num_seqs * num_heads * (parts > 0 ? parts : 1);
  int wph = 1;
  while (wph < 8 && units * wph < 2048) wph *= 2;
  int v = 0;
  if (!bf && block_size == 16 && (head_size == 64 || head_size == 128)) {
    v = (wph == 1) ? find_variant_v2(head_size, 16, (num_heads % 4 == 0) ? 4 : 1, 1, false, f8)
                   : find_variant_v2(head_size, 16, 1, wph, false, f8);
  }
  if (!v) v = find_variant_v2(head_size, block_size, 1, wph == 1 ? 1 : 4, bf, f8);
  return v ? v : find_variant_v2(head_size, block_size, 1, 1, bf, f8);
}

static pa_reduce_t reduce_kernel_for(int head_size, bool bf) {
  if (bf) return bf16_reduce_kernel(head_size);
  if (head_size == 64) return (pa_reduce_t)pa_v2_reduce_kernel<64>;
  if (head_size == 128) return (pa_reduce_t)pa_v2_reduce_kernel<128>;
  return extra_reduce_kernel(head_size, false);
}

int launch_pa_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
                        const void* key_cache, const void* value_cache, int32_t num_seqs,
                        int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
                        const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
                        int32_t max_seq_len, int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                        int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                        int32_t device, void* stream, int32_t variant, bool bf, int f8, float kv_scale,
                        const int32_t* bsp) {
  if (int rc = extras_gate("paged_attention_v2", bf, f8, bsp)) return rc;
  if (bsp) {
    if (int rc = check_sparse("paged_attention_v2", bsp)) return rc;
    if (f8) return fail(VMI_E_VARIANT, "paged_attention_v2: block-sparse attention is built for fp16 / bf16 caches");
  }
  if (!key_cache || !value_cache ||
      (num_seqs != 0 && (!out || !exp_sums || !max_logits || !tmp_out || !query || !block_tables || !seq_lens)))
    return fail(VMI_E_NULL_POINTER, "paged_attention_v2: NULL tensor pointer");
  if (!head_size_supported(head_size))
    return fail(VMI_E_HEAD_SIZE, "Unsupported head size: %d", head_size);
  if (!block_size_supported(block_size))
    return fail(VMI_E_BLOCK_SIZE, "Unsupported block size: %d", block_size);
  if (num_seqs < 0 || num_heads <= 0 || max_seq_len < 0 || max_num_blocks_per_seq < 0)
    return fail(VMI_E_SHAPE, "paged_attention_v2: negative size");
  if (num_seqs > 65535 || num_heads > 65535)
    return fail(VMI_E_SHAPE, "paged_attention_v2: num_seqs/num_heads above the 65535 grid limit");
  if (num_kv_heads <= 0 || num_heads % num_kv_heads != 0)
    return fail(VMI_E_KV_HEADS, "paged_attention_v2: num_heads=%d not divisible by num_kv_heads=%d",
                num_heads, num_kv_heads);
  const int kv_am = f8 ? 15 : 7;
  if (!aligned16(query) || !aligned16(key_cache) || !aligned16(value_cache) || (q_stride & 7) ||
      (kv_block_stride & kv_am) || (kv_head_stride & kv_am))
    return fail(VMI_E_ALIGNMENT, "paged_attention_v2: pointers/strides must be 16-byte aligned");
  if (f8 && block_size == 8)
    return fail(VMI_E_BLOCK_SIZE, "Unsupported block size: 8 with an fp8 KV cache (block sizes 16 and 32 are built)");
  const int parts = (max_seq_len + 511) / 512;  // attention_kernels.cu:885
  if (num_seqs == 0 || parts == 0) return VMI_OK;
  if (parts > 65535) return fail(VMI_E_MAX_SEQ_LEN, "paged_attention_v2: too many partitions");
  g_cus = device_cus(device);
  Variant* sparse_v = nullptr;
  if (bsp) {
    sparse_v = find_sparse(head_size, block_size, (long)num_seqs * num_heads * parts >= full_chip_waves() ? 1 : 4, bf, true);
    if (!sparse_v) return fail(VMI_E_VARIANT, "paged_attention_v2: no block-sparse kernel for head size %d / block size %d", head_size, block_size);
  } else if (variant == 0) {
    variant = pick_variant_v2(num_seqs, num_heads, head_size, block_size, max_seq_len, bf, f8, num_heads / num_kv_heads);
  }
  if (!sparse_v && (variant < 1 || variant > nvariants_v2()))
    return fail(VMI_E_VARIANT, "paged_attention_v2: unknown variant %d", variant);
  Variant& v = sparse_v ? *sparse_v : variant_v2(variant);
  if (v.F8 != f8)
    return fail(VMI_E_VARIANT, "paged_attention_v2: variant %s is for an %s KV cache", v.name, v.F8 ? "fp8" : "fp16/bf16");
  if (v.BF != bf)
    return fail(VMI_E_VARIANT, "paged_attention_v2: variant %s is for %s elements", v.name,
                v.BF ? "bfloat16" : "float16");
  if (v.D != head_size || v.BS != block_size)
    return fail(VMI_E_VARIANT, "paged_attention_v2: variant %s is for head size %d / block size %d", v.name,
                v.D, v.BS);
  if (v.WPH > 1 && num_heads % (v.HPW * v.HPT) != 0)
    return fail(VMI_E_VARIANT, "paged_attention_v2: variant %s needs num_heads %% %d == 0", v.name, v.HPW * v.HPT);
  if (v.GQS && (num_heads / num_kv_heads) % v.HPT != 0)
    return fail(VMI_E_VARIANT, "paged_attention_v2: variant %s shares a KV head between %d query heads, got "
                "num_heads / num_kv_heads = %d", v.name, v.HPT, num_heads / num_kv_heads);
  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");

  const int lpad = 512;  // one partition of logits (:886)
  const size_t lds = (size_t)v.HPW * v.HPT *
                     ((size_t)lpad * 4 + 2 * v.WPH * 4 + (size_t)v.WPH * v.D * 4 + (v.WPH > 1 ? (size_t)lpad * 2 : 0) +
                      (v.SPARSE ? (size_t)lpad / 2 : 0));
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
  p.key = nullptr;
  p.value = nullptr;
  p.key_stride = 0;
  p.value_stride = 0;
  p.num_seqs = num_seqs;
  p.q_flags = 0;
  p.app_flags = 0;
  p.kv_scale = kv_scale;
  fill_sparse(p, bsp);
  dim3 grid((num_heads + v.HPW * v.HPT - 1) / (v.HPW * v.HPT), num_seqs, parts);  // :890
  hipLaunchKernelGGL(v.fn, grid, dim3(v.HPW * v.WPH * 64), lds, static_cast<hipStream_t>(stream), p);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "paged_attention_v2 launch");

  const size_t rlds = (size_t)(2 * parts + 4) * sizeof(float);  // :894
  dim3 rgrid(num_heads, num_seqs);                               // :893
  pa_reduce_t red = reduce_kernel_for(head_size, bf);
  if (!red) return fail(VMI_E_HEAD_SIZE, "Unsupported head size: %d", head_size);
  hipLaunchKernelGGL(red, rgrid, dim3(128), rlds, static_cast<hipStream_t>(stream), static_cast<h16*>(out),
                     static_cast<const float*>(exp_sums), static_cast<const float*>(max_logits),
                     static_cast<const h16*>(tmp_out), seq_lens, parts);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "paged_attention_v2 reduce launch");
  return VMI_OK;
}

// the quantising scatter behind vmi_reshape_and_cache_fp8 (and, in the extras library, its bfloat16 / E5M2 forms)
int reshape_and_cache_fp8_impl(const void* key, const void* value, void* key_cache, void* value_cache,
                                      const int64_t* slot_mapping, int32_t num_tokens, int32_t num_heads,
                                      int32_t head_size, int32_t block_size, int32_t x, int64_t key_stride,
                                      int64_t value_stride, float kv_scale, int32_t device, void* stream, bool bf,
                                      bool e5) {
  if (!key_cache || !value_cache || (num_tokens != 0 && (!key || !value || !slot_mapping)))
    return fail(VMI_E_NULL_POINTER, "reshape_and_cache (fp8): NULL tensor pointer");
  if (x != 16) return fail(VMI_E_X, "reshape_and_cache (fp8): key_cache.size(4) must be 16, got %d", x);
  if (num_tokens < 0 || num_heads <= 0 || head_size <= 0 || (head_size & 15))
    return fail(VMI_E_SHAPE, "reshape_and_cache (fp8): bad sizes (num_tokens=%d num_heads=%d head_size=%d)",
                num_tokens, num_heads, head_size);
  if (block_size <= 0) return fail(VMI_E_BLOCK_SIZE, "Unsupported block size: %d", block_size);
  if (!(kv_scale > 0.f)) return fail(VMI_E_SHAPE, "reshape_and_cache (fp8): kv_scale must be positive, got %g", (double)kv_scale);
  if (!aligned16(key_cache)) return fail(VMI_E_ALIGNMENT, "reshape_and_cache (fp8): key_cache must be 16-byte aligned");
  if (num_tokens == 0) return VMI_OK;
  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  const bool vec = aligned16(key) && aligned16(value) && !(key_stride & 7) && !(value_stride & 7);
  const int n16 = (num_heads * head_size) >> 4;
  int threads = ((n16 + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const fp8_scatter_fn fn = (bf || e5) ? fp8_scatter_extra_kernel(vec, bf, e5)   // out-of-scope instantiations
                            : vec      ? (fp8_scatter_fn)reshape_and_cache_fp8_kernel<true, false, false>
                                       : (fp8_scatter_fn)reshape_and_cache_fp8_kernel<false, false, false>;
  if (!fn) return not_built(e5 ? "reshape_and_cache over fp8-E5M2 pages" : "reshape_and_cache (fp8) over bfloat16 rows");
  hipLaunchKernelGGL(fn, dim3(num_tokens), dim3(threads), 0, st, static_cast<const h16*>(key),
                     static_cast<const h16*>(value), static_cast<uint8_t*>(key_cache),
                     static_cast<uint8_t*>(value_cache), slot_mapping, key_stride, value_stride, num_heads, head_size,
                     block_size, kv_scale);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "reshape_and_cache (fp8) launch");
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

int vmi_paged_attention_v1_f16_ws(void* out, const void* query, const void* key_cache, const void* value_cache,
                                  int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
                                  float scale, const int32_t* block_tables, const int32_t* seq_lens,
                                  int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                                  const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                                  int64_t kv_head_stride, int32_t device, void* stream, void* workspace,
                                  int64_t workspace_bytes, int32_t variant) {
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant, false, false, nullptr, nullptr, 0, 0, 0, 1.0f,
                           nullptr, workspace, workspace_bytes);
}

int64_t vmi_paged_attention_v1_workspace_bytes(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                               int32_t max_seq_len) {
  (void)num_seqs; (void)num_heads; (void)max_seq_len;   // (sized for any launch the split kernels take: they are bounded
  if (head_size != 64 && head_size != 128) return 0;    //  by what is resident, not by the batch)
  return (int64_t)vmi::pa_split_layout(head_size).bytes;
}

int vmi_paged_attention_v1_workspace_reset(void* workspace, int64_t workspace_bytes, int32_t device, void* stream) {
  using namespace vmi;
  if (!workspace || workspace_bytes <= 0) return fail(VMI_E_NULL_POINTER, "workspace_reset: NULL workspace");
  DeviceGuard guard(device);
  if (guard.err != hipSuccess) return hip_fail(guard.err, "hipSetDevice");
  // only the words a launch polls have to be zero: status, counters, granules (the partial rows are written before read)
  const size_t n = pa_split_layout(128).partials_off;
  const hipError_t e = hipMemsetAsync(workspace, 0, (size_t)workspace_bytes < n ? (size_t)workspace_bytes : n,
                                      static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return hip_fail(e, "workspace_reset hipMemsetAsync");
  return VMI_OK;
}

int vmi_paged_attention_v1_pick_variant_ws(int32_t num_seqs, int32_t num_heads, int32_t num_kv_heads, int32_t head_size,
                                           int32_t block_size, int32_t max_seq_len) {
  if (!vmi::head_size_supported(head_size) || !vmi::block_size_supported(block_size)) return 0;
  if (num_kv_heads <= 0) num_kv_heads = num_heads;
  if (num_heads % num_kv_heads) return 0;
  int plain = vmi::pick_variant_gqa(num_seqs, num_heads, num_heads / num_kv_heads, head_size, block_size, max_seq_len, false, 0);
  if (!plain) plain = vmi::pick_variant(num_seqs, num_heads, head_size, block_size, max_seq_len);
  plain = vmi::fit_lds(plain, head_size, block_size, ((max_seq_len + 31) / 32) * 32, false, 0);
  const int sv = vmi::pick_split(num_seqs, num_heads, head_size, block_size, max_seq_len, plain, num_heads / num_kv_heads);
  return sv ? sv : plain;
}

int vmi_paged_attention_v1_append_f16(void* out, const void* query, void* key_cache, void* value_cache,
                                      int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                      int32_t num_kv_heads, float scale, const int32_t* block_tables,
                                      const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
                                      int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                                      int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                      int32_t device, void* stream, const void* key, const void* value,
                                      int64_t key_stride, int64_t value_stride, int32_t variant) {
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant, false, true, key, value, key_stride,
                           value_stride);
}

int vmi_paged_attention_v1_newest_f16(void* out, const void* query, const void* key_cache, const void* value_cache,
                                      int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                      int32_t num_kv_heads, float scale, const int32_t* block_tables,
                                      const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
                                      int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                                      int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                                      int32_t device, void* stream, const void* key, const void* value,
                                      int64_t key_stride, int64_t value_stride, int32_t variant) {
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant, false, true, key, value, key_stride,
                           value_stride, 0, 1.0f, nullptr, nullptr, 0, true);
}

int vmi_paged_attention_v1_fp8(void* out, const void* query, const void* key_cache, const void* value_cache,
                               int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
                               float scale, const int32_t* block_tables, const int32_t* seq_lens,
                               int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                               const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                               int64_t kv_head_stride, int32_t device, void* stream, float kv_scale,
                               int32_t variant) {
  if (!(kv_scale > 0.f) || kv_scale != kv_scale)
    return vmi::fail(VMI_E_SHAPE, "paged_attention_v1 (fp8 cache): kv_scale must be positive, got %g", (double)kv_scale);
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant, false, false, nullptr, nullptr, 0, 0,
                           true, kv_scale);
}

int vmi_paged_attention_v1_fp8_ws(void* out, const void* query, const void* key_cache, const void* value_cache,
                                  int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
                                  float scale, const int32_t* block_tables, const int32_t* seq_lens,
                                  int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                                  const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
                                  int64_t kv_head_stride, int32_t device, void* stream, float kv_scale, void* workspace,
                                  int64_t workspace_bytes, int32_t variant) {
  if (!(kv_scale > 0.f) || kv_scale != kv_scale)
    return vmi::fail(VMI_E_SHAPE, "paged_attention_v1 (fp8 cache): kv_scale must be positive, got %g", (double)kv_scale);
  return vmi::launch_pa_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                           num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                           max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                           kv_head_stride, device, stream, variant, false, false, nullptr, nullptr, 0, 0,
                           true, kv_scale, nullptr, workspace, workspace_bytes);
}

int vmi_paged_attention_v2_fp8(void* out, void* exp_sums, void* max_logits, void* tmp_out, const void* query,
                               const void* key_cache, const void* value_cache, int32_t num_seqs,
                               int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
                               const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
                               int32_t max_seq_len, int32_t max_num_blocks_per_seq, const float* alibi_slopes,
                               int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                               int32_t device, void* stream, float kv_scale, int32_t variant) {
  if (!(kv_scale > 0.f))
    return vmi::fail(VMI_E_SHAPE, "paged_attention_v2 (fp8 cache): kv_scale must be positive, got %g", (double)kv_scale);
  return vmi::launch_pa_v2(out, static_cast<float*>(exp_sums), static_cast<float*>(max_logits), tmp_out, query,
                           key_cache, value_cache, num_seqs, num_heads, head_size, num_kv_heads, scale,
                           block_tables, seq_lens, block_size, max_seq_len, max_num_blocks_per_seq, alibi_slopes,
                           q_stride, kv_block_stride, kv_head_stride, device, stream, variant, false, true,
                           kv_scale);
}

int vmi_paged_attention_v1_pick_variant_fp8(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                            int32_t block_size, int32_t max_seq_len, int32_t mean_seq_len) {
  if (!vmi::head_size_supported(head_size) || (block_size != 16 && block_size != 32)) return 0;
  return vmi::pick_variant_fp8(num_seqs, num_heads, head_size, block_size, max_seq_len, mean_seq_len);
}

int vmi_paged_attention_v1_variant_count(void) { return vmi::nvariants_v1(); }

const char* vmi_paged_attention_v1_variant_name(int32_t variant) {
  if (variant < 1 || variant > vmi::nvariants_v1()) return "";
  return vmi::variant_v1(variant).name;
}

int vmi_paged_attention_v1_pick_variant(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                        int32_t block_size, int32_t max_seq_len) {
  if (!vmi::head_size_supported(head_size) || !vmi::block_size_supported(block_size)) return 0;
  return vmi::pick_variant(num_seqs, num_heads, head_size, block_size, max_seq_len);
}

int vmi_set_pv_mfma(int on) {
  const int prev = vmi::g_pv_mfma;
  vmi::g_pv_mfma = on ? 1 : 0;
  return prev;
}

int vmi_paged_attention_v1_last_variant(void) { return vmi::g_last_variant; }
int vmi_paged_attention_v1_last_partner(void) { return vmi::g_last_partner; }

#ifdef VMI_DIAG
int vmi_debug_set_queue_flags(int flags) {
  const int prev = vmi::g_queue_flags;
  vmi::g_queue_flags = vmi::clamp_queue_flags(flags);
  return prev;
}
int vmi_debug_set_split_flags(int flags) {
  const int prev = vmi::g_split_flags;
  vmi::g_split_flags = flags;
  return prev;
}
#endif

int vmi_has_extras(void) { return vmi::g_has_extras ? 1 : 0; }

int vmi_is_diag_build(void) {
#ifdef VMI_DIAG
  return 1;
#else
  return 0;
#endif
}

int vmi_paged_attention_v1_variant_fits(int32_t variant, int32_t max_seq_len, int32_t for_append) {
  if (variant < 1 || variant > vmi::nvariants_v1() || max_seq_len < 0) return 0;
  if (for_append && !(for_append == 2 ? vmi::app_read_variant_v1(variant) : vmi::app_variant_v1(variant))) return 0;
  const int lpad = ((max_seq_len + 31) / 32) * 32;
  return vmi::variant_lds_bytes(vmi::variant_v1(variant), lpad) <= (size_t)160 * 1024 ? 1 : 0;
}

int vmi_paged_attention_v1_pick_variant_gqa(int32_t num_seqs, int32_t num_heads, int32_t num_kv_heads,
                                            int32_t head_size, int32_t block_size, int32_t max_seq_len,
                                            int32_t is_bf16, int32_t is_fp8) {
  if (!vmi::head_size_supported(head_size) || !vmi::block_size_supported(block_size) || num_kv_heads <= 0 ||
      num_heads % num_kv_heads)
    return 0;
  const int v = vmi::pick_variant_gqa(num_seqs, num_heads, num_heads / num_kv_heads, head_size, block_size,
                                      max_seq_len, is_bf16 != 0, is_fp8 == 2 ? 2 : (is_fp8 != 0));
  if (v) return v;
  return is_fp8 ? vmi::pick_variant_fp8(num_seqs, num_heads, head_size, block_size, max_seq_len, 0, is_bf16 != 0,
                                        is_fp8 == 2 ? 2 : 1)
                : vmi::pick_variant(num_seqs, num_heads, head_size, block_size, max_seq_len, is_bf16 != 0);
}

int vmi_paged_attention_v1_pick_variant_hint(int32_t num_seqs, int32_t num_heads, int32_t head_size,
                                             int32_t block_size, int32_t max_seq_len,
                                             int32_t mean_seq_len, int32_t is_bf16) {
  if (!vmi::head_size_supported(head_size) || !vmi::block_size_supported(block_size)) return 0;
  return vmi::pick_variant(num_seqs, num_heads, head_size, block_size, max_seq_len, is_bf16 != 0,
                           mean_seq_len);
}

int vmi_reshape_and_cache_f16(const void* key, const void* value, void* key_cache,
                              void* value_cache, const int64_t* slot_mapping, int32_t num_tokens,
                              int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
                              int64_t key_stride, int64_t value_stride, int32_t device,
                              void* stream) {
  using namespace vmi;
  if (!key_cache || !value_cache || (num_tokens != 0 && (!key || !value || !slot_mapping)))  // no tokens: a no-op below
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
  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  const bool vec = aligned16(key) && aligned16(value) && !(key_stride & 7) && !(value_stride & 7);
  const int n8 = (num_heads * head_size) >> 3;
  int threads = ((n8 + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  dim3 grid(num_tokens), block(threads);
  // prefill-sized calls with 16-B aligned rows: whole-block form (falls back per run inside the kernel)
  if (vec && num_tokens >= 2 * block_size && (block_size == 8 || block_size == 16 || block_size == 32)) {
    typedef void (*blocks_fn)(const h16*, const h16*, h16*, h16*, const int64_t*, int64_t, int64_t, int, int, int, int, int);
    // dims per wave: 32-dim slices while whole-head waves would not fill the chip (decode batches: cfg3 6.8 us instead
    // of 8.5), whole heads for prompt-sized calls (16384 tokens x 32 x 128: 118 us vs 133 with slices)
    const long whole_head_waves = (long)((num_tokens + block_size - 1) / block_size) * num_heads;
    const int dw = (head_size % 32 == 0 && whole_head_waves < 4096) ? 32 : head_size;
    // decode-sized calls (round 5): the launch is a latency chain, not a stream — slots, rows, scattered 2-byte stores — so
    // (a) ONE wave per workgroup while that still leaves fewer workgroups than 4 per CU: cfg2 48 workgroups instead of 12,
    // cfg3 384 instead of 96, i.e. that many CUs' store paths share the scatter; (b) the rows are requested next to the slots
    const long waves = (long)((num_tokens + block_size - 1) / block_size) * num_heads * (head_size / dw);
    const bool small = dw == 32 && waves <= 4L * device_cus(device);
    const int wpb = small ? 1 : 4;
    const blocks_fn fn = block_size == 8    ? (blocks_fn)reshape_and_cache_blocks_kernel<8>
                         : block_size == 16 ? (blocks_fn)reshape_and_cache_blocks_kernel<16>
                                            : (blocks_fn)reshape_and_cache_blocks_kernel<32>;
    const size_t lds = (size_t)wpb * block_size * (dw + 8) * 2;
    if (lds <= 160 * 1024) {
      if (lds > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(reshape_and_cache_blocks)");
      }
      hipLaunchKernelGGL(fn, dim3((num_tokens + block_size - 1) / block_size, (num_heads * (head_size / dw) + wpb - 1) / wpb), dim3(64 * wpb), lds,
                         static_cast<hipStream_t>(stream), static_cast<const h16*>(key),
                         static_cast<const h16*>(value), static_cast<h16*>(key_cache),
                         static_cast<h16*>(value_cache), slot_mapping, key_stride, value_stride, num_tokens,
                         num_heads, head_size, dw, small ? 1 : 0);
      e = hipGetLastError();
      if (e != hipSuccess) return hip_fail(e, "reshape_and_cache (blocks) launch");
      return VMI_OK;
    }
  }
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

int vmi_reshape_and_cache_fp8(const void* key, const void* value, void* key_cache, void* value_cache,
                              const int64_t* slot_mapping, int32_t num_tokens, int32_t num_heads,
                              int32_t head_size, int32_t block_size, int32_t x, int64_t key_stride,
                              int64_t value_stride, float kv_scale, int32_t device, void* stream) {
  return vmi::reshape_and_cache_fp8_impl(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads, head_size,
                                    block_size, x, key_stride, value_stride, kv_scale, device, stream, false);
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

int vmi_paged_attention_v2_variant_count(void) { return vmi::nvariants_v2(); }

const char* vmi_paged_attention_v2_variant_name(int32_t variant) {
  if (variant < 1 || variant > vmi::nvariants_v2()) return "";
  return vmi::variant_v2(variant).name;
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
  DeviceGuard guard(device);
  hipError_t e = guard.err;
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
  DeviceGuard guard(device);
  hipError_t e = guard.err;
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

int vmi_swap_blocks_batched(const void* src_key, const void* src_value, void* dst_key, void* dst_value,
                            const int64_t* block_mapping, int32_t num_pairs, int64_t block_bytes, int32_t device,
                            void* stream) {
  using namespace vmi;
  if (num_pairs < 0 || block_bytes <= 0 || (block_bytes & 15))
    return fail(VMI_E_SHAPE, "swap_blocks_batched: bad sizes (pairs=%d block_bytes=%lld)", num_pairs, (long long)block_bytes);
  if (num_pairs == 0) return VMI_OK;
  if (!src_key || !src_value || !dst_key || !dst_value || !block_mapping)
    return fail(VMI_E_NULL_POINTER, "swap_blocks_batched: NULL pointer");
  if (!aligned16(src_key) || !aligned16(src_value) || !aligned16(dst_key) || !aligned16(dst_value))
    return fail(VMI_E_ALIGNMENT, "swap_blocks_batched: cache pointers must be 16-byte aligned");
  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipLaunchKernelGGL(swap_blocks_kernel, dim3(num_pairs, 2), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint8_t*>(src_key), static_cast<const uint8_t*>(src_value),
                     static_cast<uint8_t*>(dst_key), static_cast<uint8_t*>(dst_value), block_mapping, block_bytes);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "swap_blocks_batched launch");
  return VMI_OK;
}

#ifdef VMI_DIAG
int vmi_diag_stream_read(const void* src, int64_t bytes, void* sink, int32_t blocks, int32_t nt,
                         int32_t device, void* stream) {
  using namespace vmi;
  if (!src || !sink || bytes < 16 || blocks <= 0) return fail(VMI_E_SHAPE, "diag_stream_read: bad args");
  DeviceGuard guard(device);
  hipError_t e = guard.err;
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

int vmi_diag_gather_read(const void* src, int64_t bytes, void* sink, int32_t chunk_kb, int32_t inflight_kb,
                         int32_t blocks, int32_t nt, int32_t device, void* stream) {
  using namespace vmi;
  if (!src || !sink || bytes < 65536 || blocks <= 0) return fail(VMI_E_SHAPE, "diag_gather_read: bad args");
  DeviceGuard guard(device);
  hipError_t e = guard.err;
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  const uint32_t nchunks = (uint32_t)(bytes / ((int64_t)chunk_kb * 1024));
  uint32_t stride = 2654435761u % nchunks;  // Knuth multiplicative hash, made coprime below
  if (stride < 2) stride = 7;
  auto gcd = [](uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; };
  while (gcd(stride, nchunks) != 1) ++stride;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const u32x4* p = static_cast<const u32x4*>(src);
  uint32_t* sk = static_cast<uint32_t*>(sink);
  bool ok = false;
#define VMI_GR(KB, IF)                                                                                      \
  if (chunk_kb == KB && inflight_kb == IF) {                                                                \
    ok = true;                                                                                              \
    if (nt) hipLaunchKernelGGL((gather_read_kernel<KB, IF, true>), dim3(blocks), dim3(256), 0, st, p, nchunks, stride, sk); \
    else hipLaunchKernelGGL((gather_read_kernel<KB, IF, false>), dim3(blocks), dim3(256), 0, st, p, nchunks, stride, sk);   \
  }
#define VMI_GR_ROW(KB) VMI_GR(KB, 1) VMI_GR(KB, 2) VMI_GR(KB, 4) VMI_GR(KB, 8) VMI_GR(KB, 16)
  VMI_GR_ROW(1) VMI_GR_ROW(2) VMI_GR_ROW(4) VMI_GR_ROW(8) VMI_GR_ROW(16) VMI_GR_ROW(32) VMI_GR_ROW(64)
#undef VMI_GR_ROW
#undef VMI_GR
  if (!ok) return fail(VMI_E_SHAPE, "diag_gather_read: chunk_kb in {1..64}, inflight_kb in {1,2,4,8,16} (powers of two)");
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "diag_gather_read launch");
  return VMI_OK;
}

// include/vmi_paged_attention_diag.h: where the core menu's pa_v1_kernel instantiations write their stage stamps
int vmi_diag_set_stage_stamps(void* records, int32_t device) {
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return -1;
  uint64_t* ptr = static_cast<uint64_t*>(records);
  const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(vmi::g_stage_stamps), &ptr, sizeof(ptr));   // (synchronous)
  (void)hipSetDevice(prev);
  return e == hipSuccess ? 0 : -(int)e;
}

#endif  // VMI_DIAG

}  // extern "C"
