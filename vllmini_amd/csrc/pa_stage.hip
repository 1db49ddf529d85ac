// pa_stage.hip — the LDS-staged form of paged_attention_v1: an EXPERIMENT kept in the build so that it can be measured
// again (BASELINE.json's north_star names "KV pages staged through LDS"; profiles/r02e_lds_staging.md has the numbers).
//
// Same decomposition as the one-wave-per-(sequence, head) kernels — same lane maps, same arithmetic, bit-identical
// results — but a page travels HBM -> LDS by `global_load_lds_dwordx4` (16 B per lane, 1 KiB per instruction, no VGPR
// in between) into a per-wave ring of R slots, and from there to registers by `ds_read_b128` when its turn comes.
// What LDS staging could buy here: a deeper prefetch queue without registers (R - 1 blocks in flight per wave whatever
// the register budget) and no address VGPRs held across the wait.  What it costs: every KV byte crosses the LDS twice.
// Never picked by the heuristic; names "stage_*", selectable by variant id.
#include "pa_kernel.hpp"

namespace vmi {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `n` vector-memory operations of this wave are outstanding (n is wave-uniform, 0..12)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
  switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 9: wait_vmcnt<9>(); break;
    case 12: wait_vmcnt<12>(); break;
    default: wait_vmcnt<0>(); break;
  }
}

// D head size, R ring slots (R - 1 blocks in flight), F8: 0 = fp16 pages, 1 = fp8 E4M3 pages (kv_scale == 1 only:
// the measurement configuration; any other scale is refused by the launcher)
// grid = (ceil(H / 4), num_seqs), block = 256.  LDS per wave: lpad*4 (logits / probabilities) + R*NL KiB (ring).
template <int D, int R, int F8>
__global__ void __launch_bounds__(256) pa_stage_kernel(const PAParams p) {
  constexpr int BS = 16;
  constexpr int EPU = F8 ? 16 : 8;       // cache elements per 16-byte unit
  constexpr int ES = F8 ? 1 : 2;         // bytes per cache element
  constexpr int NL = D * BS / EPU / 64;  // 1-KiB transfers per (block, head) tile
  static_assert(NL >= 1, "tile must fill whole 1-KiB transfers");
  constexpr int UPR = BS / EPU;          // V: 16-B units per dim row (2 for 16-bit pages, 1 for fp8)
  constexpr int RPL = 64 / UPR;          // V: rows per transfer

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.y;
  const int head = blockIdx.x * 4 + wave;
  if (head >= p.num_heads) return;
  const size_t per_wave = (size_t)p.lpad * 4 + (size_t)R * NL * 1024;
  float* logits = reinterpret_cast<float*>(smem + (size_t)wave * per_wave);
  uint16_t* ph = reinterpret_cast<uint16_t*>(logits);
  char* ring = reinterpret_cast<char*>(logits) + (size_t)p.lpad * 4;

  const int32_t* bt = p.block_tables + (int64_t)seq * p.max_blocks_per_seq;
  int32_t bt_reg = lane < p.max_blocks_per_seq ? bt[lane] : 0;
  int bt_sg = 0;
  int L = p.seq_lens[seq];
  L = L > p.lpad ? p.lpad : L;
  uint16_t* outp = reinterpret_cast<uint16_t*>(p.out) + ((int64_t)seq * p.num_heads + head) * D;
  if (L <= 0) {
    for (int d = lane; d < D; d += 64) outp[d] = 0;
    return;
  }
  const int nblk = (L + BS - 1) / BS;
  const int qpk = p.num_heads / p.num_kv_heads;
  const int c4 = lane >> 4, tk = lane & 15;
  const int64_t hoff_b = ((int64_t)(head / qpk) * p.kv_head_stride + lane * EPU) * ES;  // bytes inside a block
  const h16* qp = p.q + (int64_t)seq * p.q_stride + (int64_t)head * D;
  u32x4 qreg[NL][F8 ? 2 : 1];
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int w = 0; w < (F8 ? 2 : 1); ++w)
      qreg[i][w] = *reinterpret_cast<const u32x4*>(qp + (4 * i + c4) * EPU + 8 * w);
  const float slope = p.alibi ? p.alibi[head] : 0.f;
  f32x2_t qf[F8 ? NL : 1][8];  // fp8 pages: q as fp32 pairs for v_pk_fma_f32 (dot16_f8_s1)
  if constexpr (F8) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const h16x8 qh = __builtin_bit_cast(h16x8, qreg[i][w]);
#pragma unroll
        for (int e = 0; e < 4; ++e) qf[i][4 * w + e] = f32x2_t{(float)qh[2 * e], (float)qh[2 * e + 1]};
      }
  }

  auto table_for = [&](int b) {
    const int sg = b >> 6;
    if (sg != bt_sg) {
      const int j = sg * 64 + lane;
      bt_reg = j < p.max_blocks_per_seq ? bt[j] : 0;
      bt_sg = sg;
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(bt_reg));  // (once per 64 blocks: drains the ring)
    }
  };
  // request block b of `cache` into ring slot `slot`: NL transfers of 1 KiB, global -> LDS, non-temporal
  auto issue = [&](int slot, const void* cache, int b) {
    table_for(b);
    const int64_t phys = __builtin_amdgcn_readlane(bt_reg, b & 63);
    const char* src = static_cast<const char*>(cache) + phys * p.kv_block_stride * ES + hoff_b;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + i * 1024),
                                       (void __attribute__((address_space(3)))*)(ring + (slot * NL + i) * 1024), 16, 0, 2);
  };
  auto fetch = [&](int slot, u32x4(&r)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i) r[i] = *reinterpret_cast<const u32x4_alias*>(ring + (slot * NL + i) * 1024 + lane * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // in registers before the slot may be refilled
  };

  // ============================ K pass ============================
  float qk_max = -FLT_MAX;
#pragma unroll
  for (int k = 0; k < R - 1; ++k)
    if (k < nblk) issue(k, p.kc, k);
  for (int b = 0; b < nblk; ++b) {
    if (b + R - 1 < nblk) issue((b + R - 1) % R, p.kc, b + R - 1);
    const int younger = (nblk - 1 - b) < (R - 1) ? (nblk - 1 - b) : (R - 1);  // blocks requested after block b
    wait_vmcnt_le(younger * NL);
    u32x4 r[NL];
    fetch(b % R, r);
    const int token = b * BS + tk;
    const bool masked = token >= L;
    float accv[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if constexpr (F8) accv[i] = dot16_f8_s1<false>(qf[i], r[i]);
      else accv[i] = dot8<false>(qreg[i][0], r[i]);
    }
    float acc = accv[0];
#pragma unroll
    for (int i = 1; i < NL; ++i) acc += accv[i];
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    float qk = p.scale * acc;
    qk += (slope != 0.f) ? slope * (float)(token - L + 1) : 0.f;
    if (lane < BS) logits[token] = masked ? 0.f : qk;
    qk_max = masked ? qk_max : fmaxf(qk_max, qk);
  }

  // first V blocks (the LAST ones) go out before the softmax
#pragma unroll
  for (int k = 0; k < R - 1; ++k)
    if (k < nblk) issue(k, p.vc, nblk - 1 - k);

  const float m = wave_max(qk_max);
  float e_sum = 0.f;
  for (int i = lane; i < L; i += 64) {
    const float e = __expf(logits[i] - m);
    logits[i] = e;
    e_sum += e;
  }
  const float inv_sum = __builtin_amdgcn_rcpf(wave_sum(e_sum) + 1e-6f);
  for (int t = lane; t < nblk * BS; t += 64) {
    const float e = logits[t];
    ph[t] = t < L ? to_elem<false>(e * inv_sum) : (uint16_t)0;
  }

  // ============================ V pass, last block first ============================
  float acc[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc[i] = 0.f;
  const int hf = lane % UPR;
  const int rowl = lane / UPR;
  for (int s = 0; s < nblk; ++s) {  // step s = block nblk-1-s, in slot s % R
    const int b = nblk - 1 - s;
    if (s + R - 1 < nblk) issue((s + R - 1) % R, p.vc, nblk - 1 - (s + R - 1));
    const int younger = (nblk - 1 - s) < (R - 1) ? (nblk - 1 - s) : (R - 1);
    wait_vmcnt_le(younger * NL);
    u32x4 r[NL];
    fetch(s % R, r);
    const int token0 = b * BS + hf * EPU;
    const bool last = (b == nblk - 1);
    PV8<false> pv;
    pv.load(*reinterpret_cast<const u32x4_alias*>(ph + token0));
    if constexpr (F8) {
      PV8<false> pw;
      pw.load(*reinterpret_cast<const u32x4_alias*>(ph + token0 + 8));
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const u32x4 v0 = deq8<true, false, false>(r[i][0], r[i][1], 1.f);
        const u32x4 v1 = deq8<true, false, false>(r[i][2], r[i][3], 1.f);
        if (last) {
          acc[i] += pv.template dot<true>(v0, true, token0, L);
          acc[i] += pw.template dot<true>(v1, true, token0 + 8, L);
        } else {
          acc[i] += pv.template dot<false>(v0, false, token0, L);
          acc[i] += pw.template dot<false>(v1, false, token0 + 8, L);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        if (last) acc[i] += pv.template dot<true>(r[i], true, token0, L);
        else acc[i] += pv.template dot<false>(r[i], false, token0, L);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int mm = 1; mm < UPR; mm <<= 1) acc[i] += __shfl_xor(acc[i], mm);
  if (hf == 0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int row = RPL * i + rowl;
      if (row < D) outp[row] = to_elem<false>(acc[i]);
    }
  }
}

#define VMI_ROW_S(NAME, D, R, F8) \
  {NAME, D, 16, 4, 1, R, true, 1, false, (pa_kernel_t)pa_stage_kernel<D, R, F8>, 0, 0, 0, F8, false, false, false, false, true},

Variant g_stage_variants[] = {
    VMI_ROW_S("stage_d64_r2", 64, 2, 0)
    VMI_ROW_S("stage_d64_r3", 64, 3, 0)
    VMI_ROW_S("stage_d64_r4", 64, 4, 0)
    VMI_ROW_S("stage_d128_r2", 128, 2, 0)
    VMI_ROW_S("stage_d128_r3", 128, 3, 0)
    VMI_ROW_S("stage_fp8_d64_r2", 64, 2, 1)
    VMI_ROW_S("stage_fp8_d64_r3", 64, 3, 1)
    VMI_ROW_S("stage_fp8_d64_r4", 64, 4, 1)
    VMI_ROW_S("stage_fp8_d128_r3", 128, 3, 1)
};
const int g_stage_nvariants = (int)(sizeof(g_stage_variants) / sizeof(g_stage_variants[0]));

}  // namespace vmi
