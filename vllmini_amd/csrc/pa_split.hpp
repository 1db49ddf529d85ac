// pa_split.hpp — paged_attention_v1 with ONE (sequence, head) spread over several workgroups of one launch (round 5).
//
// Why.  The reference launches one workgroup per (head, sequence) (attention_kernels.cu:734-735) and so did every kernel
// here until now: a head's partial results could only meet in LDS, i.e. on one CU.  With fewer (sequence, head) items
// than the chip has CUs — batch 1 ... 16 at 12 heads, the regime the reference's own scheduler runs (scheduler.py:60) —
// most CUs idle, and with 1.5 items per CU (BASELINE configs[1]: 384 items on 256 CUs) half the CUs carry twice the
// load of the others.  The reference's own answer is paged_attention_v2 (:529-669, 828-990): 512-token partitions,
// each normalised BY ITSELF and rounded to fp16, then merged by a second kernel — other rounding points than v1's, a
// second launch, and no help below 1024 tokens.
//
// What.  An item is dealt to NW waves (block b -> wave b mod NW), NW / 4 workgroups of four waves, all resident.  Every
// wave runs the K pass over its own blocks and publishes its (max, sum of exp) as ONE 8-byte granule in a caller-owned
// workspace; every wave then polls the item's NW granules, and so knows the item's GLOBAL maximum and exp sum before it
// rounds a single probability: p = half(exp(logit - max) * 1 / (sum + 1e-6)) is formed exactly where and how the
// reference forms it (:334-346, 398-400).  V pass over the same blocks, fp32 partial outputs reduced in LDS per
// workgroup, written through to the workspace; the LAST workgroup of the item to arrive (one counter per item) adds the
// partials in workgroup order — a fixed order, whatever the arrival order — stores the row and puts the item's granules
// and counter back to zero.  One launch, no zeroing launch in front, no merge kernel behind.
//
// Visibility (MI355X: 8 XCDs with private L2s, per-CU L1 never refreshed by other CUs): every shared word is written and
// read with agent-scope relaxed atomics (global_store / global_load ... sc1: write-through, L1 bypassed), payload stores
// are drained (s_waitcnt vmcnt(0)) in front of the arrival atomic, and the granule IS its own flag (sum of exp >= 1, so
// its upper word is never zero once written).  Every spin is bounded: a poll that gives up sets status[0] and the
// launch finishes with wrong numbers instead of hanging the device.
//
// Items.  One query head of one sequence, or — grouped-query attention, HPT = 4 — four query heads of ONE KV head: every
// K / V tile of a wave's blocks is loaded (over fp8 pages: decoded) once for the four, q.K^T of a block on the matrix cores,
// every head with its own granule, probabilities and partial row.  Pages: fp16, or fp8 E4M3 (F8 = 1).
//
// Rounds.  The waiting is safe when every workgroup of the launch is on the chip.  A launch of more workgroups than are
// resident (launch bounds, LDS) or than the workspace has words for runs the ROUNDS = true twin of the kernel on a grid of
// whole items that IS resident: a workgroup serves grid-strided items, an item's workgroups always together; the workspace
// is then indexed by the place in the grid, two halves alternating by round (the loop at the end of the kernel).
//
// The workspace belongs to the caller (ownership row: "if a split-KV path needs scratch, the
// Python wrapper allocates it with torch on the same stream"); the library retains nothing.  Layout: pa_split_layout().
#pragma once

#include "pa_kernel.hpp"

namespace vmi {

struct PASplit {
  unsigned long long* slots;  // [items][HPT][nw] {float max, float exp_sum} granules; 0 = not published yet  (in rounds: [2][places] for [items])
  float* partials;            // [items][nw / wpg][HPT * D]  fp32 partial outputs of the item's workgroups
  unsigned int* counters;     // [items]          arrivals of the item's workgroups; back to 0 when the item is done; [4096 + 2 * place + parity]: rounds finished
  unsigned int* status;       // [0]: number of polls that gave up (must stay 0)
  int32_t nw;                 // waves per item (<= 256, a multiple of the waves per workgroup)
  int32_t wtok;               // logits per wave held in LDS: 16 * max(8, ceil(ceil(max_seq_len / 16) / nw)) (split_wtok)
  int32_t flags;              // SPF_*
};
constexpr int SPF_GMAJOR = 1;        // workgroup index = g * items + item (an item's workgroups far apart in dispatch order)
constexpr int SLOTS_PER_LANE = 4;     // granules a lane reads in the poll
constexpr int SPLIT_MAX_WAVES = 64 * SLOTS_PER_LANE;
constexpr unsigned SPLIT_SPIN_LIMIT = 1u << 20;
constexpr int SPLIT_MAX_ITEMS = 8192;  // arrival counters in the workspace: [0, 4096) by item / place-and-round, [4096, 8192) rounds finished by place

#ifdef VMI_DIAG
// diagnostic library: eight stamps per wave (100 MHz clock): entry, lengths known, first K group consumed, K pass done,
// granule published, exchange complete, V pass done, end — then HW_ID, XCC_ID | blocks << 8
static __device__ uint64_t* g_split_stamps = nullptr;
#define VMI_SSTAMP(k) do { if (tl_) ts_[k] = wall_clock64(); } while (0)
#else
#define VMI_SSTAMP(k) ((void)0)
#endif

typedef unsigned long long __attribute__((address_space(1))) gu64_t;
typedef unsigned int __attribute__((address_space(1))) gu32_t;
typedef float __attribute__((address_space(1))) gf32_t;

// Workgroups per CU the kernels are compiled for (launch bounds; what the host counts as resident).  The kernels that go in
// rounds run where LDS, not registers, bounds the occupancy (long contexts): half the workgroups, twice the registers.
__host__ __device__ constexpr int split_wgs_per_cu(int D, int HPT, bool rounds) {
  const int n = HPT > 1 ? (D == 64 ? 4 : 2) : (D == 64 ? 6 : 3);
  return rounds ? (n + 1) / 2 : n;
}

// D head size (64 | 128), U blocks per register group, NT non-temporal page loads, VA V groups requested in front of
// the exchange (1 | 2), F8 = 1: the pages hold fp8 E4M3 bytes (K [NB, H, D/16, 16, 16], V [NB, H, D, 16]; every element becomes
// half(float(fp8) * kv_scale) first, reference quant_utils.cuh:295-300, then the fp16 arithmetic applies unchanged).
// HPT > 1 (grouped-query attention): an item is HPT query heads of ONE KV head (num_heads / num_kv_heads a multiple of HPT) —
// a wave loads each K / V tile of its blocks ONCE and uses it for the HPT heads (the reference re-reads it per query head,
// attention_kernels.cu:153); every head keeps its own logits, granules, probabilities and partial rows, so a head's
// arithmetic is what HPT = 1 computes.  Up to 64 waves per item there (one granule per lane and head in the poll).
// Block size 16, fp16 query.  grid = items * (nw / wpg), block = wpg * 64.
// Launch bounds: six (head size 128: three; grouped: four / two) workgroups per CU — what the host counts as resident.
// LDS = wpg * HPT * (wtok * 4 (logits) + wtok * 2 (probabilities) + D * 4 (partial out)) + 16 (the "I am last" flag).
template <int D, int U, bool NT, int VA, int F8 = 0, int HPT = 1, bool ROUNDS = false>
__global__ void __launch_bounds__(256, split_wgs_per_cu(D, HPT, ROUNDS)) pa_split_kernel(const PAParams p, const PASplit sp) {
  constexpr int BS = 16;
  constexpr int EPU = F8 ? 16 : 8;       // cache elements per 16-byte unit
  constexpr int ES = F8 ? 1 : 2;         // bytes per cache element
  constexpr int NL = D * BS / EPU / 64;  // 1-KiB loads per (block, head) tile: 2 | 4 (fp8 pages: 1 | 2)
  constexpr int CPL = 4;                 // K: 16-B chunks (EPU dims) per load
  constexpr int UPR = BS / EPU;          // V: 16-B units per dim row
  constexpr int RPL = 64 / UPR;          // V: rows per load
  constexpr int QW = F8 ? 2 : 1;         // 16-byte pieces of q facing one K unit
  constexpr int SPL = HPT > 1 ? 1 : SLOTS_PER_LANE;  // granules a lane reads per head in the poll
  static_assert(D == 64 || D == 128, "head size 64 or 128");
  extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef VMI_DIAG
  uint64_t* const tl_ = g_split_stamps;
  uint64_t ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  VMI_SSTAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpg = blockDim.x >> 6;
  const int NW = sp.nw;
  const int G = NW / wpg;
  const int hgroups = p.num_heads / HPT;     // items per sequence
  const int items = p.num_seqs * hgroups;
  // Workgroup vb of the launch's items * G serves workgroup vb % G of item vb / G.  Usually the grid IS that many workgroups (all
  // resident).  With more items than are resident the grid is a multiple of G that is, and a workgroup serves vb = blockIdx.x,
  // blockIdx.x + gridDim.x, ... in ROUNDS: the G workgroups of an item sit at the same places of the grid in the same round, so
  // an item's waves always run together.  The workspace is indexed by the PLACE in the grid (not by the item), two regions
  // alternating by round (see the loop at the end).
  const int total = items * G;
  constexpr bool looped = ROUNDS;   // (a kernel of its own: the loop costs the one-round kernels registers)
  // A poll that gives up (SPLIT_SPIN_LIMIT: co-residency violated — another kernel holds the CUs, two split launches share a
  // workspace, dispatch out of order) must not pass unnoticed: besides the count in status[0] the item's softmax sum becomes
  // NaN, so every output element of the item is NaN (the launch used to finish with plausible wrong numbers).
  bool poisoned = false;
  auto process = [&](int vb, int wsi) {
  int item, g;
  if (sp.flags & SPF_GMAJOR) {
    g = vb / items;
    item = vb - g * items;
  } else {
    item = vb / G;
    g = vb - item * G;
  }
  const int seq = item / hgroups;
  const int head0 = (item - seq * hgroups) * HPT;
  const int w = g * wpg + wave;  // this wave's place among the item's NW

  // requested before seq_len is known (any entry of the row is readable): my first 64 blocks' physical ids
  const int32_t* bt = p.block_tables + (int64_t)seq * p.max_blocks_per_seq;
  int bt_sg = 0;
  int32_t bt_reg = (w + lane * NW < p.max_blocks_per_seq) ? bt[w + lane * NW] : 0;
  int L = p.seq_lens[seq];
  L = L > p.lpad ? p.lpad : L;  // (seq_len > max_seq_len: truncated to the LDS reserved, as pa_v1_kernel does)

  const int c4 = lane >> 4;  // chunk within a K load
  const int tk = lane & 15;  // token within the block
  const int qpk = p.num_heads / p.num_kv_heads;
  const int64_t hoff = (int64_t)(head0 / qpk) * p.kv_head_stride + lane * EPU;   // cache strides are in elements
  // HPT = 1: this lane's EPU dims of q facing each K load (fp32 FMA chains on the VALU, the reference's arithmetic).
  // HPT > 1: q.K^T of a block is a (16 tokens) x (HPT heads, padded to 16) x (D dims) product on the matrix cores — the K tile as
  // loaded IS the A operand of v_mfma_f32_16x16x32_f16 (lane = chunk * 16 + token holds 8 dims of one token), B is q with
  // lane & 15 = head; products of fp16 operands are exact in fp32 and the accumulation is fp32: the reference's arithmetic up to
  // summation order (as in pa_v1_kernel's grouped-query kernels, pa_kernel.hpp QK_MFMA).
  constexpr bool QKM = HPT > 1;
  float slope[QKM ? 1 : HPT];
  u32x4 qreg[QKM ? 1 : HPT][NL][QW];
  u32x4 qB[QKM ? NL : 1][QKM ? QW : 1];   // (fp8 pages: a K unit is 16 dims = two 8-dim operands)
  if constexpr (QKM) {
    const int n = lane & 15;
    const bool has = n < HPT;
    const h16* qp = p.q + (int64_t)seq * p.q_stride + (int64_t)(head0 + (has ? n : 0)) * D;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
      for (int qw = 0; qw < QW; ++qw) qB[i][qw] = has ? *reinterpret_cast<const u32x4*>(qp + (CPL * i + c4) * EPU + 8 * qw) : zero4;
    slope[0] = (has && p.alibi) ? p.alibi[head0 + n] : 0.f;   // of head (lane & 15)
  } else {
#pragma unroll
    for (int hh = 0; hh < HPT; ++hh) {
      slope[hh] = p.alibi ? p.alibi[head0 + hh] : 0.f;
      const h16* qp = p.q + (int64_t)seq * p.q_stride + (int64_t)(head0 + hh) * D;
#pragma unroll
      for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int qw = 0; qw < QW; ++qw) qreg[hh][i][qw] = *reinterpret_cast<const u32x4*>(qp + (CPL * i + c4) * EPU + 8 * qw);
    }
  }

  uint16_t* outp = reinterpret_cast<uint16_t*>(p.out) + ((int64_t)seq * p.num_heads + head0) * D;  // HPT adjacent rows
  if (L <= 0) {  // exp_sum = 0 -> the rows are zero (reference: no tokens)
    if (w == 0)
      for (int d = lane; d < HPT * D; d += 64) outp[d] = 0;
    return;
  }
  const int nblk = (L + BS - 1) / BS;
  // The host sized NW for max_seq_len — which the reference's scheduler sets to the CAPACITY of a block table
  // (scheduler.py:97), not to the lengths at hand.  A wave wants at least four blocks (r05_split_kernels.md: 1 - 2 blocks per
  // wave lose to the exchange they pay for), so this item uses NWe = 4 * floor(blocks / 16) of its NW waves (>= one
  // workgroup); every wave of the item derives the same NWe from the same seq_len.
  int NWe = nblk / (4 * wpg) * wpg;
  NWe = NWe < wpg ? wpg : (NWe > NW ? NW : NWe);
  if (NWe != NW) bt_sg = -1;                 // the table entries requested above were dealt for NW waves
  const int na = nblk < NWe ? nblk : NWe;   // waves of the item that own a block
  const int ga = (na + wpg - 1) / wpg;      // workgroups of the item that hold such a wave
  if (g >= ga) return;                      // the whole workgroup has nothing to do (uniform)
  const int nmy = w < na ? (nblk - w + NWe - 1) / NWe : 0;  // my blocks: b = w + idx * NWe
  VMI_SSTAMP(1);

  const int wtok = sp.wtok;
  float* lg0 = reinterpret_cast<float*>(smem) + (size_t)wave * HPT * wtok;                               // my logits, + hh * wtok
  uint16_t* ph0 = reinterpret_cast<uint16_t*>(reinterpret_cast<float*>(smem) + (size_t)wpg * HPT * wtok) + (size_t)wave * HPT * wtok;
  float* osm = reinterpret_cast<float*>(smem + (size_t)wpg * HPT * wtok * 6);                            // [wpg][HPT][D]

  float acc[HPT][NL];
#pragma unroll
  for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
    for (int i = 0; i < NL; ++i) acc[hh][i] = 0.f;
  const int hf = lane % UPR;
  const int rowl = lane / UPR;

  if (nmy > 0) {
    const int ngroups = (nmy + U - 1) / U;
    auto table_for = [&](int gi) {
      const int sg = (gi * U) >> 6;
      if (sg != bt_sg) {
        const int b = w + (sg * 64 + lane) * NWe;
        bt_reg = (b < p.max_blocks_per_seq) ? bt[b] : 0;
        bt_sg = sg;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bt_reg));  // (inside the branch: see pa_kernel.hpp table_for)
      }
    };
    auto load_group = [&](u32x4(&r)[U][NL], const h16* cache, int gi) {
      table_for(gi);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        int idx = gi * U + j;
        idx = idx < nmy ? idx : nmy - 1;  // padding slots re-read my last block
        const int64_t phys = __builtin_amdgcn_readlane(bt_reg, idx & 63);
        const char* blk = reinterpret_cast<const char*>(cache) + (phys * p.kv_block_stride + hoff) * ES;
#pragma unroll
        for (int i = 0; i < NL; ++i) r[j][i] = ld16<NT>(blk + i * 1024);
      }
    };
    float qk_max[HPT];
#pragma unroll
    for (int hh = 0; hh < HPT; ++hh) qk_max[hh] = -FLT_MAX;
    float qmaxB = -FLT_MAX;  // QKM: running max of head (lane & 15) over this lane's token rows
    auto compute_k = [&](u32x4(&r)[U][NL], int gi) {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int idx = gi * U + j;
        if (idx < nmy) {
          if constexpr (QKM) {
            f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NL; ++i)
            {
              if constexpr (F8) {  // the unit's 16 bytes -> half(float(fp8) * kv_scale), two 8-dim operands
                const u32x4 k0 = deq8<false>(r[j][i][0], r[j][i][1], p.kv_scale);
                const u32x4 k1 = deq8<false>(r[j][i][2], r[j][i][3], p.kv_scale);
                d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, k0), __builtin_bit_cast(h16x8, qB[i][0]), d4, 0, 0, 0);
                d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, k1), __builtin_bit_cast(h16x8, qB[i][QW - 1]), d4, 0, 0, 0);
              } else {
                d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, r[j][i]), __builtin_bit_cast(h16x8, qB[i][0]), d4, 0, 0, 0);
              }
            }
            // C/D layout: column = lane & 15 (head), rows 4 * (lane >> 4) + reg (tokens of this block)
            const int tok4 = (w + idx * NWe) * BS + 4 * c4;
            f32x4 lg4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float qk = p.scale * d4[e];
              qk += (slope[0] != 0.f) ? slope[0] * (float)(tok4 + e - L + 1) : 0.f;
              const bool m = tok4 + e >= L;
              lg4[e] = m ? 0.f : qk;
              qmaxB = m ? qmaxB : fmaxf(qmaxB, qk);
            }
            if ((lane & 15) < HPT) *reinterpret_cast<f32x4_alias*>(lg0 + (lane & 15) * wtok + idx * BS + 4 * c4) = lg4;
          } else {
            const int token = (w + idx * NWe) * BS + tk;
            const bool masked = token >= L;
#pragma unroll
            for (int hh = 0; hh < HPT; ++hh) {
              float accv[NL];
#pragma unroll
              for (int i = 0; i < NL; ++i) {
                if constexpr (F8) accv[i] = dot16_f8<false, false, false>(qreg[hh][i][0], qreg[hh][i][QW - 1], r[j][i], p.kv_scale);
                else accv[i] = dot8<false>(qreg[hh][i][0], r[j][i]);
              }
              float a = accv[0];
#pragma unroll
              for (int i = 1; i < NL; ++i) a += accv[i];
              a += __shfl_xor(a, 16);
              a += __shfl_xor(a, 32);
              float qk = p.scale * a;
              qk += (slope[hh] != 0.f) ? slope[hh] * (float)(token - L + 1) : 0.f;
              if (lane < BS) lg0[hh * wtok + idx * BS + tk] = masked ? 0.f : qk;
              qk_max[hh] = masked ? qk_max[hh] : fmaxf(qk_max[hh], qk);
            }
          }
        }
      }
    };

    u32x4 ra[U][NL], rb[U][NL];
    const bool single = ngroups == 1;
    if (single) {  // everything I own fits one register group: K and V pages in one round trip
      load_group(ra, p.kc, 0);
      load_group(rb, p.vc, 0);
      compute_k(ra, 0);
      VMI_SSTAMP(2);
    } else {
      load_group(ra, p.kc, 0);
      int gi = 0;
      for (; gi + 2 <= ngroups; gi += 2) {
        load_group(rb, p.kc, gi + 1);
        compute_k(ra, gi);
#ifdef VMI_DIAG
        if (gi == 0) VMI_SSTAMP(2);
#endif
        if (gi + 2 < ngroups) load_group(ra, p.kc, gi + 2);
        compute_k(rb, gi + 1);
      }
      if (gi < ngroups) compute_k(ra, gi);
      // the V pass walks my groups from the last one back (the sequence's last block was written a moment ago by
      // reshape_and_cache: its wait hides behind the exchange); VA groups are requested in front of the exchange
      load_group(ra, p.vc, ngroups - 1);
      if constexpr (VA >= 2) load_group(rb, p.vc, ngroups - 2);
    }
    VMI_SSTAMP(3);

    if constexpr (QKM) {  // per-head maxima live in lanes (lane & 15) = head: fold the four row groups, then hand out
      qmaxB = fmaxf(qmaxB, __shfl_xor(qmaxB, 16));
      qmaxB = fmaxf(qmaxB, __shfl_xor(qmaxB, 32));
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) qk_max[hh] = __shfl(qmaxB, hh);
    }
    // ---- my share of the softmax statistics, per head: max over my tokens, sum of exp relative to it ----
    float M[HPT], S[HPT];
#pragma unroll
    for (int hh = 0; hh < HPT; ++hh) {
      const float m_w = wave_max(qk_max[hh]);
      float e_sum = 0.f;
      for (int t = lane; t < nmy * BS; t += 64) {
        const int token = (w + (t >> 4) * NWe) * BS + (t & 15);
        e_sum += token < L ? __expf(lg0[hh * wtok + t] - m_w) : 0.f;
      }
      M[hh] = m_w;
      S[hh] = wave_sum(e_sum);
    }

    if (na > 1) {
      gu64_t* sl = (gu64_t*)sp.slots + (size_t)wsi * HPT * NW;   // + hh * NW
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        if (lane == hh) {
          const unsigned long long g8 = (unsigned long long)__builtin_bit_cast(uint32_t, M[hh]) |
                                        ((unsigned long long)__builtin_bit_cast(uint32_t, S[hh]) << 32);  // sum >= 1: never 0
          __hip_atomic_store(sl + hh * NW + w, g8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      VMI_SSTAMP(4);
      // one lane per granule (and head), up to SPL passes of 64: every wave of the item reads all na granules of every head
      unsigned long long x[HPT][SPL];
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
        for (int q = 0; q < SPL; ++q) x[hh][q] = 1ull << 32;
      for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
          for (int q = 0; q < SPL; ++q) {
            if (lane + 64 * q < na) x[hh][q] = __hip_atomic_load(sl + hh * NW + lane + 64 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (x[hh][q] >> 32) != 0ull;
          }
        if (__all(ok)) break;
        if (spins >= SPLIT_SPIN_LIMIT) {  // never on a healthy launch; finish — with NaN outputs — rather than hang the device
          if (lane == 0) atomicAdd(sp.status, 1u);
          poisoned = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        float mj[SPL], sj[SPL], mloc = -FLT_MAX;
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
          const bool has = lane + 64 * q < na;
          mj[q] = has ? __builtin_bit_cast(float, (uint32_t)x[hh][q]) : -FLT_MAX;
          sj[q] = has ? __builtin_bit_cast(float, (uint32_t)(x[hh][q] >> 32)) : 0.f;
          mloc = fmaxf(mloc, mj[q]);
        }
        M[hh] = wave_max(mloc);
        float sloc = 0.f;
#pragma unroll
        for (int q = 0; q < SPL; ++q) sloc += sj[q] * __expf(mj[q] - M[hh]);
        S[hh] = wave_sum(sloc);  // the same values in the same lanes in every wave of the item: one S for all
      }
    }
    if (poisoned) {
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) S[hh] = __builtin_nanf("");
    }
    VMI_SSTAMP(5);

    // ---- probabilities of my tokens, rounded to fp16 once (:398-400); positions past the context become 0 ----
#pragma unroll
    for (int hh = 0; hh < HPT; ++hh) {
      const float inv = __builtin_amdgcn_rcpf(S[hh] + 1e-6f);  // :342
      for (int t = lane; t < nmy * BS; t += 64) {
        const int token = (w + (t >> 4) * NWe) * BS + (t & 15);
        ph0[hh * wtok + t] = token < L ? to_elem<false>(__expf(lg0[hh * wtok + t] - M[hh]) * inv) : (uint16_t)0;
      }
    }

    // ---- V pass ----
    auto compute_v = [&](auto masked, u32x4(&r)[U][NL], int gi) {
      constexpr bool MASK = decltype(masked)::value;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int idx = gi * U + j;
        if (idx < nmy) {
          const int b = w + idx * NWe;
          const int token0 = b * BS + hf * EPU;
          const bool last = (b == nblk - 1);
          u32x4 vd[F8 ? NL : 1][2];  // fp8 pages: the tile decoded once, whatever the number of heads
          if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
              vd[i][0] = deq8<false>(r[j][i][0], r[j][i][1], p.kv_scale);
              vd[i][1] = deq8<false>(r[j][i][2], r[j][i][3], p.kv_scale);
            }
          }
#pragma unroll
          for (int hh = 0; hh < HPT; ++hh) {
            const uint16_t* php = ph0 + hh * wtok + idx * BS;
            PV8<false> pv;
            pv.load(*reinterpret_cast<const u32x4_alias*>(php + hf * EPU));
            if constexpr (F8) {  // a 16-byte unit is a whole 16-token row: two of the reference's 8-token groups
              PV8<false> pw;
              pw.load(*reinterpret_cast<const u32x4_alias*>(php + 8));
#pragma unroll
              for (int i = 0; i < NL; ++i) {
                acc[hh][i] += pv.template dot<MASK>(vd[F8 ? i : 0][0], last, token0, L);
                acc[hh][i] += pw.template dot<MASK>(vd[F8 ? i : 0][1], last, token0 + 8, L);
              }
            } else {
#pragma unroll
              for (int i = 0; i < NL; ++i) acc[hh][i] += pv.template dot<MASK>(r[j][i], last, token0, L);
            }
          }
        }
      }
    };
    if (single) {
      compute_v(std::true_type{}, rb, 0);
    } else {
      const int last = ngroups - 1;
      if constexpr (VA < 2) load_group(rb, p.vc, last - 1);
      compute_v(std::true_type{}, ra, last);
      int s = 1;
      for (; s + 1 < ngroups; s += 2) {
        load_group(ra, p.vc, last - (s + 1));
        compute_v(std::false_type{}, rb, last - s);
        if (s + 2 < ngroups) load_group(rb, p.vc, last - (s + 2));
        compute_v(std::false_type{}, ra, last - (s + 1));
      }
      if (s < ngroups) compute_v(std::false_type{}, rb, last - s);
    }
#pragma unroll
    for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
      for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int mm = 1; mm < UPR; mm <<= 1) acc[hh][i] += __shfl_xor(acc[hh][i], mm);  // the 8-token groups of a row held by other lanes
  }
  VMI_SSTAMP(6);

  // ---- partial outputs: the workgroup's waves meet in LDS, the item's workgroups in the workspace ----
  constexpr int HD = HPT * D;   // floats of an item's output rows
  const int nwa = (na - g * wpg) < wpg ? (na - g * wpg) : wpg;  // waves of this workgroup that own blocks
  if (hf == 0) {
#pragma unroll
    for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
      for (int i = 0; i < NL; ++i) osm[(wave * HPT + hh) * D + RPL * i + rowl] = acc[hh][i];
  }
  lds_barrier();
  int* last_flag = reinterpret_cast<int*>(osm + wpg * HD);
  gu64_t* my_slots = (gu64_t*)sp.slots + (size_t)wsi * HPT * NW;
  auto reset_slots = [&]() {  // every wave of the item has read the granules by now: back to "not published"
#pragma unroll
    for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
      for (int q = 0; q < SPL; ++q)
        if (lane + 64 * q < na) __hip_atomic_store(my_slots + hh * NW + lane + 64 * q, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (wave == 0) {
    float part[HD / 64];
#pragma unroll
    for (int k = 0; k < HD / 64; ++k) {
      float ssum = 0.f;
      for (int wv = 0; wv < nwa; ++wv) ssum += osm[wv * HD + lane + 64 * k];
      part[k] = ssum;
    }
    if (ga == 1) {
#pragma unroll
      for (int k = 0; k < HD / 64; ++k) outp[lane + 64 * k] = to_elem<false>(part[k]);
      if (na > 1) reset_slots();  // (my own waves exchanged through the workspace; all of them are past the barrier)
      if (looped) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      gf32_t* mine = (gf32_t*)sp.partials + ((size_t)wsi * G + g) * HD;
#pragma unroll
      for (int k = 0; k < HD / 64; ++k) __hip_atomic_store(mine + lane + 64 * k, part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the payload has left before the arrival is counted
      unsigned old = 0;
      if (lane == 0) {
        old = __hip_atomic_fetch_add((gu32_t*)sp.counters + wsi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last_flag = old == (unsigned)(ga - 1);
      }
    }
  }
  if (ga > 1) {
    lds_barrier();
    if (*last_flag) {
      // The item's last workgroup adds the partial rows — its four waves a quarter each (wave k: workgroups k, k + 4, ...),
      // then wave 0 the four sums: an order fixed by ga alone, whatever the arrival order.
      const gf32_t* all = (const gf32_t*)sp.partials + (size_t)wsi * G * HD;
      float o[HD / 64];
#pragma unroll
      for (int k = 0; k < HD / 64; ++k) o[k] = 0.f;
      constexpr int QB = HPT > 1 ? 2 : 8;   // workgroups' rows in flight per trip
      for (int g0 = wave; g0 < ga; g0 += QB * wpg) {
        float t8[QB][HD / 64];
#pragma unroll
        for (int q = 0; q < QB; ++q)
#pragma unroll
          for (int k = 0; k < HD / 64; ++k)
            t8[q][k] = (g0 + q * wpg < ga) ? __hip_atomic_load(all + (size_t)(g0 + q * wpg) * HD + lane + 64 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
        for (int q = 0; q < QB; ++q)
#pragma unroll
          for (int k = 0; k < HD / 64; ++k) o[k] += t8[q][k];
      }
#pragma unroll
      for (int k = 0; k < HD / 64; ++k) osm[wave * HD + lane + 64 * k] = o[k];
      lds_barrier();
      if (wave == 0) {
#pragma unroll
        for (int k = 0; k < HD / 64; ++k) {
          float ssum = 0.f;
          for (int wv = 0; wv < wpg; ++wv) ssum += osm[wv * HD + lane + 64 * k];
          outp[lane + 64 * k] = to_elem<false>(ssum);
        }
        reset_slots();
        if (lane == 0) __hip_atomic_store((gu32_t*)sp.counters + wsi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (looped) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the words are zero before this workgroup moves on
      }
    }
  }
#ifdef VMI_DIAG
  if (tl_) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ts_[7] = wall_clock64();
    if (lane == 0) {
      uint64_t* rec = tl_ + ((size_t)vb * wpg + wave) * 10;
#pragma unroll
      for (int k = 0; k < 8; ++k) rec[k] = ts_[k];
      rec[8] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);                          // HW_REG_HW_ID
      rec[9] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((uint64_t)nmy << 8);  // HW_REG_XCC_ID
    }
  }
#endif
  };  // process

  if constexpr (!ROUNDS) {
    process((int)blockIdx.x, (int)blockIdx.x / G);
  } else {
    // The G workgroups at one place of the grid go through the rounds together, but not in step: one that has nothing to do in
    // a round (an empty sequence, a context too short for its waves) is through it at once.  Round r + 2 uses the words of
    // round r again, so nobody starts it before all G have FINISHED round r.  Two running counts per place, of the (workgroup,
    // round)s finished in even and in odd rounds: before round r a workgroup waits for G * (r / 2) in the count of r's parity
    // — so many can only come from rounds r - 2, r - 4, ..., complete: nobody adds to that count from round r or later before
    // somebody has entered round r.  Almost always there already.  The last arrival of a count zeroes it.
    const int places = (int)gridDim.x / G;  // items in flight per round
    const int place = (int)blockIdx.x / G;
    gu32_t* done = (gu32_t*)sp.counters + SPLIT_MAX_ITEMS / 2 + 2 * place;   // [parity]
    const unsigned nrounds = (unsigned)((total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);   // (the same for all G)
    unsigned round = 0;
    for (int vb = (int)blockIdx.x; vb < total; vb += (int)gridDim.x, ++round) {
      if (round >= 2) {
        unsigned spins = 0;
        while (__hip_atomic_load(done + (round & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G * (round >> 1)) {
          if (++spins >= SPLIT_SPIN_LIMIT) {
            if (lane == 0) atomicAdd(sp.status, 1u);
            poisoned = true;   // (this round's item, and every later one of this workgroup, comes out as NaN)
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      process(vb, place + (int)(round & 1) * places);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my words of this round are where they belong
      lds_barrier();  // ... for all four waves; and the workgroup's LDS (partial rows, the "I am last" flag) is free again
      if (wave == 0 && lane == 0) {
        const unsigned all = (unsigned)G * ((nrounds + 1 - (round & 1)) >> 1);   // this parity's (workgroup, round)s of the launch
        const unsigned old = __hip_atomic_fetch_add(done + (round & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == all) __hip_atomic_store(done + (round & 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// Bytes of workspace and where its parts lie.  Sized for any launch the split kernels take: at most `max_wgs` workgroups
// (they must all be resident) of four waves.
struct SplitLayout {
  size_t status_off, counters_off, slots_off, partials_off, bytes;
};
constexpr int SPLIT_MAX_WGS = 2048;    // 8 resident 256-thread workgroups per CU x 256 CUs
static inline SplitLayout pa_split_layout(int head_size) {
  SplitLayout l;
  l.status_off = 0;                                                         // 256 B header
  l.counters_off = 256;
  l.slots_off = l.counters_off + (size_t)SPLIT_MAX_ITEMS * 4;
  l.partials_off = l.slots_off + (size_t)SPLIT_MAX_WGS * 4 * 8;
  l.bytes = l.partials_off + (size_t)SPLIT_MAX_WGS * (size_t)head_size * 4;
  return l;
}

typedef void (*pa_split_kernel_t)(const PAParams, const PASplit);

}  // namespace vmi
