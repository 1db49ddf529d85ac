// pa_queue.hpp — the BALANCED form of paged_attention_v1 for a full chip (gfx950 only).
//
// pa_v1_kernel (pa_kernel.hpp) gives every (sequence, head) its own resident wave.  On equal lengths that is the
// fastest form found (cfg3: 0.82 of the HBM roofline), on RAGGED batches nothing rebalances the chip when the short
// sequences are done: the long ones run on alone, each limited to its own bytes in flight per memory round trip
// (cfg3 with seq_lens ~ U{1..1024}: 0.61).  The launch geometry cannot react — the operator's argument list
// (reference attention_kernels.cu:805-826) carries the lengths only as a DEVICE tensor.
//
// pa_q_kernel decides on the device.  One launch geometry — 3 workgroups of 4 waves per CU, all resident — and two
// modes, chosen by every wave from the same seq_lens (so all waves agree without talking to each other):
//
//   S  (equal lengths, one item per wave)   wave w owns item w, 1 block per register group: pa_v1_kernel's best form,
//                                           same lane maps, same order of operations.
//   Q  (ragged, or more items than waves)   2 waves per workgroup stay as WORKERS, the rest retire (fp8 pages and head
//                                           size 128: all 4 stay, WQ_SOLO below).  A worker runs
//                                           items one after the other, 2 blocks per register group, so the chip holds
//                                           the same bytes in flight with half the waves.  Items are handed out
//                                           longest first: the sequences are ranked by a deterministic 64-bucket
//                                           counting sort that every workgroup repeats for itself in LDS, and worker w
//                                           takes ranks w, 2W-1-w, 2W+w, ... (a snake over the ranks: each worker
//                                           pairs a long item with a short one — what longest-processing-time-first
//                                           list scheduling would hand it, without any communication).
//                                           The table slice, length and q of the NEXT item are requested at the
//                                           K -> V change of the current one and its first K group right after the
//                                           softmax (third register buffer), so a worker's page stream has no bubble
//                                           at an item boundary.
//                                           Solo workers take their FIRST item in index order (requested at the top of
//                                           the kernel) while a retired wave ranks the rest; 4-wave TEAMS (heavy-tailed
//                                           and bimodal batches) rank everything up front, run the LONG items (more than
//                                           a quarter of the longest) four waves to an item and hand the SHORT ones out
//                                           in quads, one per wave (round 3).  All of it is described where it is coded.
//
// Mode S is entered before any of mode Q's preparation code (see the end of the kernel): what lies between the top of
// the kernel and a mode's first page request is paid by every launch (profiles/r02m_late_ranking.md).
//
// Tried and dropped (profiles/r02a_queue_probe_cfg3_ticket_vs_static.log): handing items out through ONE device-scope
// ticket counter (atomic add per item).  cfg3 ragged: 105 us against 73 us for the static snake, uniform 153 against
// 129 — a single word sustains ~88 atomics/us (MI355X_MICROARCH.md, "dequeue"), 3072 pulls are 35 us of it, and vmcnt
// retires in order, so a slow atomic also blocks every page group requested behind it.
//
// Arithmetic: an item is computed by ONE wave exactly as pa_v1_kernel<D, *, 1, 1, ...> computes it — K lane map
// lane = chunk*16 + token, dot8 chains + xor butterfly, fp32 softmax, probabilities rounded once to fp16/bf16,
// PV8 per 8-token group, fp32 accumulation over the blocks from the LAST block down to the first.  That order does
// not depend on the group size, so S and Q results are bit-identical to each other and to the u1 kernels: what a
// sequence gets never depends on its batch neighbours (tests/test_parity_gpu.py::test_queue_kernel_*).
// Reference rounding points: attention_kernels.cu:115-136 (context bounds), 302-305 (masked logits), 334-342
// (softmax), 398-430 (fp16 p, zeroed tail of V), dtype_float16.cuh:252-260, 439-457 (packed products and sums).
//
// No state outside the launch: no counters, no workspace, nothing to reset — safe under hipGraph replay and from any
// number of streams and host threads.
#pragma once

#include "pa_kernel.hpp"

namespace vmi {

constexpr int QSORT_MAX = 2048;  // sequences ranked in LDS (2 B each); larger batches are served in index order
constexpr int QLATE_MAX = 512;   // ... ranked by ONE retired wave beside the workers' first items (else by all four, up front)

// q_flags (PAParams): experiment / test knobs; 0 = automatic
//   bits 0-1  mode      0 auto, 1 force S (when every item has a wave), 2 force Q
//   bits 2-4  WQ        workers per workgroup in mode Q (0 -> 2)
//   bit  10   teams only: no solo quads for the short items of a heavy-tailed batch
//   bit  11   no ranking (index order)
//   bits 12-13 team     0 auto, 1 force solo workers, 2 force teams (mode Q only)
//   bit  14   no statistics (with a forced mode)      bit 15   rank everything up front (solo workers too)
constexpr int QF_MODE(int f) { return f & 3; }
constexpr int QF_TEAM(int f) { return (f >> 12) & 3; }
constexpr int QF_WQ(int f) { return (f >> 2) & 7; }
constexpr int QF_NOHYBRID = 1 << 10;  // teams take EVERY item (round 2's team mode), short ones included
constexpr int QF_NOSORT = 1 << 11;
constexpr int QF_EARLYSORT = 1 << 15;  // rank every sequence before the first item (no first round in index order)
constexpr int QF_NOSTATS = 1 << 14;  // experiment: do not read the lengths at all (with a forced mode)

// grid = (G), block = 256.  LDS = 4*lpad*4 (logits / probabilities, one region per wave) + QSORT_MAX*2 (ranking)
//                                 + QSORT_MAX*2 (per-chunk bucket counts of the counting sort) + QSORT_MAX*2 (the
//                                 clamped lengths) + 4*64*8 (bucket masks)
//                                 + 8*4 + 4*D*4 (a team's max / sum exchange and partial outputs).
// KM: q.K^T of the K pass on the matrix cores.  The K tile as loaded IS the B operand of v_mfma_f32_16x16x32_f16 (lane =
//     chunk*16 + token holds 8 dims of one token = B[k = 8*chunk ..][n = token]); A is q with every row the same (lane
//     (chunk, m) holds the q dims of k-group `chunk`), so every row of the 16x16 result is the block's 16 logits and
//     register 0 of lane l is the logit of token l & 15 — exactly what the fp32 FMA chain + xor butterfly leaves there.
//     Products of 16-bit operands are exact in fp32 and the accumulation is fp32: the reference's arithmetic up to
//     summation order (dtype_float16.cuh:292-298); the V pass keeps its fp16 rounding points on the VALU.  Built for the
//     fp8 kernels, whose K pass is VALU-pressed (16 decodes + 8 packed FMAs + the butterfly per 16 dims): the decode to
//     half pairs feeds the MFMA directly.  M = 1 of 16 rows is useful work — the matrix pipe is idle otherwise.
// (diagnostic library only — this header is compiled once more with -DVMI_DIAG for it: the kernel's body becomes a function
//  and the kernel a wrapper that writes each wave's start and end time to g_wave_timeline, see the end of this file; with
//  VMI_DIAG undefined the text of the kernel is exactly what it was)
#ifdef VMI_DIAG
extern __device__ uint64_t* g_wave_timeline;  // [gridDim.x * 4][4]: start, end (100 MHz ticks), HW_ID, XCC_ID; nullptr = off
#endif
template <int D, bool BF, bool NT, int US, int UQ, int F8 = 0, bool KM = false, int UT = 1, bool APP = false>
// (second launch bound = minimum waves per SIMD: 3 workgroups of 4 waves per CU must all be resident in mode S, i.e.
//  <= 168 VGPRs; head size 128 — twice the registers per block — runs 2 workgroups per CU, 256 VGPRs)
#ifdef VMI_DIAG
__device__ __forceinline__ void pa_q_body(const PAParams& p) {
#else
__global__ void __launch_bounds__(256, (US >= 4 || (UQ >= 4 && !F8) || D > 64) ? 2 : 3) pa_q_kernel(const PAParams p) {
#endif
  constexpr int BS = 16;
  // F8: the pages hold fp8 bytes (1 = E4M3, 2 = E5M2; kv_scale == 1 only — the launcher sends any other scale to
  // pa_v1_kernel): a 16-byte unit carries 16 elements, a (block, head) tile is D*16 BYTES; every element becomes
  // half(float(fp8)) first (reference quant_utils.cuh:295-300) — with kv_scale 1 that is the byte's own value
  constexpr bool E5 = F8 == 2;
  constexpr int EPU = F8 ? 16 : 8;       // cache elements per 16-byte unit
  constexpr int ES = F8 ? 1 : 2;         // bytes per cache element
  constexpr int NL = D * BS / EPU / 64;  // 1-KiB loads per (block, head) tile of K — and of V
  static_assert((D * BS / EPU) % 64 == 0 && NL >= 1, "a tile must fill whole 1-KiB loads");
  static_assert(!(F8 && BF), "fp8 pages: float16 query only in these kernels");
  static_assert(!(KM && BF), "KM is built for float16 operands");
  static_assert(!(APP && (F8 || KM)), "the fused append is built for 16-bit pages");
  constexpr int UPR = BS / EPU;  // V: 16-B units per dim row
  constexpr int RPL = 64 / UPR;  // V: rows per load
  constexpr int QW = F8 ? 2 : 1;  // 16-byte pieces of q facing one K unit

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* smem_f = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* logits = smem_f + (size_t)wave * p.lpad;
  uint16_t* ph = reinterpret_cast<uint16_t*>(logits);  // probabilities, in place over the consumed fp32 values
  uint16_t* order = reinterpret_cast<uint16_t*>(smem_f + (size_t)4 * p.lpad);

  const int B = p.num_seqs, H = p.num_heads;
  const int N = B * H;                    // items
  const int nwaves = gridDim.x * 4;
  const int w_nat = blockIdx.x * 4 + wave;  // this wave's item in mode S
  const int qpk = H / p.num_kv_heads;

  // ---- everything a wave needs to know about an item before its first page can be requested -------------------
  struct Meta {
    int seq, head, L;    // wave-uniform
    int32_t bt;          // lane j: physical id of my block j (first 64 of them; my blocks are sub + j*T)
    u32x4 q[NL][QW];     // this lane's EPU dims of q facing each K load
    float slope;
  };
  const int c4 = lane >> 4;  // K: chunk within a load
  const int tk = lane & 15;  // K: token within the block
  auto meta_issue = [&](Meta& m, int seq, int head, int T, int sub) {
    m.seq = seq;
    m.head = head;
    const int32_t* bt = p.block_tables + (int64_t)seq * p.max_blocks_per_seq;
    m.bt = sub + lane * T < p.max_blocks_per_seq ? bt[sub + lane * T] : 0;
    m.L = p.seq_lens[seq];
    const h16* qp = p.q + (int64_t)seq * p.q_stride + (int64_t)head * D;
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
      for (int w = 0; w < QW; ++w) m.q[i][w] = *reinterpret_cast<const u32x4*>(qp + (4 * i + c4) * EPU + 8 * w);
    m.slope = p.alibi ? p.alibi[head] : 0.f;
  };

  // ---- mode: every wave derives it from the same lengths ---------------------------------------------------------
  // (the natural item's metadata is requested first, so in mode S the decision costs no extra round trip)
  Meta cur;
  const bool nat_ok = w_nat < N;
  {
    const int s0 = nat_ok ? w_nat / H : 0;
    meta_issue(cur, s0, nat_ok ? w_nat - s0 * H : 0, 1, 0);
  }
  // ---- APP ("append-read": vmi_paged_attention_v1_newest_f16 on a full chip).  The attention takes the token at position
  //      L-1 from this step's key / value rows (PAParams key / value) instead of the cache: an item's K tile of block
  //      (L-1)/16 is loaded with the lanes of token (L-1)%16 pointed at the key row (no register, no extra request), its V
  //      tile gets the token's element of each dim row patched in before the first V product.  Whatever the page holds in
  //      that slot — old bytes, or the row if it was stored already — is never used, so `out` is bit-identical to
  //      reshape_and_cache + paged_attention_v1 and the cache write is free to happen LATER: the caller stores a token's
  //      rows of ALL layers with one reshape_and_cache (GPT2PagedDecoder(deferred_scatter=True)).  The patch costs
  //      nothing: cfg3 120.7 us with it, 121.3 without (profiles/r06_append_read.md).
  //      The WRITE was tried inside this kernel and is not built: stores at the START of the launch (requested with the
  //      first metadata, issued behind the first page request) +20 us — 73 728 partial lines leave L2 into the middle of
  //      the read stream; as each wave's LAST act +7 us on equal lengths and +9 on ragged ones — more than the stand-alone
  //      reshape_and_cache launch costs in front of the attention (6.1 us); on a second stream beside the attention the
  //      pair runs 140 us instead of 127 (cross-stream dependencies: scripts/overlap_scatter_probe.py).  The writing
  //      entry (vmi_paged_attention_v1_append_f16) therefore stays with pa_v1_kernel's whole-tile stores (+3 us). ----
  const int flags = p.q_flags;
  // ... and so is the item that would be this wave's FIRST as a solo worker of mode Q: item <worker index>, in INDEX order
  // (see "first round" below) — two waves per workgroup ask for a table slice and a q they may not need.
  // Solo workers per workgroup: 2 over 16-bit pages (each with two blocks per register group: the bytes in flight of mode S
  // with half the waves) — 4 over fp8 pages, whose half-size tiles want more requests in flight: wherever the kernel
  // chooses solo workers, four of them are 4 - 13 % faster there (cfg3 fp8 U{1..L} 42.7 -> 40.2 us, batch 512 82.1 -> 72.0,
  // cfg4 fp8 202 -> 180; over fp16 pages 67.0 -> 70.9: profiles/r03x_fp8_four_solo_workers.md).
  // Head size 128 over 16-bit pages (two workgroups per CU: two workers each would be four waves per CU) likewise: cfg4
  // U{1..L} 358.5 -> 351.9 us, batch 256 715 -> 677, U[1/8..1] 369 -> 353 / 742 -> 704.
  constexpr int WQ_SOLO = (F8 || D > 64) ? 4 : 2;
  const int WQd = QF_WQ(flags) ? QF_WQ(flags) : WQ_SOLO;
  const int wq_solo = blockIdx.x * WQd + wave;
  const int sq0 = wq_solo < N ? wq_solo / H : 0;          // its sequence and head
  const int hq0 = wq_solo < N ? wq_solo - sq0 * H : 0;
  Meta firstq;
  if (wave < WQd && ((F8 || D > 64) ? WQd < 4 : true) && !(flags & QF_EARLYSORT)) meta_issue(firstq, sq0, hq0, 1, 0);  // (4 workers: no late ranking)
  bool queue = N > nwaves;
  int maxL = 0;
  float sumL = 0.f;
  bool have_sum = false;
  bool ragged = false;
  int nlong_est = 0;
  // Statistics: every wave reads ALL the lengths once (8 loads in flight per trip of the loop) and leaves them, clamped,
  // in LDS for the counting sort below — the passes of the sort must not go back to memory (three dependent round
  // trips in front of the first page).
  uint16_t* len16 = order + 2 * QSORT_MAX;
  const bool rankable = B <= QSORT_MAX && !(flags & QF_NOSORT) && !(flags & QF_NOSTATS);
  if ((rankable || !queue || (flags & QF_GATE_RAGGED)) && !(flags & QF_NOSTATS)) {
    float sum;
    batch_stats(p.seq_lens, B, p.lpad, lane, maxL, sum, [&](int i, int c) {
      if (rankable) len16[i] = (uint16_t)c;  // (all four waves write the same values)
    });
    sumL = sum;
    have_sum = true;
    ragged = batch_is_ragged(maxL, sum, B);
    if ((flags & QF_GATE_RAGGED) && !ragged) return;  // gated double launch: the kernel in front of me did this batch
    queue = queue || ragged;
    if (rankable && ragged) {  // how many sequences are LONG (more than a quarter of the longest)?  From my own LDS copy
      for (int c = 0; c < B; c += 64)  // of the lengths: a wave reads what it wrote itself, no barrier
        nlong_est += (int)__popcll(__ballot(c + lane < B && 4 * (int)len16[c + lane] > maxL));
    }
  }
  if (QF_MODE(flags) == 1 && N <= nwaves) queue = false;
  if (QF_MODE(flags) == 2) queue = true;
  queue = __builtin_amdgcn_readfirstlane(queue);
  // Mode Q, solo workers or teams?  A worker that runs its items alone needs (longest item) <= (its share of the
  // batch) to finish with the others; when a few sequences are much longer than the rest — the usual shape of a
  // serving batch — the four waves of a workgroup work on ONE item together instead (blocks dealt round-robin, three
  // LDS barriers per item), which makes the longest item four times shorter.
  // Round 3: the same when the batch is BIMODAL — at most 70 % of the sequences are long (more than a quarter of the
  // longest).  Solo workers then hold a few long items each, and a worker that happens to get one more than its
  // neighbours finishes that much later (half the sequences full, half 1/16: 94 us with solo workers against 81 us;
  // cfg4 413 against 383), whatever the ratio to its share says.  Continuous spreads (U{1..L}: 75 % long) stay with solo
  // workers, whose snake pairs a long item with a short one (67 us against 74).  profiles/r03c_heavy_tailed_batches.md
  bool team = false;
  if (queue && have_sum) {
    // (`share` is counted in TWO-worker shares also where four solo workers run (fp8 pages, head size 128) — deliberately, not
    //  by oversight: it is the calibration of the 1.15 criterion, not a model of the schedule.  With the real worker count the
    //  share halves and a plain U{1..L} batch (longest = 2 x mean) would be sent to teams, which is 25 % SLOWER there over fp8
    //  pages: 51.2 against 41.0 us, profiles/r04_fp8_ragged_accounting.md.  Likewise nlong_est (4 * len > maxL) and the sort's
    //  long / short split (buckets 0..47 of 64, scaled by maxL + 1) may differ by the sequences within 1/64 of maxL / 4 of the
    //  boundary: the gate only asks "are at most 70 % long", the split decides who is — neither affects results.)
    const int wq0 = QF_WQ(flags) ? QF_WQ(flags) : 2;
    const float share = sumL * (float)H / (float)(gridDim.x * wq0);  // tokens per solo worker
    team = (float)maxL > 1.15f * share || (ragged && rankable && nlong_est * 10 <= 7 * B);
  }
  if (QF_TEAM(flags) == 1) team = false;
  if (QF_TEAM(flags) == 2) team = queue;
  team = __builtin_amdgcn_readfirstlane(team);
  const int WQ = team ? 4 : (QF_WQ(flags) ? QF_WQ(flags) : WQ_SOLO);
  const bool ranked = queue && rankable;
  const int nworkers = queue ? (team ? gridDim.x : gridDim.x * WQ) : nwaves;
  const int wq = queue ? (team ? blockIdx.x : blockIdx.x * WQ + wave) : w_nat;  // worker index

  // ---- rank the sequences from index `lo` on, longest first: a counting sort with 64 length buckets (bucket k in lane
  //      k), index order inside a bucket — deterministic, so every workgroup computes the same table for itself.
  //      `nsw` waves share the 64-sequence chunks (this one is number `me` of them): all four with a workgroup barrier
  //      between the passes, or ONE wave on its own, without any barrier.  B <= QSORT_MAX. ----
  //      Returns how many of the ranked sequences are LONG — longer than a quarter of the longest (buckets 0..47) — i.e.
  //      the rank at which the short ones start (wave-uniform, the same in every wave that calls this).
  auto rank_sequences = [&](int lo, int me, int nsw) -> int {
    uint16_t* cnt = order + QSORT_MAX;  // [chunk][bucket]: sequences of the chunk in the bucket
    const int nch = (B + 63) >> 6;
    const float bscale = 64.f / (float)(maxL + 1);
    auto bucket_of = [&](int i) -> int {  // bucket of sequence i (64 = not ranked: before lo or past the end of the batch)
      if (i >= B || i < lo) return 64;
      const int bk = (int)((float)(maxL - (int)len16[i]) * bscale);  // 0 = longest
      return bk > 63 ? 63 : bk;
    };
    // Per chunk: which lanes share my bucket (a 64-bit mask per bucket, built with ONE LDS atomic OR — the result of
    // an OR does not depend on the order the lanes are served in) and how many sequences each bucket holds.
    uint64_t* bm = reinterpret_cast<uint64_t*>(order + 3 * QSORT_MAX) + wave * 64;  // this wave's 64 masks
    auto masks_of = [&](int bk) {
      bm[lane] = 0;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (DS operations of one wave execute in order)
      if (bk < 64) __hip_atomic_fetch_or(&bm[bk], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    };
    for (int c = me; c < nch; c += nsw) {
      masks_of(bucket_of(c * 64 + lane));
      cnt[c * 64 + lane] = (uint16_t)__popcll(bm[lane]);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    if (nsw > 1) __syncthreads();
    int tot = 0;
    for (int c = 0; c < nch; ++c) tot += cnt[c * 64 + lane];
    int incl = tot;  // inclusive scan over the buckets
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      incl += lane >= d ? o : 0;
    }
    int run = incl - tot;  // lane k: rank of the first sequence of bucket k in the chunk at hand
    for (int c = 0; c < nch; ++c) {
      if (c % nsw == me) {
        const int bk = bucket_of(c * 64 + lane);
        masks_of(bk);
        const uint64_t same = bm[bk & 63];
        const int pos = __shfl(run, bk & 63) + __popcll(same & ((1ull << lane) - 1ull));
        if (bk < 64) order[pos] = (uint16_t)(c * 64 + lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
      run += cnt[c * 64 + lane];
    }
    return __builtin_amdgcn_readlane(incl, 47);
  };

  // ---- solo workers, FIRST ROUND IN INDEX ORDER ("late" ranking).  Ranking everything first puts the sort and one more
  //      round trip (the ranked first item's table slice and q) in front of a worker's first page.  Instead worker w
  //      starts with item w as the batch lists them — requested above, before the lengths were even read — and the
  //      sequences the first round does NOT cover (index >= R0) are ranked meanwhile by one of the waves that retire in
  //      this mode; the workers look at its table for the first time when their first item's K pass is over (a flag in
  //      LDS says it is complete).  What the snake needs from the first round is only each worker's place among its
  //      peers: v = (how many first-round sequences are longer than mine) * H + head.  Round k of the rest then goes by
  //      v exactly as the snake goes by w: the worker that started with the longest item gets the shortest of the next
  //      W, and so on.  Needs every first-round sequence to be wholly inside the round (W a multiple of H).
  //      Teams keep the full ranking up front: tried with the team ranking the rest itself under its first pages'
  //      flight, bimodal batches gained 1.5-2 us and exponential / lognormal ones lost 1-2 (r02m_late_ranking.md). ----
  const int R0 = nworkers / H;
  const bool late = queue && !team && ranked && WQ < 4 && nworkers % H == 0 && B <= QLATE_MAX && !(flags & QF_EARLYSORT);
  const bool more = N > nworkers;  // there are rounds after the first
  volatile uint32_t* sorted = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<uint64_t*>(order + 3 * QSORT_MAX) + 4 * 64);
  int vrank = 0;

  auto seq_of_rank = [&](int r) -> int {  // wave-uniform
    return ranked ? (int)__builtin_amdgcn_readfirstlane((int)order[r]) : r;
  };
  // item t (0 <= t < N) in hand-out order -> (seq, head): the H heads of a rank are consecutive items
  auto ids_of = [&](int t, int& seq, int& head) {
    const int r = t / H;
    head = t - r * H;
    seq = seq_of_rank(r);
  };

  const int hf = lane % UPR;    // V: which 16-byte unit of the dim row this lane owns
  const int rowl = lane / UPR;  // V: dim row within a load

  // LDS of a team: ONE logits array for the item (region 0), the probabilities behind it (region 1: in place would
  // overwrite slots another wave has not read yet), the max / sum exchange and the waves' partial outputs
  float* red = reinterpret_cast<float*>(reinterpret_cast<uint64_t*>(order + 3 * QSORT_MAX) + 4 * 64);  // [8]
  float* osm = red + 8;                                                                                  // [4][D]

  // QMODE = false is mode S: one item, nothing to hand out or to prefetch — compiled without any of that.
  // `handout(k, seq, head)` names this worker's k-th item (k = 1, 2, ... after `first`) or returns false (wave-uniform).
  auto run = [&](auto utag, auto teamtag, auto qtag, const Meta& first, auto&& handout) {
    constexpr int UU = decltype(utag)::value;
    constexpr bool TEAM = decltype(teamtag)::value;
    constexpr bool QMODE = decltype(qtag)::value;
    constexpr int T = TEAM ? 4 : 1;       // waves per item; my blocks are sub + idx*T
    const int sub = TEAM ? wave : 0;
    float* lg = TEAM ? smem_f : logits;
    uint16_t* pr = TEAM ? reinterpret_cast<uint16_t*>(smem_f + p.lpad) : ph;
    u32x4 rn[UU][NL], ra[UU][NL], rb[UU][NL];

    // per-item state (wave-uniform unless noted)
    int L = 0, nblk = 0, nmy = 0;  // sequence length, its blocks, MY blocks
    int32_t bt_reg = 0;  // lane j: physical id of my block bt_sg*64 + j
    int bt_sg = 0;
    const int32_t* bt = nullptr;
    int64_t hoff = 0;  // this lane's BYTE offset inside a block: kv head tile + its 16-byte unit
    u32x4 qreg[NL][QW];
    f32x2_t qf[(F8 && !KM) ? NL : 1][8];  // fp8 pages: q as fp32 pairs (v_pk_fma_f32 against the decoded bytes)
    float slope = 0.f;
    uint16_t* outp = nullptr;
    // APP: where the item's newest token lives (block lbA, offset offA; lbA = -1: nowhere) and the rows it comes from
    int lbA = -1, offA = 0, aseq = 0, akvh = 0;
    uint32_t vnew[APP ? NL : 1];
    auto key_row_of = [&](int seq_, int kvh_) -> const char* {  // this lane's 16-B chunk of load 0 (load i: + 64 * i bytes)
      return reinterpret_cast<const char*>(p.key + (int64_t)seq_ * p.key_stride + (int64_t)kvh_ * D) + c4 * 16;
    };

    auto adopt = [&](const Meta& m) {  // make m the current item
      L = __builtin_amdgcn_readfirstlane(m.L);
      if constexpr (APP) {
        lbA = L >= 1 ? (L - 1) / BS : -1;  // (of the FULL length: a truncated item never meets its newest token)
        offA = L >= 1 ? (L - 1) % BS : 0;
        aseq = m.seq;
        akvh = m.head / qpk;
      }
      L = L > p.lpad ? p.lpad : L;  // seq_len > max_seq_len: truncated to the LDS that was reserved
      nblk = (L + BS - 1) / BS;
      nmy = nblk > sub ? (nblk - sub + T - 1) / T : 0;
      bt = p.block_tables + (int64_t)m.seq * p.max_blocks_per_seq;
      bt_reg = m.bt;
      bt_sg = 0;
      hoff = ((int64_t)(m.head / qpk) * p.kv_head_stride + lane * EPU) * ES;
#pragma unroll
      for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int w = 0; w < QW; ++w) qreg[i][w] = m.q[i][w];
      if constexpr (F8 && !KM) {
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const h16x8 qh = __builtin_bit_cast(h16x8, qreg[i][w]);
#pragma unroll
            for (int e = 0; e < 4; ++e) qf[i][4 * w + e] = f32x2_t{(float)qh[2 * e], (float)qh[2 * e + 1]};
          }
      }
      slope = m.slope;
      outp = reinterpret_cast<uint16_t*>(p.out) + ((int64_t)m.seq * H + m.head) * D;
    };
    auto table_for = [&](int g) {
      const int sg = (g * UU) >> 6;
      if (sg != bt_sg) {  // once per 64 of my blocks
        const int b = sub + (sg * 64 + lane) * T;
        bt_reg = b < p.max_blocks_per_seq ? bt[b] : 0;
        bt_sg = sg;
        // The wait for this load belongs INSIDE the branch.  Left to the compiler it lands at the join in front of
        // the v_readlane — as s_waitcnt vmcnt(0) on EVERY page group, which drains the group in flight before the next
        // one is requested, i.e. no register double buffer at all.
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bt_reg));
      }
    };
    static_assert(64 % UU == 0, "a register group must not straddle two table slices");
    auto load_group = [&](u32x4(&r)[UU][NL], auto ktag, int g) {  // ktag: std::true_type = the K cache, false_type = V
      constexpr bool ISK = decltype(ktag)::value;
      const char* cache = reinterpret_cast<const char*>(ISK ? p.kc : p.vc);
      table_for(g);
#pragma unroll
      for (int j = 0; j < UU; ++j) {
        int idx = g * UU + j;
        idx = idx < nmy ? idx : nmy - 1;  // padding slots re-read my last block (never out of bounds)
        const int64_t phys = __builtin_amdgcn_readlane(bt_reg, idx & 63);
        const char* blk = cache + phys * p.kv_block_stride * ES + hoff;
        if constexpr (APP && ISK) {
          if (sub + idx * T == lbA) {  // wave-uniform: the block of the newest token — its lanes read the key row
            const char* kr = key_row_of(aseq, akvh);
#pragma unroll
            for (int i = 0; i < NL; ++i) r[j][i] = ld16<NT>(tk == offA ? kr + i * 64 : blk + i * 1024);
            continue;
          }
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) r[j][i] = ld16<NT>(blk + i * 1024);
      }
    };
    constexpr std::true_type KC{};
    constexpr std::false_type VC{};

    float qk_max;
    auto compute_k = [&](u32x4(&r)[UU][NL], int g) {
#pragma unroll
      for (int j = 0; j < UU; ++j) {
        const int idx = g * UU + j;
        if (idx < nmy) {  // wave-uniform
          const int token = (sub + idx * T) * BS + tk;
          const bool masked = token >= L;
          float acc;
          if constexpr (KM) {
            f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NL; ++i)
#pragma unroll
              for (int w = 0; w < QW; ++w) {
                u32x4 kb = r[j][i];
                if constexpr (F8) kb = deq8<true, false, E5>(r[j][i][2 * w], r[j][i][2 * w + 1], 1.f);
                d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, qreg[i][w]),
                                                            __builtin_bit_cast(h16x8, kb), d4, 0, 0, 0);
              }
            acc = d4[0];  // (all 16 rows are the same product: any register of any lane group holds token lane & 15)
          } else {
            float accv[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) {
              if constexpr (F8) accv[i] = dot16_f8_s1<E5>(qf[i], r[j][i]);
              else accv[i] = dot8<BF>(qreg[i][0], r[j][i]);
            }
            acc = accv[0];
#pragma unroll
            for (int i = 1; i < NL; ++i) acc += accv[i];
            acc += __shfl_xor(acc, 16);
            acc += __shfl_xor(acc, 32);
          }
          float qk = p.scale * acc;
          qk += (slope != 0.f) ? slope * (float)(token - L + 1) : 0.f;
          if (lane < BS) lg[token] = masked ? 0.f : qk;
          qk_max = masked ? qk_max : fmaxf(qk_max, qk);
        }
      }
    };

    float acc[NL];
    auto compute_v = [&](auto masked, u32x4(&r)[UU][NL], int g) {
      constexpr bool MASK = decltype(masked)::value;
#pragma unroll
      for (int jj = 0; jj < UU; ++jj) {
        const int j = UU - 1 - jj;  // blocks in descending order whatever the group size
        const int idx = g * UU + j;
        if (idx < nmy) {
          const int b = sub + idx * T;
          const int token0 = b * BS + hf * EPU;
          const bool last = (b == nblk - 1);
          if constexpr (APP && MASK) {  // the newest token lives in the sequence's last block: final group only
            if (b == lbA && hf == (offA >> 3)) {
              const int e = offA & 7;
#pragma unroll
              for (int i = 0; i < NL; ++i)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                  const uint32_t old = r[j][i][w];
                  const uint32_t patched = (e & 1) ? ((old & 0x0000ffffu) | (vnew[i] << 16)) : ((old & 0xffff0000u) | vnew[i]);
                  r[j][i][w] = ((e >> 1) == w) ? patched : old;
                }
            }
          }
          PV8<BF> pv;
          pv.load(*reinterpret_cast<const u32x4_alias*>(pr + token0));
          if constexpr (F8) {  // a unit is 16 tokens: two 8-token groups, each with its own probability vector
            PV8<BF> pw;
            pw.load(*reinterpret_cast<const u32x4_alias*>(pr + token0 + 8));
#pragma unroll
            for (int i = 0; i < NL; ++i)
              acc[i] += pv.template dot<MASK>(deq8<true, false, E5>(r[j][i][0], r[j][i][1], 1.f), last, token0, L);
#pragma unroll
            for (int i = 0; i < NL; ++i)
              acc[i] += pw.template dot<MASK>(deq8<true, false, E5>(r[j][i][2], r[j][i][3], 1.f), last, token0 + 8, L);
          } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) acc[i] += pv.template dot<MASK>(r[j][i], last, token0, L);
          }
        }
      }
    };

    // the first K group of the NEXT item (issued as soon as the current item's first V group has been consumed)
    auto prefetch_next = [&](const Meta& m, bool has) {
      if (!has) return;
      int l2 = m.L;
      l2 = l2 > p.lpad ? p.lpad : l2;
      l2 = __builtin_amdgcn_readfirstlane(l2);
      const int nb2 = (l2 + BS - 1) / BS;
      const int nmy2 = nb2 > sub ? (nb2 - sub + T - 1) / T : 0;
      if (nmy2 <= 0) return;
      const int64_t hoff2 = ((int64_t)(m.head / qpk) * p.kv_head_stride + lane * EPU) * ES;
      const int lf2 = __builtin_amdgcn_readfirstlane(m.L);
      const int lbA2 = (APP && lf2 >= 1) ? (lf2 - 1) / BS : -1, offA2 = (lf2 - 1) & (BS - 1);
#pragma unroll
      for (int j = 0; j < UU; ++j) {
        const int idx = j < nmy2 ? j : nmy2 - 1;
        const int64_t phys = __builtin_amdgcn_readlane(m.bt, idx);
        const char* blk = reinterpret_cast<const char*>(p.kc) + phys * p.kv_block_stride * ES + hoff2;
        if constexpr (APP) {
          if (sub + idx * T == lbA2) {
            const char* kr = key_row_of(m.seq, m.head / qpk);
#pragma unroll
            for (int i = 0; i < NL; ++i) rn[j][i] = ld16<NT>(tk == offA2 ? kr + i * 64 : blk + i * 1024);
            continue;
          }
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) rn[j][i] = ld16<NT>(blk + i * 1024);
      }
    };

    // ---- first item: its metadata was requested above ----
    adopt(first);
    if (nmy > 0) load_group(rn, KC, 0);
    int round = 0;  // items this worker has finished

    for (;;) {
      Meta nxt;  // (deliberately uninitialised: a default value would become a phi, i.e. register copies that wait
                 //  for the loads the moment they are issued)
      bool has_next = false;
      auto fetch_next = [&]() {  // wave-uniform decision + the requests for the next item's metadata
        if constexpr (QMODE) {
          int s, h;
          has_next = handout(round + 1, s, h);
          meta_issue(nxt, s, h, T, sub);  // requested unconditionally (item 0 when there is no next one): no phi
        }
      };

      if (L <= 0) {  // reference: exp_sum = 0 -> every output 0  (the whole team takes this branch together)
        if (sub == 0)
          for (int d = lane; d < D; d += 64) outp[d] = 0;
        fetch_next();
        if (!has_next) break;
        if constexpr (QMODE) {
          adopt(nxt);
          if (nmy > 0) load_group(rn, KC, 0);
          ++round;
        }
        continue;
      }

      const int ngroups = (nmy + UU - 1) / UU;  // 0 for a team wave without blocks
      const int lastg = ngroups - 1;
      qk_max = -FLT_MAX;
#pragma unroll
      for (int i = 0; i < NL; ++i) acc[i] = 0.f;

      // =========================== K pass: logits -> LDS, running max ========================
      // group 0 is in rn (requested during the previous item); the rest ping-pongs between ra and rb
      if (ngroups == 1) {
        load_group(ra, VC, 0);  // one group in all: K and V cost one round trip between them
        compute_k(rn, 0);
      } else if (ngroups > 1) {
        // The V pass starts BEFORE the K pass ends: its first group (the LAST one: the block reshape_and_cache has just
        // written is requested early, profiles/r01o_call_pair_gap.md) goes into rn — free since group 0 was consumed —
        // ahead of the final K computation, the second one right behind it, so two groups stay in flight across the
        // K -> V change and the softmax instead of the queue running empty there.
        load_group(ra, KC, 1);
        compute_k(rn, 0);
        // (written out as a two-step loop plus its two possible tails: folding the tails into the loop with
        //  conditional loads was tried for code size and made the register allocator spill 600 bytes per lane)
        int g = 1;
        for (; g + 2 < ngroups; g += 2) {
          load_group(rb, KC, g + 1);
          compute_k(ra, g);
          load_group(ra, KC, g + 2);
          compute_k(rb, g + 1);
        }
        if (g + 2 == ngroups) {
          load_group(rb, KC, g + 1);
          compute_k(ra, g);
          load_group(rn, VC, lastg);
          compute_k(rb, g + 1);
        } else {
          load_group(rn, VC, lastg);
          compute_k(ra, g);
        }
        load_group(ra, VC, lastg - 1);
      }
      fetch_next();  // the next item's table slice, length and q are requested
      if constexpr (APP) {  // ... and the newest token's V elements of my dim rows, by the wave that owns its block
        if (lbA >= 0 && lbA < nblk && (lbA % T) == sub) {
          const h16* vr = p.value + (int64_t)aseq * p.value_stride + (int64_t)akvh * D;
#pragma unroll
          for (int i = 0; i < NL; ++i) vnew[i] = (uint32_t)__builtin_bit_cast(uint16_t, vr[RPL * i + rowl]);
        }
      }

      // =========================== softmax over the logits in LDS ============================
      float inv_sum;
      if constexpr (TEAM) {
        // every wave exponentiates and sums the tokens of ITS blocks (no wave reads another's logits); the maxima
        // and the sums meet in LDS
        float m = wave_max(qk_max);
        if (lane == 0) red[sub] = m;
        lds_barrier();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float e_sum = 0.f;
        for (int t = lane; t < nmy * BS; t += 64) {
          const int i = (sub + (t >> 4) * T) * BS + (t & 15);
          if (i < L) {
            const float e = __expf(lg[i] - m);
            lg[i] = e;
            e_sum += e;
          }
        }
        e_sum = wave_sum(e_sum);
        if (lane == 0) red[4 + sub] = e_sum;
        lds_barrier();
        inv_sum = __builtin_amdgcn_rcpf((((red[4] + red[5]) + red[6]) + red[7]) + 1e-6f);
        for (int t = lane; t < nmy * BS; t += 64) {
          const int i = (sub + (t >> 4) * T) * BS + (t & 15);
          pr[i] = i < L ? to_elem<BF>(lg[i] * inv_sum) : (uint16_t)0;
        }
      } else {
        const float m = wave_max(qk_max);
        float e_sum = 0.f;
        for (int i = lane; i < L; i += 64) {
          const float e = __expf(lg[i] - m);
          lg[i] = e;
          e_sum += e;
        }
        inv_sum = __builtin_amdgcn_rcpf(wave_sum(e_sum) + 1e-6f);
        // p = exp * inv_sum -> fp16 (bf16), once per token, in place: the 64 lanes read fp32 values [t0, t0+64) and
        // then write bytes [2*t0, 2*t0+128), i.e. fp32 slots [t0/2, t0/2+32) — already consumed, or read by this access
        for (int t = lane; t < nblk * BS; t += 64) {
          const float e = lg[t];
          pr[t] = t < L ? to_elem<BF>(e * inv_sum) : (uint16_t)0;
        }
      }
      // The next item's metadata is waited for HERE, where only the first V groups (needed next anyway) are in flight
      // with it: no later use of it can then make the compiler drain the V stream.
      if constexpr (QMODE) {
        asm volatile("" : "+v"(nxt.bt), "+v"(nxt.L), "+v"(nxt.slope));
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
          for (int w = 0; w < QW; ++w) asm volatile("" : "+v"(nxt.q[i][w]));
      }

      // =========================== V pass, last group first ===================================
      if (ngroups == 1) {
        compute_v(std::true_type{}, ra, 0);
        if constexpr (QMODE) prefetch_next(nxt, has_next);
      } else if (ngroups > 1) {
        compute_v(std::true_type{}, rn, lastg);  // the only group that can hold the sequence's last block
        // rn is free again: the next item's first K group goes out now and has the rest of the V pass to arrive
        if constexpr (QMODE) prefetch_next(nxt, has_next);
        int s = 1;  // step s = group lastg - s; odd steps in ra, even steps in rb
        for (; s + 2 < ngroups; s += 2) {
          load_group(rb, VC, lastg - (s + 1));
          compute_v(std::false_type{}, ra, lastg - s);
          load_group(ra, VC, lastg - (s + 2));
          compute_v(std::false_type{}, rb, lastg - (s + 1));
        }
        if (s + 2 == ngroups) {
          load_group(rb, VC, lastg - (s + 1));
          compute_v(std::false_type{}, ra, lastg - s);
          compute_v(std::false_type{}, rb, lastg - (s + 1));
        } else {
          compute_v(std::false_type{}, ra, lastg - s);
        }
      } else {
        if constexpr (QMODE) prefetch_next(nxt, has_next);  // a team wave without blocks in this item may have some in the next
      }

      // the two lanes of a row hold its 8-token groups
#pragma unroll
      for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int mm = 1; mm < UPR; mm <<= 1) acc[i] += __shfl_xor(acc[i], mm);
      if constexpr (TEAM) {
        if (hf == 0) {
#pragma unroll
          for (int i = 0; i < NL; ++i) osm[sub * D + RPL * i + rowl] = acc[i];
        }
        lds_barrier();
        // (the next item's first barrier — every wave passes it only after this read — keeps osm from being
        //  overwritten early)
        if (sub == 0) {
          for (int d = lane; d < D; d += 64)
            outp[d] = to_elem<BF>(((osm[d] + osm[D + d]) + osm[2 * D + d]) + osm[3 * D + d]);
        }
      } else {
        if (hf == 0) {
#pragma unroll
          for (int i = 0; i < NL; ++i) outp[RPL * i + rowl] = to_elem<BF>(acc[i]);
        }
      }
      if (!has_next) break;
      if constexpr (QMODE) {
        adopt(nxt);  // its first K group is already in flight in rn
        ++round;
      }
    }
  };

  // Mode S goes first, before any of mode Q's preparation: the metadata of its item (requested at the very top) then
  // lives only as far as here.  Kept across the ranking code below it was spilled and reloaded on the way into the
  // item, and equal lengths ran 2.7 us slower than in pa_v1_kernel (124.5 -> 121.8 us by HIP events, same box, with
  // identical hot loops and identical rocprofv3 kernel time: profiles/r02m_late_ranking.md).
  if (!queue) {
    if (wq >= N) return;
    run(std::integral_constant<int, US>{}, std::false_type{}, std::false_type{}, cur, [](int, int&, int&) { return false; });
    return;
  }

  int nlong = B;  // teams: ranks [0, nlong) are the long sequences (everything when the batch is not ranked)
  if (!late) {
    if (ranked) nlong = rank_sequences(0, wave, 4);
    __syncthreads();
  } else if (more) {
    if (tid == 0) *sorted = 0;
    lds_barrier();
  }

  // ---- hand-out of a solo worker's items after its first one; ONE functor for the three schedules (so that `run` is
  //      instantiated once for solo workers), the schedule is wave-uniform ----
  int hmode = 1;           // 0: late-ranked snake by vrank, 1: snake over the ranks by wq, 2: quads behind the long items
  int64_t NL_items = N;    // hmode 2: long items come first in rank order ...
  int64_t nunits = N;      // ... then the quads: units in all
  int k1 = 0;              // hmode 2: this workgroup's first unit behind the long items
  const int G = gridDim.x, g = blockIdx.x;
  auto unit = [&](int k) -> int64_t { return (int64_t)k * G + ((k & 1) ? (G - 1 - g) : g); };  // the snake over workgroups
  auto quad_item = [&](int k) -> int64_t {  // my wave's item of this workgroup's (k1 + k)-th unit, N if there is none
    const int64_t u = unit(k1 + k);
    const int64_t t = NL_items + 4 * (u - NL_items) + wave;
    return u < nunits && t < N ? t : (int64_t)N;
  };
  auto solo_handout = [&](int k, int& s, int& h) -> bool {
    if (hmode == 0) {  // the snake over the ranks of the sequences behind the first round, by my place v in it
      const int rd = k - 1;
      const int64_t t64 = (int64_t)rd * nworkers + ((rd & 1) ? vrank : nworkers - 1 - vrank);
      const bool has = t64 < N - nworkers;
      if (has && rd == 0)  // the retired wave's ranking must be complete before its first use
        while (*sorted != 1) __builtin_amdgcn_s_sleep(1);
      const int t = has ? (int)t64 : 0;
      const int r = t / H;
      h = t - r * H;
      s = has ? (int)__builtin_amdgcn_readfirstlane((int)order[r]) : 0;
      return has;
    }
    const int64_t t64 = hmode == 1 ? (int64_t)k * nworkers + ((k & 1) ? (nworkers - 1 - wq) : wq) : quad_item(k);
    const bool has = t64 < N;
    ids_of(has ? (int)t64 : 0, s, h);
    return has;
  };

  Meta first;  // (a variable of its own: sharing `cur` with mode S made the two paths' values one register, spilled)
  if (team) {
    // ---- TEAMS FOR THE LONG ITEMS, SOLO QUADS FOR THE SHORT ONES (round 3).  A heavy-tailed batch (a few long sequences
    //      among many short ones) made every item a team item in round 2: the long ones became four times shorter, but
    //      each short one still paid a team's fixed cost (three barriers, a softmax and an output exchange for a handful
    //      of blocks) — 2.3-3.7 us per item, seven items in a row on the workgroups without a long one.  Now the unit
    //      handed to a workgroup is either ONE long item (length > maxL/4: ranks [0, nlong), done by the four waves
    //      together) or a QUAD of four short items of consecutive rank (one per wave, each wave on its own: no barrier,
    //      four items' fixed costs side by side).  A short item run by one wave is no longer than the longest item run
    //      by four, so units are comparable and the same snake over the ranked units balances them.  The units of a
    //      workgroup come in rank order: first its long items (phase 1), then its quads (phase 2). ----
    const bool hybrid = ranked && !(flags & QF_NOHYBRID);
    NL_items = hybrid ? (int64_t)nlong * H : (int64_t)N;
    nunits = NL_items + ((int64_t)N - NL_items + 3) / 4;
    if (g < NL_items) {
      Meta tfirst;
      int s0, h0;
      ids_of(g, s0, h0);
      meta_issue(tfirst, s0, h0, 4, wave);
      run(std::integral_constant<int, UT>{}, std::true_type{}, std::true_type{}, tfirst, [&](int k, int& s, int& h) {
        const int64_t u = unit(k);
        const bool has = u < NL_items;
        ids_of(has ? (int)u : 0, s, h);
        return has;
      });
    }
    if (NL_items >= N) return;
    while (unit(k1) < NL_items) ++k1;
    if (quad_item(0) >= N) return;
    hmode = 2;
    int s0, h0;
    ids_of((int)quad_item(0), s0, h0);
    meta_issue(first, s0, h0, 1, 0);
  } else {
    if (wave >= WQ) {  // not a worker in this mode
      if (late && more && wave == WQ) {
        rank_sequences(R0, 0, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *sorted = 1;
      }
      return;
    }
    // ONE ROUND of four solo workers per workgroup on a 256-CU chip (fp8 pages, head size 128: as many workers as items) —
    // which ranks share a CU?  Workgroup b runs on XCD b mod 8, the j-th workgroup of an XCD on CU slot j mod 32
    // (profiles/r04_underfilled_chip.md), so in rank order the three workgroups of CU u would hold places u, 256 + u, 512 + u
    // of the ranking (in quads of ranks): the first CUs twice the bytes of the last.  Instead they take places u,
    // 256 + (u + 128) mod 256 and 512 + (254 - 2u | 511 - 2u): every CU's — hence every SIMD's — places add up to 382 / 383.
    // Only WHICH worker runs an item changes (results bit-identical); cfg3 fp8 U{1..L} 42.1 -> 40.5 us by events on one box,
    // 39.53 -> 39.35 by rocprofv3 on another (profiles/r04_fp8_ragged_accounting.md section 4; adopted in round 5).
    int wqe = wq;
    if (WQ == 4 && !late && G == 768 && N <= nworkers) {
      const int j = g >> 3, u = (g & 7) * 32 + (j & 31), kk = j >> 5;
      const int pl = kk == 0 ? u : (kk == 1 ? 256 + ((u + 128) & 255) : 512 + (u < 128 ? 254 - 2 * u : 511 - 2 * u));
      wqe = pl * 4 + wave;
    }
    if (wqe >= N) return;  // more workers than items
    if (late) {
      hmode = 0;
      first = firstq;  // first item = item wq in index order: asked for at the top of the kernel
      const int myL = (int)len16[sq0];  // (every wave staged all the lengths itself: no barrier needed)
      int a = 0;
      for (int j0 = 0; j0 < R0; j0 += 64) {
        const int j = j0 + lane;
        const int lj = j < R0 ? (int)len16[j] : -1;
        a += (int)__popcll(__ballot(lj > myL || (lj == myL && j < sq0)));
      }
      vrank = a * H + hq0;
    } else {  // first item of worker wq = rank order position wq
      int s0, h0;
      ids_of(wqe, s0, h0);
      meta_issue(first, s0, h0, 1, 0);
    }
  }
  run(std::integral_constant<int, UQ>{}, std::false_type{}, std::true_type{}, first, solo_handout);
}

#ifdef VMI_DIAG
// The diagnostic flavour of the kernel: the same body, bracketed by two reads of the constant 100 MHz clock, one record per
// wave (vmi_diag_set_wave_timeline).  A wave that retires early in mode Q records when it left.
template <int D, bool BF, bool NT, int US, int UQ, int F8 = 0, bool KM = false, int UT = 1, bool APP = false>
__global__ void __launch_bounds__(256, (US >= 4 || (UQ >= 4 && !F8) || D > 64) ? 2 : 3) pa_q_kernel(const PAParams p) {
  const uint64_t t0 = wall_clock64();
  pa_q_body<D, BF, NT, US, UQ, F8, KM, UT, APP>(p);
  uint64_t* tl = g_wave_timeline;
  if (tl && (threadIdx.x & 63) == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the wave's stores have left
    uint64_t* rec = tl + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    rec[0] = t0;
    rec[1] = wall_clock64();
    rec[2] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: wave / SIMD / CU / SH / SE ids
    rec[3] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
  }
}
#endif

}  // namespace vmi
