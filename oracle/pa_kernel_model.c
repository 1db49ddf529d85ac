/*
 * pa_kernel_model.c — CPU restatement of the reference's paged-attention decode arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and
 * only as the checker.  The product path (vllmini_amd/) never imports it.
 *
 * What it restates (all paths relative to /root/reference/paged_attention_ext/paged_attention_cuda/):
 *   vmi_oracle_paged_attention_v1_f16  <- attention_kernels.cu:86-496 (paged_attention_kernel,
 *                                         PARTITION_SIZE = 0) as launched by :690-767 with
 *                                         NUM_THREADS = 128 and, for the NVIDIA build the reference
 *                                         ships (setup.py:30-45, no USE_ROCM), WARP_SIZE = 32 (:36-40)
 *   vmi_oracle_reshape_and_cache_f16   <- cache_kernels.cu:152-207
 *   fp16 helpers                       <- dtype_float16.cuh:87-126 (cvt), 219-260 (packed mul),
 *                                         160-168 (packed add), 331-431 (fma into fp32),
 *                                         434-457 (sum), 459-478 (from_float)
 *   bf16 helpers (dtype = 1)           <- dtype_bfloat16.cuh:207-214 (bf16 products), :337-345 (fma into
 *                                         fp32), :387-405 (fp32 sums), :408-438 (from_float, RNE)
 *
 * It is a lane-by-lane model: every thread's private partial sums, every shuffle butterfly and
 * the cross-warp tree are evaluated in the order the CUDA kernel evaluates them, so fp32
 * summation order and every fp16 rounding point match the reference.  What cannot match bit
 * for bit is the GPU's approximate transcendental hardware: __expf (:335, ex2.approx) and
 * __fdividef (:342, rcp.approx) are replaced by libm expf and an IEEE divide (<= 2 ulp fp32).
 *
 * PARITY PIN STATUS: PINNED by fixtures generated here from the IMPORTED reference Python (tests/golden/gen_golden.py,
 * committed with its outputs): ref_eager.npz — the eager attention the reference's own test compares its kernel with
 * (vllmini/tests/kernels/paged_attention.py:102-138, atol 1e-2; same expression as vllmini/model/gpt2.py:71-78) —,
 * ref_selftest.json — that unittest executed against this model through a stub paged_attention_cuda —, seam_trace.npz —
 * every op call of the reference scheduler stack for config 1 — and the layout round trip that test asserts (:63-82);
 * tests/test_oracle.py checks all of them.  What no vector in this pipeline can pin is the CUDA kernel's OWN output: it
 * cannot be built or run here (no nvcc, no NVIDIA GPU; its USE_ROCM branch includes files missing from the tree,
 * quantization/fp8/amd/quant_utils.cuh:8-10) and the reference ships no golden vectors for it.  Fidelity to the kernel's
 * rounding points therefore rests on this line-by-line restatement (every cited line can be read against the source),
 * inside the bound the reference's own test sets between its kernel and its eager path.
 *
 * Generic over head_size (multiple of 8*THREAD_GROUP_SIZE... the reference set 64..256) and
 * block_size in {8, 16, 32} by following the reference's constexpr formulas (:138-168, :360-369).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define WARP_SIZE 32
#define NUM_THREADS 128
#define NUM_WARPS (NUM_THREADS / WARP_SIZE)

/* ---------------------------------------------------------------------------------------
 * IEEE binary16 <-> binary32, software, round-to-nearest-even, subnormals preserved.
 * (cvt.f32.f16 / cvt.rn.f16.f32 — dtype_float16.cuh:87-126)
 * ------------------------------------------------------------------------------------- */
static inline float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while (!(man & 0x400u));
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static inline uint16_t f2h(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | (absx > 0x7f800000u ? 0x200u | ((absx >> 13) & 0x3ffu) : 0));
  }
  if (absx >= 0x477ff000u) { /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (absx < 0x33000001u) { /* <= 2^-25: rounds to zero (2^-25 is a tie -> even = 0) */
    return (uint16_t)sign;
  }
  int32_t e = (int32_t)(absx >> 23) - 127;
  uint32_t man = (absx & 0x7fffffu) | 0x800000u; /* 24-bit significand */
  uint32_t shift;
  uint32_t hexp;
  if (e < -14) { /* subnormal half */
    shift = (uint32_t)(13 + (-14 - e));
    hexp = 0;
  } else {
    shift = 13;
    hexp = (uint32_t)(e + 15);
  }
  uint32_t q = man >> shift;
  uint32_t rem = man & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) ++q;
  uint32_t r;
  if (hexp == 0) {
    r = q; /* may carry into the exponent field: that is the correct encoding */
  } else {
    r = ((hexp - 1) << 10) + q; /* q includes the hidden bit (0x400); carry propagates */
  }
  return (uint16_t)(sign | r);
}

/* ---------------------------------------------------------------------------------------
 * bfloat16 <-> binary32 (dtype_bfloat16.cuh: __bfloat162float, __float2bfloat16 / __float22bfloat162_rn):
 * widening is a 16-bit shift, narrowing is round-to-nearest-even on the upper 16 bits.
 * ------------------------------------------------------------------------------------- */
static inline float b2f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline uint16_t f2b(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
/*
 * fp8 E4M3 ("fn": no infinities, NaN = S.1111.111, max finite 448) — the interpretation the reference selects for
 * kv_cache_dtype "fp8" / "fp8_e4m3" (quantization/fp8/nvidia/quant_utils.cuh:529-566, __NV_E4M3).
 *   decode: exact (__nv_cvt_fp8_to_halfraw, quant_utils.cuh:295-300)
 *   encode: round to nearest even, saturate to +-448 (__NV_SATFINITE, quant_utils.cuh:458-464), NaN stays NaN
 */
static inline float fp8e4m3_to_f32(uint8_t b) {
  const int sign = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) return sign ? -NAN : NAN;
  if (e == 0) v = ldexpf((float)m, -9);            /* subnormal: m/8 * 2^-6 */
  else v = ldexpf((float)(8 + m), e - 10);         /* (1 + m/8) * 2^(e-7) */
  return sign ? -v : v;
}
static inline uint8_t f32_to_fp8e4m3_satfinite(float f) {
  const uint8_t sign = signbit(f) ? 0x80 : 0;
  if (isnan(f)) return (uint8_t)(sign | 0x7f);
  float a = fabsf(f);
  if (a >= 464.f) return (uint8_t)(sign | 0x7e);   /* above the midpoint past 448 (and +-inf): saturate */
  if (a == 0.f) return sign;
  int e;
  (void)frexpf(a, &e);                              /* a = f * 2^e, f in [0.5, 1) -> floor(log2 a) = e - 1 */
  e -= 1;
  if (e < -6) e = -6;                               /* subnormal range shares the step of the smallest normal */
  int q = (int)rintf(ldexpf(a, 3 - e));             /* a in units of 2^(e-3); RNE (default rounding mode) */
  if (q == 16) { q = 8; e += 1; }
  if (e > 8 || (e == 8 && q > 14)) return (uint8_t)(sign | 0x7e);
  if (q < 8) return (uint8_t)(sign | q);            /* subnormal (e == -6) */
  return (uint8_t)(sign | ((e + 7) << 3) | (q - 8));
}
/*
 * fp8 E5M2 (kv_cache_dtype "fp8_e5m2", __NV_E5M2: quant_utils.cuh:552-558): the upper byte of an IEEE half —
 * infinities and NaNs included — so decoding is exact by construction (__nv_cvt_fp8_to_halfraw).
 *   encode: round to nearest even on 2 mantissa bits, saturate to +-57344 (__NV_SATFINITE; +-inf saturate too),
 *           NaN -> a NaN code (0x7f | sign here; the exact NaN payload is not part of any result)
 */
static inline float h2f(uint16_t h);
static inline float fp8e5m2_to_f32(uint8_t b) { return h2f((uint16_t)((uint16_t)b << 8)); }
static inline uint8_t f32_to_fp8e5m2_satfinite(float f) {
  const uint8_t sign = signbit(f) ? 0x80 : 0;
  if (isnan(f)) return (uint8_t)(sign | 0x7f);
  float a = fabsf(f);
  if (a >= 61440.f) return (uint8_t)(sign | 0x7b); /* at or above the midpoint past 57344 (and +-inf): saturate */
  if (a == 0.f) return sign;
  int e;
  (void)frexpf(a, &e);
  e -= 1;                                           /* floor(log2 a) */
  if (e < -14) e = -14;                             /* subnormal range shares the step of the smallest normal: 2^-16 */
  int q = (int)rintf(ldexpf(a, 2 - e));             /* a in units of 2^(e-2); RNE */
  if (q == 8) { q = 4; e += 1; }
  if (e > 15) return (uint8_t)(sign | 0x7b);
  if (q < 4) return (uint8_t)(sign | q);            /* subnormal (e == -14) */
  return (uint8_t)(sign | ((e + 15) << 2) | (q - 4));
}
float vmi_oracle_fp8e5m2_to_f32(uint8_t b) { return fp8e5m2_to_f32(b); }
uint8_t vmi_oracle_f32_to_fp8e5m2(float f) { return f32_to_fp8e5m2_satfinite(f); }
float vmi_oracle_fp8e4m3_to_f32(uint8_t b) { return fp8e4m3_to_f32(b); }
uint8_t vmi_oracle_f32_to_fp8e4m3(float f) { return f32_to_fp8e4m3_satfinite(f); }

uint16_t vmi_oracle_f2b(float f) { return f2b(f); }
float vmi_oracle_b2f(uint16_t b) { return b2f(b); }

/* element type of query / caches / out: the reference dispatches on it (quant_utils.cuh:529-566) */
#define VMI_F16 0
#define VMI_BF16 1
static inline float e2f(uint16_t x, int dt) { return dt == VMI_BF16 ? b2f(x) : h2f(x); }
static inline uint16_t f2e(float x, int dt) { return dt == VMI_BF16 ? f2b(x) : f2h(x); }

/* fp16 op = exact fp32 op rounded once to fp16: innocuous double rounding since 24 >= 2*11+2. */
static inline uint16_t hmul(uint16_t a, uint16_t b) { return f2h(h2f(a) * h2f(b)); }
static inline uint16_t hadd(uint16_t a, uint16_t b) { return f2h(h2f(a) + h2f(b)); }

/* exported for the unit tests of the converters */
uint16_t vmi_oracle_f2h(float f) { return f2h(f); }
float vmi_oracle_h2f(uint16_t h) { return h2f(h); }
uint16_t vmi_oracle_hmul(uint16_t a, uint16_t b) { return hmul(a, b); }
uint16_t vmi_oracle_hadd(uint16_t a, uint16_t b) { return hadd(a, b); }

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* xor-butterfly over `n` lanes starting at `hi` (mask = hi, hi/2, ..., lo), as the
 * VLLM_SHFL_XOR_SYNC loops do: every lane adds its partner's value of the previous step. */
static void butterfly_sum(float* v, int n, int hi, int lo) {
  float t[WARP_SIZE];
  for (int mask = hi; mask >= lo && mask >= 1; mask /= 2) {
    for (int l = 0; l < n; ++l) t[l] = v[l] + v[l ^ mask];
    memcpy(v, t, (size_t)n * sizeof(float));
  }
}

/* block_sum<NUM_WARPS> — attention_kernels.cu:49-82.  per_thread[NUM_THREADS] -> scalar. */
static float block_sum(const float* per_thread) {
  float red[NUM_WARPS];
  for (int w = 0; w < NUM_WARPS; ++w) {
    float v[WARP_SIZE];
    memcpy(v, per_thread + w * WARP_SIZE, sizeof(v));
    butterfly_sum(v, WARP_SIZE, WARP_SIZE / 2, 1);
    red[w] = v[0]; /* lane 0 stores (:62-64) */
  }
  float v[WARP_SIZE];
  for (int l = 0; l < WARP_SIZE; ++l) v[l] = l < NUM_WARPS ? red[l] : 0.f; /* (:70-72) other lanes keep
      their own partial, but only lanes < NUM_WARPS feed lane 0 through masks < NUM_WARPS */
  butterfly_sum(v, WARP_SIZE, NUM_WARPS / 2, 1);
  return v[0]; /* broadcast of lane 0 (:81) */
}

/*
 * One (seq, head) of paged_attention_kernel.  Returns 0, or -1 on allocation failure.
 */
/* kv_fp8: the caches hold fp8 E4M3 bytes (kv_cache_dtype "fp8"); every element read from them becomes
 * float_to_half(half_to_float(fp8 -> half) * kv_scale) first (quant_utils.cuh:295-300), then the fp16 arithmetic
 * runs unchanged; x = 16 / sizeof(cache_t) = 16 (attention_kernels.cu:200). */
static inline uint16_t kv_elem(const void* cache, int64_t idx, int kv_fp8, float kv_scale, int dt) {
  if (!kv_fp8) return ((const uint16_t*)cache)[idx];
  /* float16: float_to_half(float(fp8) * scale) (:295-300); bfloat16: __float2bfloat16(float(fp8) * scale) (:350-359) */
  const uint8_t b = ((const uint8_t*)cache)[idx];
  return f2e((kv_fp8 == 2 ? fp8e5m2_to_f32(b) : fp8e4m3_to_f32(b)) * kv_scale, dt); /* kv_fp8: 1 = E4M3, 2 = E5M2 */
}

static int attend_one(uint16_t* out_row, const uint16_t* q_ptr, const void* k_cache,
                      const void* v_cache, int kv_head_idx, float scale,
                      const int32_t* block_table, int seq_len, int HEAD_SIZE, int BLOCK_SIZE,
                      float alibi_slope, int64_t kv_block_stride, int64_t kv_head_stride,
                      int PARTITION_SIZE, int partition_idx, float* max_logit_out, float* exp_sum_out,
                      int dt, int kv_fp8, float kv_scale, const int32_t* bsp, int bs_block_offset) {
  /* bsp (block-sparse attention, IS_BLOCK_SPARSE = vert_stride > 1, :822): {tp_rank, local_blocks, vert_stride,
   * blocksparse_block_size, head_sliding_step} or NULL; bs_block_offset per (head) as :216-224 */
  const int q_bs_block_id = bsp ? (seq_len - 1) / bsp[3] : 0; /* :215 */
  const int USE_PARTITIONING = PARTITION_SIZE > 0;                    /* :114 */
  const int num_seq_blocks = (seq_len + BLOCK_SIZE - 1) / BLOCK_SIZE; /* :121 */
  const int num_blocks_per_partition = USE_PARTITIONING ? PARTITION_SIZE / BLOCK_SIZE : num_seq_blocks; /* :122-123 */
  const int start_block_idx = USE_PARTITIONING ? partition_idx * num_blocks_per_partition : 0;          /* :126-127 */
  const int end_block_idx = imin(start_block_idx + num_blocks_per_partition, num_seq_blocks);           /* :128-129 */
  const int num_blocks = end_block_idx - start_block_idx;                                               /* :130 */
  const int start_token_idx = start_block_idx * BLOCK_SIZE;                                             /* :133 */
  const int end_token_idx = imin(start_token_idx + num_blocks * BLOCK_SIZE, seq_len);                   /* :134-135 */
  const int THREAD_GROUP_SIZE = imax(WARP_SIZE / BLOCK_SIZE, 1);     /* :138 */
  const int VEC_SIZE = imax(16 / (THREAD_GROUP_SIZE * 2), 1);        /* :162 */
  const int NUM_ELEMS_PER_THREAD = HEAD_SIZE / THREAD_GROUP_SIZE;    /* :167 */
  const int NUM_VECS_PER_THREAD = NUM_ELEMS_PER_THREAD / VEC_SIZE;   /* :168 */
  const int x = kv_fp8 ? 16 : 8;                                     /* :200: 16 / sizeof(cache_t) */

  const int padded = imax(num_blocks * BLOCK_SIZE, 1);
  float* logits = (float*)malloc((size_t)padded * sizeof(float));
  if (!logits) return -1;

  float qk_max = -FLT_MAX; /* :201 */

  /* ---- K pass (:227-308).  Logits are per token; which warp handles a block only affects
   * the (exact) max reduction, so blocks are walked in order. ---- */
  for (int block_idx = start_block_idx; block_idx < end_block_idx; ++block_idx) {
    if (bsp) { /* :232-254: blocks that are neither "remote" (vertical stripe) nor "local" are skipped */
      const int k_bs_block_id = block_idx * BLOCK_SIZE / bsp[3];
      const int is_remote = ((k_bs_block_id + bs_block_offset) % bsp[2] == 0);
      const int is_local = (k_bs_block_id > q_bs_block_id - bsp[1]);
      if (!is_remote && !is_local) {
        for (int o = 0; o < BLOCK_SIZE; ++o) logits[block_idx * BLOCK_SIZE + o - start_token_idx] = -FLT_MAX; /* :241-251 */
        continue;
      }
    }
    const int64_t physical_block_number = (int64_t)block_table[block_idx]; /* :257-258 */
    for (int physical_block_offset = 0; physical_block_offset < BLOCK_SIZE; ++physical_block_offset) {
      const int token_idx = block_idx * BLOCK_SIZE + physical_block_offset; /* :268 */
      const int64_t k_ptr = physical_block_number * kv_block_stride +
                            (int64_t)kv_head_idx * kv_head_stride + physical_block_offset * x; /* :273-275 */
      float qk_group[WARP_SIZE]; /* per thread of the thread group */
      for (int tgo = 0; tgo < THREAD_GROUP_SIZE; ++tgo) {
        float qk_vec[8];
        for (int j = 0; j < NUM_VECS_PER_THREAD; ++j) {
          const int vec_idx = tgo + j * THREAD_GROUP_SIZE;    /* :184, :276 */
          const int offset1 = (vec_idx * VEC_SIZE) / x;       /* :277 */
          const int offset2 = (vec_idx * VEC_SIZE) % x;       /* :278 */
          const int64_t kv = k_ptr + offset1 * BLOCK_SIZE * x + offset2; /* :281-289 */
          const uint16_t* qv = q_ptr + vec_idx * VEC_SIZE;    /* :186 */
          for (int e = 0; e < VEC_SIZE; ++e) {
            const float fq = e2f(qv[e], dt), fk = e2f(kv_elem(k_cache, kv + e, kv_fp8, kv_scale, dt), dt); /* dtype_bfloat16.cuh:341-345 for bf16 */
            if (j == 0) {
              qk_vec[e] = fq * fk; /* mul<A_vec>(q[0], k[0]) — attention_utils.cuh:34 */
            } else {
              qk_vec[e] = fmaf(fq, fk, qk_vec[e]); /* fma(q[ii], k[ii], qk_vec) — :36-38 */
            }
          }
        }
        float s = qk_vec[0]; /* sum(A_vec): left to right (dtype_float32.cuh:188-206) */
        for (int e = 1; e < VEC_SIZE; ++e) s += qk_vec[e];
        qk_group[tgo] = s;
      }
      /* xor butterfly over the thread group (attention_utils.cuh:42-45); offset-0 lane is read */
      butterfly_sum(qk_group, THREAD_GROUP_SIZE, THREAD_GROUP_SIZE / 2, 1);
      float qk = scale * qk_group[0];                                             /* :294 */
      qk += (alibi_slope != 0) ? alibi_slope * (float)(token_idx - seq_len + 1) : 0; /* :297 */
      const int mask = token_idx >= seq_len;                                      /* :302 */
      logits[token_idx - start_token_idx] = mask ? 0.f : qk;                      /* :303 */
      qk_max = mask ? qk_max : fmaxf(qk_max, qk);                                 /* :305 */
    }
  }

  /* ---- softmax (:310-346) ---- */
  const int num_tokens = end_token_idx - start_token_idx;                         /* :136 */
  float per_thread[NUM_THREADS];
  for (int t = 0; t < NUM_THREADS; ++t) {
    float exp_sum = 0.f;
    for (int i = t; i < num_tokens; i += NUM_THREADS) { /* :334-338 */
      const float val = expf(logits[i] - qk_max);       /* __expf */
      logits[i] = val;
      exp_sum += val;
    }
    per_thread[t] = exp_sum;
  }
  const float exp_sum = block_sum(per_thread);          /* :339 */
  const float inv_sum = 1.f / (exp_sum + 1e-6f);        /* :342 __fdividef */
  for (int i = 0; i < num_tokens; ++i) logits[i] *= inv_sum; /* :343-345 */
  if (USE_PARTITIONING) {                                     /* :349-357 */
    *max_logit_out = qk_max;
    *exp_sum_out = exp_sum;
  }

  /* ---- V pass (:359-434) ---- */
  const int V_VEC_SIZE = imin(8, BLOCK_SIZE);                               /* :360 */
  const int NUM_V_VECS_PER_ROW = BLOCK_SIZE / V_VEC_SIZE;                   /* :366 */
  const int NUM_ROWS_PER_ITER = WARP_SIZE / NUM_V_VECS_PER_ROW;             /* :367 */
  const int NUM_ROWS_PER_THREAD = (HEAD_SIZE + NUM_ROWS_PER_ITER - 1) / NUM_ROWS_PER_ITER; /* :368-369 */

  /* accs[warp][lane][i] */
  float* accs = (float*)calloc((size_t)NUM_WARPS * WARP_SIZE * NUM_ROWS_PER_THREAD, sizeof(float));
  if (!accs) {
    free(logits);
    return -1;
  }
#define ACC(w, l, i) accs[((size_t)(w) * WARP_SIZE + (l)) * NUM_ROWS_PER_THREAD + (i)]

  for (int warp_idx = 0; warp_idx < NUM_WARPS; ++warp_idx) {
    for (int block_idx = start_block_idx + warp_idx; block_idx < end_block_idx; block_idx += NUM_WARPS) { /* :380-381 */
      if (bsp) { /* :385-393 */
        const int v_bs_block_id = block_idx * BLOCK_SIZE / bsp[3];
        if (!((v_bs_block_id + bs_block_offset) % bsp[2] == 0) && !(v_bs_block_id > q_bs_block_id - bsp[1])) continue;
      }
      const int64_t physical_block_number = (int64_t)block_table[block_idx];
      const int64_t v_ptr = physical_block_number * kv_block_stride +
                            (int64_t)kv_head_idx * kv_head_stride; /* :402-403 */
      for (int lane = 0; lane < WARP_SIZE; ++lane) {
        const int physical_block_offset = (lane % NUM_V_VECS_PER_ROW) * V_VEC_SIZE; /* :396 */
        const int token_idx = block_idx * BLOCK_SIZE + physical_block_offset;       /* :397 */
        uint16_t logits_vec[8];
        for (int j = 0; j < V_VEC_SIZE; ++j)
          logits_vec[j] = f2e(logits[token_idx - start_token_idx + j], dt); /* :398-400 */
        for (int i = 0; i < NUM_ROWS_PER_THREAD; ++i) {
          const int row_idx = lane / NUM_V_VECS_PER_ROW + i * NUM_ROWS_PER_ITER; /* :406 */
          if (row_idx < HEAD_SIZE) {
            const int offset = row_idx * BLOCK_SIZE + physical_block_offset; /* :408 */
            uint16_t v_vec[8];
            for (int j = 0; j < V_VEC_SIZE; ++j) v_vec[j] = kv_elem(v_cache, v_ptr + offset + j, kv_fp8, kv_scale, dt); /* :410-418 */
            if (block_idx == num_seq_blocks - 1) { /* :420-430 */
              for (int j = 0; j < V_VEC_SIZE; ++j) v_vec[j] = (token_idx + j < seq_len) ? v_vec[j] : 0;
            }
            /* dot(logits_vec, v_vec) = sum(mul<uint4>(a, b)) — attention_generic.cuh:40-43,
             * dtype_float16.cuh:252-260 (packed fp16 products), :451-457 (packed fp16 adds of the
             * four half2 words), :439-443 (fp32 add of the two halves) */
            float d;
            if (dt == VMI_BF16) {
              /* bf16: mul<bf16_8_t> = four __hmul2 (each product rounded to bf16, dtype_bfloat16.cuh:207-214),
               * sum(bf16_8_t) = sum(x)+sum(y)+sum(z)+sum(w) with sum(bf162) = float(lo)+float(hi) (:392-405) */
              float s4[4] = {0.f, 0.f, 0.f, 0.f};
              for (int w = 0; w < V_VEC_SIZE / 2; ++w) {
                const float p0 = b2f(f2b(b2f(logits_vec[2 * w]) * b2f(v_vec[2 * w])));
                const float p1 = b2f(f2b(b2f(logits_vec[2 * w + 1]) * b2f(v_vec[2 * w + 1])));
                s4[w] = p0 + p1;
              }
              d = s4[0];
              for (int w = 1; w < V_VEC_SIZE / 2; ++w) d += s4[w];
            } else {
              uint16_t prod[8] = {0};
              for (int j = 0; j < V_VEC_SIZE; ++j) prod[j] = hmul(logits_vec[j], v_vec[j]);
              uint16_t c0 = prod[0], c1 = prod[1];
              for (int j = 2; j < V_VEC_SIZE; j += 2) {
                c0 = hadd(c0, prod[j]);
                c1 = hadd(c1, prod[j + 1]);
              }
              d = h2f(c0) + h2f(c1);
            }
            ACC(warp_idx, lane, i) += d; /* :431 */
          }
        }
      }
    }
    /* reduction within the warp over the lanes of a row (:436-445) */
    for (int i = 0; i < NUM_ROWS_PER_THREAD; ++i) {
      float v[WARP_SIZE];
      for (int l = 0; l < WARP_SIZE; ++l) v[l] = ACC(warp_idx, l, i);
      butterfly_sum(v, WARP_SIZE, NUM_V_VECS_PER_ROW / 2, 1);
      for (int l = 0; l < WARP_SIZE; ++l) ACC(warp_idx, l, i) = v[l];
    }
  }

  /* reduction across warps (:451-481): upper half adds into lower half, repeatedly */
  for (int i = NUM_WARPS; i > 1; i /= 2) {
    const int mid = i / 2;
    for (int w = 0; w < mid; ++w) {
      for (int r = 0; r < NUM_ROWS_PER_THREAD; ++r) {
        for (int l = 0; l < WARP_SIZE; l += NUM_V_VECS_PER_ROW) {
          ACC(w, l, r) += ACC(w + mid, l, r);
        }
      }
    }
  }

  /* final store (:483-495), from_float -> fp16 RNE */
  for (int r = 0; r < NUM_ROWS_PER_THREAD; ++r) {
    for (int l = 0; l < WARP_SIZE; l += NUM_V_VECS_PER_ROW) {
      const int row_idx = l / NUM_V_VECS_PER_ROW + r * NUM_ROWS_PER_ITER;
      if (row_idx < HEAD_SIZE) out_row[row_idx] = f2e(ACC(0, l, r), dt);
    }
  }
#undef ACC
  free(accs);
  free(logits);
  return 0;
}

/*
 * Whole-operator entry; argument meaning identical to vmi_paged_attention_v1_f16
 * (include/vmi_paged_attention.h) with host pointers.  seq_begin/seq_end select a slice of
 * sequences so callers can run slices on several host threads.
 * Returns 0 on success, 1 for an unsupported head/block size, -1 on allocation failure.
 */
/* :216-224: the stripe pattern slides with the (tensor-parallel-global) query head, or KV head for a negative step */
static int bs_offset(const int32_t* bsp, int num_heads, int num_kv_heads, int head_idx, int kv_head_idx) {
  if (!bsp) return 0;
  return bsp[4] >= 0 ? (bsp[0] * num_heads + head_idx) * bsp[4] + 1 : (bsp[0] * num_kv_heads + kv_head_idx) * (-bsp[4]) + 1;
}

static int pa_v1_impl(
    uint16_t* out, const uint16_t* query, const void* key_cache, const void* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t seq_begin, int32_t seq_end, int32_t dt,
    int kv_fp8, float kv_scale, const int32_t* bsp) {
  if (block_size != 8 && block_size != 16 && block_size != 32) return 1; /* :789-803 */
  switch (head_size) {                                                   /* :738-766 */
    case 64: case 80: case 96: case 112: case 128: case 192: case 256: break;
    default: return 1;
  }
  if (seq_end > num_seqs) seq_end = num_seqs;
  const int num_queries_per_kv = num_heads / num_kv_heads; /* :152 */
  for (int seq_idx = seq_begin; seq_idx < seq_end; ++seq_idx) {
    const int seq_len = seq_lens[seq_idx]; /* :115 */
    const int32_t* block_table = block_tables + (int64_t)seq_idx * max_num_blocks_per_seq; /* :207 */
    for (int head_idx = 0; head_idx < num_heads; ++head_idx) {
      const int kv_head_idx = head_idx / num_queries_per_kv;                    /* :153 */
      const float alibi_slope = alibi_slopes ? alibi_slopes[head_idx] : 0.f;    /* :154-155 */
      const uint16_t* q_ptr = query + (int64_t)seq_idx * q_stride + (int64_t)head_idx * head_size; /* :179 */
      uint16_t* out_row = out + ((int64_t)seq_idx * num_heads + head_idx) * head_size; /* :485-487 */
      int rc = attend_one(out_row, q_ptr, key_cache, value_cache, kv_head_idx, scale, block_table,
                          seq_len, head_size, block_size, alibi_slope, kv_block_stride,
                          kv_head_stride, 0, 0, NULL, NULL, dt, kv_fp8, kv_scale, bsp,
                          bs_offset(bsp, num_heads, num_kv_heads, head_idx, kv_head_idx));
      if (rc) return rc;
    }
  }
  return 0;
}

int vmi_oracle_paged_attention_v1(
    uint16_t* out, const uint16_t* query, const uint16_t* key_cache, const uint16_t* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t seq_begin, int32_t seq_end, int32_t dt) {
  return pa_v1_impl(out, query, key_cache, value_cache, num_seqs, num_heads, head_size, num_kv_heads, scale,
                    block_tables, seq_lens, block_size, max_num_blocks_per_seq, alibi_slopes, q_stride,
                    kv_block_stride, kv_head_stride, seq_begin, seq_end, dt, 0, 1.0f, NULL);
}

/* block-sparse attention (blocksparse_vert_stride > 1): the operator's last five arguments, "auto" cache */
int vmi_oracle_paged_attention_v1_blocksparse(
    uint16_t* out, const uint16_t* query, const uint16_t* key_cache, const uint16_t* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t seq_begin, int32_t seq_end, int32_t dt,
    int32_t tp_rank, int32_t local_blocks, int32_t vert_stride, int32_t bs_block_size, int32_t head_sliding_step) {
  const int32_t bsp[5] = {tp_rank, local_blocks, vert_stride, bs_block_size, head_sliding_step};
  return pa_v1_impl(out, query, key_cache, value_cache, num_seqs, num_heads, head_size, num_kv_heads, scale,
                    block_tables, seq_lens, block_size, max_num_blocks_per_seq, alibi_slopes, q_stride,
                    kv_block_stride, kv_head_stride, seq_begin, seq_end, dt, 0, 1.0f, vert_stride > 1 ? bsp : NULL);
}

/* kv_cache_dtype "fp8" / "fp8_e4m3": fp16 query/out, caches of fp8 E4M3 bytes
 * (key_cache [NB, H, D/16, BS, 16], value_cache [NB, H, D, BS]; strides in elements = bytes). */
int vmi_oracle_paged_attention_v1_fp8(
    uint16_t* out, const uint16_t* query, const uint8_t* key_cache, const uint8_t* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t seq_begin, int32_t seq_end, float kv_scale) {
  return pa_v1_impl(out, query, key_cache, value_cache, num_seqs, num_heads, head_size, num_kv_heads, scale,
                    block_tables, seq_lens, block_size, max_num_blocks_per_seq, alibi_slopes, q_stride,
                    kv_block_stride, kv_head_stride, seq_begin, seq_end, VMI_F16, 1, kv_scale, NULL);
}

/* the same with bfloat16 query / out (bit patterns) */
int vmi_oracle_paged_attention_v1_fp8_bf16(
    uint16_t* out, const uint16_t* query, const uint8_t* key_cache, const uint8_t* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t seq_begin, int32_t seq_end, float kv_scale) {
  return pa_v1_impl(out, query, key_cache, value_cache, num_seqs, num_heads, head_size, num_kv_heads, scale,
                    block_tables, seq_lens, block_size, max_num_blocks_per_seq, alibi_slopes, q_stride,
                    kv_block_stride, kv_head_stride, seq_begin, seq_end, VMI_BF16, 1, kv_scale, NULL);
}

int vmi_oracle_paged_attention_v1_f16(
    uint16_t* out, const uint16_t* query, const uint16_t* key_cache, const uint16_t* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t seq_begin, int32_t seq_end) {
  return vmi_oracle_paged_attention_v1(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                                       num_kv_heads, scale, block_tables, seq_lens, block_size,
                                       max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride,
                                       kv_head_stride, seq_begin, seq_end, VMI_F16);
}

/*
 * paged_attention_v2 — attention_kernels.cu:966-990 -> launcher :845-928:
 *   paged_attention_v2_kernel (:529-562, the same paged_attention_kernel with PARTITION_SIZE = 512,
 *   grid (heads, seqs, max_num_partitions)) followed by paged_attention_v2_reduce_kernel (:564-669).
 * exp_sums / max_logits: [num_seqs, num_heads, max_num_partitions] fp32;
 * tmp_out: [num_seqs, num_heads, max_num_partitions, head_size] fp16.  Partitions at or past the
 * context are not touched (:116-119), exactly like the kernel's early return.
 */
#define V2_PARTITION_SIZE 512

static void v2_reduce_one(uint16_t* out_ptr, const float* exp_sums_ptr, const float* max_logits_ptr,
                          const uint16_t* tmp_out_ptr, int seq_len, int HEAD_SIZE, int dt) {
  const int num_partitions = (seq_len + V2_PARTITION_SIZE - 1) / V2_PARTITION_SIZE; /* :581 */
  if (num_partitions == 1) {                                                          /* :582-594 */
    for (int i = 0; i < HEAD_SIZE; ++i) out_ptr[i] = tmp_out_ptr[i];
    return;
  }
  if (num_partitions <= 0) { /* seq_len == 0: every loop over partitions is empty -> acc = 0 is stored */
    for (int i = 0; i < HEAD_SIZE; ++i) out_ptr[i] = f2e(0.f, dt);
    return;
  }
  float* shared_max_logits = (float*)malloc(sizeof(float) * 2 * (size_t)num_partitions);
  float* shared_exp_sums = shared_max_logits + num_partitions;
  float max_logit = -FLT_MAX;
  for (int i = 0; i < num_partitions; ++i) { /* :611-615 + reductions :619-635 (max is exact) */
    shared_max_logits[i] = max_logits_ptr[i];
    max_logit = fmaxf(max_logit, max_logits_ptr[i]);
  }
  float per_thread[NUM_THREADS];
  for (int t = 0; t < NUM_THREADS; ++t) {
    float global_exp_sum = 0.0f;
    for (int i = t; i < num_partitions; i += NUM_THREADS) { /* :644-649 */
      const float l = shared_max_logits[i];
      const float rescaled_exp_sum = exp_sums_ptr[i] * expf(l - max_logit);
      global_exp_sum += rescaled_exp_sum;
      shared_exp_sums[i] = rescaled_exp_sum;
    }
    per_thread[t] = global_exp_sum;
  }
  const float global_exp_sum = block_sum(per_thread);              /* :651 */
  const float inv_global_exp_sum = 1.0f / (global_exp_sum + 1e-6f); /* :652 */
  for (int i = 0; i < HEAD_SIZE; ++i) {                            /* :661-668 */
    float acc = 0.0f;
    for (int j = 0; j < num_partitions; ++j) {
      /* acc += to_float(tmp) * shared_exp_sums[j] * inv: (a*b) rounded, then contracted into an FMA */
      acc = fmaf(e2f(tmp_out_ptr[(size_t)j * HEAD_SIZE + i], dt) * shared_exp_sums[j], inv_global_exp_sum, acc);
    }
    out_ptr[i] = f2e(acc, dt);
  }
  free(shared_max_logits);
}

static int pa_v2_impl(
    uint16_t* out, float* exp_sums, float* max_logits, uint16_t* tmp_out, const uint16_t* query,
    const void* key_cache, const void* value_cache, int32_t num_seqs, int32_t num_heads,
    int32_t head_size, int32_t num_kv_heads, float scale, const int32_t* block_tables,
    const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t dt, int kv_fp8, float kv_scale, const int32_t* bsp) {
  if (block_size != 8 && block_size != 16 && block_size != 32) return 1;
  switch (head_size) {
    case 64: case 80: case 96: case 112: case 128: case 192: case 256: break;
    default: return 1;
  }
  const int max_num_partitions = (max_seq_len + V2_PARTITION_SIZE - 1) / V2_PARTITION_SIZE; /* :885 */
  const int num_queries_per_kv = num_heads / num_kv_heads;
  for (int seq_idx = 0; seq_idx < num_seqs; ++seq_idx) {
    const int seq_len = seq_lens[seq_idx];
    const int32_t* block_table = block_tables + (int64_t)seq_idx * max_num_blocks_per_seq;
    for (int head_idx = 0; head_idx < num_heads; ++head_idx) {
      const int kv_head_idx = head_idx / num_queries_per_kv;
      const float alibi_slope = alibi_slopes ? alibi_slopes[head_idx] : 0.f;
      const uint16_t* q_ptr = query + (int64_t)seq_idx * q_stride + (int64_t)head_idx * head_size;
      const size_t sh = ((size_t)seq_idx * num_heads + head_idx) * max_num_partitions;
      for (int part = 0; part < max_num_partitions; ++part) {
        if (part * V2_PARTITION_SIZE >= seq_len) continue; /* :116-119 */
        int rc = attend_one(tmp_out + (sh + part) * head_size, q_ptr, key_cache, value_cache,
                            kv_head_idx, scale, block_table, seq_len, head_size, block_size,
                            alibi_slope, kv_block_stride, kv_head_stride, V2_PARTITION_SIZE, part,
                            max_logits + sh + part, exp_sums + sh + part, dt, kv_fp8, kv_scale, bsp,
                            bs_offset(bsp, num_heads, num_kv_heads, head_idx, kv_head_idx));
        if (rc) return rc;
      }
      v2_reduce_one(out + ((size_t)seq_idx * num_heads + head_idx) * head_size, exp_sums + sh,
                    max_logits + sh, tmp_out + sh * head_size, seq_len, head_size, dt);
    }
  }
  return 0;
}

int vmi_oracle_paged_attention_v2(
    uint16_t* out, float* exp_sums, float* max_logits, uint16_t* tmp_out, const uint16_t* query,
    const uint16_t* key_cache, const uint16_t* value_cache, int32_t num_seqs, int32_t num_heads,
    int32_t head_size, int32_t num_kv_heads, float scale, const int32_t* block_tables,
    const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t dt) {
  return pa_v2_impl(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
                    head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                    max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride, kv_head_stride, dt, 0, 1.0f, NULL);
}

int vmi_oracle_paged_attention_v2_blocksparse(
    uint16_t* out, float* exp_sums, float* max_logits, uint16_t* tmp_out, const uint16_t* query,
    const uint16_t* key_cache, const uint16_t* value_cache, int32_t num_seqs, int32_t num_heads,
    int32_t head_size, int32_t num_kv_heads, float scale, const int32_t* block_tables,
    const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t dt,
    int32_t tp_rank, int32_t local_blocks, int32_t vert_stride, int32_t bs_block_size, int32_t head_sliding_step) {
  const int32_t bsp[5] = {tp_rank, local_blocks, vert_stride, bs_block_size, head_sliding_step};
  return pa_v2_impl(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
                    head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                    max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride, kv_head_stride, dt, 0, 1.0f,
                    vert_stride > 1 ? bsp : NULL);
}

/* split-KV over an fp8 E4M3 cache (fp16 query / out / tmp_out) */
int vmi_oracle_paged_attention_v2_fp8(
    uint16_t* out, float* exp_sums, float* max_logits, uint16_t* tmp_out, const uint16_t* query,
    const uint8_t* key_cache, const uint8_t* value_cache, int32_t num_seqs, int32_t num_heads,
    int32_t head_size, int32_t num_kv_heads, float scale, const int32_t* block_tables,
    const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, float kv_scale) {
  return pa_v2_impl(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
                    head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                    max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride, kv_head_stride, VMI_F16, 1,
                    kv_scale, NULL);
}

int vmi_oracle_paged_attention_v2_f16(
    uint16_t* out, float* exp_sums, float* max_logits, uint16_t* tmp_out, const uint16_t* query,
    const uint16_t* key_cache, const uint16_t* value_cache, int32_t num_seqs, int32_t num_heads,
    int32_t head_size, int32_t num_kv_heads, float scale, const int32_t* block_tables,
    const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride) {
  return vmi_oracle_paged_attention_v2(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache,
                                       num_seqs, num_heads, head_size, num_kv_heads, scale, block_tables,
                                       seq_lens, block_size, max_seq_len, max_num_blocks_per_seq,
                                       alibi_slopes, q_stride, kv_block_stride, kv_head_stride, VMI_F16);
}

/*
 * reshape_and_cache — cache_kernels.cu:152-207 (index math :164-199).  Pure copy.
 */
int vmi_oracle_reshape_and_cache_f16(const uint16_t* key, const uint16_t* value,
                                     uint16_t* key_cache, uint16_t* value_cache,
                                     const int64_t* slot_mapping, int32_t num_tokens,
                                     int32_t num_heads, int32_t head_size, int32_t block_size,
                                     int32_t x, int64_t key_stride, int64_t value_stride) {
  for (int64_t token_idx = 0; token_idx < num_tokens; ++token_idx) {
    const int64_t slot_idx = slot_mapping[token_idx];
    if (slot_idx < 0) continue; /* :165-169 */
    const int64_t block_idx = slot_idx / block_size;  /* :172 */
    const int64_t block_offset = slot_idx % block_size; /* :173 */
    const int n = num_heads * head_size;
    for (int i = 0; i < n; ++i) {
      const int64_t src_key_idx = token_idx * key_stride + i;
      const int64_t src_value_idx = token_idx * value_stride + i;
      const int head_idx = i / head_size;
      const int head_offset = i % head_size;
      const int x_idx = head_offset / x;
      const int x_offset = head_offset % x;
      const int64_t tgt_key_idx = block_idx * num_heads * (head_size / x) * block_size * x +
                                  (int64_t)head_idx * (head_size / x) * block_size * x +
                                  (int64_t)x_idx * block_size * x + block_offset * x + x_offset; /* :187-190 */
      const int64_t tgt_value_idx = block_idx * num_heads * head_size * block_size +
                                    (int64_t)head_idx * head_size * block_size +
                                    (int64_t)head_offset * block_size + block_offset; /* :191-194 */
      key_cache[tgt_key_idx] = key[src_key_idx];       /* :198 */
      value_cache[tgt_value_idx] = value[src_value_idx]; /* :199 */
    }
  }
  return 0;
}

/*
 * reshape_and_cache with kv_cache_dtype "fp8": cache_kernels.cu:200-205 —
 * cache = scaled_convert<uint8_t, half>(x, kv_scale) = fp8(float(x) / kv_scale), RNE, saturating
 * (quant_utils.cuh:458-464); x = 16 / sizeof(cache_t) = 16 (cache_kernels.cu:269 via key_cache.size(4)).
 */
static int reshape_fp8_impl(const uint16_t* key, const uint16_t* value,
                                     uint8_t* key_cache, uint8_t* value_cache,
                                     const int64_t* slot_mapping, int32_t num_tokens,
                                     int32_t num_heads, int32_t head_size, int32_t block_size,
                                     int32_t x, int64_t key_stride, int64_t value_stride, float kv_scale, int dt,
                                     int fmt) {
  for (int64_t token_idx = 0; token_idx < num_tokens; ++token_idx) {
    const int64_t slot_idx = slot_mapping[token_idx];
    if (slot_idx < 0) continue;
    const int64_t block_idx = slot_idx / block_size;
    const int64_t block_offset = slot_idx % block_size;
    const int n = num_heads * head_size;
    for (int i = 0; i < n; ++i) {
      const int head_idx = i / head_size;
      const int head_offset = i % head_size;
      const int x_idx = head_offset / x;
      const int x_offset = head_offset % x;
      const int64_t tgt_key_idx = block_idx * num_heads * (head_size / x) * block_size * x +
                                  (int64_t)head_idx * (head_size / x) * block_size * x +
                                  (int64_t)x_idx * block_size * x + block_offset * x + x_offset;
      const int64_t tgt_value_idx = block_idx * num_heads * head_size * block_size +
                                    (int64_t)head_idx * head_size * block_size +
                                    (int64_t)head_offset * block_size + block_offset;
      const float kq = e2f(key[token_idx * key_stride + i], dt) / kv_scale;
      const float vq = e2f(value[token_idx * value_stride + i], dt) / kv_scale;
      key_cache[tgt_key_idx] = fmt == 2 ? f32_to_fp8e5m2_satfinite(kq) : f32_to_fp8e4m3_satfinite(kq);
      value_cache[tgt_value_idx] = fmt == 2 ? f32_to_fp8e5m2_satfinite(vq) : f32_to_fp8e4m3_satfinite(vq);
    }
  }
  return 0;
}

int vmi_oracle_reshape_and_cache_fp8(const uint16_t* key, const uint16_t* value, uint8_t* key_cache,
                                     uint8_t* value_cache, const int64_t* slot_mapping, int32_t num_tokens,
                                     int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
                                     int64_t key_stride, int64_t value_stride, float kv_scale) {
  return reshape_fp8_impl(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads, head_size,
                          block_size, x, key_stride, value_stride, kv_scale, VMI_F16, 1);
}

/* bfloat16 rows: fp8(__bfloat162float(x) / kv_scale) (quant_utils.cuh:468-478) */
int vmi_oracle_reshape_and_cache_fp8_bf16(const uint16_t* key, const uint16_t* value, uint8_t* key_cache,
                                          uint8_t* value_cache, const int64_t* slot_mapping, int32_t num_tokens,
                                          int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
                                          int64_t key_stride, int64_t value_stride, float kv_scale) {
  return reshape_fp8_impl(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads, head_size,
                          block_size, x, key_stride, value_stride, kv_scale, VMI_BF16, 1);
}

/* ---- generic fp8 entries: dt = VMI_F16 / VMI_BF16 (query, rows), fmt = 1 (E4M3) / 2 (E5M2) ---- */
int vmi_oracle_reshape_and_cache_fp8x(const uint16_t* key, const uint16_t* value, uint8_t* key_cache,
                                      uint8_t* value_cache, const int64_t* slot_mapping, int32_t num_tokens,
                                      int32_t num_heads, int32_t head_size, int32_t block_size, int32_t x,
                                      int64_t key_stride, int64_t value_stride, float kv_scale, int32_t dt, int32_t fmt) {
  return reshape_fp8_impl(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads, head_size,
                          block_size, x, key_stride, value_stride, kv_scale, dt, fmt);
}
int vmi_oracle_paged_attention_v1_fp8x(
    uint16_t* out, const uint16_t* query, const uint8_t* key_cache, const uint8_t* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t seq_begin, int32_t seq_end, float kv_scale,
    int32_t dt, int32_t fmt) {
  return pa_v1_impl(out, query, key_cache, value_cache, num_seqs, num_heads, head_size, num_kv_heads, scale,
                    block_tables, seq_lens, block_size, max_num_blocks_per_seq, alibi_slopes, q_stride,
                    kv_block_stride, kv_head_stride, seq_begin, seq_end, dt, fmt, kv_scale, NULL);
}
int vmi_oracle_paged_attention_v2_fp8x(
    uint16_t* out, float* exp_sums, float* max_logits, uint16_t* tmp_out, const uint16_t* query,
    const uint8_t* key_cache, const uint8_t* value_cache, int32_t num_seqs, int32_t num_heads,
    int32_t head_size, int32_t num_kv_heads, float scale, const int32_t* block_tables,
    const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, float kv_scale, int32_t dt, int32_t fmt) {
  return pa_v2_impl(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
                    head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                    max_num_blocks_per_seq, alibi_slopes, q_stride, kv_block_stride, kv_head_stride, dt, fmt,
                    kv_scale, NULL);
}

/* =====================================================================================================
 * float32 tensors (the (float, float) branch of the dispatch, quant_utils.cuh:529-535): x = 16 / sizeof(float) = 4,
 * key_cache [NB, H, D/4, BS, 4], value_cache [NB, H, D, BS]; every operation in fp32 (dtype_float32.cuh):
 *   VEC_SIZE = max(16 / (THREAD_GROUP_SIZE * 4), 1) (:162), qk = first product, then fma chain (attention_utils.cuh:30-47),
 *   V_VEC_SIZE = min(16 / 4, BLOCK_SIZE) = 4 (:360), dot(float4) = x + y + z + w of the products (dtype_float32.cuh:194-197),
 *   logits stay fp32 (from_float, :231-235).  Same control flow as attend_one above.
 * ===================================================================================================== */
static int attend_one_f32(float* out_row, const float* q_ptr, const float* k_cache, const float* v_cache,
                          int kv_head_idx, float scale, const int32_t* block_table, int seq_len, int HEAD_SIZE,
                          int BLOCK_SIZE, float alibi_slope, int64_t kv_block_stride, int64_t kv_head_stride) {
  const int num_seq_blocks = (seq_len + BLOCK_SIZE - 1) / BLOCK_SIZE;
  const int THREAD_GROUP_SIZE = imax(WARP_SIZE / BLOCK_SIZE, 1);
  const int VEC_SIZE = imax(16 / (THREAD_GROUP_SIZE * 4), 1);
  const int NUM_ELEMS_PER_THREAD = HEAD_SIZE / THREAD_GROUP_SIZE;
  const int NUM_VECS_PER_THREAD = NUM_ELEMS_PER_THREAD / VEC_SIZE;
  const int x = 4;
  const int padded = imax(num_seq_blocks * BLOCK_SIZE, 1);
  float* logits = (float*)malloc((size_t)padded * sizeof(float));
  if (!logits) return -1;
  float qk_max = -FLT_MAX;
  for (int block_idx = 0; block_idx < num_seq_blocks; ++block_idx) {
    const int64_t physical_block_number = (int64_t)block_table[block_idx];
    for (int physical_block_offset = 0; physical_block_offset < BLOCK_SIZE; ++physical_block_offset) {
      const int token_idx = block_idx * BLOCK_SIZE + physical_block_offset;
      const int64_t k_ptr = physical_block_number * kv_block_stride + (int64_t)kv_head_idx * kv_head_stride +
                            physical_block_offset * x;
      float qk_group[WARP_SIZE];
      for (int tgo = 0; tgo < THREAD_GROUP_SIZE; ++tgo) {
        float qk_vec[4];
        for (int j = 0; j < NUM_VECS_PER_THREAD; ++j) {
          const int vec_idx = tgo + j * THREAD_GROUP_SIZE;
          const int offset1 = (vec_idx * VEC_SIZE) / x;
          const int offset2 = (vec_idx * VEC_SIZE) % x;
          const int64_t kv = k_ptr + offset1 * BLOCK_SIZE * x + offset2;
          for (int e = 0; e < VEC_SIZE; ++e) {
            const float fq = q_ptr[vec_idx * VEC_SIZE + e], fk = k_cache[kv + e];
            qk_vec[e] = j == 0 ? fq * fk : fmaf(fq, fk, qk_vec[e]);
          }
        }
        float s = qk_vec[0];
        for (int e = 1; e < VEC_SIZE; ++e) s += qk_vec[e];
        qk_group[tgo] = s;
      }
      butterfly_sum(qk_group, THREAD_GROUP_SIZE, THREAD_GROUP_SIZE / 2, 1);
      float qk = scale * qk_group[0];
      qk += (alibi_slope != 0) ? alibi_slope * (float)(token_idx - seq_len + 1) : 0;
      const int mask = token_idx >= seq_len;
      logits[token_idx] = mask ? 0.f : qk;
      qk_max = mask ? qk_max : fmaxf(qk_max, qk);
    }
  }
  float per_thread[NUM_THREADS];
  for (int t = 0; t < NUM_THREADS; ++t) {
    float exp_sum = 0.f;
    for (int i = t; i < seq_len; i += NUM_THREADS) {
      const float val = expf(logits[i] - qk_max);
      logits[i] = val;
      exp_sum += val;
    }
    per_thread[t] = exp_sum;
  }
  const float exp_sum = block_sum(per_thread);
  const float inv_sum = 1.f / (exp_sum + 1e-6f);
  for (int i = 0; i < seq_len; ++i) logits[i] *= inv_sum;

  const int V_VEC_SIZE = imin(4, BLOCK_SIZE);
  const int NUM_V_VECS_PER_ROW = BLOCK_SIZE / V_VEC_SIZE;
  const int NUM_ROWS_PER_ITER = WARP_SIZE / NUM_V_VECS_PER_ROW;
  const int NUM_ROWS_PER_THREAD = (HEAD_SIZE + NUM_ROWS_PER_ITER - 1) / NUM_ROWS_PER_ITER;
  float* accs = (float*)calloc((size_t)NUM_WARPS * WARP_SIZE * NUM_ROWS_PER_THREAD, sizeof(float));
  if (!accs) {
    free(logits);
    return -1;
  }
#define ACC(w, l, i) accs[((size_t)(w) * WARP_SIZE + (l)) * NUM_ROWS_PER_THREAD + (i)]
  for (int warp_idx = 0; warp_idx < NUM_WARPS; ++warp_idx) {
    for (int block_idx = warp_idx; block_idx < num_seq_blocks; block_idx += NUM_WARPS) {
      const int64_t physical_block_number = (int64_t)block_table[block_idx];
      const int64_t v_ptr = physical_block_number * kv_block_stride + (int64_t)kv_head_idx * kv_head_stride;
      for (int lane = 0; lane < WARP_SIZE; ++lane) {
        const int physical_block_offset = (lane % NUM_V_VECS_PER_ROW) * V_VEC_SIZE;
        const int token_idx = block_idx * BLOCK_SIZE + physical_block_offset;
        for (int i = 0; i < NUM_ROWS_PER_THREAD; ++i) {
          const int row_idx = lane / NUM_V_VECS_PER_ROW + i * NUM_ROWS_PER_ITER;
          if (row_idx < HEAD_SIZE) {
            const int offset = row_idx * BLOCK_SIZE + physical_block_offset;
            float d = 0.f;
            for (int j = 0; j < V_VEC_SIZE; ++j) {
              float v = v_cache[v_ptr + offset + j];
              if (block_idx == num_seq_blocks - 1 && token_idx + j >= seq_len) v = 0.f;
              const float pr = logits[token_idx + j] * v;
              d = j == 0 ? pr : d + pr;
            }
            ACC(warp_idx, lane, i) += d;
          }
        }
      }
    }
    for (int i = 0; i < NUM_ROWS_PER_THREAD; ++i) {
      float v[WARP_SIZE];
      for (int l = 0; l < WARP_SIZE; ++l) v[l] = ACC(warp_idx, l, i);
      butterfly_sum(v, WARP_SIZE, NUM_V_VECS_PER_ROW / 2, 1);
      for (int l = 0; l < WARP_SIZE; ++l) ACC(warp_idx, l, i) = v[l];
    }
  }
  for (int i = NUM_WARPS; i > 1; i /= 2) {
    const int mid = i / 2;
    for (int w = 0; w < mid; ++w)
      for (int r = 0; r < NUM_ROWS_PER_THREAD; ++r)
        for (int l = 0; l < WARP_SIZE; l += NUM_V_VECS_PER_ROW) ACC(w, l, r) += ACC(w + mid, l, r);
  }
  for (int r = 0; r < NUM_ROWS_PER_THREAD; ++r)
    for (int l = 0; l < WARP_SIZE; l += NUM_V_VECS_PER_ROW) {
      const int row_idx = l / NUM_V_VECS_PER_ROW + r * NUM_ROWS_PER_ITER;
      if (row_idx < HEAD_SIZE) out_row[row_idx] = ACC(0, l, r);
    }
#undef ACC
  free(accs);
  free(logits);
  return 0;
}

int vmi_oracle_paged_attention_v1_f32(
    float* out, const float* query, const float* key_cache, const float* value_cache,
    int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
    int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int32_t seq_begin, int32_t seq_end) {
  if (block_size != 8 && block_size != 16 && block_size != 32) return 1;
  switch (head_size) {
    case 64: case 80: case 96: case 112: case 128: case 192: case 256: break;
    default: return 1;
  }
  if (seq_end > num_seqs) seq_end = num_seqs;
  const int num_queries_per_kv = num_heads / num_kv_heads;
  for (int seq_idx = seq_begin; seq_idx < seq_end; ++seq_idx) {
    const int32_t* block_table = block_tables + (int64_t)seq_idx * max_num_blocks_per_seq;
    for (int head_idx = 0; head_idx < num_heads; ++head_idx) {
      int rc = attend_one_f32(out + ((int64_t)seq_idx * num_heads + head_idx) * head_size,
                              query + (int64_t)seq_idx * q_stride + (int64_t)head_idx * head_size, key_cache,
                              value_cache, head_idx / num_queries_per_kv, scale, block_table, seq_lens[seq_idx],
                              head_size, block_size, alibi_slopes ? alibi_slopes[head_idx] : 0.f, kv_block_stride,
                              kv_head_stride);
      if (rc) return rc;
    }
  }
  return 0;
}
