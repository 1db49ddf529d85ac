/*
 * c_abi_smoke.c — drives the drop-in boundary with NO Python and NO torch: plain C, hipMalloc'd buffers,
 * the C-ABI of include/vmi_paged_attention.h, and the oracle's C restatement (oracle/pa_kernel_model.c,
 * checker only) compiled into the same executable.
 *
 *   gcc tests/c_abi/c_abi_smoke.c oracle/pa_kernel_model.c -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include \
 *       -Lvllmini_amd/_C -lvmi_paged_attention -L/opt/rocm/lib -lamdhip64 -lm -o c_abi_smoke
 *   (tests/test_parity_gpu.py::test_c_abi_from_plain_c does exactly that on the GPU box.)
 *
 * Scenario: 3 sequences (lengths 5, 40, 100), 4 heads x 64, block 16: reshape_and_cache writes the newest
 * token of each sequence, paged_attention_v1 attends; both results are compared with the oracle.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vmi_paged_attention.h"

/* oracle entry points (test infrastructure) */
int vmi_oracle_paged_attention_v1_f16(uint16_t*, const uint16_t*, const uint16_t*, const uint16_t*, int32_t, int32_t,
                                      int32_t, int32_t, float, const int32_t*, const int32_t*, int32_t, int32_t,
                                      const float*, int64_t, int64_t, int64_t, int32_t, int32_t);
int vmi_oracle_reshape_and_cache_f16(const uint16_t*, const uint16_t*, uint16_t*, uint16_t*, const int64_t*, int32_t,
                                     int32_t, int32_t, int32_t, int32_t, int64_t, int64_t);
uint16_t vmi_oracle_f2h(float);
float vmi_oracle_h2f(uint16_t);

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

static uint32_t rng_state = 12345u;
static float frand(void) { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }

int main(void) {
  enum { S = 3, H = 4, D = 64, BS = 16, NB = 32, MB = 8 };
  const int lens[S] = {5, 40, 100};
  const size_t kv_elems = (size_t)NB * H * D * BS;
  uint16_t* kc = malloc(kv_elems * 2), *vc = malloc(kv_elems * 2);
  uint16_t* qkv = malloc((size_t)S * 3 * H * D * 2);
  for (size_t i = 0; i < kv_elems; ++i) { kc[i] = vmi_oracle_f2h(frand()); vc[i] = vmi_oracle_f2h(frand()); }
  for (size_t i = 0; i < (size_t)S * 3 * H * D; ++i) qkv[i] = vmi_oracle_f2h(2.f * frand());
  int32_t tables[S * MB];
  int32_t seq_lens[S];
  int64_t slots[S];
  int next = 3;
  for (int s = 0; s < S; ++s) {
    seq_lens[s] = lens[s];
    for (int j = 0; j < MB; ++j) tables[s * MB + j] = -1;
    for (int j = 0; j < (lens[s] + BS - 1) / BS; ++j) { tables[s * MB + j] = (next * 7) % NB; ++next; }
    slots[s] = (int64_t)tables[s * MB + (lens[s] - 1) / BS] * BS + (lens[s] - 1) % BS;
  }
  const int64_t row = 3 * H * D; /* q/k/v are strided views of one fused row, like gpt2.py:35-41 */

  /* ---- oracle ---- */
  uint16_t* kc_ref = malloc(kv_elems * 2), *vc_ref = malloc(kv_elems * 2);
  memcpy(kc_ref, kc, kv_elems * 2);
  memcpy(vc_ref, vc, kv_elems * 2);
  vmi_oracle_reshape_and_cache_f16(qkv + H * D, qkv + 2 * H * D, kc_ref, vc_ref, slots, S, H, D, BS, 8, row, row);
  uint16_t out_ref[S * H * D];
  vmi_oracle_paged_attention_v1_f16(out_ref, qkv, kc_ref, vc_ref, S, H, D, H, 0.125f, tables, seq_lens, BS, MB, NULL, row,
                                    (int64_t)H * D * BS, (int64_t)D * BS, 0, S);

  /* ---- device, through the C-ABI ---- */
  void *d_kc, *d_vc, *d_qkv, *d_out, *d_tab, *d_len, *d_slot;
  CHECK_HIP(hipSetDevice(0));
  CHECK_HIP(hipMalloc(&d_kc, kv_elems * 2));
  CHECK_HIP(hipMalloc(&d_vc, kv_elems * 2));
  CHECK_HIP(hipMalloc(&d_qkv, (size_t)S * row * 2));
  CHECK_HIP(hipMalloc(&d_out, sizeof(out_ref)));
  CHECK_HIP(hipMalloc(&d_tab, sizeof(tables)));
  CHECK_HIP(hipMalloc(&d_len, sizeof(seq_lens)));
  CHECK_HIP(hipMalloc(&d_slot, sizeof(slots)));
  CHECK_HIP(hipMemcpy(d_kc, kc, kv_elems * 2, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_vc, vc, kv_elems * 2, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_qkv, qkv, (size_t)S * row * 2, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_tab, tables, sizeof(tables), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_len, seq_lens, sizeof(seq_lens), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_slot, slots, sizeof(slots), hipMemcpyHostToDevice));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  uint16_t* dq = (uint16_t*)d_qkv;
  int rc = vmi_reshape_and_cache_f16(dq + H * D, dq + 2 * H * D, d_kc, d_vc, (const int64_t*)d_slot, S, H, D, BS, 8, row, row,
                                     0, stream);
  if (rc) { printf("reshape_and_cache rc=%d: %s\n", rc, vmi_last_error_string()); return 1; }
  rc = vmi_paged_attention_v1_f16(d_out, dq, d_kc, d_vc, S, H, D, H, 0.125f, (const int32_t*)d_tab, (const int32_t*)d_len, BS,
                                  MB * BS, MB, NULL, row, (int64_t)H * D * BS, (int64_t)D * BS, 0, stream);
  if (rc) { printf("paged_attention_v1 rc=%d: %s\n", rc, vmi_last_error_string()); return 1; }
  CHECK_HIP(hipStreamSynchronize(stream));
  uint16_t out[S * H * D];
  CHECK_HIP(hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(kc, d_kc, kv_elems * 2, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(vc, d_vc, kv_elems * 2, hipMemcpyDeviceToHost));

  if (memcmp(kc, kc_ref, kv_elems * 2) || memcmp(vc, vc_ref, kv_elems * 2)) { printf("FAIL: caches differ from the oracle\n"); return 1; }
  double worst = 0;
  for (int i = 0; i < S * H * D; ++i) {
    const double d = fabs((double)vmi_oracle_h2f(out[i]) - (double)vmi_oracle_h2f(out_ref[i]));
    if (d > worst) worst = d;
  }
  /* an unsupported head size must come back as a validation code, not a crash */
  rc = vmi_paged_attention_v1_f16(d_out, dq, d_kc, d_vc, S, H, 72, H, 0.125f, (const int32_t*)d_tab, (const int32_t*)d_len, BS,
                                  MB * BS, MB, NULL, row, (int64_t)H * D * BS, (int64_t)D * BS, 0, stream);
  printf("abi=%d arch=%s max|hip-oracle|=%.3e bad-head-size rc=%d (%s)\n", vmi_abi_version(), vmi_target_arch(), worst, rc,
         vmi_last_error_string());
  if (worst > 1e-3 || rc != VMI_E_HEAD_SIZE) { printf("FAIL\n"); return 1; }
  printf("PASS\n");
  return 0;
}
