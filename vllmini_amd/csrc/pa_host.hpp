// pa_host.hpp — host-side helpers shared by the translation units that define C-ABI entries (paged_attention.hip and the
// out-of-scope units pa_f32.hip / pa_extras_cache.hip / pa_extras_abi.hip): the thread's last error text, the error
// returns, the device guard, and the launchers the extras library's entries forward to.  Defined in paged_attention.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vmi {

extern thread_local char g_err[512];                 // text behind vmi_last_error_string()
int fail(int code, const char* fmt, ...);            // formats g_err, returns `code`
int hip_fail(hipError_t e, const char* what);        // "<what>: <hipGetErrorString>", returns -(int)e
// "<what> is not in this build" (VMI_E_NOT_BUILT): an out-of-scope operator asked of the product library
int not_built(const char* what);

// Make `device` current for the duration of a call and restore the caller's device afterwards
// (the reference wraps its launches in at::cuda::OptionalCUDAGuard, attention_kernels.cu:736).
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  hipError_t err;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) {
      err = hipSetDevice(device);
      changed = (err == hipSuccess);
    }
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool head_size_supported(int d);   // the reference's switch, attention_kernels.cu:738-766
bool block_size_supported(int b);  // :789-803

// The launchers behind every paged_attention_v1 / _v2 / fp8 scatter entry.  bf = bfloat16 tensors, f8 = 0 (16-bit pages) /
// 1 (fp8 E4M3) / 2 (fp8 E5M2), bsp = {tp_rank, local_blocks, vert_stride, blocksparse_block_size, head_sliding_step} or
// nullptr; a case whose kernel menu is empty in this library (the product library: bf, f8 == 2, bsp) is VMI_E_NOT_BUILT.
int launch_pa_v1(void* out, const void* query, const void* key_cache, const void* value_cache, int32_t num_seqs,
                 int32_t num_heads, int32_t head_size, int32_t num_kv_heads, float scale, const int32_t* block_tables,
                 const int32_t* seq_lens, int32_t block_size, int32_t max_seq_len, int32_t max_num_blocks_per_seq,
                 const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                 int32_t device, void* stream, int32_t variant, bool bf = false, bool append = false,
                 const void* key = nullptr, const void* value = nullptr, int64_t key_stride = 0, int64_t value_stride = 0,
                 int f8 = 0, float kv_scale = 1.0f, const int32_t* bsp = nullptr, void* workspace = nullptr,
                 int64_t workspace_bytes = 0, bool append_no_write = false);
int launch_pa_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query, const void* key_cache,
                 const void* value_cache, int32_t num_seqs, int32_t num_heads, int32_t head_size, int32_t num_kv_heads,
                 float scale, const int32_t* block_tables, const int32_t* seq_lens, int32_t block_size,
                 int32_t max_seq_len, int32_t max_num_blocks_per_seq, const float* alibi_slopes, int64_t q_stride,
                 int64_t kv_block_stride, int64_t kv_head_stride, int32_t device, void* stream, int32_t variant,
                 bool bf = false, int f8 = 0, float kv_scale = 1.0f, const int32_t* bsp = nullptr);
int pick_variant_fp8(int num_seqs, int num_heads, int head_size, int block_size, int max_seq_len, int mean_seq_len,
                     bool bf = false, int fmt = 1, bool unit_scale = false);
int reshape_and_cache_fp8_impl(const void* key, const void* value, void* key_cache, void* value_cache,
                               const int64_t* slot_mapping, int32_t num_tokens, int32_t num_heads, int32_t head_size,
                               int32_t block_size, int32_t x, int64_t key_stride, int64_t value_stride, float kv_scale,
                               int32_t device, void* stream, bool bf, bool e5 = false);

}  // namespace vmi
