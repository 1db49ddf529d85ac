// pa_host.hpp — host-side helpers shared by the translation units that define C-ABI entries (paged_attention.hip and the
// out-of-scope units pa_f32.hip / pa_extras_cache.hip / pa_extras_absent.hip): the thread's last error text, the error
// returns, the device guard.  Defined in paged_attention.hip.
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

}  // namespace vmi
