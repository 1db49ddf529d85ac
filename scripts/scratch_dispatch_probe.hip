// What does a private segment (scratch) cost a kernel at DISPATCH?  Same kernel body — every wave spins for a fixed
// number of clock ticks — with 0 / 8 / 16 / 64 / 512 bytes of scratch per lane RESERVED but never touched at run time, 768 workgroups of 256 threads, timed
// by HIP events around each of 300 back-to-back launches.  Build: hipcc --offload-arch=gfx950 -O3 -o
// scripts/_bin/scratch_dispatch_probe scripts/scratch_dispatch_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

template <int W>
__global__ void __launch_bounds__(256) spin(long ticks, int* sink, int k) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if constexpr (W > 0) {
    if (k < 0) {  // never true at run time: the private segment is reserved for the dispatch and never touched
      volatile int priv[W];
      for (int i = 0; i < W; ++i) priv[i] = i + k;
      int s = 0;
      for (int i = 0; i < W; ++i) s += priv[(i - k) % W];
      *sink = s;
    }
  }
}

template <int W>
void run(const char* tag, int* sink) {
  const int iters = 300;
  std::vector<hipEvent_t> a(iters), b(iters);
  for (int i = 0; i < iters; ++i) {
    hipEventCreate(&a[i]);
    hipEventCreate(&b[i]);
  }
  const long ticks = 100 * 50;  // wall_clock64 runs at 100 MHz: 50 us
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(spin<W>, dim3(768), dim3(256), 0, 0, ticks, sink, i);
  hipDeviceSynchronize();
  for (int i = 0; i < iters; ++i) {
    hipEventRecord(a[i], 0);
    hipLaunchKernelGGL(spin<W>, dim3(768), dim3(256), 0, 0, ticks, sink, i);
    hipEventRecord(b[i], 0);
  }
  hipDeviceSynchronize();
  std::vector<float> ms(iters);
  for (int i = 0; i < iters; ++i) hipEventElapsedTime(&ms[i], a[i], b[i]);
  std::sort(ms.begin(), ms.end());
  double sum = 0;
  for (float x : ms) sum += x;
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(spin<W>));
  printf("%-28s scratch %4zu B/lane  mean %7.2f us  median %7.2f  min %7.2f\n", tag, (size_t)fa.localSizeBytes, sum / iters * 1e3,
         ms[iters / 2] * 1e3, ms[0] * 1e3);
}

int main() {
  int* sink;
  hipMalloc(&sink, 4);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("no private segment", sink);
    run<2>("2 dwords", sink);
    run<4>("4 dwords", sink);
    run<16>("16 dwords", sink);
    run<128>("128 dwords", sink);
  }
  return 0;
}
