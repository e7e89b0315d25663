// Development probe: what does the chip sustain for pure 16-byte-per-lane stores (the GEMM epilogue's output stream)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/hbm_write_probe.hip -o tools/probes/hbm_write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0 plain store, 1 non-temporal store, 2 load (sum), 3 copy
__global__ __launch_bounds__(256) void stream_kernel(u32x4* __restrict__ dst, const u32x4* __restrict__ src, size_t n16, unsigned* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    if (MODE == 0) dst[i] = u32x4{(unsigned)i, 1, 2, 3};
    else if (MODE == 1) __builtin_nontemporal_store(u32x4{(unsigned)i, 1, 2, 3}, dst + i);
    else if (MODE == 2) acc += src[i];
    else dst[i] = src[i];
  }
  if (MODE == 2 && acc.x == 0x12345678u) *sink = acc.y;
}

int main() {
  const size_t MAXB = 1ull << 30;
  u32x4 *d, *s;
  unsigned* sink;
  CK(hipMalloc(&d, MAXB)); CK(hipMalloc(&s, MAXB)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(d, 0, MAXB)); CK(hipMemset(s, 1, MAXB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sizes[] = {6ull << 20, 25ull << 20, 51ull << 20, 128ull << 20, 512ull << 20};
  const int grids[] = {200, 256, 512, 1024, 2048, 4096};
  const char* names[] = {"store", "nt store", "load", "copy"};
  for (int mode = 0; mode < 4; ++mode)
    for (size_t bytes : sizes) {
      printf("%-9s %4zu MB:", names[mode], bytes >> 20);
      for (int g : grids) {
        auto launch = [&]() {
          if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(g), dim3(256), 0, 0, d, s, bytes / 16, sink);
          else if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(g), dim3(256), 0, 0, d, s, bytes / 16, sink);
          else if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(g), dim3(256), 0, 0, d, s, bytes / 16, sink);
          else hipLaunchKernelGGL(stream_kernel<3>, dim3(g), dim3(256), 0, 0, d, s, bytes / 16, sink);
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        const int it = 10;
        for (int i = 0; i < it; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / it;
        printf("  g%-4d %6.1f us %5.2f TB/s |", g, us, (mode == 3 ? 2.0 : 1.0) * bytes / us * 1e-6);
      }
      printf("\n");
    }
  return 0;
}
