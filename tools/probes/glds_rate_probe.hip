// Development probe: how many bytes per microsecond does ONE workgroup per CU pull from the L2 into LDS with global_load_lds
// (and into registers with global_load_dwordx4), as a function of wavefronts per workgroup and instructions in flight per wavefront?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/glds_rate_probe.hip -o tools/probes/glds_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// every workgroup reads `span` bytes starting at src + (blockIdx.x % nreg) * span, `iters` times over; DEPTH instructions per wavefront
// between waits.  MODE 0: global_load_lds into a per-wave LDS slot ring, 1: global_load_dwordx4 to registers (xor-accumulated)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void pull_kernel(const char* __restrict__ src, size_t span, int nreg, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = src + (size_t)(blockIdx.x % nreg) * span;
  const size_t per_round = (size_t)nw * DEPTH * 1024;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    for (size_t off = 0; off + per_round <= span; off += per_round) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const char* q = base + off + ((size_t)(d * nw + wave)) * 1024 + lane * 16;
        if (MODE == 0) __builtin_amdgcn_global_load_lds((gptr_t)q, (lptr_t)(dsm + (wave * DEPTH + d) * 1024), 16, 0, 0);
        else acc ^= *reinterpret_cast<const u32x4*>(q);
      }
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (MODE == 1 && acc.x == 0x12345u) *sink = acc.y;
  if (MODE == 0 && dsm[threadIdx.x] == 77 && iters < 0) *sink = 1;
}

int main() {
  const size_t BYTES = 512ull << 20;
  char* s; unsigned* sink;
  CK(hipMalloc(&s, BYTES)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(s, 1, BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int lds = 128 * 1024;
  auto run = [&](int mode, int depth, int nw, size_t span, int nreg, int grid) {
    const int iters = (int)((64ull << 20) / span) > 0 ? (int)((64ull << 20) / span) : 1;  // 64 MB pulled per workgroup
    auto launch = [&]() {
#define L(M, D) hipLaunchKernelGGL((pull_kernel<M, D>), dim3(grid), dim3(nw * 64), lds, 0, s, span, nreg, iters, sink)
      if (mode == 0 && depth == 4) L(0, 4); else if (mode == 0 && depth == 8) L(0, 8); else if (mode == 0 && depth == 16) L(0, 16);
      else if (mode == 1 && depth == 4) L(1, 4); else if (mode == 1 && depth == 8) L(1, 8); else L(1, 16);
    };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes_wg = (double)(span / ((size_t)nw * depth * 1024)) * nw * depth * 1024 * iters;
    printf("  %s depth %2d waves %2d span %5zu KB x %3d regions grid %3d: %7.1f GB/s per CU, %6.2f TB/s chip\n", mode ? "gload" : "glds ", depth, nw,
           span >> 10, nreg, grid, bytes_wg / (ms * 1e-3) * 1e-9, bytes_wg * grid / (ms * 1e-3) * 1e-12);
  };
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pull_kernel<0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pull_kernel<0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pull_kernel<0, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pull_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pull_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pull_kernel<1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  printf("# every workgroup (one per CU: 128 KB of LDS) re-reads its own L2-resident region\n");
  for (int mode = 0; mode < 2; ++mode)
    for (int nw : {4, 8, 16})
      for (int depth : {4, 8, 16}) {
        if (nw * depth * 1024 > lds) continue;
        run(mode, depth, nw, 512 << 10, 8, 256);    // 8 regions of 512 KB: 4 MB total, L2-resident in every XCD
      }
  printf("# one CU alone\n");
  for (int mode = 0; mode < 2; ++mode)
    for (int nw : {8, 16}) run(mode, 8, nw, 512 << 10, 8, 1);
  printf("# 200 workgroups, regions from a 64 MB set (infinity-cache resident, not L2)\n");
  for (int mode = 0; mode < 2; ++mode)
    for (int nw : {8, 16}) run(mode, 8, nw, 256 << 10, 256, 200);
  return 0;
}
