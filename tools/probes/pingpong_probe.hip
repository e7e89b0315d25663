// Development probe: cost of one barrier-delimited phase of an 8-wavefront workgroup (one per CU) in which
//   mode 0: all wavefronts issue 32 MFMAs                         mode 1: all wavefronts read 12 fragments (ds_read_b128)
//   mode 2: half A reads 12 fragments while half B issues 32 MFMAs, roles swap every phase (ping-pong)
//   mode 3: mode 2 + the reading half requests 16 KB per phase by global_load_lds (L2-resident source)
//   mode 4: every wavefront: 12 reads then 32 MFMAs (no role split)          mode 5: mode 4 + 8 KB of requests per wavefront every 4th phase
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/pingpong_probe.hip -o tools/probes/pingpong_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>
__global__ __launch_bounds__(512, 2) void pp_kernel(const char* __restrict__ src, int phases, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = wave >> 2, wn = wave & 3;
  for (int i = tid; i < 128 * 1024 / 16; i += 512) reinterpret_cast<uint4*>(dsm)[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  __syncthreads();
  f32x4_t acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa[8], fb[4];
  for (int i = 0; i < 8; ++i) fa[i] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 4; ++j) fb[j] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
  const uint32_t a_base = (half * 128 + (lane & 15)) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
  const uint32_t b_base = 32768 + (wn * 64 + (lane & 15)) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
  const char* gsrc = src + (size_t)(blockIdx.x & 7) * (512 << 10) + wn * 1024 + lane * 16;
  auto rd = [&](int ph) {
    const char* st = dsm + (ph & 2 ? 65536 : 0) + (ph & 1 ? 64 : 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(st + b_base + j * 2048);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(st + a_base + i * 2048);
    asm volatile("" ::: "memory");
  };
  auto mma = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[i]),
                                                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[j]), acc[i][j], 0, 0, 0);
  };
  auto req = [&](int ph, int n) {  // n x 1 KB into the stage not being read
    char* base = dsm + (ph & 2 ? 0 : 65536) + half * 32768 + wn * 1024;
    for (int i = 0; i < n; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + ((size_t)((ph * 8 + i) & 63)) * 4096), (lptr_t)(base + i * 4096), 16, 0, 0);
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    for (int ph = 0; ph < phases; ++ph) { mma(); __builtin_amdgcn_s_barrier(); }
  } else if (MODE == 1) {
    for (int ph = 0; ph < phases; ++ph) { rd(ph); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  } else if (MODE == 2 || MODE == 3) {
    if (half == 0) {
      for (int ph = 0; ph < phases; ph += 2) {
        rd(ph); if (MODE == 3) req(ph, 4); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
        mma(); if (MODE == 3 && (ph & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
      }
    } else {
      for (int ph = 0; ph < phases; ph += 2) {
        mma(); __builtin_amdgcn_s_barrier();
        rd(ph + 1); if (MODE == 3) req(ph + 1, 4); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE == 3 && (ph & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  } else {
    for (int ph = 0; ph < phases; ++ph) {
      if (MODE == 5 && (ph & 3) == 0) req(ph, 8);
      rd(ph); mma();
      if (MODE == 5 && (ph & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  char* src; float* out; unsigned long long* cyc;
  CK(hipMalloc(&src, 8 << 20)); CK(hipMemset(src, 0x3c, 8 << 20));
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8));
  const int lds = 128 * 1024, phases = 256, grid = 200;
  unsigned long long h[256];
  const char* names[] = {"all: 32 MFMAs", "all: 12 ds_read_b128", "ping-pong: 12 reads | 32 MFMAs", "ping-pong + 16 KB of global_load_lds per phase",
                         "all: 12 reads then 32 MFMAs", "all: 12 reads, 32 MFMAs, 64 KB of global_load_lds per 4 phases"};
#define RUN(M) do { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pp_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(pp_kernel<M>, dim3(grid), dim3(512), lds, 0, src, phases, out, cyc); \
    CK(hipDeviceSynchronize()); CK(hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost)); \
    double s = 0; for (int i = 0; i < grid; ++i) s += h[i]; \
    printf("mode %d  %-62s %7.1f cycles per phase\n", M, names[M], s / grid / phases); } while (0)
  RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5);
  return 0;
}
