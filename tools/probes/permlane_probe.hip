// Development probe: lane semantics of v_permlane16_swap_b32 / v_permlane32_swap_b32 on gfx950 (r = swap(vdst = a, src = b))
//   hipcc --offload-arch=gfx950 -O3 tools/probes/permlane_probe.hip -o tools/probes/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0];
  o[threadIdx.x + 64] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x + 128] = s[0];
  o[threadIdx.x + 192] = s[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"permlane16_swap r[0] (vdst)", "permlane16_swap r[1] (src)", "permlane32_swap r[0] (vdst)", "permlane32_swap r[1] (src)"};
  for (int q = 0; q < 4; ++q) {
    printf("%s: (a = lane, b = lane + 100)\n", names[q]);
    for (int l = 0; l < 64; ++l) printf("%4u%s", h[q * 64 + l], (l & 15) == 15 ? "\n" : "");
  }
  return 0;
}
