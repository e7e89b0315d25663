// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds u16 values equal to their element index; every lane issues the
// transposing read at a per-lane byte address and the four returned 16-bit values per lane are printed.
//   build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.hip -o tools/probes/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(const int* addr, uint16_t* out, int n_elems) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < n_elems; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)(&lds[0]) + (uint32_t)addr[threadIdx.x];
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = (uint16_t)(v.x & 0xffff);
  out[threadIdx.x * 4 + 1] = (uint16_t)(v.x >> 16);
  out[threadIdx.x * 4 + 2] = (uint16_t)(v.y & 0xffff);
  out[threadIdx.x * 4 + 3] = (uint16_t)(v.y >> 16);
}

static void run(const char* title, int (*f)(int lane)) {
  int h[64];
  for (int l = 0; l < 64; ++l) h[l] = f(l);
  int* d; uint16_t* o; uint16_t ho[256];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, 4096);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  printf("== %s\n", title);
  for (int l = 0; l < 64; ++l) printf("lane %2d addr(elem) %4d -> %4d %4d %4d %4d\n", l, h[l] / 2, ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3]);
  hipFree(d); hipFree(o);
}

int main() {
  // (1) lane-linear 8-byte addresses: lane l at element 4*l
  run("linear: byte addr = 8*lane", [](int l) { return 8 * l; });
  // (2) a [k][n] row-major tile with 64-element rows (128 B): lane -> row (l&15)/4 + 4*(l>>4)... i.e. 16-lane group g covers k rows 4g..4g+3, cols (l&3)*4
  run("tile64: addr = ((4*(l>>4) + ((l&15)>>2))*64 + (l&3)*4)*2", [](int l) { return ((4 * (l >> 4) + ((l & 15) >> 2)) * 64 + (l & 3) * 4) * 2; });
  return 0;
}
