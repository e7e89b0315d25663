// Pieces shared by the GEMM kernels (gemm.hip: register-staged / direct-to-LDS / grouped weight-gradient kernels;
// conv_igemm.hip: implicit-GEMM convolutions): the LDS tile image and the activation functions of the fused epilogues.
#pragma once
#include "common.h"
#include "espresso_amd.h"

// Development probe (tools/probes/gemm_timing.hip compiles gemm.hip / gemm_w8.hip with -DEA_GEMM_TIMING): thread 0 of every workgroup stamps the
// 100 MHz device clock at entry / first tile in LDS / end of the k loop / end of the epilogue.  Compiled out of the library.
#ifdef EA_GEMM_TIMING
__device__ unsigned long long* g_ea_timing = nullptr;  // [workgroup][8]
__device__ unsigned long long* g_ea_seg = nullptr;     // [workgroup][half][5] shader-clock cycles per segment of the ping-pong loop
__device__ __forceinline__ void ea_stamp(int slot) {
  if (threadIdx.x == 0 && g_ea_timing) {
    const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    g_ea_timing[wg * 8 + slot] = wall_clock64();
    if (slot == 0) {
      uint32_t hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_ea_timing[wg * 8 + 4] = ((unsigned long long)xcc << 32) | hw;
    }
  }
}
#define EA_STAMP(slot) ea_stamp(slot)
#else
#define EA_STAMP(slot)
#endif


namespace {

constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;  // 128 B per LDS row

// 16-byte chunk c of LDS row r lives at chunk (c ^ (r&7) ^ ((r>>4)&7)).  The (r&7) term makes the ds_read_b128
// fragment reads (16 consecutive rows, fixed c) conflict-free; the (r>>4) term is constant inside such a
// 16-row group (reads unaffected) and spreads the transposing ds_write_b64 of the k-strided staging path,
// whose 16 lanes hit rows 8 apart (same r&7), over 8 different bank slots instead of one.
__device__ __forceinline__ uint32_t lds_off(int row, int chunk) {
  return (uint32_t)(row * ROW_BYTES + ((chunk ^ (row & 7) ^ ((row >> 4) & 7)) << 4));
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == EA_ACT_RELU) return fmaxf(v, 0.f);
  if (act == EA_ACT_SILU) return silu_f(v);
  return v;
}
__device__ __forceinline__ float apply_dact(float z, int act) {
  if (act == EA_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == EA_ACT_SILU) return dsilu_f(z);
  return 1.f;
}
// the same on a chunk of eight, with ONE uniform branch on the activation (the per-element form above compiled to a scalar
// branch per element inside the GEMM epilogue's unrolled loops: ~25 taken branches per 8-column chunk)
__device__ __forceinline__ void apply_act8(float (&v)[8], int act) {
  if (act == EA_ACT_SILU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
  } else if (act == EA_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
}
__device__ __forceinline__ void mul_dact8(float (&v)[8], const float (&z)[8], int act) {  // v *= act'(z)
  if (act == EA_ACT_SILU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= dsilu_f(z[e]);
  } else if (act == EA_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= z[e] > 0.f ? 1.f : 0.f;
  }
}

}  // namespace
