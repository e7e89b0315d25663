// Shared device helpers for the espresso_amd HIP kernels (gfx950 / CDNA4 only).
// Wavefront = 64 lanes; all reductions below are written for that width.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#define EA_WAVE 64

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf2f(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even, NaN preserved (same rounding torch uses for .to(bfloat16)): gfx950 converts in hardware
// (v_cvt_pk_bf16_f32, one instruction per pair instead of ~6 VALU ops per element in software)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __bf16 bf16v2_t __attribute__((ext_vector_type(2)));
  typedef float f32v2_t __attribute__((ext_vector_type(2)));
  const f32v2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16v2_t));
}

// Sum over the 64 lanes, returned to every lane, on the DPP path: four row-local steps (quad permutes, row rotates), two
// row-broadcast steps, one v_readlane — ~7 dependent VALU operations instead of six ds_bpermute round trips through the LDS
// crossbar (what __shfl_xor compiles to).  The summation ORDER differs from wave_sum's butterfly (fp32: last-bit differences).
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto dpp = [](float x, auto ctrl, auto rmask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rmask)::value, 0xF, false));
  };
  using std::integral_constant;
  v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xF>{});   // quad_perm [1,0,3,2]
  v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xF>{});   // quad_perm [2,3,0,1]
  v += dpp(v, integral_constant<int, 0x124>{}, integral_constant<int, 0xF>{});  // row_ror:4
  v += dpp(v, integral_constant<int, 0x128>{}, integral_constant<int, 0xF>{});  // row_ror:8  -> every lane holds its row's sum
  v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{});  // row_bcast:15 into rows 1 and 3
  v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{});  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<= 1024). `sm` needs 16 floats.
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sm[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  float r = -INFINITY;
  for (int i = 0; i < nw; ++i) r = fmaxf(r, sm[i]);
  return r;
}

// Counter-based dropout RNG: a 32-bit avalanche mix (murmur3 finaliser) of (seed, element index) -> uniform 32 bits.
// The same (seed, idx) is re-evaluated in the backward kernels, so no mask is ever stored.  All arithmetic is 32-bit
// (gfx950 has no 64-bit integer multiplier: a splitmix64 costs ~35 VALU ops per element, this one ~10, and the dropout
// of the attention probabilities made the fused attention kernels VALU-bound); the high halves of seed / idx only
// enter through a term that is loop invariant in every caller.
__host__ __device__ __forceinline__ uint32_t ea_hash(uint64_t seed, uint64_t idx) {
  const uint32_t hi = ((uint32_t)(idx >> 32) * 0x85EBCA77u) ^ (uint32_t)(seed >> 32) ^ ((uint32_t)seed * 0xC2B2AE3Du);
  uint32_t x = (uint32_t)idx * 0x9E3779B1u + (uint32_t)seed;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= hi;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
// keep-scale: returns 0 if dropped, 1/(1-p) if kept.  thr = p * 2^32 (host computed).
__device__ __forceinline__ float ea_keep(uint64_t seed, uint64_t idx, uint32_t thr, float inv_keep) {
  return ea_hash(seed, idx) >= thr ? inv_keep : 0.f;
}

// (v_rcp_f32, 1 ulp, instead of the IEEE division sequence — two v_div_scale, v_rcp, four FMAs, v_div_fmas, v_div_fixup per
// call: the GEMM epilogues alone held 1 344 of them; the result is rounded to bf16 by every caller)
// keep-scales of 8 CONSECUTIVE elements idx0 .. idx0+7 — bit for bit ea_keep(seed, idx0 + e): the first multiply of the hash is
// linear in the index ((lo + e) * K = lo * K + e * K mod 2^32) and the high-word term is the same for all eight when the low
// word does not wrap, so a chunk costs 1 + 16 quarter-rate multiplies instead of 32 and no 64-bit additions
__device__ __forceinline__ void ea_keep8(uint64_t seed, uint64_t idx0, uint32_t thr, float inv_keep, float (&k)[8]) {
  const uint32_t lo = (uint32_t)idx0;
  if (__builtin_expect(lo > 0xFFFFFFF7u, 0)) {  // the low word wraps inside the chunk: element-wise (laid out of line)
#pragma unroll
    for (int e = 0; e < 8; ++e) k[e] = ea_keep(seed, idx0 + (uint64_t)e, thr, inv_keep);
    return;
  }
  const uint32_t hi = ((uint32_t)(idx0 >> 32) * 0x85EBCA77u) ^ (uint32_t)(seed >> 32) ^ ((uint32_t)seed * 0xC2B2AE3Du);
  uint32_t x0 = lo * 0x9E3779B1u + (uint32_t)seed;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    uint32_t x = x0 + (uint32_t)e * 0x9E3779B1u;
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= hi;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    k[e] = x >= thr ? inv_keep : 0.f;
  }
}

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
  float s = __builtin_amdgcn_rcpf(1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}

__device__ __forceinline__ float log_add(float a, float b) {
  // log(exp(a)+exp(b)) robust to -inf
  float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + log1pf(__expf(-fabsf(a - b)));
}

// EA_DEBUG_SYNC=1 in the environment: print the launch site and wait for the device after every launch (a memory fault then
// aborts right after the line that names the offending launch).
static inline int ea_check_launch(const char* file, int line, const char* func) {
  static const int dbg = [] { const char* e = getenv("EA_DEBUG_SYNC"); return e && e[0] == '1' ? 1 : 0; }();
  if (dbg) {
    fprintf(stderr, "[ea] %s:%d %s\n", file, line, func);
    fflush(stderr);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
#define EA_CHECK_LAUNCH() ea_check_launch(__FILE__, __LINE__, __func__)
