// Pieces shared by the 8-wavefront kernels (gemm_w8.hip: the large-tile GEMMs; joint_rnnt.hip: the transducer joint's vocabulary
// projection fused with the RNN-T loss): LDS image of the direct-to-LDS ring, counted waits, the pinned instruction order of the
// software-pipelined k-tile, and the 16-byte stores of the register epilogue.
#pragma once
#include "common.h"
#include "gemm_common.h"

#include "gemm_epilogue.h"  // (u32x4_t)

namespace {

// LDS image of an operand tile: [rows][64 k] bf16, 128-byte rows; the 16-byte slot s of row r holds k-chunk s ^ (r & 7): the 16 rows
// x one chunk of a ds_read_b128 fragment read fall on 16 different bank slots, and every 16-row group uses the same permutation, so
// the fragments of a wavefront are ONE base address + immediates.
__device__ __forceinline__ uint32_t w8_off(int row, int chunk) { return (uint32_t)(row * 128 + ((chunk ^ (row & 7)) << 4)); }

template <int N>
__device__ __forceinline__ void w8_wait_vmcnt() {
  static_assert(N == 0 || N == 2 || N == 3 || N == 4 || N == 6 || N == 8 || N == 9 || N == 12 || N == 16 || N == 18, "vmcnt value not listed");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
}

// instruction order of the pipelined k-tile body: [reads of group g + 1][MFMAs of group g] for g = 0 .. NG - 1
template <int NG, int GPK, int AG, int NJ, int G = 0>
__device__ __forceinline__ void w8_sched_groups() {
  if constexpr (G < NG) {
    if constexpr (G + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, AG + ((G + 1) % GPK == 0 ? NJ : 0), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, AG * NJ, 0);
    w8_sched_groups<NG, GPK, AG, NJ, G + 1>();
  }
}

// (x, y) of tile j and of tile j + 1 -> this lane's 16-byte piece
__device__ __forceinline__ u32x4_t w8_swap_pair(uint2 tj, uint2 tj1) {
  const auto r0 = __builtin_amdgcn_permlane16_swap(tj.x, tj1.x, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(tj.y, tj1.y, false, false);
  u32x4_t o;
  o.x = r0[0]; o.y = r1[0]; o.z = r0[1]; o.w = r1[1];
  return o;
}
__device__ __forceinline__ void w8_store16(bf16_t* q, u32x4_t u, bool nt) {
  if (nt) __builtin_nontemporal_store(u, reinterpret_cast<u32x4_t*>(q));
  else *reinterpret_cast<u32x4_t*>(q) = u;
}


}  // namespace
