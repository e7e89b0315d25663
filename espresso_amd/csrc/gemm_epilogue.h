// Fused GEMM epilogue on 8 consecutive output columns of one row: shared by the 4-wave kernels of gemm.hip and the 8-wave
// large-tile kernel of gemm_w8.hip (bias, activation, dropout, residual, second output, query split, split-K slabs, fp32 output).
#pragma once
#include "common.h"
#include "espresso_amd.h"
#include "gemm_common.h"

namespace {

// ---- epilogue on 8 consecutive output columns of one row --------------------------------------
__device__ __forceinline__ void load8_bf16(const bf16_t* q, bool vec, int cnt, float (&o)[8]) {
  if (vec) {
    const uint4 u = *reinterpret_cast<const uint4*>(q);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] = __uint_as_float(w[e] << 16);
      o[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = e < cnt ? bf2f(q[e]) : 0.f;
  }
}
__device__ __forceinline__ void unpack8_bf16(const uint4& u, float (&o)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[2 * e] = __uint_as_float(w[e] << 16);
    o[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
  }
}
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store8_bf16(bf16_t* q, bool vec, int cnt, const float (&v)[8], bool nt = false) {
  if (vec) {
    u32x4_t u;
    u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
    // large outputs are streamed past the L2 (the consumer is another kernel, mostly on another XCD): their write-allocate
    // otherwise evicts the stationary operand the co-resident tiles share
    if (nt) __builtin_nontemporal_store(u, reinterpret_cast<u32x4_t*>(q));
    else *reinterpret_cast<u32x4_t*>(q) = u;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (e < cnt) q[e] = f2bf(v[e]);
  }
}

// bias of the 8 output columns a thread owns (the same columns in every pass of the epilogue): loaded once, 2 x 16 bytes
__device__ __forceinline__ void load_bias8(const EaGemmParams& p, int n, float (&b)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) b[e] = 0.f;
  if (!p.bias || n >= p.N) return;
  if (n + 8 <= p.N && (((uintptr_t)p.bias) & 15) == 0) {
    const float4 x0 = *reinterpret_cast<const float4*>(p.bias + n);
    const float4 x1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    b[0] = x0.x; b[1] = x0.y; b[2] = x0.z; b[3] = x0.w; b[4] = x1.x; b[5] = x1.y; b[6] = x1.z; b[7] = x1.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (n + e < p.N) b[e] = p.bias[n + e];
  }
}

// positional biases of the 8 query columns a thread owns (query-split epilogue): loaded once like the bias (they were 16 scalar
// loads per 8-column chunk and pass: 38.8 us against 26.2 us for the same projection with the plain epilogue, isolated)
__device__ __forceinline__ void load_pos8(const EaGemmParams& p, int n, float (&u)[8], float (&w)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) u[e] = w[e] = 0.f;
  if (!p.q_u || n >= p.qsplit_n) return;  // (qsplit_n % 128 == 0: a thread's 8 columns are all query columns or none)
  const bool al = ((((uintptr_t)p.pos_u) | ((uintptr_t)p.pos_v)) & 15) == 0;
  if (p.pos_u) {
    if (al) {
      const float4 x0 = *reinterpret_cast<const float4*>(p.pos_u + n), x1 = *reinterpret_cast<const float4*>(p.pos_u + n + 4);
      u[0] = x0.x; u[1] = x0.y; u[2] = x0.z; u[3] = x0.w; u[4] = x1.x; u[5] = x1.y; u[6] = x1.z; u[7] = x1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) u[e] = p.pos_u[n + e];
    }
  }
  if (p.pos_v) {
    if (al) {
      const float4 x0 = *reinterpret_cast<const float4*>(p.pos_v + n), x1 = *reinterpret_cast<const float4*>(p.pos_v + n + 4);
      w[0] = x0.x; w[1] = x0.y; w[2] = x0.z; w[3] = x0.w; w[4] = x1.x; w[5] = x1.y; w[6] = x1.z; w[7] = x1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = p.pos_v[n + e];
    }
  }
}

// FAST (host-checked, fast_epilogue_ok): bf16 output, no split-K, N a multiple of 128 and every operand of the epilogue 16-byte
// aligned with 8-element pitches — the chunk is always whole and vector-accessible, so the ragged / scalar / fp32-output /
// split-K paths (three quarters of the kernel's 10 000 instructions, re-walked by every workgroup) are not instantiated.
template <bool FAST = false>
__device__ __forceinline__ void epilogue_chunk(const EaGemmParams& p, int z, int ks_id, int zhi, int zlo, long coff, int m,
                                               int n, float (&v)[8], const float (&bias8)[8], const float (&posu8)[8],
                                               const float (&posv8)[8], bool vec_ok, bool nt = false, int pre_kind = 0,
                                               uint4 pre = uint4{0, 0, 0, 0}) {
  // pre_kind (FAST only): 1 = `pre` holds this chunk's 8 residual values, 2 = its 8 auxiliary values — fetched by the kernel for
  // all four passes at once before the accumulators go through LDS, instead of one dependent round trip per pass in here
  const int cnt = FAST ? 8 : min(8, p.N - n);
  const bool vec = FAST ? true : (vec_ok && cnt == 8);
  const bool has_drop = p.drop_thr != 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bias8[e];
  if (p.q_u && n < p.qsplit_n) {
    // rel-pos query columns (whole 128-column tiles: the branch is uniform per workgroup): q is rounded to bf16 exactly as the
    // plain epilogue would store it, then the two biased, scaled copies the attention kernels read are written instead of it
    float u8[8], b8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float q = __uint_as_float((uint32_t)f2bf(v[e]) << 16);
      u8[e] = (q + posu8[e]) * p.qscale;
      b8[e] = (q + posv8[e]) * p.qscale;
    }
    const bool qvec = FAST ? true : ((p.ld_q & 7) == 0 && ((((uintptr_t)p.q_u) | ((uintptr_t)p.q_v)) & 15) == 0);
    store8_bf16(reinterpret_cast<bf16_t*>(p.q_u) + (long)m * p.ld_q + n, qvec, 8, u8, nt);
    if (p.q_v) store8_bf16(reinterpret_cast<bf16_t*>(p.q_v) + (long)m * p.ld_q + n, qvec, 8, b8, nt);
    return;
  }
  const uint64_t didx = ((uint64_t)z * (uint64_t)p.M + (uint64_t)m) * (uint64_t)p.N + (uint64_t)n;
  const long co = coff + (long)m * p.ldc + n;
  if (!FAST && p.splitk > 1) {  // split-K partial slab [ks][batch][M][N] fp32 (dense, ld = N); combined by splitk_reduce_kernel
    float* W = reinterpret_cast<float*>(p.workspace) + (((long)ks_id * p.batch + z) * p.M + m) * (long)p.N + n;
    if (cnt == 8 && (p.N & 3) == 0) {
      *reinterpret_cast<float4*>(W) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(W + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < cnt) W[e] = v[e];
    }
    return;
  }
  float keep8[8];
  if (has_drop) {
    ea_keep8(p.drop_seed, didx, p.drop_thr, p.drop_scale, keep8);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) keep8[e] = 1.f;  // (x * 1.0f is exact: one uniform branch here instead of one per element below)
  }
  if (p.aux) {
    float zz[8];
    if (FAST && pre_kind == 2) unpack8_bf16(pre, zz);
    else
    load8_bf16(reinterpret_cast<const bf16_t*>(p.aux) + (long)zhi * p.sX_hi + (long)zlo * p.sX_lo + (long)m * p.ldaux + n,
               FAST ? true : (vec && (p.ldaux & 7) == 0 && ((((uintptr_t)p.aux) & 15) == 0) && (((p.sX_hi | p.sX_lo) & 7) == 0)), cnt, zz);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= keep8[e];
    mul_dact8(v, zz, p.act);
  } else {
    if (p.C2) {
      // the pre-activation copy is only read again by the backward pass, milliseconds and gigabytes later: streamed past the caches
      // so that it does not evict the operands and the activated copy the next GEMM reads
      store8_bf16(reinterpret_cast<bf16_t*>(p.C) + co, vec, cnt, v, true);
      apply_act8(v, p.act);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= keep8[e];
      store8_bf16(reinterpret_cast<bf16_t*>(p.C2) + coff + (long)m * p.ldc2 + n,
                  FAST ? true : (vec && (p.ldc2 & 7) == 0 && ((((uintptr_t)p.C2) & 15) == 0)), cnt, v, nt);
      return;
    }
    apply_act8(v, p.act);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= keep8[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
  if (p.resid) {
    const long ro = (long)zhi * p.sR_hi + (long)zlo * p.sR_lo + (long)m * p.ldr + n;
    if (!FAST && p.resid_f32) {
      const float* r = reinterpret_cast<const float*>(p.resid) + ro;
      if (cnt == 8 && ((((uintptr_t)r) & 15) == 0)) {
        const float4 r0 = *reinterpret_cast<const float4*>(r), r1 = *reinterpret_cast<const float4*>(r + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) if (e < cnt) v[e] += r[e];
      }
    } else {
      float rr[8];
      if (FAST && pre_kind == 1) unpack8_bf16(pre, rr);
      else
      load8_bf16(reinterpret_cast<const bf16_t*>(p.resid) + ro,
                 FAST ? true : (vec && (p.ldr & 7) == 0 && ((((uintptr_t)p.resid) & 15) == 0) && (((p.sR_hi | p.sR_lo) & 7) == 0)), cnt, rr);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rr[e];
    }
  }
  if (!FAST && p.c_f32) {
    float* C = reinterpret_cast<float*>(p.C) + co;
    if (cnt == 8 && ((((uintptr_t)C) & 15) == 0)) {
      if (p.accumulate) {
        const float4 c0 = *reinterpret_cast<const float4*>(C), c1 = *reinterpret_cast<const float4*>(C + 4);
        v[0] += c0.x; v[1] += c0.y; v[2] += c0.z; v[3] += c0.w; v[4] += c1.x; v[5] += c1.y; v[6] += c1.z; v[7] += c1.w;
      }
      *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(C + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < cnt) C[e] = p.accumulate ? C[e] + v[e] : v[e];
    }
  } else {
    store8_bf16(reinterpret_cast<bf16_t*>(p.C) + co, vec, cnt, v, nt);
  }
}

}  // namespace
