// LayerNorm forward/backward (eps 1e-5, affine) for gfx950.
// Replaces fairseq/modules/layer_norm.py:28-33 (apex FusedLayerNorm / torch.nn.LayerNorm) as used
// by every Conformer/Transformer block (fairseq/modules/conformer_layer.py:86,141;
// espresso/modules/conformer_with_relative_positional_embedding_encoder_layer.py:46,70) and by
// layernorm_embedding + dropout + pad-zeroing in
// espresso/models/transformer/speech_transformer_encoder.py:348-357.
//
// HBM-bound: one wavefront per row, 16-byte loads (8 bf16 per lane per pass), fp32 statistics via
// 64-lane butterfly reductions, no LDS.  Backward re-reads x and dy once, writes dx once, and
// accumulates dgamma/dbeta per block in registers -> LDS -> one partial row per block in a workspace slab that a
// second small kernel folds into the gradient (fp32 global atomics run at ~33 G/s on gfx950: 2*C atomics per block
// cost more than the whole streaming pass; the atomic path remains for callers without a workspace).
#include "common.h"
#include "espresso_amd.h"

namespace {

constexpr int MAXC8_LIMIT = 4;  // supports C <= 64*8*4 = 2048

// eight consecutive elements of a bf16 or fp32 row <-> fp32 registers (fp32 islands of the reference's autocast run: the joint
// network's LayerNorm outputs and their gradients stay fp32, speech_transformer_transducer_base.py:292-294)
template <typename T>
__device__ inline void load8(const T* p, float (&v)[8]) {
  if constexpr (sizeof(T) == 2) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = __uint_as_float(w[e] << 16);
      v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
  } else {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
template <typename T>
__device__ inline void store8(T* p, const float (&o)[8]) {
  if constexpr (sizeof(T) == 2) {
    uint4 u;
    u.x = pack_bf2(o[0], o[1]); u.y = pack_bf2(o[2], o[3]); u.z = pack_bf2(o[4], o[5]); u.w = pack_bf2(o[6], o[7]);
    *reinterpret_cast<uint4*>(p) = u;
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
}

// The LayerNorm backward arithmetic of eight columns, shared by ln_bwd_kernel and ln_bwd2_kernel so that the paired kernel is
// bit-identical to the two launches it replaces: contraction is switched off and every fused multiply-add is spelled out (left to
// the compiler, a*b+c contracts differently in the two kernels and the bf16 rounding of dx flips on ~1e-4 of the elements).
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __uint_as_float(w[e] << 16);
    f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
  }
}
// xhat, g = dy * gamma; dgamma += dy * xhat, dbeta += dy; row sums s1 += g, s2 += g * xhat (this lane's share)
__device__ __forceinline__ void ln_bwd_accum8(const uint4& ux, const float (&dv)[8], const float (&gam)[8], float mean, float rstd,
                                              float (&dg)[8], float (&db)[8], float (&xh)[8], float (&g)[8], float& s1, float& s2) {
#pragma clang fp contract(off)
  float xv[8];
  unpack8(ux, xv);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float h = (xv[e] - mean) * rstd;
    xh[e] = h;
    dg[e] = __builtin_fmaf(dv[e], h, dg[e]);
    db[e] = db[e] + dv[e];
    const float gv = dv[e] * gam[e];
    g[e] = gv;
    s1 = s1 + gv;
    s2 = __builtin_fmaf(gv, h, s2);
  }
}
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ residual-path gradient]
__device__ __forceinline__ void ln_bwd_finish8(const float (&xh)[8], const float (&g)[8], float rstd, float s1, float s2,
                                               const uint4* add, float (&o)[8]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = rstd * __builtin_fmaf(-xh[e], s2, g[e] - s1);
  if (add) {
    float a[8];
    unpack8(*add, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = o[e] + a[e];
  }
}

// rows are processed one per wave. C % 8 == 0 required.  MAXC8 = 16-byte chunks per lane (C <= 512*MAXC8).
template <int MAXC8, typename TY = bf16_t>
__global__ __launch_bounds__(256) void ln_fwd_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    TY* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int M, int C,
    float eps, const uint8_t* __restrict__ row_zero, uint64_t seed, uint32_t thr, float inv_keep) {
  const int lane = threadIdx.x & 63;
  const int nch = C >> 3;  // 16-byte chunks per row
  // affine parameters of this lane's chunks stay in registers across the rows the wavefront walks (re-reading 2*C fp32 per
  // row is 4x the bf16 row itself)
  float ga[MAXC8][8], be[MAXC8][8];
#pragma unroll
  for (int i = 0; i < MAXC8; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ga[i][e] = ch < nch ? gamma[ch * 8 + e] : 0.f;
      be[i][e] = ch < nch ? beta[ch * 8 + e] : 0.f;
    }
  }
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
  float v[MAXC8][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC8; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const uint4 u = *reinterpret_cast<const uint4*>(x + (long)row * C + ch * 8);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][2 * e] = __uint_as_float(w[e] << 16);
        v[i][2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
        s += v[i][2 * e] + v[i][2 * e + 1];
      }
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC8; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
  const bool zero = row_zero && row_zero[row];
#pragma unroll
  for (int i = 0; i < MAXC8; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = ch * 8 + e;
        float t = (v[i][e] - mean) * rstd * ga[i][e] + be[i][e];
        if (thr) t *= ea_keep(seed, (uint64_t)row * C + c, thr, inv_keep);
        o[e] = zero ? 0.f : t;
      }
      store8(y + (long)row * C + ch * 8, o);
    }
  }
  }
}

// C <= 512: a wavefront takes RW rows AT ONCE (one 16-byte chunk per lane and row): the RW loads are in flight together and
// the 2*RW butterfly reductions interleave, instead of one dependent load -> reduce -> reduce -> store chain per row (the
// one-row kernel is latency-bound: every wave of the grid is resident, each walks 1-2 rows, 11 us for 6 MB in and 6 MB out).
template <int RW>
__global__ __launch_bounds__(256) void ln_fwd_rows_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    bf16_t* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int M, int C,
    float eps, const uint8_t* __restrict__ row_zero, uint64_t seed, uint32_t thr, float inv_keep) {
  const int lane = threadIdx.x & 63;
  const int nch = C >> 3;
  const bool on = lane < nch;
  float ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ga[e] = on ? gamma[lane * 8 + e] : 0.f;
    be[e] = on ? beta[lane * 8 + e] : 0.f;
  }
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= M) return;
  float v[RW][8], s[RW];
  bool zr[RW];  // padding flags, requested with the rows (a load at its first use, in the store loop, is a round trip per row)
#pragma unroll
  for (int r = 0; r < RW; ++r) zr[r] = row_zero && row_zero[min(row0 + r, M - 1)];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = min(row0 + r, M - 1);
    uint4 u = make_uint4(0, 0, 0, 0);
    if (on) u = *reinterpret_cast<const uint4*>(x + (long)row * C + lane * 8);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    s[r] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[r][2 * e] = __uint_as_float(w[e] << 16);
      v[r][2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
      s[r] += v[r][2 * e] + v[r][2 * e + 1];
    }
  }
  float mean[RW], q[RW];
  const float inv_c = __builtin_amdgcn_rcpf((float)C);  // (one v_rcp per wave instead of an IEEE division sequence per row and sum)
#pragma unroll
  for (int r = 0; r < RW; ++r) mean[r] = wave_sum_dpp(s[r]) * inv_c;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    q[r] = 0.f;
    if (on) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[r][e] - mean[r];
        q[r] += d * d;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = row0 + r;
    const float rstd = rsqrtf(wave_sum_dpp(q[r]) * inv_c + eps);
    if (row >= M) continue;
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean[r];
      if (rstd_out) rstd_out[row] = rstd;
    }
    if (!on) continue;
    const bool zero = zr[r];
    float o[8], k8[8];
    if (thr) ea_keep8(seed, (uint64_t)row * C + lane * 8, thr, inv_keep, k8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = (v[r][e] - mean[r]) * rstd * ga[e] + be[e];
      if (thr) t *= k8[e];
      o[e] = zero ? 0.f : t;
    }
    uint4 u;
    u.x = pack_bf2(o[0], o[1]); u.y = pack_bf2(o[2], o[3]); u.z = pack_bf2(o[4], o[5]); u.w = pack_bf2(o[6], o[7]);
    *reinterpret_cast<uint4*>(y + (long)row * C + lane * 8) = u;
  }
}

// Two LayerNorms back to back over the same rows (round 6): y1 = LN(x; g1, b1) rounded to bf16 — the output of a Conformer layer,
// espresso/modules/conformer_with_relative_positional_embedding_encoder_layer.py:143 — and y2 = LN(y1; g2, b2), the first operation of
// the NEXT layer (fairseq/modules/conformer_layer.py:141 FeedForwardModule.layer_norm).  One launch instead of two and y1 is not read
// back; both results are bit-identical to two ea_layernorm_fwd calls (the second norm starts from the ROUNDED y1).  C <= 512.
template <int RW>
__global__ __launch_bounds__(256) void ln_fwd2_rows_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ g1, const float* __restrict__ b1, bf16_t* __restrict__ y1,
    float* __restrict__ mean1, float* __restrict__ rstd1, const float* __restrict__ g2, const float* __restrict__ b2,
    bf16_t* __restrict__ y2, float* __restrict__ mean2, float* __restrict__ rstd2, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int nch = C >> 3;
  const bool on = lane < nch;
  const int cl = on ? lane : 0;  // (clamped: every parameter load unconditional)
  float ga[8], be[8], gb[8], bb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ga[e] = g1[cl * 8 + e]; be[e] = b1[cl * 8 + e];
    gb[e] = g2[cl * 8 + e]; bb[e] = b2[cl * 8 + e];
  }
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= M) return;
  float v[RW][8], s[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = min(row0 + r, M - 1);
    const uint4 u = *reinterpret_cast<const uint4*>(x + (long)row * C + cl * 8);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    s[r] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[r][2 * e] = on ? __uint_as_float(w[e] << 16) : 0.f;
      v[r][2 * e + 1] = on ? __uint_as_float(w[e] & 0xffff0000u) : 0.f;
      s[r] += v[r][2 * e] + v[r][2 * e + 1];
    }
  }
  const float inv_c = __builtin_amdgcn_rcpf((float)C);
  float mean[RW], q[RW], rs[RW];
  // the same statistics sequence as ln_fwd_rows_kernel, run twice
  auto stats = [&]() {
#pragma unroll
    for (int r = 0; r < RW; ++r) mean[r] = wave_sum_dpp(s[r]) * inv_c;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      q[r] = 0.f;
      if (on) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[r][e] - mean[r];
          q[r] += d * d;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) rs[r] = rsqrtf(wave_sum_dpp(q[r]) * inv_c + eps);
  };
  stats();
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = row0 + r;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (v[r][e] - mean[r]) * rs[r] * ga[e] + be[e];
    uint4 u;
    u.x = pack_bf2(o[0], o[1]); u.y = pack_bf2(o[2], o[3]); u.z = pack_bf2(o[4], o[5]); u.w = pack_bf2(o[6], o[7]);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    s[r] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // the second norm sees what a reader of y1 would see
      v[r][2 * e] = on ? __uint_as_float(w[e] << 16) : 0.f;
      v[r][2 * e + 1] = on ? __uint_as_float(w[e] & 0xffff0000u) : 0.f;
      s[r] += v[r][2 * e] + v[r][2 * e + 1];
    }
    if (row < M) {
      if (lane == 0) { mean1[row] = mean[r]; rstd1[row] = rs[r]; }
      if (on) *reinterpret_cast<uint4*>(y1 + (long)row * C + lane * 8) = u;
    }
  }
  stats();
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = row0 + r;
    if (row >= M) continue;
    if (lane == 0) { mean2[row] = mean[r]; rstd2[row] = rs[r]; }
    if (!on) continue;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (v[r][e] - mean[r]) * rs[r] * gb[e] + bb[e];
    uint4 u;
    u.x = pack_bf2(o[0], o[1]); u.y = pack_bf2(o[2], o[3]); u.z = pack_bf2(o[4], o[5]); u.w = pack_bf2(o[6], o[7]);
    *reinterpret_cast<uint4*>(y2 + (long)row * C + lane * 8) = u;
  }
}

// Each block owns ROWS_PER_BLOCK consecutive rows (4 waves round-robin), accumulates dgamma/dbeta
// partials per lane-column, reduces across the 4 waves through LDS and writes one partial row (or issues atomics).
// optional second output of the backward kernel: out[i] = a * dropout(dx[i]) with its own counter-based mask (the next block's
// "gradient after the residual dropout"), written from the values that are in registers anyway
struct LnOut2 {
  bf16_t* out;
  float a;
  uint64_t seed;
  uint32_t thr;
  float inv_keep;
};
template <int MAXC8, typename TD = bf16_t>
__global__ __launch_bounds__(256) void ln_bwd_kernel(
    const bf16_t* __restrict__ x, const TD* __restrict__ dy, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, bf16_t* __restrict__ dx,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int C, int rows_per_block,
    const uint8_t* __restrict__ row_zero, uint64_t seed, uint32_t thr, float inv_keep,
    const bf16_t* __restrict__ dx_add, float* __restrict__ partial, const LnOut2 o2, const int two_rows) {
  extern __shared__ float red[];  // [4][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 3;
  float dg[MAXC8][8], db[MAXC8][8];
#pragma unroll
  for (int i = 0; i < MAXC8; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) dg[i][e] = db[i][e] = 0.f;

  float gam[MAXC8][8];
#pragma unroll
  for (int i = 0; i < MAXC8; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < 8; ++e) gam[i][e] = ch < nch ? gamma[ch * 8 + e] : 0.f;
  }
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  const float inv_c = __builtin_amdgcn_rcpf((float)C);
  struct RowIn {
    uint4 ux[MAXC8], ua[MAXC8];
    float dy[MAXC8][8];
    float mean, rstd;
    bool zero;
  };
  // every load of the row is requested up front — the residual-path gradient too, which used to be fetched only after the two
  // wave reductions (a second exposed round trip per row), and the padding flag last (its first use is a branch: a wait)
  auto load_row = [&](int row, RowIn& r) {
#pragma unroll
    for (int i = 0; i < MAXC8; ++i) {
      const int ch = lane + 64 * i;
      r.ux[i] = r.ua[i] = uint4{0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 8; ++e) r.dy[i][e] = 0.f;
      if (ch < nch) {
        r.ux[i] = *reinterpret_cast<const uint4*>(x + (long)row * C + ch * 8);
        load8(dy + (long)row * C + ch * 8, r.dy[i]);
        if (dx_add) r.ua[i] = *reinterpret_cast<const uint4*>(dx_add + (long)row * C + ch * 8);
      }
    }
    r.mean = mean_in[row];
    r.rstd = rstd_in[row];
    r.zero = row_zero && row_zero[row];
  };
  auto process_row = [&](int row, const RowIn& r) {
    const float mean = r.mean, rstd = r.rstd;
    const bool zero = r.zero;
    float xh[MAXC8][8], g[MAXC8][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC8; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float k8[8], dv[8];
        if (thr) ea_keep8(seed, (uint64_t)row * C + ch * 8, thr, inv_keep, k8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          dv[e] = zero ? 0.f : r.dy[i][e];
          if (thr) dv[e] *= k8[e];
        }
        ln_bwd_accum8(r.ux[i], dv, gam[i], mean, rstd, dg[i], db[i], xh[i], g[i], s1, s2);
      }
    }
    s1 = wave_sum_dpp(s1) * inv_c;
    s2 = wave_sum_dpp(s2) * inv_c;
#pragma unroll
    for (int i = 0; i < MAXC8; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float o[8];
        ln_bwd_finish8(xh[i], g[i], rstd, s1, s2, dx_add ? &r.ua[i] : nullptr, o);
        uint4 u;
        u.x = pack_bf2(o[0], o[1]);
        u.y = pack_bf2(o[2], o[3]);
        u.z = pack_bf2(o[4], o[5]);
        u.w = pack_bf2(o[6], o[7]);
        *reinterpret_cast<uint4*>(dx + (long)row * C + ch * 8) = u;
        if (o2.out) {
          float dr[8], p2[8], kk8[8];
          unpack8(u, dr);  // the rounded dx
          if (o2.thr) ea_keep8(o2.seed, (uint64_t)row * C + ch * 8, o2.thr, o2.inv_keep, kk8);
#pragma unroll
          for (int e = 0; e < 8; ++e) p2[e] = o2.a * dr[e] * (o2.thr ? kk8[e] : 1.f);
          uint4 u2;
          u2.x = pack_bf2(p2[0], p2[1]);
          u2.y = pack_bf2(p2[2], p2[3]);
          u2.z = pack_bf2(p2[4], p2[5]);
          u2.w = pack_bf2(p2[6], p2[7]);
          *reinterpret_cast<uint4*>(o2.out + (long)row * C + ch * 8) = u2;
        }
      }
    }
  };
  if (MAXC8 == 1 && two_rows) {
    // C <= 512: a wavefront's two rows of the block (r, r + 4) are in flight TOGETHER (round 6; the forward kernel's lesson,
    // ln_fwd_rows_kernel): one after the other, each row was a dependent load -> reduce -> reduce -> store chain, 11.5 us for the
    // recipe's 6 240 x 512 rows.  The rows are still accumulated in the same order: dgamma / dbeta partials are bit-identical.
    for (int row = r0 + wave; row < r1; row += 8) {
      RowIn ra, rb;
      const bool two = row + 4 < r1;  // (uniform)
      load_row(row, ra);
      if (two) load_row(row + 4, rb);
      process_row(row, ra);
      if (two) process_row(row + 4, rb);
    }
  } else {
    for (int row = r0 + wave; row < r1; row += 4) {
      RowIn ra;
      load_row(row, ra);
      process_row(row, ra);
    }
  }
  // cross-wave reduction of dgamma/dbeta
#pragma unroll
  for (int i = 0; i < MAXC8; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 2 + 0) * C + ch * 8 + e] = dg[i][e];
        red[(wave * 2 + 1) * C + ch * 8 + e] = db[i][e];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[(w * 2 + 0) * C + c];
      b += red[(w * 2 + 1) * C + c];
    }
    if (partial) {
      partial[(long)blockIdx.x * 2 * C + c] = a;
      partial[(long)blockIdx.x * 2 * C + C + c] = b;
    } else {
      atomicAdd(dgamma + c, a);
      atomicAdd(dbeta + c, b);
    }
  }
}

// Backward of the pair above in one pass over the rows (C <= 512):
//   d   = bf16( LNbwd(x1, dy1; g1, mean1, rstd1) + dx_add )     (what ea_layernorm_bwd_dx stores as the next layer's input gradient)
//   dx  = bf16( LNbwd(x2, d;   g2, mean2, rstd2) )              (the previous layer's final LayerNorm, its incoming gradient = d)
//   out2 = a * dropout(dx)                                      (second output, as ea_layernorm_bwd_dx2)
// with the dgamma / dbeta partials of both norms in two workspace slabs.  d never reaches memory; every rounding point of the two
// separate launches is kept, so dx, out2 and the partials are bit-identical to them.
__global__ __launch_bounds__(256) void ln_bwd2_kernel(
    const bf16_t* __restrict__ x1, const bf16_t* __restrict__ dy1, const float* __restrict__ gamma1,
    const float* __restrict__ mean1, const float* __restrict__ rstd1, const bf16_t* __restrict__ dx_add,
    float* __restrict__ partial1, const bf16_t* __restrict__ x2, const float* __restrict__ gamma2,
    const float* __restrict__ mean2, const float* __restrict__ rstd2, float* __restrict__ partial2,
    bf16_t* __restrict__ dx, int M, int C, int rows_per_block, const LnOut2 o2) {
  extern __shared__ float red[];  // [4 waves][4 vectors][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 3;
  const bool on = lane < nch;
  const int cl = on ? lane : 0;
  float dg1[8], db1[8], dg2[8], db2[8], gam1[8], gam2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    dg1[e] = db1[e] = dg2[e] = db2[e] = 0.f;
    gam1[e] = gamma1[cl * 8 + e];
    gam2[e] = gamma2[cl * 8 + e];
  }
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  const float inv_c = __builtin_amdgcn_rcpf((float)C);
  struct RowIn {
    uint4 ux1, ud1, ua, ux2;
    float m1, s1, m2, s2;
  };
  auto load_row = [&](int row, RowIn& r) {  // (branch-free: clamped column for the idle lanes of a narrow row)
    const long o = (long)row * C + cl * 8;
    r.ux1 = *reinterpret_cast<const uint4*>(x1 + o);
    r.ud1 = *reinterpret_cast<const uint4*>(dy1 + o);
    r.ua = *reinterpret_cast<const uint4*>(dx_add + o);
    r.ux2 = *reinterpret_cast<const uint4*>(x2 + o);
    r.m1 = mean1[row]; r.s1 = rstd1[row]; r.m2 = mean2[row]; r.s2 = rstd2[row];
  };
  // one LayerNorm backward over this lane's eight columns (the arithmetic of ln_bwd_kernel, shared helpers)
  auto norm_bwd = [&](const uint4& ux, const float (&dv)[8], const float (&gam)[8], float mean, float rstd, float (&dg)[8],
                      float (&db)[8], const uint4* add, float (&o)[8]) {
    float xh[8], g[8];
    float s1 = 0.f, s2 = 0.f;
    if (on) {
      ln_bwd_accum8(ux, dv, gam, mean, rstd, dg, db, xh, g, s1, s2);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) xh[e] = g[e] = 0.f;
    }
    s1 = wave_sum_dpp(s1) * inv_c;
    s2 = wave_sum_dpp(s2) * inv_c;
    ln_bwd_finish8(xh, g, rstd, s1, s2, add, o);
  };
  auto process_row = [&](int row, const RowIn& r) {
    float dv[8], o[8];
    unpack8(r.ud1, dv);
    norm_bwd(r.ux1, dv, gam1, r.m1, r.s1, dg1, db1, &r.ua, o);
    uint4 ud;
    ud.x = pack_bf2(o[0], o[1]); ud.y = pack_bf2(o[2], o[3]); ud.z = pack_bf2(o[4], o[5]); ud.w = pack_bf2(o[6], o[7]);
    unpack8(ud, dv);  // the rounded gradient the second norm would have read back
    norm_bwd(r.ux2, dv, gam2, r.m2, r.s2, dg2, db2, nullptr, o);
    if (!on) return;
    uint4 u;
    u.x = pack_bf2(o[0], o[1]); u.y = pack_bf2(o[2], o[3]); u.z = pack_bf2(o[4], o[5]); u.w = pack_bf2(o[6], o[7]);
    *reinterpret_cast<uint4*>(dx + (long)row * C + lane * 8) = u;
    if (o2.out) {
      float dr[8], p2[8], kk8[8];
      unpack8(u, dr);
      if (o2.thr) ea_keep8(o2.seed, (uint64_t)row * C + lane * 8, o2.thr, o2.inv_keep, kk8);
#pragma unroll
      for (int e = 0; e < 8; ++e) p2[e] = o2.a * dr[e] * (o2.thr ? kk8[e] : 1.f);
      uint4 u2;
      u2.x = pack_bf2(p2[0], p2[1]); u2.y = pack_bf2(p2[2], p2[3]); u2.z = pack_bf2(p2[4], p2[5]); u2.w = pack_bf2(p2[6], p2[7]);
      *reinterpret_cast<uint4*>(o2.out + (long)row * C + lane * 8) = u2;
    }
  };
  for (int row = r0 + wave; row < r1; row += 8) {  // a wavefront's two rows of the block are in flight together
    RowIn ra, rb;
    const bool two = row + 4 < r1;  // (uniform)
    load_row(row, ra);
    if (two) load_row(row + 4, rb);
    process_row(row, ra);
    if (two) process_row(row + 4, rb);
  }
  if (on) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(wave * 4 + 0) * C + lane * 8 + e] = dg1[e];
      red[(wave * 4 + 1) * C + lane * 8 + e] = db1[e];
      red[(wave * 4 + 2) * C + lane * 8 + e] = dg2[e];
      red[(wave * 4 + 3) * C + lane * 8 + e] = db2[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] += red[(w * 4 + k) * C + c];
    partial1[(long)blockIdx.x * 2 * C + c] = a[0];
    partial1[(long)blockIdx.x * 2 * C + C + c] = a[1];
    partial2[(long)blockIdx.x * 2 * C + c] = a[2];
    partial2[(long)blockIdx.x * 2 * C + C + c] = a[3];
  }
}

// partial: [nrows][2C] -> dgamma[c] += sum partial[:, c], dbeta[c] += sum partial[:, C + c].
// grid (2C/64, RSPLIT); block 256 = 4 row-lanes x 64 columns.
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int nrows, int C) {
  __shared__ float sm[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;  // [0, 2C)
  const int per = (nrows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(nrows, r0 + per);
  float a = 0.f;
  for (int r = r0 + ry; r < r1; r += 4) a += partial[(long)r * 2 * C + col];
  sm[ry][cx] = a;
  __syncthreads();
  if (ry == 0) {
    a = sm[0][cx] + sm[1][cx] + sm[2][cx] + sm[3][cx];
    if (col < C) atomicAdd(dgamma + col, a);
    else atomicAdd(dbeta + col - C, a);
  }
}

// the same over several LayerNorms at once: blockIdx.z picks the item (all LayerNorms of one encoder layer's backward)
struct LnRedDev { const float* partial; float* dgamma; float* dbeta; int nrows, C; };
struct LnRedTable { LnRedDev it[EA_LNRED_MAX]; };
__global__ __launch_bounds__(256) void ln_param_reduce_group_kernel(const LnRedTable tb) {
  __shared__ float sm[4][64];
  const LnRedDev d = tb.it[blockIdx.z];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;  // [0, 2C)
  if (blockIdx.x * 64 >= 2 * d.C) return;
  const int per = (d.nrows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per, r1 = min(d.nrows, r0 + per);
  float a = 0.f;
  if (col < 2 * d.C)
    for (int r = r0 + ry; r < r1; r += 4) a += d.partial[(long)r * 2 * d.C + col];
  sm[ry][cx] = a;
  __syncthreads();
  if (ry == 0 && col < 2 * d.C) {
    a = sm[0][cx] + sm[1][cx] + sm[2][cx] + sm[3][cx];
    if (col < d.C) atomicAdd(d.dgamma + col, a);
    else atomicAdd(d.dbeta + col - d.C, a);
  }
}

}  // namespace

extern "C" int ea_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                                float* mean, float* rstd, int M, int C, float eps,
                                const uint8_t* row_zero, uint64_t drop_seed, uint32_t drop_thr,
                                float drop_scale, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8 != 0 || C > 64 * 8 * MAXC8_LIMIT) return -2;
  static const int rw_mode = [] { const char* e = getenv("EA_LN_ROWS"); return e ? atoi(e) : 4; }();  // (diagnostic: 0 = one row per wave)
  if (C <= 512 && M >= 2048 && rw_mode > 0) {
    if (rw_mode >= 4)
      hipLaunchKernelGGL((ln_fwd_rows_kernel<4>), dim3((M + 15) / 16), dim3(256), 0, stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean,
                         rstd, M, C, eps, row_zero, drop_seed, drop_thr, drop_scale);
    else
      hipLaunchKernelGGL((ln_fwd_rows_kernel<2>), dim3((M + 7) / 8), dim3(256), 0, stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean,
                         rstd, M, C, eps, row_zero, drop_seed, drop_thr, drop_scale);
    return EA_CHECK_LAUNCH();
  }
  int fblocks = (M + 3) / 4;
  if (fblocks > 1024) fblocks = (fblocks + 1) / 2 > 1024 ? (fblocks + 1) / 2 : 1024;  // >= 2 rows per wavefront on large inputs
#define EA_LN_FWD(NC)                                                                                                  \
  hipLaunchKernelGGL((ln_fwd_kernel<NC>), dim3(fblocks), dim3(256), 0, stream, (const bf16_t*)x, gamma, beta,          \
                     (bf16_t*)y, mean, rstd, M, C, eps, row_zero, drop_seed, drop_thr, drop_scale)
  if (C <= 512) EA_LN_FWD(1);
  else if (C <= 1024) EA_LN_FWD(2);
  else EA_LN_FWD(4);
#undef EA_LN_FWD
  return EA_CHECK_LAUNCH();
}

// the same with an fp32 output (x stays bf16: a Linear's output): the joint network's two LayerNorms, whose outputs the
// reference adds and rectifies in fp32 (speech_transformer_transducer_base.py:292-294 under autocast)
extern "C" int ea_layernorm_fwd_f32out(const void* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                       int M, int C, float eps, const uint8_t* row_zero, uint64_t drop_seed, uint32_t drop_thr,
                                       float drop_scale, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8 != 0 || C > 64 * 8 * MAXC8_LIMIT) return -2;
  int fblocks = (M + 3) / 4;
  if (fblocks > 1024) fblocks = (fblocks + 1) / 2 > 1024 ? (fblocks + 1) / 2 : 1024;
#define EA_LN_FWD32(NC)                                                                                                \
  hipLaunchKernelGGL((ln_fwd_kernel<NC, float>), dim3(fblocks), dim3(256), 0, stream, (const bf16_t*)x, gamma, beta, y, \
                     mean, rstd, M, C, eps, row_zero, drop_seed, drop_thr, drop_scale)
  if (C <= 512) EA_LN_FWD32(1);
  else if (C <= 1024) EA_LN_FWD32(2);
  else EA_LN_FWD32(4);
#undef EA_LN_FWD32
  return EA_CHECK_LAUNCH();
}

static inline int ln_bwd_rows_per_block(int M, bool have_ws) {
  if (have_ws) return 8;
  // atomic path: ~512 blocks keep the dgamma/dbeta atomics (2*C per block) bounded
  int rpb = (M + 511) / 512;
  rpb = ((rpb + 3) / 4) * 4;
  return rpb < 4 ? 4 : rpb;
}

extern "C" long ea_layernorm_bwd_workspace_bytes(int M, int C) {
  const int rpb = ln_bwd_rows_per_block(M, true);
  return (long)((M + rpb - 1) / rpb) * 2 * C * (long)sizeof(float);
}

static int ln_bwd_launch(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                         float* dgamma, float* dbeta, int M, int C, const uint8_t* row_zero, uint64_t drop_seed, uint32_t drop_thr,
                         float drop_scale, const void* dx_add, void* workspace, hipStream_t stream, bool reduce_params,
                         const LnOut2& o2, bool dy_f32 = false) {
  if (M <= 0) return 0;
  if (C % 8 != 0 || C > 64 * 8 * MAXC8_LIMIT) return -2;
  const int rpb = ln_bwd_rows_per_block(M, workspace != nullptr);
  const int nblk = (M + rpb - 1) / rpb;
  static const int two_rows = [] { const char* e = getenv("EA_LN_BWD_TWO_ROWS"); return e ? atoi(e) : 1; }();  // (diagnostic A/B switch)
  if (dy_f32) {
#define EA_LN_BWD32(NC)                                                                                                \
  hipLaunchKernelGGL((ln_bwd_kernel<NC, float>), dim3(nblk), dim3(256), (size_t)8 * C * sizeof(float), stream,         \
                     (const bf16_t*)x, (const float*)dy, gamma, mean, rstd, (bf16_t*)dx, dgamma, dbeta, M, C, rpb,     \
                     row_zero, drop_seed, drop_thr, drop_scale, (const bf16_t*)dx_add, (float*)workspace, o2, two_rows)
    if (C <= 512) EA_LN_BWD32(1);
    else if (C <= 1024) EA_LN_BWD32(2);
    else EA_LN_BWD32(4);
#undef EA_LN_BWD32
    if (workspace && reduce_params) return ea_layernorm_param_reduce(workspace, dgamma, dbeta, M, C, stream);
    return EA_CHECK_LAUNCH();
  }
#define EA_LN_BWD(NC)                                                                                                  \
  hipLaunchKernelGGL((ln_bwd_kernel<NC>), dim3(nblk), dim3(256), (size_t)8 * C * sizeof(float), stream,                \
                     (const bf16_t*)x, (const bf16_t*)dy, gamma, mean, rstd, (bf16_t*)dx, dgamma, dbeta, M, C, rpb,    \
                     row_zero, drop_seed, drop_thr, drop_scale, (const bf16_t*)dx_add, (float*)workspace, o2, two_rows)
  if (C <= 512) EA_LN_BWD(1);
  else if (C <= 1024) EA_LN_BWD(2);
  else EA_LN_BWD(4);
#undef EA_LN_BWD
  if (workspace && reduce_params) return ea_layernorm_param_reduce(workspace, dgamma, dbeta, M, C, stream);
  return EA_CHECK_LAUNCH();
}

// Final LayerNorm of one Conformer layer + first LayerNorm of the next in one launch (ln_fwd2_rows_kernel above); C <= 512.
extern "C" int ea_layernorm_fwd2(const void* x, const float* g1, const float* b1, void* y1, float* mean1, float* rstd1,
                                 const float* g2, const float* b2, void* y2, float* mean2, float* rstd2, int M, int C, float eps,
                                 hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8 != 0 || C > 512) return -2;
  if (!y1 || !y2 || !mean1 || !rstd1 || !mean2 || !rstd2) return -2;
  hipLaunchKernelGGL((ln_fwd2_rows_kernel<4>), dim3((M + 15) / 16), dim3(256), 0, stream, (const bf16_t*)x, g1, b1, (bf16_t*)y1, mean1,
                     rstd1, g2, b2, (bf16_t*)y2, mean2, rstd2, M, C, eps);
  return EA_CHECK_LAUNCH();
}

// ... and the backward of the pair (ln_bwd2_kernel above): `ws1` / `ws2` receive the dgamma / dbeta partials of the two norms
// (ea_layernorm_bwd_workspace_bytes each; fold them with ea_layernorm_param_reduce).  out2 may be NULL.
extern "C" int ea_layernorm_bwd2_dx(const void* x1, const void* dy1, const float* gamma1, const float* mean1, const float* rstd1,
                                    const void* dx_add, void* ws1, const void* x2, const float* gamma2, const float* mean2,
                                    const float* rstd2, void* ws2, void* dx, int M, int C, void* out2, float a2, uint64_t seed2,
                                    uint32_t thr2, float scale2, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8 != 0 || C > 512 || !ws1 || !ws2 || !dx_add) return -2;
  const int rpb = ln_bwd_rows_per_block(M, true);
  const int nblk = (M + rpb - 1) / rpb;
  hipLaunchKernelGGL(ln_bwd2_kernel, dim3(nblk), dim3(256), (size_t)16 * C * sizeof(float), stream, (const bf16_t*)x1, (const bf16_t*)dy1,
                     gamma1, mean1, rstd1, (const bf16_t*)dx_add, (float*)ws1, (const bf16_t*)x2, gamma2, mean2, rstd2, (float*)ws2,
                     (bf16_t*)dx, M, C, rpb, LnOut2{(bf16_t*)out2, a2, seed2, thr2, scale2});
  return EA_CHECK_LAUNCH();
}

// second pass of the workspace path: dgamma / dbeta += column sums of the per-workgroup partials (optimizer-only product)
extern "C" int ea_layernorm_param_reduce(const void* workspace, float* dgamma, float* dbeta, int M, int C, hipStream_t stream) {
  if (M <= 0) return 0;
  const int rpb = ln_bwd_rows_per_block(M, true);
  const int nblk = (M + rpb - 1) / rpb;
  int rs = nblk / 32;
  if (rs < 1) rs = 1;
  if (rs > 32) rs = 32;
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(2 * C / 64, rs), dim3(256), 0, stream, (const float*)workspace, dgamma,
                     dbeta, nblk, C);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_layernorm_param_reduce_group(const EaLnReduceGroup* g, hipStream_t stream) {
  if (g->count <= 0) return 0;
  if (g->count > EA_LNRED_MAX) return -2;
  LnRedTable tb;
  int maxc2 = 0, maxblk = 0, n = 0;
  for (int i = 0; i < g->count; ++i) {
    const EaLnReduceItem& it = g->item[i];
    if (it.M <= 0) continue;
    const int rpb = ln_bwd_rows_per_block(it.M, true);
    const int nblk = (it.M + rpb - 1) / rpb;
    tb.it[n++] = LnRedDev{(const float*)it.workspace, it.dgamma, it.dbeta, nblk, it.C};
    if (2 * it.C > maxc2) maxc2 = 2 * it.C;
    if (nblk > maxblk) maxblk = nblk;
  }
  if (n == 0) return 0;
  static const int rs_cap = [] { const char* e = getenv("EA_LN_REDUCE_RS"); return e ? atoi(e) : 32; }();  // (tuning knob: row splits per item)
  int rs = maxblk / 32;
  if (rs < 1) rs = 1;
  if (rs > rs_cap) rs = rs_cap;
  hipLaunchKernelGGL(ln_param_reduce_group_kernel, dim3((maxc2 + 63) / 64, rs, n), dim3(256), 0, stream, tb);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean,
                                const float* rstd, void* dx, float* dgamma, float* dbeta, int M, int C,
                                const uint8_t* row_zero, uint64_t drop_seed, uint32_t drop_thr,
                                float drop_scale, const void* dx_add, void* workspace, hipStream_t stream) {
  return ln_bwd_launch(x, dy, gamma, mean, rstd, dx, dgamma, dbeta, M, C, row_zero, drop_seed, drop_thr, drop_scale, dx_add,
                       workspace, stream, true, LnOut2{nullptr, 1.f, 0, 0, 1.f});
}
// ea_layernorm_bwd with an fp32 incoming gradient (of an fp32 LayerNorm output, ea_layernorm_fwd_f32out)
extern "C" int ea_layernorm_bwd_f32dy(const void* x, const float* dy, const float* gamma, const float* mean,
                                      const float* rstd, void* dx, float* dgamma, float* dbeta, int M, int C,
                                      const uint8_t* row_zero, uint64_t drop_seed, uint32_t drop_thr,
                                      float drop_scale, const void* dx_add, void* workspace, hipStream_t stream) {
  return ln_bwd_launch(x, dy, gamma, mean, rstd, dx, dgamma, dbeta, M, C, row_zero, drop_seed, drop_thr, drop_scale, dx_add,
                       workspace, stream, true, LnOut2{nullptr, 1.f, 0, 0, 1.f}, true);
}
// dx only (workspace required): the caller runs ea_layernorm_param_reduce later, possibly on another stream
extern "C" int ea_layernorm_bwd_dx(const void* x, const void* dy, const float* gamma, const float* mean,
                                   const float* rstd, void* dx, float* dgamma, float* dbeta, int M, int C,
                                   const uint8_t* row_zero, uint64_t drop_seed, uint32_t drop_thr,
                                   float drop_scale, const void* dx_add, void* workspace, hipStream_t stream) {
  if (!workspace) return -2;
  return ln_bwd_launch(x, dy, gamma, mean, rstd, dx, dgamma, dbeta, M, C, row_zero, drop_seed, drop_thr, drop_scale, dx_add,
                       workspace, stream, false, LnOut2{nullptr, 1.f, 0, 0, 1.f});
}
// ea_layernorm_bwd_dx with a second output out2[i] = a2 * dropout(dx[i]; seed2, thr2, scale2) (same element order and mask
// as ea_scale_dropout_bf16 applied to dx): saves the separate pass over the gradient that the next residual block starts with
extern "C" int ea_layernorm_bwd_dx2(const void* x, const void* dy, const float* gamma, const float* mean,
                                    const float* rstd, void* dx, float* dgamma, float* dbeta, int M, int C,
                                    const void* dx_add, void* workspace, void* out2, float a2, uint64_t seed2, uint32_t thr2,
                                    float scale2, hipStream_t stream) {
  if (!workspace) return -2;
  return ln_bwd_launch(x, dy, gamma, mean, rstd, dx, dgamma, dbeta, M, C, nullptr, 0, 0, 1.f, dx_add, workspace, stream, false,
                       LnOut2{(bf16_t*)out2, a2, seed2, thr2, scale2});
}
