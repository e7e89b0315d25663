// Relative-position (Transformer-XL style) attention glue kernels for gfx950.
//
// Reference semantics: fairseq/modules/multihead_attention.py:679-688 (q + pos_bias_u / pos_bias_v,
// scaling), :788-831 (content logits + positional logits, the as_strided "skew":
// pos[i][j] = raw[i][(T-1) - i + j]), :835-867 (additive attn_mask, key-padding -inf, fp32 softmax),
// :874 (attention dropout).  The matrix products run on the MFMA GEMM (gemm.hip); these kernels
// are the HBM-bound pieces in between and apply the skew as an index transform so the
// (BH, T, 2S-1) tensor is only ever read, never re-materialised in shifted form.
//
// Score tensors are laid out [H][B][T][ld] (z = h*B + b) with ld padded to a multiple of 8 so
// that the GEMM's 16-byte loads stay aligned; padded columns are written as zeros.
#include "common.h"
#include "espresso_amd.h"

namespace {

// qkv: [M][3C] bf16 (q | k | v).  qu = (q + u) * s, qv = (q + v) * s, both [M][C] bf16.
__global__ __launch_bounds__(256) void relpos_q_prep_kernel(const bf16_t* __restrict__ qkv, long ldq,
                                                            const float* __restrict__ u,
                                                            const float* __restrict__ v,
                                                            bf16_t* __restrict__ qu, bf16_t* __restrict__ qv,
                                                            int M, int C, float s) {
  const int nch = C >> 3;
  const long total = (long)M * nch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / nch), ch = (int)(i % nch);
    const uint4 uq = *reinterpret_cast<const uint4*>(qkv + (long)m * ldq + ch * 8);
    const uint32_t w[4] = {uq.x, uq.y, uq.z, uq.w};
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float q = (e & 1) ? __uint_as_float(w[e >> 1] & 0xffff0000u) : __uint_as_float(w[e >> 1] << 16);
      a[e] = (q + (u ? u[ch * 8 + e] : 0.f)) * s;
      b[e] = (q + (v ? v[ch * 8 + e] : 0.f)) * s;
    }
    uint4 o;
    o.x = pack_bf2(a[0], a[1]); o.y = pack_bf2(a[2], a[3]); o.z = pack_bf2(a[4], a[5]); o.w = pack_bf2(a[6], a[7]);
    *reinterpret_cast<uint4*>(qu + (long)m * C + ch * 8) = o;
    if (qv) {
      o.x = pack_bf2(b[0], b[1]); o.y = pack_bf2(b[2], b[3]); o.z = pack_bf2(b[4], b[5]); o.w = pack_bf2(b[6], b[7]);
      *reinterpret_cast<uint4*>(qv + (long)m * C + ch * 8) = o;
    }
  }
}

// One wave per (z, i) row.  MAXJ elements per lane -> S <= 64*MAXJ.
constexpr int MAXJ = 16;  // S <= 1024 (max_source_positions 3600 / 4 = 900)

__global__ __launch_bounds__(256) void relpos_softmax_fwd_kernel(
    const float* __restrict__ ac, const float* __restrict__ bd, const int* __restrict__ klen,
    const float* __restrict__ attn_mask, bf16_t* __restrict__ P, bf16_t* __restrict__ Pd, int B, int T,
    int S, int ld_ac, int ld_bd, int ld_p, int causal, uint64_t seed, uint32_t thr, float inv_keep, long nrows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int i = (int)(row % T);
  const int z = (int)(row / T);
  const int b = z % B;
  const int kl = klen ? klen[b] : S;
  const float* arow = ac + row * ld_ac;
  const float* brow = bd ? bd + row * ld_bd + (T - 1 - i) : nullptr;
  float sc[MAXJ];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < MAXJ; ++t) {
    const int j = lane + 64 * t;
    float s = -INFINITY;
    if (j < S) {
      s = arow[j];
      if (brow) s += brow[j];
      if (attn_mask) s += attn_mask[(long)i * S + j];
      if (j >= kl) s = -INFINITY;
      if (causal && j > i + (S - T)) s = -INFINITY;
    }
    sc[t] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < MAXJ; ++t) {
    const float e = (sc[t] == -INFINITY) ? 0.f : __expf(sc[t] - mx);
    sc[t] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;  // all-masked rows give NaN exactly like the reference softmax
#pragma unroll
  for (int t = 0; t < MAXJ; ++t) {
    const int j = lane + 64 * t;
    if (j < ld_p) {
      float p = (j < S) ? sc[t] * inv : 0.f;
      P[row * ld_p + j] = f2bf(p);
      if (Pd) {
        if (thr && j < S) p *= ea_keep(seed, (uint64_t)row * S + j, thr, inv_keep);
        Pd[row * ld_p + j] = f2bf(p);
      }
    }
  }
}

// dS = P * (dP - sum_j dP*P),  dP = dPd * keep.   dAC[i][j] = dS;  dBD[i][r] = dS[i][r-(T-1)+i]
__global__ __launch_bounds__(256) void relpos_softmax_bwd_kernel(
    const bf16_t* __restrict__ P, const float* __restrict__ dPd, bf16_t* __restrict__ dAC,
    bf16_t* __restrict__ dBD, int T, int S, int ld_p, int ld_dp, int ld_bd, uint64_t seed, uint32_t thr,
    float inv_keep, long nrows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int i = (int)(row % T);
  float p[MAXJ], dp[MAXJ];
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < MAXJ; ++t) {
    const int j = lane + 64 * t;
    p[t] = 0.f;
    dp[t] = 0.f;
    if (j < S) {
      p[t] = bf2f(P[row * ld_p + j]);
      float g = dPd[row * ld_dp + j];
      if (thr) g *= ea_keep(seed, (uint64_t)row * S + j, thr, inv_keep);
      dp[t] = g;
      dot += g * p[t];
    }
  }
  dot = wave_sum(dot);
  if (dBD) {
    // zero the columns of this dBD row that no (i,j) maps to: r < T-1-i or r >= T-1-i+S
    const int lo = T - 1 - i, hi = lo + S;
    for (int r = lane; r < ld_bd; r += 64)
      if (r < lo || r >= hi) dBD[row * ld_bd + r] = 0;
  }
#pragma unroll
  for (int t = 0; t < MAXJ; ++t) {
    const int j = lane + 64 * t;
    if (j < ld_p) {
      const float ds = (j < S) ? p[t] * (dp[t] - dot) : 0.f;
      const bf16_t o = f2bf(ds);
      dAC[row * ld_p + j] = o;
      if (dBD && j < S) dBD[row * ld_bd + (T - 1 - i) + j] = o;
    }
  }
}

// out[m*ldo + c] = a[m*lda + c] + b[m*ldb + c]   (bf16, C % 8 == 0, all lds % 8 == 0)
__global__ __launch_bounds__(256) void add2_strided_kernel(const bf16_t* __restrict__ a, long lda,
                                                           const bf16_t* __restrict__ b, long ldb,
                                                           bf16_t* __restrict__ out, long ldo, int M, int C) {
  const int nch = C >> 3;
  const long total = (long)M * nch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / nch), ch = (int)(i % nch);
    const uint4 ua = *reinterpret_cast<const uint4*>(a + (long)m * lda + ch * 8);
    const uint4 ub = *reinterpret_cast<const uint4*>(b + (long)m * ldb + ch * 8);
    const uint32_t wa[4] = {ua.x, ua.y, ua.z, ua.w}, wb[4] = {ub.x, ub.y, ub.z, ub.w};
    uint32_t wo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = __uint_as_float(wa[e] << 16) + __uint_as_float(wb[e] << 16);
      const float hi = __uint_as_float(wa[e] & 0xffff0000u) + __uint_as_float(wb[e] & 0xffff0000u);
      wo[e] = pack_bf2(lo, hi);
    }
    *reinterpret_cast<uint4*>(out + (long)m * ldo + ch * 8) = make_uint4(wo[0], wo[1], wo[2], wo[3]);
  }
}

}  // namespace

static inline int egrid(long n) {
  long b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int ea_relpos_q_prep(const void* qkv, long ldq, const float* u, const float* v, void* qu,
                                void* qv, int M, int C, float scaling, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8 || ldq % 8) return -2;
  hipLaunchKernelGGL(relpos_q_prep_kernel, dim3(egrid((long)M * (C / 8))), dim3(256), 0, stream,
                     (const bf16_t*)qkv, ldq, u, v, (bf16_t*)qu, (bf16_t*)qv, M, C, scaling);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_relpos_softmax_fwd(const float* ac, const float* bd, const int* key_len,
                                     const float* attn_mask, void* P, void* Pd, int H, int B, int T, int S,
                                     int ld_ac, int ld_bd, int ld_p, int causal, uint64_t drop_seed,
                                     uint32_t drop_thr, float drop_scale, hipStream_t stream) {
  const long nrows = (long)H * B * T;
  if (nrows <= 0) return 0;
  if (S > 64 * MAXJ || ld_p > 64 * MAXJ) return -2;
  hipLaunchKernelGGL(relpos_softmax_fwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, stream, ac, bd,
                     key_len, attn_mask, (bf16_t*)P, (bf16_t*)Pd, B, T, S, ld_ac, ld_bd, ld_p, causal,
                     drop_seed, drop_thr, drop_scale, nrows);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_relpos_softmax_bwd(const void* P, const float* dPd, void* dAC, void* dBD, int H, int B,
                                     int T, int S, int ld_p, int ld_dp, int ld_bd, uint64_t drop_seed,
                                     uint32_t drop_thr, float drop_scale, hipStream_t stream) {
  const long nrows = (long)H * B * T;
  if (nrows <= 0) return 0;
  if (S > 64 * MAXJ || ld_p > 64 * MAXJ) return -2;
  hipLaunchKernelGGL(relpos_softmax_bwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, stream,
                     (const bf16_t*)P, dPd, (bf16_t*)dAC, (bf16_t*)dBD, T, S, ld_p, ld_dp, ld_bd, drop_seed,
                     drop_thr, drop_scale, nrows);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_add2_strided_bf16(const void* a, long lda, const void* b, long ldb, void* out, long ldo,
                                    int M, int C, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8 || lda % 8 || ldb % 8 || ldo % 8) return -2;
  hipLaunchKernelGGL(add2_strided_kernel, dim3(egrid((long)M * (C / 8))), dim3(256), 0, stream,
                     (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, M, C);
  return EA_CHECK_LAUNCH();
}
