// Conv2d sub-sampling front-end (ConvBNReLU) pieces for gfx950.
// Reference: espresso/modules/speech_convolutions.py:22-129 — 4 x (Conv2d 3x3 pad 1 + BatchNorm2d +
// ReLU) over (B, C, T, F), strides (1,1),(2,2),(1,1),(2,2), then (B,T',C*F') flatten and zeroing of
// padded frames.  BatchNorm statistics include padded frames (the reference does not mask).
//
// MI355X layout: channels-last [B][T][F][C] bf16.  The first conv (C_in = 1, K = 9) is a direct
// HBM-bound kernel; convs 2-4 are lowered to im2col + the MFMA GEMM (gemm.hip) with K = 9*C_in,
// so the output of the GEMM *is* the channels-last activation.  BatchNorm+ReLU reuse the
// channel-wise kernels in convmodule.hip (ea_bn_act_fwd / ea_bn_act_bwd).
#include "common.h"
#include "espresso_amd.h"

namespace {

// X [B][T][F] fp32 -> Z [B][To][Fo][CO] bf16 (pre-BN, bias added); stats[0..CO) += sum, [CO..2CO) += sumsq.
// A thread owns 8 output channels (one 16-byte store per position) and walks a contiguous run of positions; the 8 lanes of
// a position share the nine input taps (same addresses: one L1 transaction).  256 threads = 32 runs x 8 channel chunks per
// 64-channel slab; HBM-bound on the Z store (B*To*Fo*CO*2 bytes).
constexpr int C1_RUN = 16;  // positions per thread run
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W /*[CO][3][3]*/,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ Z,
                                                        double* __restrict__ stats, int T, int F, int To, int Fo, int CO,
                                                        int sy, int sx, long npos) {
  __shared__ float red[4][2][64];
  const int grp = threadIdx.x >> 3, ch = threadIdx.x & 7, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long p0 = ((long)blockIdx.x * 32 + grp) * C1_RUN;
  for (int cs = 0; cs < CO; cs += 64) {
    const int c0 = cs + ch * 8;
    float w[8][9], bv[8], s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int k = 0; k < 9; ++k) w[e][k] = W[(c0 + e) * 9 + k];
      bv[e] = bias[c0 + e];
      s[e] = q[e] = 0.f;
    }
    int fo = (int)(p0 % Fo), to = (int)((p0 / Fo) % To);  // carried along the run (64-bit divisions per position were the hot spot)
    long b = p0 / ((long)Fo * To);
    for (int r = 0; r < C1_RUN; ++r, ++fo) {
      const long p = p0 + r;
      if (p >= npos) break;
      if (fo == Fo) {
        fo = 0;
        if (++to == To) { to = 0; ++b; }
      }
      float x[9];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int t = to * sy + ky - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int f = fo * sx + kx - 1;
          x[ky * 3 + kx] = (t >= 0 && t < T && f >= 0 && f < F) ? X[(b * T + t) * F + f] : 0.f;
        }
      }
      float z[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = bv[e];
#pragma unroll
        for (int k = 0; k < 9; ++k) a += w[e][k] * x[k];
        z[e] = a;
      }
      uint4 o;
      o.x = pack_bf2(z[0], z[1]); o.y = pack_bf2(z[2], z[3]); o.z = pack_bf2(z[4], z[5]); o.w = pack_bf2(z[6], z[7]);
      *reinterpret_cast<uint4*>(Z + p * CO + c0) = o;
      if (stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float zr = bf2f(f2bf(z[e]));  // statistics of the rounded values the next kernel reads
          s[e] += zr;
          q[e] += zr * zr;
        }
      }
    }
    if (stats) {
      // lanes l, l+8, ... hold the same channels: fold the 8 position groups of a wavefront, then the 4 wavefronts via LDS
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
          s[e] += __shfl_xor(s[e], off, 64);
          q[e] += __shfl_xor(q[e], off, 64);
        }
      }
      if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[wave][0][lane * 8 + e] = s[e];
          red[wave][1][lane * 8 + e] = q[e];
        }
      }
      __syncthreads();
      if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        const float a = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
        atomicAdd(stats + which * CO + cs + c, (double)a);
      }
      __syncthreads();
    }
  }
}

// dW[co][ky][kx] += sum_pos dZ[pos][co] * X[...];  dbias[co] += sum_pos dZ[pos][co].  Same thread map as the forward kernel
// (8 channels per thread, 16-byte dZ loads, the 8 lanes of a position share the taps); HBM-bound on the dZ read.
constexpr int C1W_RUN = 64;
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ X, const bf16_t* __restrict__ dZ,
                                                          float* __restrict__ dW, float* __restrict__ dbias, int T, int F,
                                                          int To, int Fo, int CO, int sy, int sx, long npos) {
  __shared__ float red[4][10][64];
  const int grp = threadIdx.x >> 3, ch = threadIdx.x & 7, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long p0 = ((long)blockIdx.x * 32 + grp) * C1W_RUN;
  for (int cs = 0; cs < CO; cs += 64) {
    const int c0 = cs + ch * 8;
    float acc[10][8];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    // 4 positions per step with their dZ loads issued together: the loop was one dependent 16-byte load per iteration
    // (64 round trips to HBM per thread: 297 us for a 251 MB read; this kernel is the LAST one of the backward pass, nothing hides it)
    // (fo, to, b) of the run's first position by division once, then carried along: three 64-bit divisions per position cost
    // more VALU time than the 80 FMAs they feed
    int fo = (int)(p0 % Fo), to = (int)((p0 / Fo) % To);
    long b = p0 / ((long)Fo * To);
    for (int r0 = 0; r0 < C1W_RUN; r0 += 4) {
      if (p0 + r0 >= npos) break;
      uint4 ddq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long p = p0 + r0 + q;
        ddq[q] = p < npos ? *reinterpret_cast<const uint4*>(dZ + p * CO + c0) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t dw[4] = {ddq[q].x, ddq[q].y, ddq[q].z, ddq[q].w};  // (past the end: zero gradient, any valid (b, to, fo))
        float d[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          d[2 * e] = __uint_as_float(dw[e] << 16);
          d[2 * e + 1] = __uint_as_float(dw[e] & 0xffff0000u);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int t = to * sy + ky - 1;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int f = fo * sx + kx - 1;
            const float x = (t >= 0 && t < T && f >= 0 && f < F) ? X[(b * T + t) * F + f] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[ky * 3 + kx][e] += d[e] * x;
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[9][e] += d[e];
        if (++fo == Fo) {
          fo = 0;
          if (++to == To) { to = 0; if (p0 + r0 + q + 1 < npos) ++b; }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = acc[k][e];
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        if (lane < 8) red[wave][k][lane * 8 + e] = v;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 640; i += 256) {
      const int k = i >> 6, c = i & 63;
      const float a = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
      if (k < 9) atomicAdd(dW + (cs + c) * 9 + k, a);
      else if (dbias) atomicAdd(dbias + cs + c, a);
    }
    __syncthreads();
  }
}

// A [B][T][F][C] bf16 -> col [B*To*Fo][9*C] bf16 ; k = (ky*3+kx)*C + c ; C % 8 == 0
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ A, bf16_t* __restrict__ col, int T, int F,
                                                     int C, int To, int Fo, int sy, int sx, long total_chunks) {
  const int cpr = 9 * (C >> 3);  // 16-byte chunks per col row
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_chunks; i += (long)gridDim.x * blockDim.x) {
    const long row = i / cpr;
    const int rem = (int)(i % cpr);
    const int kk = rem / (C >> 3), ch = rem % (C >> 3);
    const int ky = kk / 3, kx = kk % 3;
    const int fo = (int)(row % Fo);
    const int to = (int)((row / Fo) % To);
    const long b = row / ((long)Fo * To);
    const int t = to * sy + ky - 1, f = fo * sx + kx - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t >= 0 && t < T && f >= 0 && f < F) v = *reinterpret_cast<const uint4*>(A + (((b * T + t) * F + f) * C + ch * 8));
    *reinterpret_cast<uint4*>(col + row * (9L * C) + (long)kk * C + ch * 8) = v;
  }
}

// dA [B][T][F][C] = sum over taps of dcol rows (gather form of col2im)
__global__ __launch_bounds__(256) void col2im_kernel(const bf16_t* __restrict__ dcol, bf16_t* __restrict__ dA, int T, int F,
                                                     int C, int To, int Fo, int sy, int sx, long total_chunks) {
  const int cpp = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_chunks; i += (long)gridDim.x * blockDim.x) {
    const long pos = i / cpp;
    const int ch = (int)(i % cpp);
    const int f = (int)(pos % F);
    const int t = (int)((pos / F) % T);
    const long b = pos / ((long)F * T);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int tn = t + 1 - ky;
      if (tn < 0 || tn % sy) continue;
      const int to = tn / sy;
      if (to >= To) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int fn = f + 1 - kx;
        if (fn < 0 || fn % sx) continue;
        const int fo = fn / sx;
        if (fo >= Fo) continue;
        const long row = (b * To + to) * Fo + fo;
        const uint4 u = *reinterpret_cast<const uint4*>(dcol + row * (9L * C) + (long)(ky * 3 + kx) * C + ch * 8);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] += __uint_as_float(w[e] << 16);
          acc[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
        }
      }
    }
    uint4 o;
    o.x = pack_bf2(acc[0], acc[1]); o.y = pack_bf2(acc[2], acc[3]); o.z = pack_bf2(acc[4], acc[5]); o.w = pack_bf2(acc[6], acc[7]);
    *reinterpret_cast<uint4*>(dA + pos * C + ch * 8) = o;
  }
}

// stats[0..C) += sum_m X[m][c], stats[C..2C) += sum_m X[m][c]^2   (X bf16 [M][C])
__global__ __launch_bounds__(256) void colstats_kernel(const bf16_t* __restrict__ X, double* __restrict__ stats, long M, int C,
                                                       int rows_per_block) {
  __shared__ float sm[8][2][64];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 64 + cx * 2;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(M, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
  if (c0 + 1 < C) {
    for (long r = r0 + ry; r < r1; r += 8) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(X + r * C + c0);
      const float a = __uint_as_float(w << 16), b = __uint_as_float(w & 0xffff0000u);
      s0 += a; s1 += b; q0 += a * a; q1 += b * b;
    }
  }
  sm[ry][0][cx * 2] = s0; sm[ry][0][cx * 2 + 1] = s1;
  sm[ry][1][cx * 2] = q0; sm[ry][1][cx * 2 + 1] = q1;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a += sm[i][which][c];
    const int cc = blockIdx.x * 64 + c;
    if (cc < C) atomicAdd(stats + which * C + cc, (double)a);
  }
}

static inline int egrid(long n) {
  long b = (n + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int ea_conv1_fwd(const float* X, const float* W, const float* bias, void* Z, double* stats, int B, int T, int F,
                            int CO, int sy, int sx, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if (CO % 64) return -2;
  const int To = (T - 1) / sy + 1, Fo = (F - 1) / sx + 1;
  const long npos = (long)B * To * Fo;
  const int ppb = 32 * C1_RUN;
  hipLaunchKernelGGL(conv1_fwd_kernel, dim3((unsigned)((npos + ppb - 1) / ppb)), dim3(256), 0, stream, X, W, bias,
                     (bf16_t*)Z, stats, T, F, To, Fo, CO, sy, sx, npos);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_conv1_wgrad(const float* X, const void* dZ, float* dW, float* dbias, int B, int T, int F, int CO,
                              int sy, int sx, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if (CO % 64) return -2;
  const int To = (T - 1) / sy + 1, Fo = (F - 1) / sx + 1;
  const long npos = (long)B * To * Fo;
  const int ppb = 32 * C1W_RUN;
  hipLaunchKernelGGL(conv1_wgrad_kernel, dim3((unsigned)((npos + ppb - 1) / ppb)), dim3(256), 0, stream, X,
                     (const bf16_t*)dZ, dW, dbias, T, F, To, Fo, CO, sy, sx, npos);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_im2col3x3(const void* A, void* col, int B, int T, int F, int C, int sy, int sx, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if (C % 8) return -2;
  const int To = (T - 1) / sy + 1, Fo = (F - 1) / sx + 1;
  const long total = (long)B * To * Fo * 9 * (C / 8);
  hipLaunchKernelGGL(im2col_kernel, dim3(egrid(total)), dim3(256), 0, stream, (const bf16_t*)A, (bf16_t*)col, T, F, C, To,
                     Fo, sy, sx, total);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_col2im3x3(const void* dcol, void* dA, int B, int T, int F, int C, int sy, int sx, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if (C % 8) return -2;
  const int To = (T - 1) / sy + 1, Fo = (F - 1) / sx + 1;
  const long total = (long)B * T * F * (C / 8);
  hipLaunchKernelGGL(col2im_kernel, dim3(egrid(total)), dim3(256), 0, stream, (const bf16_t*)dcol, (bf16_t*)dA, T, F, C,
                     To, Fo, sy, sx, total);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_colstats_bf16(const void* X, double* stats, long M, int C, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 2) return -2;
  int rpb = (int)((M + 1023) / 1024);
  if (rpb < 64) rpb = 64;
  dim3 grid((C + 63) / 64, (unsigned)((M + rpb - 1) / rpb));
  hipLaunchKernelGGL(colstats_kernel, grid, dim3(256), 0, stream, (const bf16_t*)X, stats, M, C, rpb);
  return EA_CHECK_LAUNCH();
}
