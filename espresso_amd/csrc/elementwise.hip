// Streaming element-wise / column-reduction kernels (HBM-bound, 16-byte accesses).
//  - fp32 -> bf16 cast of master weights (the per-step AMP cast the reference gets from
//    torch.cuda.amp.autocast, fairseq/tasks/fairseq_task.py:516)
//  - scale + dropout (backward of "dropout(x) * s" epilogues; FairseqDropout,
//    fairseq/modules/fairseq_dropout.py:23-25)
//  - column sums for bias gradients (autograd of nn.Linear bias)
//  - axpby for residual adds that cannot ride a GEMM epilogue.
#include "common.h"
#include "espresso_amd.h"

namespace {

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src,
                                                            bf16_t* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const float4 a = *reinterpret_cast<const float4*>(src + i);
      const float4 b = *reinterpret_cast<const float4*>(src + i + 4);
      uint4 u;
      u.x = pack_bf2(a.x, a.y);
      u.y = pack_bf2(a.z, a.w);
      u.z = pack_bf2(b.x, b.y);
      u.w = pack_bf2(b.z, b.w);
      *reinterpret_cast<uint4*>(dst + i) = u;
    } else {
      for (long j = i; j < n; ++j) dst[j] = f2bf(src[j]);
    }
  }
}

// row-pitched variant: src fp32 [M][ld_src] (N valid columns) -> dst bf16 [M][ld_dst]; columns N .. ld_dst-1 are zero-filled (the
// vocabulary gradient, N = 5004, goes to a pitch of 5008 so that every later 16-byte access of the weight / data gradient GEMMs
// stays aligned).  One block per row slice of 2048 columns.
__global__ __launch_bounds__(256) void cast_f32_bf16_rows_kernel(const float* __restrict__ src, long ld_src, bf16_t* __restrict__ dst,
                                                                 long ld_dst, int N) {
  const long m = blockIdx.y;
  const float* s = src + m * ld_src;
  bf16_t* d = dst + m * ld_dst;
  const bool vec = (ld_src & 3) == 0 && (ld_dst & 7) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  for (int n = (blockIdx.x * 256 + threadIdx.x) * 8; n < ld_dst; n += gridDim.x * 256 * 8) {
    if (vec && n + 8 <= N) {
      const float4 a = *reinterpret_cast<const float4*>(s + n);
      const float4 b = *reinterpret_cast<const float4*>(s + n + 4);
      uint4 u;
      u.x = pack_bf2(a.x, a.y);
      u.y = pack_bf2(a.z, a.w);
      u.z = pack_bf2(b.x, b.y);
      u.w = pack_bf2(b.z, b.w);
      *reinterpret_cast<uint4*>(d + n) = u;
    } else {
      for (int j = n; j < n + 8 && j < ld_dst; ++j) d[j] = j < N ? f2bf(s[j]) : (bf16_t)0;
    }
  }
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ src,
                                                            float* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = bf2f(src[i]);
}

// out = a*x*keep(idx) + b*y   (x,y,out bf16; y optional)
__global__ __launch_bounds__(256) void scale_dropout_kernel(const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ y,
                                                            bf16_t* __restrict__ out, long n, float a,
                                                            float b, uint64_t seed, uint32_t thr,
                                                            float inv_keep) {
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const uint4 ux = *reinterpret_cast<const uint4*>(x + i);
      uint4 uy = make_uint4(0, 0, 0, 0);
      if (y) uy = *reinterpret_cast<const uint4*>(y + i);
      const uint32_t wx[4] = {ux.x, ux.y, ux.z, ux.w};
      const uint32_t wy[4] = {uy.x, uy.y, uy.z, uy.w};
      float o[8], k8[8];
      if (thr) ea_keep8(seed, (uint64_t)i, thr, inv_keep, k8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xv = (e & 1) ? __uint_as_float(wx[e >> 1] & 0xffff0000u) : __uint_as_float(wx[e >> 1] << 16);
        const float yv = (e & 1) ? __uint_as_float(wy[e >> 1] & 0xffff0000u) : __uint_as_float(wy[e >> 1] << 16);
        const float k = thr ? k8[e] : 1.f;
        o[e] = a * xv * k + b * yv;
      }
      uint4 u;
      u.x = pack_bf2(o[0], o[1]);
      u.y = pack_bf2(o[2], o[3]);
      u.z = pack_bf2(o[4], o[5]);
      u.w = pack_bf2(o[6], o[7]);
      *reinterpret_cast<uint4*>(out + i) = u;
    } else {
      for (long j = i; j < n; ++j) {
        float k = 1.f;
        if (thr) k = ea_keep(seed, (uint64_t)j, thr, inv_keep);
        out[j] = f2bf(a * bf2f(x[j]) * k + (y ? b * bf2f(y[j]) : 0.f));
      }
    }
  }
}

// out[n] (+)= sum_m X[m*ld + n]   ; block = 256 threads = 32 column-pairs x 8 row-lanes
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ X, float* __restrict__ out,
                                                     int M, int N, long ld, int rows_per_block) {
  __shared__ float sm[8][64];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 64 + cx * 2;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float a0 = 0.f, a1 = 0.f;
  if (c0 + 1 < N && (ld & 1) == 0) {
    for (int r = r0 + ry; r < r1; r += 8) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(X + (long)r * ld + c0);
      a0 += __uint_as_float(w << 16);
      a1 += __uint_as_float(w & 0xffff0000u);
    }
  } else {
    for (int r = r0 + ry; r < r1; r += 8) {
      if (c0 < N) a0 += bf2f(X[(long)r * ld + c0]);
      if (c0 + 1 < N) a1 += bf2f(X[(long)r * ld + c0 + 1]);
    }
  }
  sm[ry][cx * 2] = a0;
  sm[ry][cx * 2 + 1] = a1;
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += sm[i][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < N) atomicAdd(out + c, s);
  }
}

// zero rows of a bf16 [M][C] matrix where row_zero[m] != 0
__global__ __launch_bounds__(256) void zero_rows_kernel(bf16_t* __restrict__ x, const uint8_t* __restrict__ row_zero,
                                                        int M, int C) {
  const int row = blockIdx.x;
  if (!row_zero[row]) return;
  for (int c = threadIdx.x; c < C; c += 256) x[(long)row * C + c] = 0;
}

// out[m][:] = scale * W[tok[m]][:] + pos[position[m]][:]   (W, pos fp32; out bf16; one row per wave)
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int* __restrict__ tok, const int* __restrict__ position,
                                                            const float* __restrict__ W, const float* __restrict__ pos,
                                                            bf16_t* __restrict__ out, int M, int C, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* w = W + (long)tok[row] * C;
  const float* pe = pos ? pos + (long)position[row] * C : nullptr;
  for (int c = (threadIdx.x & 63) * 2; c < C; c += 128) {
    float a = scale * w[c], b = scale * w[c + 1];
    if (pe) { a += pe[c]; b += pe[c + 1]; }
    *reinterpret_cast<uint32_t*>(out + (long)row * C + c) = pack_bf2(a, b);
  }
}
// dW[tok[m]][:] += scale * dy[m][:]
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int* __restrict__ tok, const bf16_t* __restrict__ dy,
                                                            float* __restrict__ dW, int M, int C, float scale, int pad_idx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int t = tok[row];
  if (t == pad_idx) return;  // nn.Embedding(padding_idx): no gradient to the pad row
  float* w = dW + (long)t * C;
  for (int c = (threadIdx.x & 63); c < C; c += 64) atomicAdd(w + c, scale * bf2f(dy[(long)row * C + c]));
}

}  // namespace

static inline int grid_for(long n, int per_thread) {
  long b = (n + 256L * per_thread - 1) / (256L * per_thread);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// Batched 2-D transposes of bf16 matrices (dst[c][r] = src[r][c]) through a 64x64 LDS tile: up to 8 matrices per launch.
struct TransposeBatch {
  const bf16_t* src[8];
  bf16_t* dst[8];
  int rows[8], cols[8], tile0[9];  // tile0[i] = first linear tile of matrix i ; tile0[n] = total
  int n;
};
__global__ __launch_bounds__(256) void transpose_batch_kernel(const TransposeBatch tb) {
  __shared__ bf16_t tile[64][66];
  int mi = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i < tb.n && (int)blockIdx.x >= tb.tile0[i]) mi = i;
  const int R = tb.rows[mi], Cc = tb.cols[mi];
  const int tcols = (Cc + 63) / 64;
  const int t = blockIdx.x - tb.tile0[mi];
  const int r0 = (t / tcols) * 64, c0 = (t % tcols) * 64;
  const bf16_t* __restrict__ src = tb.src[mi];
  bf16_t* __restrict__ dst = tb.dst[mi];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4)
    if (r0 + i < R && c0 + tx < Cc) tile[i][tx] = src[(long)(r0 + i) * Cc + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 64; i += 4)
    if (c0 + i < Cc && r0 + tx < R) dst[(long)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

extern "C" int ea_transpose_bf16_batch(const void* const* src, void* const* dst, const int* rows, const int* cols, int n,
                                       hipStream_t stream) {
  if (n <= 0) return 0;
  if (n > 8) return -2;
  TransposeBatch tb;
  int tot = 0;
  for (int i = 0; i < 8; ++i) {
    tb.src[i] = i < n ? (const bf16_t*)src[i] : nullptr;
    tb.dst[i] = i < n ? (bf16_t*)dst[i] : nullptr;
    tb.rows[i] = i < n ? rows[i] : 0;
    tb.cols[i] = i < n ? cols[i] : 0;
    tb.tile0[i] = tot;
    if (i < n) tot += ((rows[i] + 63) / 64) * ((cols[i] + 63) / 64);
  }
  tb.tile0[8] = tot;
  tb.n = n;
  if (tot == 0) return 0;
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(tot), dim3(256), 0, stream, tb);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_cast_f32_to_bf16(const float* src, void* dst, long n, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n, 8)), dim3(256), 0, stream, src, (bf16_t*)dst, n);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_cast_f32_to_bf16_rows(const float* src, long ld_src, void* dst, long ld_dst, long M, int N, hipStream_t stream) {
  if (M <= 0 || N <= 0) return 0;
  if (ld_dst < N || ld_src < N || M > 65535) return -2;
  const int bx = (int)((ld_dst + 2047) / 2048);
  hipLaunchKernelGGL(cast_f32_bf16_rows_kernel, dim3(bx, (unsigned)M), dim3(256), 0, stream, src, ld_src, (bf16_t*)dst, ld_dst, N);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_cast_bf16_to_f32(const void* src, float* dst, long n, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n, 1)), dim3(256), 0, stream, (const bf16_t*)src, dst, n);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_dropout_hash_host(uint64_t seed, uint64_t idx0, long n, uint32_t* out) {
  if (!out || n < 0) return -1;
  for (long i = 0; i < n; ++i) out[i] = ea_hash(seed, idx0 + (uint64_t)i);
  return 0;
}
extern "C" int ea_scale_dropout_bf16(const void* x, const void* y, void* out, long n, float a, float b,
                                     uint64_t seed, uint32_t thr, float inv_keep, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scale_dropout_kernel, dim3(grid_for(n, 8)), dim3(256), 0, stream, (const bf16_t*)x,
                     (const bf16_t*)y, (bf16_t*)out, n, a, b, seed, thr, inv_keep);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_colsum_bf16(const void* X, float* out, int M, int N, long ld, hipStream_t stream) {
  if (M <= 0 || N <= 0) return 0;
  int rpb = (M + 255) / 256;
  if (rpb < 64) rpb = 64;
  dim3 grid((N + 63) / 64, (M + rpb - 1) / rpb);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, stream, (const bf16_t*)X, out, M, N, ld, rpb);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_zero_rows_bf16(void* x, const uint8_t* row_zero, int M, int C, hipStream_t stream) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(zero_rows_kernel, dim3(M), dim3(256), 0, stream, (bf16_t*)x, row_zero, M, C);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_embedding_fwd(const int* tokens, const int* positions, const float* W, const float* pos_table, void* out,
                                int M, int C, float scale, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 2) return -2;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, tokens, positions, W, pos_table, (bf16_t*)out, M, C, scale);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_embedding_bwd(const int* tokens, const void* dy, float* dW, int M, int C, float scale, int pad_idx,
                                hipStream_t stream) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, tokens, (const bf16_t*)dy, dW, M, C, scale, pad_idx);
  return EA_CHECK_LAUNCH();
}
