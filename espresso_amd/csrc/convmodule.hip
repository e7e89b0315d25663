// Conformer convolution module — the HBM-bound middle (GLU -> depthwise Conv1d k<=31 -> BatchNorm1d
// -> SiLU) between the two point-wise convolutions, which run on the MFMA GEMM.
// Reference: fairseq/modules/conformer_layer.py:79-101 (forward), ctor :48-77.  As in the reference,
// NO padding mask is applied inside the module: padded frames flow through the depthwise conv and
// enter the BatchNorm batch statistics.
//
// Layout: activations are [B][T][C] (row m = b*T + t), channels contiguous.  Each thread owns two
// adjacent channels (4-byte bf16x2 accesses -> 256 B per wavefront) and slides a register window
// along time, so every input element is read once per tile (+halo) and the 31 taps stay in VGPRs.
#include "common.h"
#include "espresso_amd.h"

namespace {

constexpr int KMAX = 31;
constexpr int TT = 32;  // output timesteps per block

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

// Y [M][2C] -> U = a*sigmoid(g) [M][C] (saved), Z = dwconv(U) [M][C] (pre-BN), stats += (sum, sumsq)
template <int KW>
__global__ __launch_bounds__(128) void glu_dwconv_fwd_kernel(const bf16_t* __restrict__ Y,
                                                             const float* __restrict__ w,  // [C][KW]
                                                             bf16_t* __restrict__ U, bf16_t* __restrict__ Z,
                                                             float* __restrict__ stats, int T, int C) {
  constexpr int PAD = (KW - 1) / 2;
  const int c = (blockIdx.x * 128 + threadIdx.x) * 2;
  if (c >= C) return;
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * TT;
  float w0[KW], w1[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) {
    w0[k] = w[(long)c * KW + k];
    w1[k] = w[(long)(c + 1) * KW + k];
  }
  float x0[KW], x1[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) x0[k] = x1[k] = 0.f;
  float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
  const long rowbase = (long)b * T;
  // stream inputs t0-PAD .. t0+TT-1+PAD ; output t = tin - PAD once the window is full
  for (int tin = t0 - PAD; tin < t0 + TT + PAD; ++tin) {
    float u0 = 0.f, u1 = 0.f;
    if (tin >= 0 && tin < T) {
      const bf16_t* yr = Y + (rowbase + tin) * (2L * C);
      const uint32_t a = *reinterpret_cast<const uint32_t*>(yr + c);
      const uint32_t g = *reinterpret_cast<const uint32_t*>(yr + C + c);
      u0 = __uint_as_float(a << 16) * sigmoid_f(__uint_as_float(g << 16));
      u1 = __uint_as_float(a & 0xffff0000u) * sigmoid_f(__uint_as_float(g & 0xffff0000u));
      if (tin >= t0 && tin < t0 + TT) *reinterpret_cast<uint32_t*>(U + (rowbase + tin) * C + c) = pack_bf2(u0, u1);
      // conv consumes the bf16-rounded U (what backward will see)
      u0 = bf2f(f2bf(u0));
      u1 = bf2f(f2bf(u1));
    }
#pragma unroll
    for (int k = 0; k < KW - 1; ++k) {
      x0[k] = x0[k + 1];
      x1[k] = x1[k + 1];
    }
    x0[KW - 1] = u0;
    x1[KW - 1] = u1;
    const int tout = tin - PAD;
    if (tout >= t0 && tout < T) {
      float z0 = 0.f, z1 = 0.f;
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        z0 += w0[k] * x0[k];
        z1 += w1[k] * x1[k];
      }
      *reinterpret_cast<uint32_t*>(Z + (rowbase + tout) * C + c) = pack_bf2(z0, z1);
      z0 = bf2f(f2bf(z0));
      z1 = bf2f(f2bf(z1));
      s0 += z0; s1 += z1; q0 += z0 * z0; q1 += z1 * z1;
    }
  }
  if (stats) {
    atomicAdd(stats + c, s0);
    atomicAdd(stats + c + 1, s1);
    atomicAdd(stats + C + c, q0);
    atomicAdd(stats + C + c + 1, q1);
  }
}

// stats (sum,sumsq over n rows) -> mean/rstd ; running stats update (momentum, unbiased var)
__global__ void bn_finalize_kernel(const float* __restrict__ stats, float* __restrict__ mean_rstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   int C, float n, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = stats[c] / n;
  float var = stats[C + c] / n - mean * mean;
  var = fmaxf(var, 0.f);
  mean_rstd[c] = mean;
  mean_rstd[C + c] = rsqrtf(var + eps);
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    const float unb = n > 1.f ? var * n / (n - 1.f) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
  }
}
// eval mode: mean/rstd from running stats
__global__ void bn_from_running_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                       float* __restrict__ mean_rstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean_rstd[c] = running_mean[c];
  mean_rstd[C + c] = rsqrtf(running_var[c] + eps);
}

// H = act( (Z - mean) * rstd * gamma + beta ),  act = SiLU (act=2), ReLU (act=1) or identity
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const bf16_t* __restrict__ Z, const float* __restrict__ mean_rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16_t* __restrict__ Hout, long M, int C, int act) {
  const int nch = C >> 3;
  const long total = M * nch;
  const long stride = (long)gridDim.x * blockDim.x;
  // host guarantees stride % nch == 0: the channel chunk of a thread never changes -> hoist its constants
  const int ch = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) % nch);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = ch * 8 + e;
    sc[e] = mean_rstd[C + c] * gamma[c];
    sh[e] = beta[c] - mean_rstd[c] * sc[e];
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long m = i / nch;
    const uint4 u = *reinterpret_cast<const uint4*>(Z + m * C + ch * 8);
    const uint32_t wv[4] = {u.x, u.y, u.z, u.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float z = (e & 1) ? __uint_as_float(wv[e >> 1] & 0xffff0000u) : __uint_as_float(wv[e >> 1] << 16);
      const float y = z * sc[e] + sh[e];
      o[e] = act == 2 ? silu_f(y) : (act == 1 ? fmaxf(y, 0.f) : y);
    }
    uint4 r;
    r.x = pack_bf2(o[0], o[1]); r.y = pack_bf2(o[2], o[3]); r.z = pack_bf2(o[4], o[5]); r.w = pack_bf2(o[6], o[7]);
    *reinterpret_cast<uint4*>(Hout + m * C + ch * 8) = r;
  }
}

// BN backward pass 1: red[0][c] += sum dy, red[1][c] += sum dy*xhat, dy = dH * act'(y)
// block = 256 threads = TC channel pairs x (256/TC) row lanes; rows split over blockIdx.y
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const bf16_t* __restrict__ Z, const bf16_t* __restrict__ dH,
                                                                const float* __restrict__ mean_rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ red, long M, int C, int act,
                                                                int rows_per_block, int TC) {
  __shared__ float sm[2][256][2];
  const int cx = threadIdx.x % TC, ry = threadIdx.x / TC, TR = 256 / TC;
  const int c = (blockIdx.x * TC + cx) * 2;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(M, r0 + rows_per_block);
  float sa0 = 0.f, sa1 = 0.f, sb0 = 0.f, sb1 = 0.f;
  if (c < C) {
    const float m0 = mean_rstd[c], m1 = mean_rstd[c + 1], i0 = mean_rstd[C + c], i1 = mean_rstd[C + c + 1];
    const float g0 = gamma[c], g1 = gamma[c + 1], b0 = beta[c], b1 = beta[c + 1];
    for (long m = r0 + ry; m < r1; m += TR) {
      const uint32_t zz = *reinterpret_cast<const uint32_t*>(Z + m * C + c);
      const uint32_t dd = *reinterpret_cast<const uint32_t*>(dH + m * C + c);
      const float xh0 = (__uint_as_float(zz << 16) - m0) * i0, xh1 = (__uint_as_float(zz & 0xffff0000u) - m1) * i1;
      const float y0 = xh0 * g0 + b0, y1 = xh1 * g1 + b1;
      float d0 = __uint_as_float(dd << 16), d1 = __uint_as_float(dd & 0xffff0000u);
      d0 *= act == 2 ? dsilu_f(y0) : (act == 1 ? (y0 > 0.f ? 1.f : 0.f) : 1.f);
      d1 *= act == 2 ? dsilu_f(y1) : (act == 1 ? (y1 > 0.f ? 1.f : 0.f) : 1.f);
      sa0 += d0; sa1 += d1; sb0 += d0 * xh0; sb1 += d1 * xh1;
    }
  }
  sm[0][threadIdx.x][0] = sa0; sm[0][threadIdx.x][1] = sa1;
  sm[1][threadIdx.x][0] = sb0; sm[1][threadIdx.x][1] = sb1;
  __syncthreads();
  if (threadIdx.x < TC && c < C) {
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    for (int r = 0; r < TR; ++r) {
      a0 += sm[0][r * TC + cx][0]; a1 += sm[0][r * TC + cx][1];
      b0 += sm[1][r * TC + cx][0]; b1 += sm[1][r * TC + cx][1];
    }
    atomicAdd(red + c, a0);
    atomicAdd(red + c + 1, a1);
    atomicAdd(red + C + c, b0);
    atomicAdd(red + C + c + 1, b1);
  }
}

// BN backward pass 2: dZ = rstd*gamma*(dy - sum_dy/n - xhat*sum_dyxh/n)   (training)
//                     dZ = rstd*gamma*dy                                   (eval: n <= 0)
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const bf16_t* __restrict__ Z, const bf16_t* __restrict__ dH,
                                                               const float* __restrict__ mean_rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ red, bf16_t* __restrict__ dZ,
                                                               long M, int C, int act, float n) {
  const int nch = C >> 3;
  const long total = M * nch;
  const float invn = n > 0.f ? 1.f / n : 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  const int ch = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) % nch);  // constant per thread (stride % nch == 0)
  float mu[8], rs[8], ga[8], be[8], r0[8], r1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = ch * 8 + e;
    mu[e] = mean_rstd[c]; rs[e] = mean_rstd[C + c]; ga[e] = gamma[c]; be[e] = beta[c];
    r0[e] = red[c] * invn; r1[e] = red[C + c] * invn;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long m = i / nch;
    const uint4 uz = *reinterpret_cast<const uint4*>(Z + m * C + ch * 8);
    const uint4 ud = *reinterpret_cast<const uint4*>(dH + m * C + ch * 8);
    const uint32_t wz[4] = {uz.x, uz.y, uz.z, uz.w}, wd[4] = {ud.x, ud.y, ud.z, ud.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float z = (e & 1) ? __uint_as_float(wz[e >> 1] & 0xffff0000u) : __uint_as_float(wz[e >> 1] << 16);
      float d = (e & 1) ? __uint_as_float(wd[e >> 1] & 0xffff0000u) : __uint_as_float(wd[e >> 1] << 16);
      const float xh = (z - mu[e]) * rs[e];
      const float y = xh * ga[e] + be[e];
      d *= act == 2 ? dsilu_f(y) : (act == 1 ? (y > 0.f ? 1.f : 0.f) : 1.f);
      o[e] = rs[e] * ga[e] * (d - r0[e] - xh * r1[e]);
    }
    uint4 r;
    r.x = pack_bf2(o[0], o[1]); r.y = pack_bf2(o[2], o[3]); r.z = pack_bf2(o[4], o[5]); r.w = pack_bf2(o[6], o[7]);
    *reinterpret_cast<uint4*>(dZ + m * C + ch * 8) = r;
  }
}

// dU[t] = sum_k w[k] * dZ[t + PAD - k]; then GLU backward -> dY [M][2C]
template <int KW>
__global__ __launch_bounds__(128) void glu_dwconv_bwd_data_kernel(const bf16_t* __restrict__ dZ, const bf16_t* __restrict__ Y,
                                                                  const float* __restrict__ w, bf16_t* __restrict__ dY,
                                                                  int T, int C) {
  constexpr int PAD = (KW - 1) / 2;
  const int c = (blockIdx.x * 128 + threadIdx.x) * 2;
  if (c >= C) return;
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * TT;
  float w0[KW], w1[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) {  // flipped taps
    w0[k] = w[(long)c * KW + (KW - 1 - k)];
    w1[k] = w[(long)(c + 1) * KW + (KW - 1 - k)];
  }
  float x0[KW], x1[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) x0[k] = x1[k] = 0.f;
  const long rowbase = (long)b * T;
  for (int tin = t0 - PAD; tin < t0 + TT + PAD; ++tin) {
    float d0 = 0.f, d1 = 0.f;
    if (tin >= 0 && tin < T) {
      const uint32_t dd = *reinterpret_cast<const uint32_t*>(dZ + (rowbase + tin) * C + c);
      d0 = __uint_as_float(dd << 16);
      d1 = __uint_as_float(dd & 0xffff0000u);
    }
#pragma unroll
    for (int k = 0; k < KW - 1; ++k) {
      x0[k] = x0[k + 1];
      x1[k] = x1[k + 1];
    }
    x0[KW - 1] = d0;
    x1[KW - 1] = d1;
    const int tout = tin - PAD;
    if (tout >= t0 && tout < T) {
      // window holds dZ[tout-PAD .. tout+PAD] at x[0..KW-1];  dU[tout] = sum_j w[j]*dZ[tout+PAD-j]
      float u0 = 0.f, u1 = 0.f;
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        u0 += w0[k] * x0[k];
        u1 += w1[k] * x1[k];
      }
      const bf16_t* yr = Y + (rowbase + tout) * (2L * C);
      const uint32_t a = *reinterpret_cast<const uint32_t*>(yr + c);
      const uint32_t g = *reinterpret_cast<const uint32_t*>(yr + C + c);
      const float a0 = __uint_as_float(a << 16), a1 = __uint_as_float(a & 0xffff0000u);
      const float sg0 = sigmoid_f(__uint_as_float(g << 16)), sg1 = sigmoid_f(__uint_as_float(g & 0xffff0000u));
      bf16_t* dyr = dY + (rowbase + tout) * (2L * C);
      *reinterpret_cast<uint32_t*>(dyr + c) = pack_bf2(u0 * sg0, u1 * sg1);
      *reinterpret_cast<uint32_t*>(dyr + C + c) = pack_bf2(u0 * a0 * sg0 * (1.f - sg0), u1 * a1 * sg1 * (1.f - sg1));
    }
  }
}

// dw[c][k] += sum_{b,t} dZ[b,t,c] * U[b,t-PAD+k,c]
// Weight gradient of the depthwise conv: one channel per thread, 32-step time tiles -> many wavefronts to hide the
// dependent-load latency; each block writes its partial [C][KW] slab (no atomics), a second kernel sums slabs.
constexpr int TTW = 32;
template <int KW>
__global__ __launch_bounds__(128) void dwconv_bwd_weight_kernel(const bf16_t* __restrict__ dZ, const bf16_t* __restrict__ U,
                                                                float* __restrict__ part, int T, int C) {
  constexpr int PAD = (KW - 1) / 2;
  const int c = blockIdx.x * 128 + threadIdx.x;
  if (c >= C) return;
  const int b = blockIdx.z;
  const int t0 = blockIdx.y * TTW;
  float a0[KW], x0[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) a0[k] = x0[k] = 0.f;
  const long rowbase = (long)b * T;
  for (int tin = t0 - PAD; tin < t0 + TTW + PAD; ++tin) {
    float u0 = 0.f;
    if (tin >= 0 && tin < T) u0 = bf2f(U[(rowbase + tin) * C + c]);
#pragma unroll
    for (int k = 0; k < KW - 1; ++k) x0[k] = x0[k + 1];
    x0[KW - 1] = u0;
    const int tout = tin - PAD;
    if (tout >= t0 && tout < T) {
      const float d0 = bf2f(dZ[(rowbase + tout) * C + c]);
#pragma unroll
      for (int k = 0; k < KW; ++k) a0[k] += d0 * x0[k];
    }
  }
  float* out = part + ((long)(blockIdx.z * gridDim.y + blockIdx.y) * C + c) * KW;
#pragma unroll
  for (int k = 0; k < KW; ++k) out[k] = a0[k];
}
__global__ __launch_bounds__(256) void dwconv_weight_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nslab,
                                                                   int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int s = 0; s < nslab; ++s) a += part[(long)s * n + i];
  dw[i] += a;
}

static inline int egrid(long n) {
  long b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}
// grid whose total thread count is a multiple of nch (so a thread keeps one channel chunk)
static inline int egrid_ch(long n, int nch) {
  long b = egrid(n);
  while ((b * 256) % nch) ++b;
  return (int)b;
}

}  // namespace

#define EA_KW_DISPATCH(KW, FN, ...)                  \
  switch (KW) {                                      \
    case 3: FN<3>(__VA_ARGS__); break;               \
    case 7: FN<7>(__VA_ARGS__); break;               \
    case 15: FN<15>(__VA_ARGS__); break;             \
    case 31: FN<31>(__VA_ARGS__); break;             \
    default: return -2;                              \
  }

template <int KW>
static void launch_glu_dwconv_fwd(dim3 grid, hipStream_t stream, const bf16_t* Y, const float* w, bf16_t* U, bf16_t* Z,
                                  float* stats, int T, int C) {
  hipLaunchKernelGGL((glu_dwconv_fwd_kernel<KW>), grid, dim3(128), 0, stream, Y, w, U, Z, stats, T, C);
}
template <int KW>
static void launch_glu_dwconv_bwd_data(dim3 grid, hipStream_t stream, const bf16_t* dZ, const bf16_t* Y, const float* w,
                                       bf16_t* dY, int T, int C) {
  hipLaunchKernelGGL((glu_dwconv_bwd_data_kernel<KW>), grid, dim3(128), 0, stream, dZ, Y, w, dY, T, C);
}
template <int KW>
static void launch_dwconv_bwd_weight(dim3 grid, hipStream_t stream, const bf16_t* dZ, const bf16_t* U, float* dw, int T,
                                     int C) {
  hipLaunchKernelGGL((dwconv_bwd_weight_kernel<KW>), grid, dim3(128), 0, stream, dZ, U, dw, T, C);
}

extern "C" int ea_glu_dwconv_fwd(const void* Y, const float* w, void* U, void* Z, float* stats, int B, int T,
                                 int C, int KW, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if (C % 2) return -2;
  dim3 grid((C / 2 + 127) / 128, (T + TT - 1) / TT, B);
  EA_KW_DISPATCH(KW, launch_glu_dwconv_fwd, grid, stream, (const bf16_t*)Y, w, (bf16_t*)U, (bf16_t*)Z, stats, T, C);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_bn_finalize(const float* stats, float* mean_rstd, float* running_mean, float* running_var, int C,
                              float n, float eps, float momentum, hipStream_t stream) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, stats, mean_rstd, running_mean,
                     running_var, C, n, eps, momentum);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_bn_from_running(const float* running_mean, const float* running_var, float* mean_rstd, int C,
                                  float eps, hipStream_t stream) {
  hipLaunchKernelGGL(bn_from_running_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, running_mean, running_var,
                     mean_rstd, C, eps);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_bn_act_fwd(const void* Z, const float* mean_rstd, const float* gamma, const float* beta, void* H,
                             long M, int C, int act, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8) return -2;
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(egrid_ch(M * (C / 8), C / 8)), dim3(256), 0, stream, (const bf16_t*)Z, mean_rstd,
                     gamma, beta, (bf16_t*)H, M, C, act);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_bn_act_bwd(const void* Z, const void* dH, const float* mean_rstd, const float* gamma,
                             const float* beta, float* red /*[2][C] zeroed*/, void* dZ, float* dgamma, float* dbeta,
                             long M, int C, int act, int training, hipStream_t stream);

namespace {
__global__ void bn_param_grad_kernel(const float* __restrict__ red, float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] += red[c];
  if (dgamma) dgamma[c] += red[C + c];
}
}  // namespace

extern "C" int ea_bn_act_bwd(const void* Z, const void* dH, const float* mean_rstd, const float* gamma,
                             const float* beta, float* red, void* dZ, float* dgamma, float* dbeta, long M, int C,
                             int act, int training, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8) return -2;
  int TC = 1;
  while (TC < 256 && TC * 2 < C) TC <<= 1;  // channel-pair threads per block (power of two <= 256)
  int rpb = (int)((M + 1023) / 1024);
  const int TR = 256 / TC;
  if (rpb < 8 * TR) rpb = 8 * TR;
  dim3 g1((C / 2 + TC - 1) / TC, (unsigned)((M + rpb - 1) / rpb));
  hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, g1, dim3(256), 0, stream, (const bf16_t*)Z, (const bf16_t*)dH,
                     mean_rstd, gamma, beta, red, M, C, act, rpb, TC);
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(egrid_ch(M * (C / 8), C / 8)), dim3(256), 0, stream, (const bf16_t*)Z,
                     (const bf16_t*)dH, mean_rstd, gamma, beta, red, (bf16_t*)dZ, M, C, act,
                     training ? (float)M : 0.f);
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, red, dgamma, dbeta, C);
  return EA_CHECK_LAUNCH();
}

extern "C" long ea_dwconv_wgrad_workspace_bytes(int B, int T, int C, int KW) {
  return (long)B * ((T + TTW - 1) / TTW) * C * KW * (long)sizeof(float);
}

extern "C" int ea_glu_dwconv_bwd(const void* dZ, const void* Y, const void* U, const float* w, void* dY, float* dw,
                                 void* wgrad_ws, int B, int T, int C, int KW, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if (C % 2) return -2;
  dim3 grid((C / 2 + 127) / 128, (T + TT - 1) / TT, B);
  EA_KW_DISPATCH(KW, launch_glu_dwconv_bwd_data, grid, stream, (const bf16_t*)dZ, (const bf16_t*)Y, w, (bf16_t*)dY, T, C);
  dim3 gridw((C + 127) / 128, (T + TTW - 1) / TTW, B);
  float* part = (float*)wgrad_ws;
  EA_KW_DISPATCH(KW, launch_dwconv_bwd_weight, gridw, stream, (const bf16_t*)dZ, (const bf16_t*)U, part, T, C);
  hipLaunchKernelGGL(dwconv_weight_reduce_kernel, dim3((C * KW + 255) / 256), dim3(256), 0, stream, part, dw,
                     (int)(gridw.y * gridw.z), C * KW);
  return EA_CHECK_LAUNCH();
}
