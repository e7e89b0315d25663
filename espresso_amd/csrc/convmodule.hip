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

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }  // (v_rcp_f32, see silu_f)

// Depthwise-conv tiles: one workgroup = CT channels x TTILE output steps of one utterance; the input rows (plus the KW-1
// halo) are staged once in LDS with every global load in flight at once, then each thread produces TPT consecutive outputs
// of one channel from registers.  (The first version streamed the rows through a per-thread shift register: 62 dependent
// row loads per wavefront and only ~800 wavefronts in flight — 51 us where the traffic is worth 5 us.)
constexpr int CT = 64, TTILE = 64, TPT = TTILE / 4;

// Y [M][2C] -> U = a*sigmoid(g) [M][C] (saved), Z = dwconv(U) [M][C] (pre-BN), stats += (sum, sumsq)
template <int KW>
__global__ __launch_bounds__(256) void glu_dwconv_fwd_kernel(const bf16_t* __restrict__ Y,
                                                             const float* __restrict__ w,  // [C][KW]
                                                             bf16_t* __restrict__ U, bf16_t* __restrict__ Z,
                                                             double* __restrict__ stats, int T, int C) {
  constexpr int PAD = (KW - 1) / 2, ROWS = TTILE + KW - 1;
  __shared__ __attribute__((aligned(16))) float su[ROWS][CT];
  __shared__ float sred[2][4][CT];
  const int c0 = blockIdx.x * CT, t0 = blockIdx.y * TTILE, b = blockIdx.z;
  const long rowbase = (long)b * T;
  if ((C & 7) == 0 && c0 + CT <= C) {
    // Round 6 fast path (full-width channel tile): the loads of all staging trips are requested first, branch-free (clamped rows,
    // zeros selected afterwards), the gated tile is kept in LDS as the bf16 values the convolution consumes anyway, and Z leaves
    // as 16-byte rows through LDS instead of TPT two-byte stores per thread.  Same arithmetic, same rounding points.
    bf16_t (*sub)[CT] = reinterpret_cast<bf16_t (*)[CT]>(&su[0][0]);              // [ROWS][CT] bf16: 12 KB of the 24 KB array
    bf16_t (*sz)[CT] = reinterpret_cast<bf16_t (*)[CT]>(&su[0][0]) + ROWS;        // [TTILE][CT] bf16 behind it: 8 KB
    constexpr int NCH = (ROWS * (CT / 8) + 255) / 256;
    uint4 va[NCH], vg[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int row = min(i / (CT / 8), ROWS - 1), c8 = (i % (CT / 8)) * 8;
      const int tin = min(max(t0 - PAD + row, 0), T - 1);
      const bf16_t* yr = Y + (rowbase + tin) * (2L * C) + c0 + c8;
      va[k] = *reinterpret_cast<const uint4*>(yr);
      vg[k] = *reinterpret_cast<const uint4*>(yr + C);
    }
    const int cl = threadIdx.x & (CT - 1), grp = threadIdx.x >> 6;
    float wk[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) wk[k] = w[(long)(c0 + cl) * KW + k];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < ROWS * (CT / 8)) {
        const int row = i / (CT / 8), c8 = (i % (CT / 8)) * 8;
        const int tin = t0 - PAD + row;
        const uint32_t aw[4] = {va[k].x, va[k].y, va[k].z, va[k].w}, gw[4] = {vg[k].x, vg[k].y, vg[k].z, vg[k].w};
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pk[e] = pack_bf2(__uint_as_float(aw[e] << 16) * sigmoid_f(__uint_as_float(gw[e] << 16)),
                           __uint_as_float(aw[e] & 0xffff0000u) * sigmoid_f(__uint_as_float(gw[e] & 0xffff0000u)));
        const bool inside = tin >= 0 && tin < T;
        const uint4 u4 = inside ? make_uint4(pk[0], pk[1], pk[2], pk[3]) : make_uint4(0, 0, 0, 0);
        if (inside && row >= PAD && row < PAD + TTILE) *reinterpret_cast<uint4*>(U + (rowbase + tin) * C + c0 + c8) = u4;
        *reinterpret_cast<uint4*>(&sub[row][c8]) = u4;
      }
    }
    __syncthreads();
    float acc[TPT];
#pragma unroll
    for (int j = 0; j < TPT; ++j) acc[j] = 0.f;
#pragma unroll
    for (int r = 0; r < TPT + KW - 1; ++r) {
      const float x = bf2f(sub[grp * TPT + r][cl]);
#pragma unroll
      for (int j = 0; j < TPT; ++j)
        if (r - j >= 0 && r - j < KW) acc[j] += wk[r - j] * x;
    }
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < TPT; ++j) {
      const int tout = t0 + grp * TPT + j;
      const bf16_t zb = f2bf(acc[j]);
      sz[grp * TPT + j][cl] = zb;
      if (tout < T) {
        const float z = bf2f(zb);
        s += z;
        q += z * z;
      }
    }
    if (stats) {
      sred[0][grp][cl] = s;
      sred[1][grp][cl] = q;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = threadIdx.x + 256 * k;  // [row][chunk]
      const int r = i >> 3, c8 = (i & 7) * 8;
      if (t0 + r < T) *reinterpret_cast<uint4*>(Z + (rowbase + t0 + r) * C + c0 + c8) = *reinterpret_cast<const uint4*>(&sz[r][c8]);
    }
    if (stats && threadIdx.x < CT) {
      const int x = threadIdx.x;
      atomicAdd(stats + c0 + x, (double)(sred[0][0][x] + sred[0][1][x] + sred[0][2][x] + sred[0][3][x]));
      atomicAdd(stats + C + c0 + x, (double)(sred[1][0][x] + sred[1][1][x] + sred[1][2][x] + sred[1][3][x]));
    }
    return;
  }
  // staging: 8 channels (16 bytes of a and of g) per thread and row
  const bool vec = (C & 7) == 0;
  for (int i = threadIdx.x; i < ROWS * (CT / 8); i += 256) {
    const int row = i / (CT / 8), c8 = (i % (CT / 8)) * 8;
    const int tin = t0 - PAD + row;
    float u[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] = 0.f;
    if (tin >= 0 && tin < T && c0 + c8 < C) {
      const bf16_t* yr = Y + (rowbase + tin) * (2L * C) + c0 + c8;
      uint32_t pk[4] = {0, 0, 0, 0};
      if (vec && c0 + c8 + 8 <= C) {
        const uint4 av = *reinterpret_cast<const uint4*>(yr);
        const uint4 gv = *reinterpret_cast<const uint4*>(yr + C);
        const uint32_t aw[4] = {av.x, av.y, av.z, av.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          pk[e] = pack_bf2(__uint_as_float(aw[e] << 16) * sigmoid_f(__uint_as_float(gw[e] << 16)),
                           __uint_as_float(aw[e] & 0xffff0000u) * sigmoid_f(__uint_as_float(gw[e] & 0xffff0000u)));
        if (row >= PAD && row < PAD + TTILE)
          *reinterpret_cast<uint4*>(U + (rowbase + tin) * C + c0 + c8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      } else {
        for (int e = 0; e < 8; ++e) {
          if (c0 + c8 + e >= C) break;
          const bf16_t ub = f2bf(bf2f(yr[e]) * sigmoid_f(bf2f(yr[C + e])));
          if (row >= PAD && row < PAD + TTILE) U[(rowbase + tin) * C + c0 + c8 + e] = ub;
          pk[e >> 1] |= (uint32_t)ub << (16 * (e & 1));
        }
      }
      // the conv consumes the bf16-rounded U (what backward will see)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u[2 * e] = __uint_as_float(pk[e] << 16);
        u[2 * e + 1] = __uint_as_float(pk[e] & 0xffff0000u);
      }
    }
    *reinterpret_cast<float4*>(&su[row][c8]) = make_float4(u[0], u[1], u[2], u[3]);
    *reinterpret_cast<float4*>(&su[row][c8 + 4]) = make_float4(u[4], u[5], u[6], u[7]);
  }
  const int cl = threadIdx.x & (CT - 1), grp = threadIdx.x >> 6;
  const int c = c0 + cl;
  float wk[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) wk[k] = c < C ? w[(long)c * KW + k] : 0.f;
  __syncthreads();
  float acc[TPT];
#pragma unroll
  for (int j = 0; j < TPT; ++j) acc[j] = 0.f;
  // output j of this thread sits at tile step grp*TPT + j and reads LDS rows grp*TPT + j + k, k = 0..KW-1
#pragma unroll
  for (int r = 0; r < TPT + KW - 1; ++r) {
    const float x = su[grp * TPT + r][cl];
#pragma unroll
    for (int j = 0; j < TPT; ++j)
      if (r - j >= 0 && r - j < KW) acc[j] += wk[r - j] * x;
  }
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int j = 0; j < TPT; ++j) {
    const int tout = t0 + grp * TPT + j;
    if (tout < T && c < C) {
      const bf16_t zb = f2bf(acc[j]);
      Z[(rowbase + tout) * C + c] = zb;
      const float z = bf2f(zb);
      s += z;
      q += z * z;
    }
  }
  if (stats) {
    sred[0][grp][cl] = s;
    sred[1][grp][cl] = q;
    __syncthreads();
    if (threadIdx.x < CT && c0 + threadIdx.x < C) {
      const int x = threadIdx.x;
      // fp64 accumulators: the order of the atomics no longer shows in the fp32 statistics (run-to-run reproducible)
      atomicAdd(stats + c0 + x, (double)(sred[0][0][x] + sred[0][1][x] + sred[0][2][x] + sred[0][3][x]));
      atomicAdd(stats + C + c0 + x, (double)(sred[1][0][x] + sred[1][1][x] + sred[1][2][x] + sred[1][3][x]));
    }
  }
}

// stats (sum,sumsq over n rows) -> mean/rstd ; running stats update (momentum, unbiased var)
__global__ void bn_finalize_kernel(const double* __restrict__ stats, float* __restrict__ mean_rstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   int C, float n, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = (float)(stats[c] / (double)n);
  float var = (float)(stats[C + c] / (double)n - (stats[c] / (double)n) * (stats[c] / (double)n));
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

// Training-mode BatchNorm forward in ONE launch: every thread derives mean / rstd of its 8 channels from the fp64 batch sums
// (what bn_finalize_kernel did in a launch of its own), block 0 also records mean / rstd for the backward pass and updates the
// running statistics, and all blocks clear `zero_next` (the statistics buffer of the NEXT BatchNorm forward: two buffers take
// turns, so no fill launch precedes the accumulating kernel).
__global__ __launch_bounds__(256) void bn_act_fwd_stats_kernel(const bf16_t* __restrict__ Z, const double* __restrict__ stats,
                                                               float* __restrict__ mean_rstd, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, bf16_t* __restrict__ Hout, long M,
                                                               int C, int act, float n, float eps, float momentum,
                                                               double* __restrict__ zero_next, int zero_n) {
  extern __shared__ float s_scsh[];  // [2][C]: scale, shift of every channel (once per block, not once per thread)
  const int nch = C >> 3;
  const long total = M * nch;
  const long stride = (long)gridDim.x * blockDim.x;
  const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const double inv_n = 1.0 / (double)n;
  for (int c = threadIdx.x; c < C; c += 256) {
    // (bn_finalize_kernel's arithmetic with the fp64 divisions replaced by products with 1/n)
    const double m1 = stats[c] * inv_n;
    const float mean = (float)m1;
    float var = (float)(stats[C + c] * inv_n - m1 * m1);
    var = fmaxf(var, 0.f);
    const float rstd = rsqrtf(var + eps);
    const float scv = rstd * gamma[c];
    s_scsh[c] = scv;
    s_scsh[C + c] = beta[c] - mean * scv;
    if (blockIdx.x == 0) {  // one block records mean / rstd for the backward pass and updates the running statistics
      mean_rstd[c] = mean;
      mean_rstd[C + c] = rstd;
      if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        const float unb = n > 1.f ? var * n / (n - 1.f) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
      }
    }
  }
  __syncthreads();
  const int ch = (int)(gtid % nch);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = s_scsh[ch * 8 + e];
    sh[e] = s_scsh[C + ch * 8 + e];
  }
  for (long i = gtid; i < total; i += stride) {
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
  if (zero_next)  // (the whole buffer, not 2*C entries: the next user may have more channels than this call)
    for (long i = gtid; i < zero_n; i += stride) zero_next[i] = 0.0;
}

// BN backward pass 1: red[0][c] += sum dy, red[1][c] += sum dy*xhat, dy = dH * act'(y)
// A thread owns one 8-channel chunk (16-byte loads of Z and dH, per-channel constants hoisted) and walks rows in steps of the
// block's row lanes; per-block partial sums meet in LDS and cost 2*C atomics per block, so blocks take >= 32 rows per lane
// group (fp32 atomics run at ~33 G/s: the first version issued 2*C atomics per 8 rows and spent its time there).
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const bf16_t* __restrict__ Z, const bf16_t* __restrict__ dH,
                                                                const float* __restrict__ mean_rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ red, long M, int C, int act,
                                                                int rows_per_block, int TCH) {
  extern __shared__ float sm[];  // [lanes][TCH][16]
  const int nch = C >> 3;
  const int cx = threadIdx.x % TCH, ry = threadIdx.x / TCH, lanes = 256 / TCH;
  const int ch = blockIdx.x * TCH + cx;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(M, r0 + rows_per_block);
  float sa[8], sb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sa[e] = sb[e] = 0.f;
  if (ch < nch) {
    float mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = ch * 8 + e;
      mu[e] = mean_rstd[c]; rs[e] = mean_rstd[C + c]; ga[e] = gamma[c]; be[e] = beta[c];
    }
    auto accum = [&](const uint4& uz, const uint4& ud) {
      const uint32_t wz[4] = {uz.x, uz.y, uz.z, uz.w}, wd[4] = {ud.x, ud.y, ud.z, ud.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = (e & 1) ? __uint_as_float(wz[e >> 1] & 0xffff0000u) : __uint_as_float(wz[e >> 1] << 16);
        float d = (e & 1) ? __uint_as_float(wd[e >> 1] & 0xffff0000u) : __uint_as_float(wd[e >> 1] << 16);
        const float xh = (z - mu[e]) * rs[e];
        const float y = xh * ga[e] + be[e];
        d *= act == 2 ? dsilu_f(y) : (act == 1 ? (y > 0.f ? 1.f : 0.f) : 1.f);
        sa[e] += d;
        sb[e] += d * xh;
      }
    };
    // four of the thread's rows are requested together (round 6): one row per trip was a dependent 16-byte-load round trip per
    // row — 8 trips, 16.7 us for the Conformer layer's 6 240 x 512 rows (0.8 TB/s); same accumulation order, same bits
    constexpr int R = 4;
    long m = r0 + ry;
    for (; m + (long)(R - 1) * lanes < r1; m += (long)R * lanes) {
      uint4 uz[R], ud[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        uz[r] = *reinterpret_cast<const uint4*>(Z + (m + (long)r * lanes) * C + ch * 8);
        ud[r] = *reinterpret_cast<const uint4*>(dH + (m + (long)r * lanes) * C + ch * 8);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) accum(uz[r], ud[r]);
    }
    for (; m < r1; m += lanes) {
      const uint4 uz = *reinterpret_cast<const uint4*>(Z + m * C + ch * 8);
      const uint4 ud = *reinterpret_cast<const uint4*>(dH + m * C + ch * 8);
      accum(uz, ud);
    }
  }
  float* mine = sm + ((long)ry * TCH + cx) * 16;
#pragma unroll
  for (int e = 0; e < 8; ++e) { mine[e] = sa[e]; mine[8 + e] = sb[e]; }
  __syncthreads();
  // TCH*16 outputs per block: thread t sums column t over the row lanes
  for (int o = threadIdx.x; o < TCH * 16; o += 256) {
    const int cxo = o / 16, e = o % 16;
    const int cho = blockIdx.x * TCH + cxo;
    if (cho >= nch) continue;
    float a = 0.f;
    for (int r = 0; r < lanes; ++r) a += sm[((long)r * TCH + cxo) * 16 + e];
    atomicAdd(red + (e < 8 ? 0 : C) + cho * 8 + (e & 7), a);
  }
}

// BN backward pass 2: dZ = rstd*gamma*(dy - sum_dy/n - xhat*sum_dyxh/n)   (training)
//                     dZ = rstd*gamma*dy                                   (eval: n <= 0)
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const bf16_t* __restrict__ Z, const bf16_t* __restrict__ dH,
                                                               const float* __restrict__ mean_rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ red, bf16_t* __restrict__ dZ,
                                                               long M, int C, int act, float n,
                                                               float* __restrict__ dgamma = nullptr, float* __restrict__ dbeta = nullptr,
                                                               float* __restrict__ zero_next = nullptr, int zero_n = 0) {
  const int nch = C >> 3;
  const long total = M * nch;
  const float invn = n > 0.f ? 1.f / n : 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int ch = (int)(gtid % nch);  // constant per thread (stride % nch == 0)
  float mu[8], rs[8], ga[8], be[8], r0[8], r1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = ch * 8 + e;
    mu[e] = mean_rstd[c]; rs[e] = mean_rstd[C + c]; ga[e] = gamma[c]; be[e] = beta[c];
    r0[e] = red[c] * invn; r1[e] = red[C + c] * invn;
  }
  if (blockIdx.x == 0 && (dgamma || dbeta)) {  // fused bn_param_grad_kernel (one block; a workgroup-uniform branch off the main path)
    for (int c = threadIdx.x; c < C; c += 256) {
      if (dbeta) dbeta[c] += red[c];
      if (dgamma) dgamma[c] += red[C + c];
    }
  }
  if (zero_next && blockIdx.x == gridDim.x - 1)  // the whole sum buffer of the NEXT BatchNorm backward (two buffers take turns)
    for (int i = threadIdx.x; i < zero_n; i += 256) zero_next[i] = 0.f;
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

// dU[t] = sum_k w[k] * dZ[t + PAD - k]; then GLU backward -> dY [M][2C]   (same tiling as the forward kernel)
// Round 6: every global access is a 16-byte vector.  The thread mapping of the convolution (one channel, TPT consecutive time steps)
// made the GLU operands 2 x TPT two-byte loads and the result 2 x TPT two-byte stores per thread — 64 memory instructions per
// thread for 256 bytes, and the loads sat in the output loop behind the stores (one dependent round trip per output).  Now the
// Y tile is staged [half][t][c] in LDS with the dZ tile, each thread updates its own cells in place, and the tile leaves as
// 16-byte rows.  Full-width tiles only (C % 64 == 0 for this block); ragged channel tiles take the scalar path below.
// FUSE_BN (round 6): the BatchNorm + SiLU backward "apply" pass (bn_act_bwd_apply_kernel) happens while the tile is staged — the
// block reads Z and dH (the BatchNorm's input and the incoming gradient) instead of a stored dZ, forms dZ = BN'(dH) per 16-byte
// chunk in registers with the same arithmetic and the same bf16 rounding, feeds the convolution from LDS and ALSO stores the tile's
// own rows of dZ (the depthwise weight gradient on the side stream reads them).  One launch and one pass over [M][C] fewer per layer.
struct BnBwdFuse {
  const bf16_t* Z; const bf16_t* dH; const float* mean_rstd; const float* gamma; const float* beta; const float* red; bf16_t* dZ_out;
  float n; int act; float* dgamma; float* dbeta; float* zero_next; int zero_n;
};
template <int KW, bool FUSE_BN>
__global__ __launch_bounds__(256) void glu_dwconv_bwd_data_kernel(const bf16_t* __restrict__ dZ, const bf16_t* __restrict__ Y,
                                                                  const float* __restrict__ w, bf16_t* __restrict__ dY,
                                                                  int T, int C, const BnBwdFuse fb) {
  constexpr int PAD = (KW - 1) / 2, ROWS = TTILE + KW - 1;
  // one buffer, two views: [dZ tile bf16 ROWS x CT][Y / dY tile bf16 2 x TTILE x CT] (fast path) or the dZ tile in fp32 (ragged path)
  constexpr int SDZ_BYTES = ROWS * CT * 2, SY_BYTES = 2 * TTILE * CT * 2;
  static_assert(SDZ_BYTES % 16 == 0 && SDZ_BYTES + SY_BYTES >= ROWS * CT * 4, "views of the shared buffer");
  __shared__ __attribute__((aligned(16))) char smem[SDZ_BYTES + SY_BYTES];
  bf16_t (*sdz)[CT] = reinterpret_cast<bf16_t (*)[CT]>(smem);
  bf16_t (*sy)[TTILE][CT] = reinterpret_cast<bf16_t (*)[TTILE][CT]>(smem + SDZ_BYTES);
  const int c0 = blockIdx.x * CT, t0 = blockIdx.y * TTILE, b = blockIdx.z;
  const long rowbase = (long)b * T;
  const int cl = threadIdx.x & (CT - 1), grp = threadIdx.x >> 6;
  const int c = c0 + cl;
  if ((C & 7) == 0 && c0 + CT <= C) {
    // ---- all loads of the block first, branch-free (clamped rows, zeros selected afterwards) ----
    constexpr int NDZ = (ROWS * (CT / 8) + 255) / 256;  // 16-byte chunks of the dZ tile per thread
    uint4 vdz[NDZ], vy[4], vz[FUSE_BN ? NDZ : 1];
#pragma unroll
    for (int k = 0; k < NDZ; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int row = min(i / (CT / 8), ROWS - 1), c8 = (i % (CT / 8)) * 8;
      const int tin = min(max(t0 - PAD + row, 0), T - 1);
      if constexpr (FUSE_BN) {
        vz[k] = *reinterpret_cast<const uint4*>(fb.Z + (rowbase + tin) * C + c0 + c8);
        vdz[k] = *reinterpret_cast<const uint4*>(fb.dH + (rowbase + tin) * C + c0 + c8);
      } else {
        vdz[k] = *reinterpret_cast<const uint4*>(dZ + (rowbase + tin) * C + c0 + c8);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = threadIdx.x + 256 * k;  // [half][row][chunk]
      const int half = i >> 9, r = (i >> 3) & 63, c8 = (i & 7) * 8;
      const int tout = min(t0 + r, T - 1);
      vy[k] = *reinterpret_cast<const uint4*>(Y + (rowbase + tout) * (2L * C) + (long)half * C + c0 + c8);
    }
    float wk[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) wk[k] = w[(long)c * KW + k];
    if constexpr (FUSE_BN) {
      // a thread's chunks all sit in the same 8 channels (256 % (CT / 8) == 0): their BatchNorm constants once
      const int cc = c0 + (threadIdx.x % (CT / 8)) * 8;
      const float invn = fb.n > 0.f ? 1.f / fb.n : 0.f;
      float mu[8], rs[8], ga[8], be[8], r0[8], r1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        mu[e] = fb.mean_rstd[cc + e]; rs[e] = fb.mean_rstd[C + cc + e]; ga[e] = fb.gamma[cc + e]; be[e] = fb.beta[cc + e];
        r0[e] = fb.red[cc + e] * invn; r1[e] = fb.red[C + cc + e] * invn;
      }
#pragma unroll
      for (int k = 0; k < NDZ; ++k) {  // bn_act_bwd_apply_kernel's arithmetic on this chunk
        const uint32_t wz[4] = {vz[k].x, vz[k].y, vz[k].z, vz[k].w}, wd[4] = {vdz[k].x, vdz[k].y, vdz[k].z, vdz[k].w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float z = (e & 1) ? __uint_as_float(wz[e >> 1] & 0xffff0000u) : __uint_as_float(wz[e >> 1] << 16);
          float d = (e & 1) ? __uint_as_float(wd[e >> 1] & 0xffff0000u) : __uint_as_float(wd[e >> 1] << 16);
          const float xh = (z - mu[e]) * rs[e];
          const float y = xh * ga[e] + be[e];
          d *= fb.act == 2 ? dsilu_f(y) : (fb.act == 1 ? (y > 0.f ? 1.f : 0.f) : 1.f);
          o[e] = rs[e] * ga[e] * (d - r0[e] - xh * r1[e]);
        }
        vdz[k] = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
        const int i = threadIdx.x + 256 * k;
        const int row = i / (CT / 8), c8 = (i % (CT / 8)) * 8;
        const int tin = t0 - PAD + row;
        if (i < ROWS * (CT / 8) && row >= PAD && row < PAD + TTILE && tin < T)  // the tile's own rows: dZ as the unfused pass stored it
          *reinterpret_cast<uint4*>(fb.dZ_out + (rowbase + tin) * C + c0 + c8) = vdz[k];
      }
      if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {  // (the apply kernel's block 0: parameter gradients, next ring buffer)
        if (fb.dgamma || fb.dbeta)
          for (int q = threadIdx.x; q < C; q += 256) {
            if (fb.dbeta) fb.dbeta[q] += fb.red[q];
            if (fb.dgamma) fb.dgamma[q] += fb.red[C + q];
          }
        if (fb.zero_next)
          for (int q = threadIdx.x; q < fb.zero_n; q += 256) fb.zero_next[q] = 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < NDZ; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < ROWS * (CT / 8)) {
        const int row = i / (CT / 8), c8 = (i % (CT / 8)) * 8;
        const int tin = t0 - PAD + row;
        *reinterpret_cast<uint4*>(&sdz[row][c8]) = (tin >= 0 && tin < T) ? vdz[k] : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = threadIdx.x + 256 * k;
      *reinterpret_cast<uint4*>(&sy[i >> 9][(i >> 3) & 63][(i & 7) * 8]) = vy[k];
    }
    __syncthreads();
    float acc[TPT];
#pragma unroll
    for (int j = 0; j < TPT; ++j) acc[j] = 0.f;
#pragma unroll
    for (int r = 0; r < TPT + KW - 1; ++r) {
      const float x = bf2f(sdz[grp * TPT + r][cl]);
#pragma unroll
      for (int j = 0; j < TPT; ++j) {
        const int k = j + 2 * PAD - r;
        if (k >= 0 && k < KW) acc[j] += wk[k] * x;
      }
    }
#pragma unroll
    for (int j = 0; j < TPT; ++j) {  // this thread's own cells of the Y tile become its cells of the dY tile
      const int r = grp * TPT + j;
      const float av = bf2f(sy[0][r][cl]);
      const float sg = sigmoid_f(bf2f(sy[1][r][cl]));
      sy[0][r][cl] = f2bf(acc[j] * sg);
      sy[1][r][cl] = f2bf(acc[j] * av * sg * (1.f - sg));
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int half = i >> 9, r = (i >> 3) & 63, c8 = (i & 7) * 8;
      if (t0 + r < T)
        *reinterpret_cast<uint4*>(dY + (rowbase + t0 + r) * (2L * C) + (long)half * C + c0 + c8) = *reinterpret_cast<const uint4*>(&sy[half][r][c8]);
    }
    return;
  }
  // ---- ragged channel tile: scalar path ----
  float (*sdf)[CT] = reinterpret_cast<float (*)[CT]>(smem);
  for (int i = threadIdx.x; i < ROWS * (CT / 8); i += 256) {
    const int row = i / (CT / 8), c8 = (i % (CT / 8)) * 8;
    const int tin = t0 - PAD + row;
    float d[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = 0.f;
    if (tin >= 0 && tin < T && c0 + c8 < C) {
      const bf16_t* src = dZ + (rowbase + tin) * C + c0 + c8;
      for (int e = 0; e < 8; ++e)
        if (c0 + c8 + e < C) d[e] = bf2f(src[e]);
    }
    *reinterpret_cast<float4*>(&sdf[row][c8]) = make_float4(d[0], d[1], d[2], d[3]);
    *reinterpret_cast<float4*>(&sdf[row][c8 + 4]) = make_float4(d[4], d[5], d[6], d[7]);
  }
  float wk[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) wk[k] = c < C ? w[(long)c * KW + k] : 0.f;
  __syncthreads();
  float acc[TPT];
#pragma unroll
  for (int j = 0; j < TPT; ++j) acc[j] = 0.f;
  // dU[tout] needs dZ[tout + PAD - k]: LDS row (grp*TPT + j) + 2*PAD - k, i.e. tap k = j + 2*PAD - r for row offset r
#pragma unroll
  for (int r = 0; r < TPT + KW - 1; ++r) {
    const float x = sdf[grp * TPT + r][cl];
#pragma unroll
    for (int j = 0; j < TPT; ++j) {
      const int k = j + 2 * PAD - r;
      if (k >= 0 && k < KW) acc[j] += wk[k] * x;
    }
  }
#pragma unroll
  for (int j = 0; j < TPT; ++j) {
    const int tout = t0 + grp * TPT + j;
    if (tout < T && c < C) {
      const bf16_t* yr = Y + (rowbase + tout) * (2L * C);
      const float av = bf2f(yr[c]);
      const float sg = sigmoid_f(bf2f(yr[C + c]));
      bf16_t* dyr = dY + (rowbase + tout) * (2L * C);
      dyr[c] = f2bf(acc[j] * sg);
      dyr[C + c] = f2bf(acc[j] * av * sg * (1.f - sg));
    }
  }
}

// dw[c][k] += sum_{b,t} dZ[b,t,c] * U[b,t-PAD+k,c]
// Weight gradient of the depthwise conv: workgroup = (CT channels, one utterance), looping over 64-step time tiles staged in
// LDS; thread (channel, tap group of 8) keeps a sliding window of U in registers.  One partial [C][KW] slab per utterance
// (no atomics), a second kernel sums the B slabs.
template <int KW>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const bf16_t* __restrict__ dZ, const bf16_t* __restrict__ U,
                                                                float* __restrict__ part, int T, int C, int tiles_per_block) {
  constexpr int PAD = (KW - 1) / 2, ROWS = TTILE + KW - 1, KG = 8;
  __shared__ __attribute__((aligned(16))) float sU[ROWS + KG][CT];
  __shared__ __attribute__((aligned(16))) float sD[TTILE][CT];
  const int c0 = blockIdx.x * CT, b = blockIdx.z;
  const long rowbase = (long)b * T;
  const int cl = threadIdx.x & (CT - 1), grp = threadIdx.x >> 6;  // taps grp*8 .. grp*8+7
  float acc[KG];
#pragma unroll
  for (int k = 0; k < KG; ++k) acc[k] = 0.f;
  const int tbeg = blockIdx.y * tiles_per_block * TTILE, tend = min(T, tbeg + tiles_per_block * TTILE);
  for (int t0 = tbeg; t0 < tend; t0 += TTILE) {
    __syncthreads();
    // staging with 16-byte loads (8 channels per thread and row; the first version moved two channels per load and spent its
    // time in address arithmetic: 35 us per launch for 12 MB)
    const bool c_ok = (C & 7) == 0;
    for (int i = threadIdx.x; i < (ROWS + KG) * (CT / 8); i += 256) {
      const int row = i / (CT / 8), c8 = (i % (CT / 8)) * 8;
      const int tin = t0 - PAD + row;
      float u[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) u[e] = 0.f;
      if (row < ROWS && tin >= 0 && tin < T) {
        const bf16_t* src = U + (rowbase + tin) * C + c0 + c8;
        if (c_ok && c0 + c8 + 8 <= C) {
          const uint4 v = *reinterpret_cast<const uint4*>(src);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            u[2 * e] = __uint_as_float(w[e] << 16);
            u[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
          }
        } else {
          for (int e = 0; e < 8; ++e)
            if (c0 + c8 + e < C) u[e] = bf2f(src[e]);
        }
      }
      *reinterpret_cast<float4*>(&sU[row][c8]) = make_float4(u[0], u[1], u[2], u[3]);
      *reinterpret_cast<float4*>(&sU[row][c8 + 4]) = make_float4(u[4], u[5], u[6], u[7]);
    }
    for (int i = threadIdx.x; i < TTILE * (CT / 8); i += 256) {
      const int row = i / (CT / 8), c8 = (i % (CT / 8)) * 8;
      const int t = t0 + row;
      float d[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = 0.f;
      if (t < T) {
        const bf16_t* src = dZ + (rowbase + t) * C + c0 + c8;
        if (c_ok && c0 + c8 + 8 <= C) {
          const uint4 v = *reinterpret_cast<const uint4*>(src);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            d[2 * e] = __uint_as_float(w[e] << 16);
            d[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
          }
        } else {
          for (int e = 0; e < 8; ++e)
            if (c0 + c8 + e < C) d[e] = bf2f(src[e]);
        }
      }
      *reinterpret_cast<float4*>(&sD[row][c8]) = make_float4(d[0], d[1], d[2], d[3]);
      *reinterpret_cast<float4*>(&sD[row][c8 + 4]) = make_float4(d[4], d[5], d[6], d[7]);
    }
    __syncthreads();
    // dw[k] += dZ[t] * U[t - PAD + k]  ->  LDS row of U = (t - t0) + k ; window x[kk] = sU[tt + grp*8 + kk]
    float x[KG];
#pragma unroll
    for (int kk = 0; kk < KG - 1; ++kk) x[kk + 1] = sU[grp * KG + kk][cl];
#pragma unroll
    for (int tt = 0; tt < TTILE; ++tt) {
#pragma unroll
      for (int kk = 0; kk < KG - 1; ++kk) x[kk] = x[kk + 1];
      x[KG - 1] = sU[tt + grp * KG + KG - 1][cl];
      const float d = sD[tt][cl];
#pragma unroll
      for (int kk = 0; kk < KG; ++kk) acc[kk] += d * x[kk];
    }
  }
  const int c = c0 + cl;
  if (c < C) {
    float* out = part + (((long)b * gridDim.y + blockIdx.y) * C + c) * KW;
#pragma unroll
    for (int kk = 0; kk < KG; ++kk)
      if (grp * KG + kk < KW) out[grp * KG + kk] = acc[kk];
  }
}
__global__ __launch_bounds__(256) void dwconv_weight_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nslab,
                                                                   int n) {
  // blockIdx.y takes every gridDim.y-th slab (62 workgroups each walking ~60 slabs serially were a 20 us latency chain)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int s = blockIdx.y; s < nslab; s += gridDim.y) a += part[(long)s * n + i];
  if (gridDim.y == 1) dw[i] += a;
  else atomicAdd(dw + i, a);
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
                                  double* stats, int T, int C) {
  hipLaunchKernelGGL((glu_dwconv_fwd_kernel<KW>), grid, dim3(256), 0, stream, Y, w, U, Z, stats, T, C);
}
template <int KW>
static void launch_glu_dwconv_bwd_data(dim3 grid, hipStream_t stream, const bf16_t* dZ, const bf16_t* Y, const float* w,
                                       bf16_t* dY, int T, int C) {
  hipLaunchKernelGGL((glu_dwconv_bwd_data_kernel<KW, false>), grid, dim3(256), 0, stream, dZ, Y, w, dY, T, C, BnBwdFuse{});
}
template <int KW>
static void launch_glu_dwconv_bwd_data_bn(dim3 grid, hipStream_t stream, const bf16_t* Y, const float* w, bf16_t* dY, int T, int C,
                                          BnBwdFuse fb) {
  hipLaunchKernelGGL((glu_dwconv_bwd_data_kernel<KW, true>), grid, dim3(256), 0, stream, (const bf16_t*)nullptr, Y, w, dY, T, C, fb);
}
template <int KW>
static void launch_dwconv_bwd_weight(dim3 grid, hipStream_t stream, const bf16_t* dZ, const bf16_t* U, float* dw, int T,
                                     int C, int tiles_per_block) {
  hipLaunchKernelGGL((dwconv_bwd_weight_kernel<KW>), grid, dim3(256), 0, stream, dZ, U, dw, T, C, tiles_per_block);
}

extern "C" int ea_glu_dwconv_fwd(const void* Y, const float* w, void* U, void* Z, double* stats, int B, int T,
                                 int C, int KW, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if (C % 2) return -2;
  dim3 grid((C + CT - 1) / CT, (T + TTILE - 1) / TTILE, B);
  EA_KW_DISPATCH(KW, launch_glu_dwconv_fwd, grid, stream, (const bf16_t*)Y, w, (bf16_t*)U, (bf16_t*)Z, stats, T, C);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_bn_finalize(const double* stats, float* mean_rstd, float* running_mean, float* running_var, int C,
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
extern "C" int ea_bn_act_fwd_train(const void* Z, const double* stats, float* mean_rstd, float* running_mean, float* running_var,
                                   const float* gamma, const float* beta, void* H, long M, int C, int act, float n, float eps,
                                   float momentum, double* zero_next, int zero_n, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8) return -2;
  hipLaunchKernelGGL(bn_act_fwd_stats_kernel, dim3(egrid_ch(M * (C / 8), C / 8)), dim3(256), (size_t)2 * C * sizeof(float), stream, (const bf16_t*)Z, stats,
                     mean_rstd, running_mean, running_var, gamma, beta, (bf16_t*)H, M, C, act, n, eps, momentum, zero_next, zero_n);
  return EA_CHECK_LAUNCH();
}

namespace {
// BatchNorm (+ activation) backward of the FIRST sub-sampler layer fused with that convolution's weight gradient
// (espresso/modules/speech_convolutions.py:78-102, layer 0: 1 input channel, 3x3): dZ = BN'(dH) is formed in registers from Z and
// dH, rounded to bf16 like the stored tensor it replaces, and is consumed on the spot by
//   dW[co][ky][kx] += sum_pos dZ[pos][co] * X[b][to*sy+ky-1][fo*sx+kx-1],   dbias[co] += sum_pos dZ[pos][co].
// The first layer needs no data gradient, so dZ (251 MB at the recipe's batch) is neither written nor read back, and the weight
// gradient is no longer the last, un-overlapped kernel of the backward pass.
// The products run on MFMA as a [64 co] x [16: 9 taps, a column of ones for the bias, 6 x zero] x [32 positions] tile per wave
// step (the scalar version spent ~300 VALU operations per position and lane group: 380 us; this one is bound by reading Z and
// dH once): a wave stages its 32 x 64 dZ block in LDS as it lies in memory ([position][channel]) and reads the A fragments with
// ds_read_b64_tr_b16 (k = position runs down the rows); the B operand is the tap value of X for (position, tap), split into
// bf16 hi + lo parts (two MFMAs into one accumulator) so that the fp32 features lose nothing.
__device__ __forceinline__ int c1_sw(int r) { return (r & 3) ^ ((r >> 3) & 1); }
__device__ __forceinline__ void c1_tr_read8(const uint32_t (&ad)[8], uint2 (&o)[8]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
      "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\tds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
      : "memory");
}
__global__ __launch_bounds__(256) void conv1_bn_bwd_wgrad_kernel(const float* __restrict__ X, const bf16_t* __restrict__ Z,
                                                                 const bf16_t* __restrict__ dH, const float* __restrict__ mean_rstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ red, float* __restrict__ dW,
                                                                 float* __restrict__ dbias, int T, int F, int To, int Fo, int CO,
                                                                 int sy, int sx, long npos, int act, float n) {
  __shared__ __attribute__((aligned(16))) char tile[4][32 * 128];  // per wave: [32 positions][64 channels] bf16
  __shared__ float sred[4][64][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g4 = lane >> 4;
  const float invn = n > 0.f ? 1.f / n : 0.f;
  const long nblk = (npos + 31) / 32;
  const long bmax = (npos - 1) / ((long)Fo * To);  // last utterance index (address clamp of the tap loads)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(&tile[wave][0]);
  for (int cs = 0; cs < CO; cs += 64) {
    // staging role: lane owns channels cs + 8 * (lane & 7) .. + 7 of positions (lane >> 3) + 8 * it
    const int c8 = lane & 7, prow = lane >> 3;
    float mu[8], rs[8], ga[8], be[8], r0[8], r1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cs + c8 * 8 + e;
      mu[e] = mean_rstd[c]; rs[e] = mean_rstd[CO + c]; ga[e] = gamma[c]; be[e] = beta[c];
      r0[e] = red[c] * invn; r1[e] = red[CO + c] * invn;
    }
    f32x4_t acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (long blk = (long)blockIdx.x * 4 + wave; blk < nblk; blk += (long)gridDim.x * 4) {
      const long pb = blk * 32;
      // ---- every load of the block is requested first, branch-free (clamped addresses, values selected afterwards): round 6.
      // With the loads inside `if (p < npos)` / the tap tests, hipcc waited for each of the twelve in turn (vmcnt(0) after a
      // divergent branch): ~15 us per 8 KB block and wavefront, 2.4 TB/s for the whole kernel with 16 wavefronts per CU.
      uint4 uz4[4], ud4[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const long p = min(pb + prow + 8 * it, npos - 1);
        uz4[it] = *reinterpret_cast<const uint4*>(Z + p * CO + cs + c8 * 8);
        ud4[it] = *reinterpret_cast<const uint4*>(dH + p * CO + cs + c8 * 8);
      }
      int fo = (int)(pb % Fo), to = (int)((pb / Fo) % To);  // wave-uniform
      long b = pb / ((long)Fo * To);
      float xs[8];
      bool xv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * g4 + j;
        int f = fo + k, t = to;
        long bb = b;
        while (f >= Fo) { f -= Fo; ++t; }
        while (t >= To) { t -= To; ++bb; }
        const int tap = li < 9 ? li : 0;
        const int ti = t * sy + tap / 3 - 1, fi = f * sx + tap % 3 - 1;
        xv[j] = pb + k < npos && li < 9 && ti >= 0 && ti < T && fi >= 0 && fi < F;
        // (a clamped, always valid address: its value is dropped when !xv[j])
        const int tc = min(max(ti, 0), T - 1), fc = min(max(fi, 0), F - 1);
        const long bc = bb < bmax ? bb : bmax;
        xs[j] = X[(bc * T + tc) * F + fc];
      }
      // ---- dZ of 32 positions x 64 channels -> LDS (bf16, as stored before) ----
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = prow + 8 * it;
        const long p = pb + r;
        uint4 o = make_uint4(0, 0, 0, 0);
        {
          const uint4 uz = uz4[it], ud = ud4[it];
          const uint32_t wz[4] = {uz.x, uz.y, uz.z, uz.w}, wd[4] = {ud.x, ud.y, ud.z, ud.w};
          float d[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float z = (e & 1) ? __uint_as_float(wz[e >> 1] & 0xffff0000u) : __uint_as_float(wz[e >> 1] << 16);
            float g = (e & 1) ? __uint_as_float(wd[e >> 1] & 0xffff0000u) : __uint_as_float(wd[e >> 1] << 16);
            const float xh = (z - mu[e]) * rs[e];
            const float y = xh * ga[e] + be[e];
            g *= act == 2 ? dsilu_f(y) : (act == 1 ? (y > 0.f ? 1.f : 0.f) : 1.f);
            d[e] = rs[e] * ga[e] * (g - r0[e] - xh * r1[e]);
          }
          o.x = pack_bf2(d[0], d[1]); o.y = pack_bf2(d[2], d[3]); o.z = pack_bf2(d[4], d[5]); o.w = pack_bf2(d[6], d[7]);
        }
        if (p >= npos) o = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(&tile[wave][r * 128 + (((c8 >> 1) ^ c1_sw(r)) << 5) + (c8 & 1) * 16]) = o;
      }
      // ---- B operand: X tap values of (position 8 g4 + j, column li) as bf16 hi / lo ----
      uint32_t bh[4], bl[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * g4 + j;
        float x = xv[j] ? xs[j] : 0.f;
        if (li == 9 && pb + k < npos) x = 1.f;
        const uint32_t hi = (uint32_t)f2bf(x);
        const uint32_t lo = (uint32_t)f2bf(x - __uint_as_float(hi << 16));
        if (j & 1) { bh[j >> 1] |= hi << 16; bl[j >> 1] |= lo << 16; }
        else { bh[j >> 1] = hi; bl[j >> 1] = lo; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- A fragments: channels 16 mt + li, positions 8 g4 .. + 7 (two transposing reads of 4 positions each) ----
      uint32_t ad[8];
      uint2 o8[8];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = 8 * g4 + 4 * h + (li >> 2);
          ad[mt * 2 + h] = lds0 + r * 128 + ((mt ^ c1_sw(r)) << 5) + (li & 3) * 8;
        }
      c1_tr_read8(ad, o8);
      const bf16x8_t vbh = __builtin_bit_cast(bf16x8_t, make_uint4(bh[0], bh[1], bh[2], bh[3]));
      const bf16x8_t vbl = __builtin_bit_cast(bf16x8_t, make_uint4(bl[0], bl[1], bl[2], bl[3]));
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const bf16x8_t a8 = __builtin_bit_cast(bf16x8_t, make_uint4(o8[mt * 2].x, o8[mt * 2].y, o8[mt * 2 + 1].x, o8[mt * 2 + 1].y));
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a8),
                                                          __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, vbh), acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a8),
                                                          __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, vbl), acc[mt], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();  // the tile is rewritten in the next iteration
    }
    // ---- fold the four waves, then one atomic per output and workgroup ----
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sred[wave][mt * 16 + g4 * 4 + r][li] = acc[mt][r];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 10; i += 256) {
      const int c = i / 10, k = i % 10;
      const float a = sred[0][c][k] + sred[1][c][k] + sred[2][c][k] + sred[3][c][k];
      if (k < 9) atomicAdd(dW + (cs + c) * 9 + k, a);
      else if (dbias) atomicAdd(dbias + cs + c, a);
    }
    __syncthreads();
  }
}

__global__ void bn_param_grad_kernel(const float* __restrict__ red, float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] += red[c];
  if (dgamma) dgamma[c] += red[C + c];
}
}  // namespace

static void launch_bn_bwd_reduce(const void* Z, const void* dH, const float* mean_rstd, const float* gamma, const float* beta,
                                 float* red, long M, int C, int act, hipStream_t stream);

extern "C" int ea_bn_act_bwd(const void* Z, const void* dH, const float* mean_rstd, const float* gamma,
                             const float* beta, float* red, void* dZ, float* dgamma, float* dbeta, long M, int C,
                             int act, int training, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8) return -2;
  launch_bn_bwd_reduce(Z, dH, mean_rstd, gamma, beta, red, M, C, act, stream);
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(egrid_ch(M * (C / 8), C / 8)), dim3(256), 0, stream, (const bf16_t*)Z,
                     (const bf16_t*)dH, mean_rstd, gamma, beta, red, (bf16_t*)dZ, M, C, act,
                     training ? (float)M : 0.f);
  if (dgamma || dbeta) hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, red, dgamma, dbeta, C);
  return EA_CHECK_LAUNCH();
}

// First sub-sampler layer: BatchNorm (+ activation) backward and the 3x3 / 1-channel convolution's weight (and bias) gradient in
// one pass over Z and dH (conv1_bn_bwd_wgrad_kernel); `red` (fp32 [2 C], zeroed by the caller) receives BatchNorm's two sums.
// ea_bn_act_bwd in two launches instead of three or four: the parameter gradients are added by the apply kernel's first
// threads, which also clear `zero_next` (fp32 [2 C] or NULL) for the next call
extern "C" int ea_bn_act_bwd_fused(const void* Z, const void* dH, const float* mean_rstd, const float* gamma, const float* beta,
                                   float* red, void* dZ, float* dgamma, float* dbeta, long M, int C, int act, int training,
                                   float* zero_next, int zero_n, hipStream_t stream) {
  if (M <= 0) return 0;
  if (C % 8) return -2;
  launch_bn_bwd_reduce(Z, dH, mean_rstd, gamma, beta, red, M, C, act, stream);
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(egrid_ch(M * (C / 8), C / 8)), dim3(256), 0, stream, (const bf16_t*)Z,
                     (const bf16_t*)dH, mean_rstd, gamma, beta, red, (bf16_t*)dZ, M, C, act, training ? (float)M : 0.f, dgamma, dbeta,
                     zero_next, zero_n);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_conv1_bn_bwd(const float* X, const void* Z, const void* dH, const float* mean_rstd, const float* gamma,
                               const float* beta, float* red, float* dgamma, float* dbeta, float* dW, float* dbias, int B, int T,
                               int F, int CO, int sy, int sx, int act, int training, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if (CO % 64) return -2;
  const int To = (T - 1) / sy + 1, Fo = (F - 1) / sx + 1;
  const long npos = (long)B * To * Fo;
  launch_bn_bwd_reduce(Z, dH, mean_rstd, gamma, beta, red, npos, CO, act, stream);
  long nb = (npos + 127) / 128;  // one 32-position block per wave and step; ~4 workgroups per CU walk the rest with a grid stride
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(conv1_bn_bwd_wgrad_kernel, dim3((unsigned)nb), dim3(256), 0, stream, X, (const bf16_t*)Z,
                     (const bf16_t*)dH, mean_rstd, gamma, beta, red, dW, dbias, T, F, To, Fo, CO, sy, sx, npos, act,
                     training ? (float)npos : 0.f);
  if (dgamma || dbeta) hipLaunchKernelGGL(bn_param_grad_kernel, dim3((CO + 255) / 256), dim3(256), 0, stream, red, dgamma, dbeta, CO);
  return EA_CHECK_LAUNCH();
}

static void launch_bn_bwd_reduce(const void* Z, const void* dH, const float* mean_rstd, const float* gamma, const float* beta,
                                 float* red, long M, int C, int act, hipStream_t stream) {
  const int nch = C / 8;
  int TCH = 1;
  while (TCH < 256 && TCH < nch) TCH <<= 1;  // 8-channel chunks per block row (power of two <= 256)
  const int lanes = 256 / TCH;
  int rpb = 8 * lanes;                        // >= 8 rows per thread
  if (rpb < 32) rpb = 32;
  // every block ends with 2*8*TCH atomics onto the same 2*C sums: cap the grid at ~2048 blocks (the sub-sampler's 2 M-row,
  // 64-channel maps launched 7 660 blocks = 1 M atomics on 128 addresses and ran at 1.5 TB/s)
  const long gx = (nch + TCH - 1) / TCH;
  const long want = (M + (2048 / gx) - 1) / (2048 / gx > 0 ? 2048 / gx : 1);
  if (want > rpb) rpb = (int)((want + lanes - 1) / lanes * lanes);
  dim3 g1((unsigned)gx, (unsigned)((M + rpb - 1) / rpb));
  hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, g1, dim3(256), (size_t)256 * 16 * sizeof(float), stream, (const bf16_t*)Z,
                     (const bf16_t*)dH, mean_rstd, gamma, beta, red, M, C, act, rpb, TCH);
}
// the optimizer-only tail of ea_bn_act_bwd on its own (call ea_bn_act_bwd with dgamma = dbeta = NULL first)
extern "C" int ea_bn_param_grad(const float* red, float* dgamma, float* dbeta, int C, hipStream_t stream) {
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, red, dgamma, dbeta, C);
  return EA_CHECK_LAUNCH();
}

constexpr int DW_TILES_PER_BLOCK = 2;
static inline int dw_time_blocks(int T) {
  const int nt = (T + TTILE - 1) / TTILE;
  return (nt + DW_TILES_PER_BLOCK - 1) / DW_TILES_PER_BLOCK;
}
extern "C" long ea_dwconv_wgrad_workspace_bytes(int B, int T, int C, int KW) {
  return (long)B * dw_time_blocks(T) * C * KW * (long)sizeof(float);
}

extern "C" int ea_glu_dwconv_bwd(const void* dZ, const void* Y, const void* U, const float* w, void* dY, float* dw,
                                 void* wgrad_ws, int B, int T, int C, int KW, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if (C % 2) return -2;
  dim3 grid((C + CT - 1) / CT, (T + TTILE - 1) / TTILE, B);
  EA_KW_DISPATCH(KW, launch_glu_dwconv_bwd_data, grid, stream, (const bf16_t*)dZ, (const bf16_t*)Y, w, (bf16_t*)dY, T, C);
  if (!dw) return EA_CHECK_LAUNCH();  // data gradient only: the caller runs ea_dwconv_bwd_weight (optimizer-only) itself
  return ea_dwconv_bwd_weight(dZ, U, dw, wgrad_ws, B, T, C, KW, stream);
}
// ea_bn_act_bwd_fused followed by ea_glu_dwconv_bwd's data gradient with the BatchNorm "apply" pass folded into the convolution
// kernel's tile staging (glu_dwconv_bwd_data_kernel<KW, true>): reduce launch + ONE more launch.  dZ is still written (the
// depthwise weight gradient reads it).  Channel counts that are not a multiple of 64 take the two separate passes.
extern "C" int ea_bn_glu_dwconv_bwd_fused(const void* Z, const void* dH, const float* mean_rstd, const float* gamma, const float* beta,
                                          float* red, void* dZ, float* dgamma, float* dbeta, int act, int training, float* zero_next,
                                          int zero_n, const void* Y, const float* w, void* dY, int B, int T, int C, int KW,
                                          hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  const long M = (long)B * T;
  // (A/B switch, default OFF: measured 12.04 / 12.02 ms per step fused against 12.00 / 12.01 with the two passes — the halo rows'
  // BatchNorm arithmetic and the second operand's loads cost what the saved launch and pass gave; profiles/r06_side_kernel_grids_ab.txt)
  static const bool fuse = [] { const char* e = getenv("EA_BN_GLU_BWD_FUSED"); return e && e[0] == '1'; }();
  if (!fuse || C % CT != 0) {
    const int rc = ea_bn_act_bwd_fused(Z, dH, mean_rstd, gamma, beta, red, dZ, dgamma, dbeta, M, C, act, training, zero_next, zero_n, stream);
    return rc ? rc : ea_glu_dwconv_bwd(dZ, Y, nullptr, w, dY, nullptr, nullptr, B, T, C, KW, stream);
  }
  launch_bn_bwd_reduce(Z, dH, mean_rstd, gamma, beta, red, M, C, act, stream);
  BnBwdFuse fb{(const bf16_t*)Z, (const bf16_t*)dH, mean_rstd, gamma, beta, red, (bf16_t*)dZ, training ? (float)M : 0.f, act, dgamma, dbeta,
               zero_next, zero_n};
  dim3 grid(C / CT, (T + TTILE - 1) / TTILE, B);
  EA_KW_DISPATCH(KW, launch_glu_dwconv_bwd_data_bn, grid, stream, (const bf16_t*)Y, w, (bf16_t*)dY, T, C, fb);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_dwconv_bwd_weight(const void* dZ, const void* U, float* dw, void* wgrad_ws, int B, int T, int C, int KW,
                                    hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  dim3 gridw((C + CT - 1) / CT, dw_time_blocks(T), B);
  float* part = (float*)wgrad_ws;
  EA_KW_DISPATCH(KW, launch_dwconv_bwd_weight, gridw, stream, (const bf16_t*)dZ, (const bf16_t*)U, part, T, C, DW_TILES_PER_BLOCK);
  const int nslab = (int)(gridw.y * gridw.z);
  hipLaunchKernelGGL(dwconv_weight_reduce_kernel, dim3((C * KW + 255) / 256, nslab >= 16 ? 8 : 1), dim3(256), 0, stream, part, dw,
                     nslab, C * KW);
  return EA_CHECK_LAUNCH();
}
