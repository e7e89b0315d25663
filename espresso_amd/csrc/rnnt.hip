// RNN-T (transducer) loss forward + backward for gfx950.
//
// Replaces torchaudio.functional.rnnt_loss as called by espresso/criterions/transducer_loss.py:130-140
// (blank = index of "<s>", clamp = -1, fused_log_softmax = True, reduction sum/none).  torchaudio is a third-party
// dependency that is NOT in the reference tree (README.md:16 ">= 0.10.0"); the algorithm restated here is Graves 2012
// ("Sequence Transduction with Recurrent Neural Networks", eq. 16-20) with the fused log-softmax gradient of the
// warp-transducer formulation, the same recursion the torchaudio CPU/GPU kernels implement:
//   alpha(t,u) = logaddexp(alpha(t-1,u) + lp_blank(t-1,u), alpha(t,u-1) + lp_y(t,u-1)),   alpha(0,0) = 0
//   loss_b     = -(alpha(T-1,U) + lp_blank(T-1,U))
//   dL/dz[t,u,v] = exp(lp[t,u,v] + alpha(t,u) + beta(t,u) + L) - [v=blank] exp(alpha+lp_blank+beta(t+1,u)+L)
//                                                               - [v=y_{u+1}] exp(alpha+lp_y+beta(t,u+1)+L)
//
// HBM-bound on the (B,T,U+1,V) logits: (1) one wavefront per lattice node computes logsumexp over V and keeps only
// the two log-probs the recursion needs; (2) one workgroup per utterance sweeps anti-diagonals (one lane per u, LDS
// hand-off between neighbours) for alpha and beta concurrently; (3) one wavefront per node streams V once more to write
// the gradient.  The joint-network-fused variant (logits never materialised) is csrc/joint_rnnt.hip (round 6); this file stays
// the path for logits that exist (criterion called on a tensor, shapes the fused kernels do not take) and the test reference.
#include "common.h"
#include "espresso_amd.h"

namespace {

__device__ __forceinline__ float ldz(const float* p, long i) { return p[i]; }
__device__ __forceinline__ float ldz(const bf16_t* p, long i) { return bf2f(p[i]); }

// denom-free storage: lpb[b][t][u] = log p(blank | t,u), lpy[b][t][u] = log p(y_{u+1} | t,u) (u < U_b), lse[b][t][u]
template <typename TIn>
__global__ __launch_bounds__(256) void rnnt_lse_kernel(const TIn* __restrict__ logits, const int* __restrict__ targets,
                                                       const int* __restrict__ T_len, const int* __restrict__ U_len,
                                                       float* __restrict__ lse, float* __restrict__ lpb, float* __restrict__ lpy,
                                                       int T, int U1, int V, long ld, int Umax, int blank, long nnodes) {
  const int lane = threadIdx.x & 63;
  const long node = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= nnodes) return;
  const int u = (int)(node % U1);
  const int t = (int)((node / U1) % T);
  const int b = (int)(node / ((long)U1 * T));
  if (t >= T_len[b] || u > U_len[b]) return;
  const TIn* z = logits + node * ld;
  float mx = -INFINITY, s = 0.f;
  if constexpr (sizeof(TIn) == 2) {
    if ((ld & 7) == 0 && (((uintptr_t)logits) & 15) == 0) {
      // rows padded to a multiple of 8 elements: 16-byte loads, 8 logits per lane and chunk
      const int nch = (V + 7) >> 3;
      constexpr int MAXC = 12;  // chunks per lane held in registers: rows up to 6144 logits make ONE trip to memory
      if (nch <= 64 * MAXC) {
        // every load of the row is requested before the first is used (the two-pass loop below waits for each 16-byte load in
        // turn: 20 dependent round trips per 10 KB row — 465 us for the recipe's 708 MB of logits, 1.5 TB/s); the arithmetic and
        // its order are those of the loop below: same bits
        uint4 q[MAXC];
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
          const int c = lane + 64 * k;
          q[k] = c < nch ? *reinterpret_cast<const uint4*>(z + c * 8) : make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u);
        }
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
          const int c = lane + 64 * k;
          const uint32_t w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = (e & 1) ? __uint_as_float(w[e >> 1] & 0xffff0000u) : __uint_as_float(w[e >> 1] << 16);
            if (c * 8 + e < V) mx = fmaxf(mx, x);
          }
        }
        mx = wave_max(mx);
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
          const int c = lane + 64 * k;
          if (c < nch) {
            const uint32_t w[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float x = (e & 1) ? __uint_as_float(w[e >> 1] & 0xffff0000u) : __uint_as_float(w[e >> 1] << 16);
              if (c * 8 + e < V) s += expf(x - mx);
            }
          }
        }
      } else {
        for (int c = lane; c < nch; c += 64) {
          const uint4 q = *reinterpret_cast<const uint4*>(z + c * 8);
          const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = (e & 1) ? __uint_as_float(w[e >> 1] & 0xffff0000u) : __uint_as_float(w[e >> 1] << 16);
            if (c * 8 + e < V) mx = fmaxf(mx, x);
          }
        }
        mx = wave_max(mx);
        for (int c = lane; c < nch; c += 64) {
          const uint4 q = *reinterpret_cast<const uint4*>(z + c * 8);
          const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = (e & 1) ? __uint_as_float(w[e >> 1] & 0xffff0000u) : __uint_as_float(w[e >> 1] << 16);
            if (c * 8 + e < V) s += expf(x - mx);
          }
        }
      }
      s = wave_sum(s);
      const float l = mx + logf(s);
      if (lane == 0) {
        lse[node] = l;
        lpb[node] = ldz(z, blank) - l;
        lpy[node] = (u < U_len[b]) ? ldz(z, targets[(long)b * Umax + u]) - l : -INFINITY;
      }
      return;
    }
  }
  for (int v = lane; v < V; v += 64) mx = fmaxf(mx, ldz(z, v));
  mx = wave_max(mx);
  for (int v = lane; v < V; v += 64) s += expf(ldz(z, v) - mx);
  s = wave_sum(s);
  const float l = mx + logf(s);
  if (lane == 0) {
    lse[node] = l;
    lpb[node] = ldz(z, blank) - l;
    lpy[node] = (u < U_len[b]) ? ldz(z, targets[(long)b * Umax + u]) - l : -INFINITY;
  }
}

// blockDim = 2 * UP (UP = U1 rounded up to 64): first half alpha, second half beta; one lane per u.
__global__ void rnnt_scan_kernel(const float* __restrict__ lpb, const float* __restrict__ lpy, const int* __restrict__ T_len,
                                 const int* __restrict__ U_len, float* __restrict__ alpha, float* __restrict__ beta,
                                 float* __restrict__ loss, int T, int U1, int UP) {
  extern __shared__ float sh[];  // [2 dir][2 buf][UP + 2]
  const int b = blockIdx.x;
  const int Tb = T_len[b], Ub = U_len[b];
  const int dir = threadIdx.x >= UP ? 1 : 0;
  const int u = threadIdx.x - dir * UP;
  const long base = (long)b * T * U1;
  float* prevbuf = sh + (dir * 2 + 0) * (UP + 2) + 1;
  float* curbuf = sh + (dir * 2 + 1) * (UP + 2) + 1;
  if (Tb <= 0) {
    if (threadIdx.x == 0) loss[b] = INFINITY;
    return;
  }
  // diagonal d = t + u (alpha: ascending from 0 ; beta: descending from Tb-1+Ub)
  const int ndiag = Tb + Ub;
  float own_prev = -INFINITY;  // value of this lane's cell on the previous diagonal it owned (t-1 for alpha, t+1 for beta)
  prevbuf[u] = -INFINITY;
  if (u == 0) { prevbuf[-1] = -INFINITY; curbuf[-1] = -INFINITY; prevbuf[UP] = -INFINITY; curbuf[UP] = -INFINITY; }
  __syncthreads();
  // the two log-probabilities a cell needs do not depend on the recursion: they are requested FOUR DIAGONALS AHEAD (round 6 —
  // loaded where they were used, every diagonal paid a memory round trip inside the dependent chain: 0.5 - 0.65 ms per batch at
  // the recipe's lattices, with only B workgroups on the device); the barrier below waits for the LDS hand-off only, so that the
  // requests stay in flight across it
  // (every lane loads on every diagonal from a clamped cell, the value is selected afterwards: a load inside a divergent branch makes
  // hipcc's wait-count pass fall back to vmcnt(0), i.e. to waiting for the request it has just issued)
  const int uc = u <= Ub ? u : Ub;
  auto fetch = [&](int k, float& pb, float& py) {
    const int kk = k < ndiag ? k : ndiag - 1;
    const int d = dir ? (ndiag - 1 - kk) : kk;
    int t = d - uc;
    t = t < 0 ? 0 : (t > Tb - 1 ? Tb - 1 : t);
    const long idx = base + t * U1 + uc;
    if (!dir) {
      pb = lpb[t > 0 ? idx - U1 : idx];   // from (t-1,u) by blank
      py = lpy[uc > 0 ? idx - 1 : idx];   // from (t,u-1) by label
    } else {
      pb = lpb[idx];                      // to (t+1,u) by blank — or the final blank at (Tb-1, Ub)
      py = lpy[idx];                      // to (t,u+1) by label
    }
  };
  // four register pairs, one per diagonal modulo 4, refilled right after use with the pair of diagonal k + 4 — no register rotation:
  // copying a register whose load is pending is a wait for that load
  auto step = [&](int k, float& qb, float& qy) {
    if (k >= ndiag) return;  // (uniform)
    const float pb = qb, py = qy;
    fetch(k + 4, qb, qy);
    const int d = dir ? (ndiag - 1 - k) : k;
    const int t = d - u;
    const bool in = u <= Ub && t >= 0 && t < Tb;
    float val;
    if (!dir) {
      const float a = t > 0 ? own_prev + pb : -INFINITY;
      const float c = u > 0 ? prevbuf[u - 1] + py : -INFINITY;
      val = (t == 0 && u == 0) ? 0.f : log_add(a, c);
    } else {
      const float a = t < Tb - 1 ? own_prev + pb : -INFINITY;
      const float c = u < Ub ? prevbuf[u + 1] + py : -INFINITY;
      val = (t == Tb - 1 && u == Ub) ? pb : log_add(a, c);
    }
    val = in ? val : -INFINITY;
    if (in) {
      (dir ? beta : alpha)[base + (long)t * U1 + u] = val;
      own_prev = val;
    }
    curbuf[u] = val;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this lane's hand-off is in LDS (global requests are NOT waited for)
    __builtin_amdgcn_s_barrier();
    float* tmp = prevbuf; prevbuf = curbuf; curbuf = tmp;
  };
  float b0, y0, b1, y1, b2, y2, b3, y3;
  fetch(0, b0, y0);
  fetch(1, b1, y1);
  fetch(2, b2, y2);
  fetch(3, b3, y3);
  for (int k = 0; k < ndiag; k += 4) {
    step(k, b0, y0);
    step(k + 1, b1, y1);
    step(k + 2, b2, y2);
    step(k + 3, b3, y3);
  }
  if (dir == 1 && u == 0) loss[b] = -beta[base];  // beta(0,0) = log P(y|x)
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void rnnt_grad_kernel(const TIn* __restrict__ logits, const int* __restrict__ targets,
                                                        const int* __restrict__ T_len, const int* __restrict__ U_len,
                                                        const float* __restrict__ lse, const float* __restrict__ lpb,
                                                        const float* __restrict__ lpy, const float* __restrict__ alpha,
                                                        const float* __restrict__ beta, const float* __restrict__ loss,
                                                        TOut* __restrict__ grad, int T, int U1, int V, long ld, int Umax, int blank,
                                                        float scale, const float* __restrict__ scale_dev, long nnodes) {
  const int lane = threadIdx.x & 63;
  const long node = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= nnodes) return;
  const int u = (int)(node % U1);
  const int t = (int)((node / U1) % T);
  const int b = (int)(node / ((long)U1 * T));
  TOut* g = grad + node * ld;  // (the gradient has the logits' row pitch; its pad columns are written as zeros)
  const int Tb = T_len[b], Ub = U_len[b];
  const float L = loss[b];
  if (t >= Tb || u > Ub || !(L < INFINITY)) {
    for (int v = lane; v < (int)ld; v += 64) { if constexpr (sizeof(TOut) == 2) g[v] = 0; else g[v] = 0.f; }
    return;
  }
  if (scale_dev) scale *= scale_dev[0];
  const TIn* z = logits + node * ld;
  const float a = alpha[node];
  const float occ = a + beta[node] + L;  // log occupancy of (t,u)
  const float l = lse[node];
  float cb, cy = 0.f;
  if (t == Tb - 1 && u == Ub) cb = expf(a + lpb[node] + L);
  else cb = t < Tb - 1 ? expf(a + lpb[node] + beta[node + U1] + L) : 0.f;
  int y = -1;
  if (u < Ub) {
    y = targets[(long)b * Umax + u];
    cy = expf(a + lpy[node] + beta[node + 1] + L);
  }
  if constexpr (sizeof(TIn) == 2 && sizeof(TOut) == 2) {
    if ((ld & 7) == 0 && ((((uintptr_t)logits) | ((uintptr_t)grad)) & 15) == 0) {
      const int nch = (int)(ld >> 3);
      constexpr int MAXC = 12;
      uint4 pre[MAXC];  // the row's loads requested together (rows up to 6144 columns; longer rows load in the loop)
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        const int c = lane + 64 * k;
        pre[k] = c < nch ? *reinterpret_cast<const uint4*>(z + c * 8) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        const int c = lane + 64 * k;
        if (c >= nch) break;
        const uint4 q = pre[k];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int v = c * 8 + e;
          const float x = (e & 1) ? __uint_as_float(w[e >> 1] & 0xffff0000u) : __uint_as_float(w[e >> 1] << 16);
          float gv = expf(x - l + occ);
          if (v == blank) gv -= cb;
          if (v == y) gv -= cy;
          o[e] = v < V ? gv * scale : 0.f;
        }
        uint4 r;
        r.x = pack_bf2(o[0], o[1]); r.y = pack_bf2(o[2], o[3]); r.z = pack_bf2(o[4], o[5]); r.w = pack_bf2(o[6], o[7]);
        *reinterpret_cast<uint4*>(g + c * 8) = r;
      }
      for (int c = lane + 64 * MAXC; c < nch; c += 64) {
        const uint4 q = *reinterpret_cast<const uint4*>(z + c * 8);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int v = c * 8 + e;
          const float x = (e & 1) ? __uint_as_float(w[e >> 1] & 0xffff0000u) : __uint_as_float(w[e >> 1] << 16);
          float gv = expf(x - l + occ);
          if (v == blank) gv -= cb;
          if (v == y) gv -= cy;
          o[e] = v < V ? gv * scale : 0.f;
        }
        uint4 r;
        r.x = pack_bf2(o[0], o[1]); r.y = pack_bf2(o[2], o[3]); r.z = pack_bf2(o[4], o[5]); r.w = pack_bf2(o[6], o[7]);
        *reinterpret_cast<uint4*>(g + c * 8) = r;
      }
      return;
    }
  }
  for (int v = lane; v < (int)ld; v += 64) {
    float gv = 0.f;
    if (v < V) {
      gv = expf(ldz(z, v) - l + occ);
      if (v == blank) gv -= cb;
      if (v == y) gv -= cy;
      gv *= scale;
    }
    if constexpr (sizeof(TOut) == 2) g[v] = f2bf(gv); else g[v] = gv;
  }
}

// ---- joint network element-wise stages (speech_transformer_transducer_base.py:276-299) --------------------------
// Z[b][t][u][:] = relu(E[b][t][:] + D[b][u][:])   (bf16, J % 8 == 0); one 16-byte chunk per thread
__global__ __launch_bounds__(256) void joint_add_relu_kernel(const bf16_t* __restrict__ E, const bf16_t* __restrict__ D,
                                                             bf16_t* __restrict__ Z, int T, int U1, int J, long nchunks) {
  const int nch = J >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const long node = i / nch;
    const int u = (int)(node % U1);
    const long bt = node / U1;
    const long b = bt / T;
    const uint4 ue = *reinterpret_cast<const uint4*>(E + bt * J + ch * 8);
    const uint4 ud = *reinterpret_cast<const uint4*>(D + (b * U1 + u) * J + ch * 8);
    const uint32_t we[4] = {ue.x, ue.y, ue.z, ue.w}, wd[4] = {ud.x, ud.y, ud.z, ud.w};
    uint32_t wo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = fmaxf(__uint_as_float(we[e] << 16) + __uint_as_float(wd[e] << 16), 0.f);
      const float hi = fmaxf(__uint_as_float(we[e] & 0xffff0000u) + __uint_as_float(wd[e] & 0xffff0000u), 0.f);
      wo[e] = pack_bf2(lo, hi);
    }
    *reinterpret_cast<uint4*>(Z + node * J + ch * 8) = make_uint4(wo[0], wo[1], wo[2], wo[3]);
  }
}
// the same on fp32 E / D (the reference's autocast run adds and rectifies the two LayerNorm outputs in fp32; only fc_out's operand
// is bf16): one 16-byte chunk of Z per thread, two 16-byte loads per operand
__global__ __launch_bounds__(256) void joint_add_relu_f32_kernel(const float* __restrict__ E, const float* __restrict__ D,
                                                                 bf16_t* __restrict__ Z, int T, int U1, int J, long nchunks) {
  const int nch = J >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const long node = i / nch;
    const int u = (int)(node % U1);
    const long bt = node / U1;
    const long b = bt / T;
    const float* pe = E + bt * J + ch * 8;
    const float* pd = D + (b * U1 + u) * J + ch * 8;
    const float4 e0 = *reinterpret_cast<const float4*>(pe), e1 = *reinterpret_cast<const float4*>(pe + 4);
    const float4 d0 = *reinterpret_cast<const float4*>(pd), d1 = *reinterpret_cast<const float4*>(pd + 4);
    *reinterpret_cast<uint4*>(Z + node * J + ch * 8) =
        make_uint4(pack_bf2(fmaxf(e0.x + d0.x, 0.f), fmaxf(e0.y + d0.y, 0.f)), pack_bf2(fmaxf(e0.z + d0.z, 0.f), fmaxf(e0.w + d0.w, 0.f)),
                   pack_bf2(fmaxf(e1.x + d1.x, 0.f), fmaxf(e1.y + d1.y, 0.f)), pack_bf2(fmaxf(e1.z + d1.z, 0.f), fmaxf(e1.w + d1.w, 0.f)));
  }
}
// dE[b][t][:] = sum_u dZ[b][t][u][:]  (mode 0, rows = B*T, inner = U1 consecutive rows)
// dD[b][u][:] = sum_t dZ[b][t][u][:]  (mode 1, rows = B*U1, inner = T rows U1*J apart)
template <typename TO>
__global__ __launch_bounds__(256) void joint_reduce_kernel(const bf16_t* __restrict__ dZ, TO* __restrict__ out, int T, int U1, int J,
                                                           int mode, long nrows) {
  const int nch = J >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nrows * nch; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % nch);
    const long row = i / nch;
    const bf16_t* src;
    long step;
    int cnt;
    if (mode == 0) { src = dZ + row * U1 * (long)J + ch * 8; step = J; cnt = U1; }
    else { const long b = row / U1; const int u = (int)(row % U1); src = dZ + ((b * T) * U1 + u) * (long)J + ch * 8; step = (long)U1 * J; cnt = T; }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < cnt; ++k) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + k * step);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += __uint_as_float(w[e] << 16);
        acc[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
      }
    }
    if constexpr (sizeof(TO) == 2) {
      *reinterpret_cast<uint4*>(out + row * J + ch * 8) =
          make_uint4(pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3]), pack_bf2(acc[4], acc[5]), pack_bf2(acc[6], acc[7]));
    } else {
      *reinterpret_cast<float4*>(out + row * J + ch * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *reinterpret_cast<float4*>(out + row * J + ch * 8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
  }
}

}  // namespace

extern "C" int ea_joint_add_relu(const void* E, const void* D, void* Z, int B, int T, int U1, int J, hipStream_t stream) {
  const long nchunks = (long)B * T * U1 * (J / 8);
  if (nchunks <= 0) return 0;
  if (J % 8) return -2;
  long blocks = (nchunks + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(joint_add_relu_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)E, (const bf16_t*)D, (bf16_t*)Z, T,
                     U1, J, nchunks);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_joint_reduce(const void* dZ, void* dE, void* dD, int B, int T, int U1, int J, hipStream_t stream) {
  if ((long)B * T * U1 <= 0) return 0;
  if (J % 8) return -2;
  const long rE = (long)B * T, rD = (long)B * U1;
  long bE = (rE * (J / 8) + 255) / 256, bD = (rD * (J / 8) + 255) / 256;
  if (bE > 16384) bE = 16384;
  if (bD > 16384) bD = 16384;
  if (dE) hipLaunchKernelGGL(joint_reduce_kernel<bf16_t>, dim3((unsigned)bE), dim3(256), 0, stream, (const bf16_t*)dZ, (bf16_t*)dE, T, U1, J, 0, rE);
  if (dD) hipLaunchKernelGGL(joint_reduce_kernel<bf16_t>, dim3((unsigned)bD), dim3(256), 0, stream, (const bf16_t*)dZ, (bf16_t*)dD, T, U1, J, 1, rD);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_joint_add_relu_f32(const float* E, const float* D, void* Z, int B, int T, int U1, int J, hipStream_t stream) {
  const long nchunks = (long)B * T * U1 * (J / 8);
  if (nchunks <= 0) return 0;
  if (J % 8) return -2;
  long blocks = (nchunks + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(joint_add_relu_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, E, D, (bf16_t*)Z, T, U1, J, nchunks);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_joint_reduce_f32(const void* dZ, float* dE, float* dD, int B, int T, int U1, int J, hipStream_t stream) {
  if ((long)B * T * U1 <= 0) return 0;
  if (J % 8) return -2;
  const long rE = (long)B * T, rD = (long)B * U1;
  long bE = (rE * (J / 8) + 255) / 256, bD = (rD * (J / 8) + 255) / 256;
  if (bE > 16384) bE = 16384;
  if (bD > 16384) bD = 16384;
  if (dE) hipLaunchKernelGGL(joint_reduce_kernel<float>, dim3((unsigned)bE), dim3(256), 0, stream, (const bf16_t*)dZ, dE, T, U1, J, 0, rE);
  if (dD) hipLaunchKernelGGL(joint_reduce_kernel<float>, dim3((unsigned)bD), dim3(256), 0, stream, (const bf16_t*)dZ, dD, T, U1, J, 1, rD);
  return EA_CHECK_LAUNCH();
}

extern "C" long ea_rnnt_workspace_bytes(int B, int T, int U1) { return 5L * B * T * U1 * (long)sizeof(float); }

// alpha / beta sweep + per-utterance loss from the two log-probabilities per lattice node (shared with csrc/joint_rnnt.hip, whose
// fused vocabulary projection produces lpb / lpy without materialising the logits)
extern "C" int ea_rnnt_scan(const float* lpb, const float* lpy, const int* logit_lengths, const int* target_lengths, float* alpha,
                            float* beta, float* loss, int B, int T, int U1, hipStream_t stream) {
  if (B <= 0) return 0;
  if (T <= 0 || U1 <= 0 || U1 > 512) return -2;
  const int UP = (U1 + 63) / 64 * 64;
  hipLaunchKernelGGL(rnnt_scan_kernel, dim3(B), dim3(2 * UP), (size_t)4 * (UP + 2) * sizeof(float), stream, lpb, lpy,
                     logit_lengths, target_lengths, alpha, beta, loss, T, U1, UP);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_rnnt_loss(const void* logits, int logits_bf16, const int* targets, const int* logit_lengths,
                            const int* target_lengths, float* loss /*[B]*/, void* workspace, int B, int T, int U1, int V, long ld,
                            int Umax, int blank, hipStream_t stream) {
  if (B <= 0) return 0;
  if (T <= 0 || U1 <= 0 || U1 > 512 || ld < V) return -2;
  const long n = (long)B * T * U1;
  float* lse = (float*)workspace;
  float* lpb = lse + n;
  float* lpy = lpb + n;
  float* alpha = lpy + n;
  float* beta = alpha + n;
  if (logits_bf16)
    hipLaunchKernelGGL(rnnt_lse_kernel<bf16_t>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)logits, targets,
                       logit_lengths, target_lengths, lse, lpb, lpy, T, U1, V, ld, Umax, blank, n);
  else
    hipLaunchKernelGGL(rnnt_lse_kernel<float>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, (const float*)logits, targets,
                       logit_lengths, target_lengths, lse, lpb, lpy, T, U1, V, ld, Umax, blank, n);
  return ea_rnnt_scan(lpb, lpy, logit_lengths, target_lengths, alpha, beta, loss, B, T, U1, stream);
}

extern "C" int ea_rnnt_grad(const void* logits, int logits_bf16, const int* targets, const int* logit_lengths,
                            const int* target_lengths, const float* loss, const void* workspace, void* grad, int grad_bf16, int B,
                            int T, int U1, int V, long ld, int Umax, int blank, float grad_scale, const float* grad_scale_dev,
                            hipStream_t stream) {
  if (B <= 0) return 0;
  if (ld < V) return -2;
  const long n = (long)B * T * U1;
  const float* lse = (const float*)workspace;
  const float* lpb = lse + n;
  const float* lpy = lpb + n;
  const float* alpha = lpy + n;
  const float* beta = alpha + n;
#define EA_RNNT_GRAD(TI, TO)                                                                                              \
  hipLaunchKernelGGL((rnnt_grad_kernel<TI, TO>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, (const TI*)logits, targets, \
                     logit_lengths, target_lengths, lse, lpb, lpy, alpha, beta, loss, (TO*)grad, T, U1, V, ld, Umax, blank, \
                     grad_scale, grad_scale_dev, n)
  if (logits_bf16) { if (grad_bf16) EA_RNNT_GRAD(bf16_t, bf16_t); else EA_RNNT_GRAD(bf16_t, float); }
  else { if (grad_bf16) EA_RNNT_GRAD(float, bf16_t); else EA_RNNT_GRAD(float, float); }
#undef EA_RNNT_GRAD
  return EA_CHECK_LAUNCH();
}
