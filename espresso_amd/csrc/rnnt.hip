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
// the gradient.  A joint-network-fused variant (logits never materialised) is the planned next step.
#include "common.h"
#include "espresso_amd.h"

namespace {

// denom-free storage: lpb[b][t][u] = log p(blank | t,u), lpy[b][t][u] = log p(y_{u+1} | t,u) (u < U_b), lse[b][t][u]
__global__ __launch_bounds__(256) void rnnt_lse_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                                       const int* __restrict__ T_len, const int* __restrict__ U_len,
                                                       float* __restrict__ lse, float* __restrict__ lpb, float* __restrict__ lpy,
                                                       int T, int U1, int V, int Umax, int blank, long nnodes) {
  const int lane = threadIdx.x & 63;
  const long node = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= nnodes) return;
  const int u = (int)(node % U1);
  const int t = (int)((node / U1) % T);
  const int b = (int)(node / ((long)U1 * T));
  if (t >= T_len[b] || u > U_len[b]) return;
  const float* z = logits + node * V;
  float mx = -INFINITY;
  for (int v = lane; v < V; v += 64) mx = fmaxf(mx, z[v]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int v = lane; v < V; v += 64) s += expf(z[v] - mx);
  s = wave_sum(s);
  const float l = mx + logf(s);
  if (lane == 0) {
    lse[node] = l;
    lpb[node] = z[blank] - l;
    lpy[node] = (u < U_len[b]) ? z[targets[(long)b * Umax + u]] - l : -INFINITY;
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
  for (int k = 0; k < ndiag; ++k) {
    const int d = dir ? (ndiag - 1 - k) : k;
    const int t = d - u;
    float val = -INFINITY;
    const bool in = u <= Ub && t >= 0 && t < Tb;
    if (in) {
      const long idx = base + (long)t * U1 + u;
      if (!dir) {
        if (t == 0 && u == 0) val = 0.f;
        else {
          const float a = t > 0 ? own_prev + lpb[idx - U1] : -INFINITY;            // from (t-1,u) by blank
          const float c = u > 0 ? prevbuf[u - 1] + lpy[idx - 1] : -INFINITY;        // from (t,u-1) by label
          val = log_add(a, c);
        }
        alpha[idx] = val;
      } else {
        if (t == Tb - 1 && u == Ub) val = lpb[idx];
        else {
          const float a = t < Tb - 1 ? own_prev + lpb[idx] : -INFINITY;             // to (t+1,u) by blank
          const float c = u < Ub ? prevbuf[u + 1] + lpy[idx] : -INFINITY;           // to (t,u+1) by label
          val = log_add(a, c);
        }
        beta[idx] = val;
      }
      own_prev = val;
    }
    curbuf[u] = in ? val : -INFINITY;
    __syncthreads();
    float* tmp = prevbuf; prevbuf = curbuf; curbuf = tmp;
  }
  if (dir == 1 && u == 0) loss[b] = -beta[base];  // beta(0,0) = log P(y|x)
}

template <typename TOut>
__global__ __launch_bounds__(256) void rnnt_grad_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                                        const int* __restrict__ T_len, const int* __restrict__ U_len,
                                                        const float* __restrict__ lse, const float* __restrict__ lpb,
                                                        const float* __restrict__ lpy, const float* __restrict__ alpha,
                                                        const float* __restrict__ beta, const float* __restrict__ loss,
                                                        TOut* __restrict__ grad, int T, int U1, int V, int Umax, int blank,
                                                        float scale, const float* __restrict__ scale_dev, long nnodes) {
  const int lane = threadIdx.x & 63;
  const long node = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= nnodes) return;
  const int u = (int)(node % U1);
  const int t = (int)((node / U1) % T);
  const int b = (int)(node / ((long)U1 * T));
  TOut* g = grad + node * V;
  const int Tb = T_len[b], Ub = U_len[b];
  const float L = loss[b];
  if (t >= Tb || u > Ub || !(L < INFINITY)) {
    for (int v = lane; v < V; v += 64) { if constexpr (sizeof(TOut) == 2) g[v] = 0; else g[v] = 0.f; }
    return;
  }
  if (scale_dev) scale *= scale_dev[0];
  const float* z = logits + node * V;
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
  for (int v = lane; v < V; v += 64) {
    float gv = expf(z[v] - l + occ);
    if (v == blank) gv -= cb;
    if (v == y) gv -= cy;
    gv *= scale;
    if constexpr (sizeof(TOut) == 2) g[v] = f2bf(gv); else g[v] = gv;
  }
}

}  // namespace

extern "C" long ea_rnnt_workspace_bytes(int B, int T, int U1) { return 5L * B * T * U1 * (long)sizeof(float); }

extern "C" int ea_rnnt_loss(const float* logits, const int* targets, const int* logit_lengths, const int* target_lengths,
                            float* loss /*[B]*/, void* workspace, int B, int T, int U1, int V, int Umax, int blank,
                            hipStream_t stream) {
  if (B <= 0) return 0;
  if (T <= 0 || U1 <= 0 || U1 > 512) return -2;
  const long n = (long)B * T * U1;
  float* lse = (float*)workspace;
  float* lpb = lse + n;
  float* lpy = lpb + n;
  float* alpha = lpy + n;
  float* beta = alpha + n;
  hipLaunchKernelGGL(rnnt_lse_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, logits, targets, logit_lengths,
                     target_lengths, lse, lpb, lpy, T, U1, V, Umax, blank, n);
  const int UP = (U1 + 63) / 64 * 64;
  hipLaunchKernelGGL(rnnt_scan_kernel, dim3(B), dim3(2 * UP), (size_t)4 * (UP + 2) * sizeof(float), stream, lpb, lpy,
                     logit_lengths, target_lengths, alpha, beta, loss, T, U1, UP);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_rnnt_grad(const float* logits, const int* targets, const int* logit_lengths, const int* target_lengths,
                            const float* loss, const void* workspace, void* grad, int grad_bf16, int B, int T, int U1, int V,
                            int Umax, int blank, float grad_scale, const float* grad_scale_dev, hipStream_t stream) {
  if (B <= 0) return 0;
  const long n = (long)B * T * U1;
  const float* lse = (const float*)workspace;
  const float* lpb = lse + n;
  const float* lpy = lpb + n;
  const float* alpha = lpy + n;
  const float* beta = alpha + n;
  if (grad_bf16)
    hipLaunchKernelGGL((rnnt_grad_kernel<bf16_t>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, logits, targets,
                       logit_lengths, target_lengths, lse, lpb, lpy, alpha, beta, loss, (bf16_t*)grad, T, U1, V, Umax, blank,
                       grad_scale, grad_scale_dev, n);
  else
    hipLaunchKernelGGL((rnnt_grad_kernel<float>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, logits, targets,
                       logit_lengths, target_lengths, lse, lpb, lpy, alpha, beta, loss, (float*)grad, T, U1, V, Umax, blank,
                       grad_scale, grad_scale_dev, n);
  return EA_CHECK_LAUNCH();
}
