// Loss kernels for gfx950: fp32 log-softmax, CTC forward-backward, label-smoothed cross-entropy.
//
// CTC — reference: espresso/criterions/ctc_loss.py:59-100 -> torch.nn.functional.ctc_loss with
//   blank = index of "<s>" (0), reduction="sum", zero_infinity, cuDNN disabled (:85) i.e. ATen's
//   native alpha/beta recursion over log_softmax(logits.float())
//   (espresso/models/transformer/speech_transformer_encoder_model.py:141-150).
// Label-smoothed CE — reference: espresso/criterions/label_smoothed_cross_entropy_v2.py:94-119
//   (uniform smoothing: eps_i = eps/(V-1), loss = (1-eps-eps_i)*nll + eps_i*smooth, pad rows zeroed).
//
// CTC design: the loss is a tiny part of the flops but touches the (T',B,V) posteriors, so it is
// organised around HBM traffic: (1) log-softmax writes lprobs once; (2) a fully parallel gather
// pulls the 2U+1 needed columns per frame into a compact lattice G[b][t][s]; (3) one workgroup per
// utterance runs the alpha scan and the beta scan CONCURRENTLY (two halves of the workgroup, one
// state per lane, lattice rows double-buffered in LDS, next rows prefetched); (4) one workgroup
// per frame turns alpha+beta into d(loss)/d(logits) by building the sparse posterior row in LDS
// and streaming exp(lprobs) once — no dense gradient-of-lprobs tensor, no global atomics.
#include "common.h"
#include "espresso_amd.h"

namespace {

// ---------------------------------------------------------------------------------------------
// log_softmax over the last dim: in [M][ld_in] (fp32 or bf16) -> out fp32 [M][V]
template <typename TIn>
__global__ __launch_bounds__(256) void log_softmax_kernel(const TIn* __restrict__ in, long ld_in,
                                                          float* __restrict__ out, int V) {
  __shared__ float sm[16];
  const long row = blockIdx.x;
  const TIn* x = in + row * ld_in;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < V; c += 256) {
    float v;
    if constexpr (sizeof(TIn) == 2) v = bf2f(x[c]); else v = x[c];
    mx = fmaxf(mx, v);
  }
  mx = block_max(mx, sm);
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) {
    float v;
    if constexpr (sizeof(TIn) == 2) v = bf2f(x[c]); else v = x[c];
    s += expf(v - mx);
  }
  s = block_sum(s, sm);
  const float lse = mx + logf(s);
  float* o = out + row * (long)V;
  for (int c = threadIdx.x; c < V; c += 256) {
    float v;
    if constexpr (sizeof(TIn) == 2) v = bf2f(x[c]); else v = x[c];
    o[c] = v - lse;
  }
}

// ---------------------------------------------------------------------------------------------
// CTC.  lprobs [B][T][V] fp32 ; targets [B][Lmax] int32 ; in_len[B], tgt_len[B]
// lattice buffers G, A(lpha), Bt(beta): [B][T][Smax] fp32, Smax = 2*Lmax+1
__device__ __forceinline__ int ctc_label(const int* tg, int s, int blank) { return (s & 1) ? tg[s >> 1] : blank; }

__global__ __launch_bounds__(256) void ctc_gather_kernel(const float* __restrict__ lprobs, const int* __restrict__ targets,
                                                         const int* __restrict__ in_len, const int* __restrict__ tgt_len,
                                                         float* __restrict__ G, int T, int V, int Lmax, int blank) {
  const int b = blockIdx.y, t = blockIdx.x;
  if (t >= in_len[b]) return;
  const int S = 2 * tgt_len[b] + 1, Smax = 2 * Lmax + 1;
  const float* lp = lprobs + ((long)b * T + t) * V;
  const int* tg = targets + (long)b * Lmax;
  float* g = G + ((long)b * T + t) * Smax;
  for (int s = threadIdx.x; s < S; s += blockDim.x) g[s] = lp[ctc_label(tg, s, blank)];
}

// One workgroup per (utterance, direction): blockIdx.y = 0 runs the alpha recursion forwards, 1 the beta recursion backwards;
// blockDim = SP (Smax rounded up to 64), one lattice state per thread, rows double-buffered in LDS.  (Round 4: the two
// directions used to share one 2*SP-thread workgroup — a four-wave barrier per frame — and the log-sum-exp ran on libm's
// expf / logf: 0.56 us per frame, 172 us for a 308-frame batch during which 21 workgroups had the device to themselves.  Two
// waves per barrier and the hardware exp2 / log2 (v_exp_f32 / v_log_f32, 1 ulp) bring a frame to ~0.15 us.)
__global__ void ctc_scan_kernel(const float* __restrict__ G, const int* __restrict__ targets,
                                const int* __restrict__ in_len, const int* __restrict__ tgt_len,
                                float* __restrict__ A, float* __restrict__ Bt, float* __restrict__ nll_out,
                                int T, int Lmax, int blank, int SP) {
  extern __shared__ float sh[];  // [2 buf][SP + 2]
  const int b = blockIdx.x;
  const int Tb = in_len[b], L = tgt_len[b];
  const int S = 2 * L + 1, Smax = 2 * Lmax + 1;
  const int dir = blockIdx.y;
  const int s = threadIdx.x;
  float* buf0 = sh + 0 * (SP + 2) + (dir ? 0 : 2);  // alpha reads s-1,s-2 ; beta reads s+1,s+2
  float* buf1 = sh + 1 * (SP + 2) + (dir ? 0 : 2);
  const int* tg = targets + (long)b * Lmax;
  if (Tb <= 0) {
    if (threadIdx.x == 0 && dir == 0) nll_out[b] = (L == 0) ? 0.f : INFINITY;
    return;
  }
  // skip transition allowed?  alpha: from s-2 if label(s) != blank and != label(s-2)
  //                           beta : to   s+2 if label(s+2) != blank and != label(s)
  bool skip = false;
  if (s < S) {
    if (!dir) { if ((s & 1) && s >= 2) skip = tg[s >> 1] != tg[(s >> 1) - 1]; }
    else      { if ((s & 1) && s + 2 < S) skip = tg[(s >> 1) + 1] != tg[s >> 1]; }
  }
  // pads of the LDS rows are -inf
  if (threadIdx.x < 2) {
    sh[0 * (SP + 2) + (dir ? SP : 0) + threadIdx.x] = -INFINITY;
    sh[1 * (SP + 2) + (dir ? SP : 0) + threadIdx.x] = -INFINITY;
  }
  const float* Gb = G + (long)b * T * Smax;
  float* Ob = (dir ? Bt : A) + (long)b * T * Smax;
  // t = first frame of this direction
  int t = dir ? Tb - 1 : 0;
  const int step = dir ? -1 : 1;
  {
    float v = -INFINITY;
    if (s < S) {
      const float lp = Gb[(long)t * Smax + s];
      if (!dir) { if (s == 0 || s == 1) v = lp; }
      else      { if (s == S - 1 || s == S - 2) v = lp; }
    }
    buf0[s] = v;
    if (s < S) Ob[(long)t * Smax + s] = v;
  }
  // the emission log-probabilities do not depend on the recursion: requested four frames ahead, and the barrier of a frame waits
  // for the LDS hand-off only (round 6: __syncthreads() also drains the vector-memory counter, i.e. the prefetch AND the store of
  // the frame's row — a memory round trip per frame, 0.5 us, where the arithmetic needs 0.15)
  // Branch-free loop body: every lane loads and stores on every frame (clamped addresses, the value selected afterwards) — with a
  // load or store inside a divergent branch hipcc's wait-count pass loses the count and waits for vmcnt(0), i.e. for the request
  // it has just issued.  Lanes s >= S keep -inf and store nothing.
  const long gstep = (long)step * Smax;
  const int sc = s < S ? s : (S - 1);
  const int so = s < S ? s : (S - 1);
  const float* g0 = Gb + (long)t * Smax + sc;   // row of frame `it` = g0 + it * gstep
  auto row = [&](int it) { return g0 + (long)(it < Tb ? it : Tb - 1) * gstep; };
  float* op = Ob + (long)t * Smax + so;
  const bool live = s < S;
  __syncthreads();
  float* rd = buf0;
  float* wr = buf1;
  const int o1 = dir ? 1 : -1, o2 = dir ? 2 : -2;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  // four registers, one per frame modulo 4, each refilled right after use with the row four frames on (no register rotation:
  // copying a register whose load is pending is a wait for that load)
  auto frame = [&](int it, float& q) {
    if (it >= Tb) return;  // (uniform)
    const float lp = q;
    q = *row(it + 4);
    op += gstep;
    const float a0 = rd[s];
    const float a1 = rd[s + o1];
    const float a2 = skip ? rd[s + o2] : -INFINITY;
    const float m = fmaxf(a0, fmaxf(a1, a2));
    const float mm = m > -INFINITY ? m : 0.f;
    float v = mm + LN2 * __builtin_amdgcn_logf(__builtin_amdgcn_exp2f((a0 - mm) * LOG2E) + __builtin_amdgcn_exp2f((a1 - mm) * LOG2E) +
                                               __builtin_amdgcn_exp2f((a2 - mm) * LOG2E)) + lp;
    v = (live && m > -INFINITY) ? v : -INFINITY;
    if (live) *op = v;  // (a store under a branch only makes the wait-count pass assume it was issued: the loads stay counted)
    wr[s] = v;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this lane's hand-off is in LDS (global requests stay in flight)
    __builtin_amdgcn_s_barrier();
    float* tmp = rd; rd = wr; wr = tmp;
  };
  float q1 = *row(1), q2 = *row(2), q3 = *row(3), q4 = *row(4);
  for (int it = 1; it < Tb; it += 4) {
    frame(it, q1);
    frame(it + 1, q2);
    frame(it + 2, q3);
    frame(it + 3, q4);
  }
  __syncthreads();
  if (threadIdx.x == 0 && dir == 0) {
    // the alpha direction finished at t = Tb-1 in rd
    const float l1 = rd[S - 1];
    const float l2 = S >= 2 ? rd[S - 2] : -INFINITY;
    const float m = fmaxf(l1, l2);
    nll_out[b] = (m == -INFINITY) ? INFINITY : -(m + logf(expf(l1 - m) + expf(l2 - m)));
  }
}

// One block per (t, b): dlogits[b,t,:] = scale * ( exp(lp)*sumpost - post_by_class )  (bf16 or fp32 out)
template <typename TOut>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lprobs, const float* __restrict__ G,
                                                       const float* __restrict__ A, const float* __restrict__ Bt,
                                                       const float* __restrict__ nll, const int* __restrict__ targets,
                                                       const int* __restrict__ in_len, const int* __restrict__ tgt_len,
                                                       TOut* __restrict__ dlogits, long ld_out, int T, int V, int Lmax,
                                                       int blank, float scale, const float* __restrict__ scale_dev,
                                                       int zero_infinity) {
  extern __shared__ float post[];  // [V]
  __shared__ float sm[16];
  const int b = blockIdx.y, t = blockIdx.x;
  TOut* out = dlogits + ((long)b * T + t) * ld_out;
  const float nl = nll[b];
  if (scale_dev) scale *= scale_dev[0];
  const bool dead = t >= in_len[b] || (zero_infinity && !(nl < INFINITY));
  for (int c = V + threadIdx.x; c < ld_out; c += 256) {
    if constexpr (sizeof(TOut) == 2) out[c] = 0; else out[c] = 0.f;
  }
  if (dead) {
    for (int c = threadIdx.x; c < V; c += 256) {
      if constexpr (sizeof(TOut) == 2) out[c] = 0; else out[c] = 0.f;
    }
    return;
  }
  for (int c = threadIdx.x; c < V; c += 256) post[c] = 0.f;
  __syncthreads();
  const int S = 2 * tgt_len[b] + 1, Smax = 2 * Lmax + 1;
  const long lo = ((long)b * T + t) * Smax;
  const int* tg = targets + (long)b * Lmax;
  float part = 0.f, blank_part = 0.f;
  for (int s = threadIdx.x; s < S; s += 256) {
    const float ab = A[lo + s] + Bt[lo + s];
    // alpha and beta both include the emission at t -> subtract it once
    const float p = (ab == -INFINITY) ? 0.f : expf(ab + nl - G[lo + s]);
    part += p;
    if (s & 1) atomicAdd(&post[tg[s >> 1]], p); else blank_part += p;
  }
  const float sumpost = block_sum(part, sm);
  const float bsum = block_sum(blank_part, sm);
  if (threadIdx.x == 0) post[blank] += bsum;
  __syncthreads();
  const float* lp = lprobs + ((long)b * T + t) * V;
  for (int c = threadIdx.x; c < V; c += 256) {
    const float g = scale * (expf(lp[c]) * sumpost - post[c]);
    if constexpr (sizeof(TOut) == 2) out[c] = f2bf(g); else out[c] = g;
  }
}

// ---------------------------------------------------------------------------------------------
// CTC greedy decoding (espresso/tools/ctc_decoder.py:172-188): per-frame argmax, collapse repeats, drop blanks.
// x: [B][T][ld] logits or log-probs (argmax is invariant); best[b][t] = argmax index (lowest index on ties),
// bestv[b][t] = max value.
template <typename TIn>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const TIn* __restrict__ x, long ld, int V, int* __restrict__ best,
                                                          float* __restrict__ bestv) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const long row = blockIdx.x;
  const TIn* r = x + row * ld;
  float mv = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = threadIdx.x; c < V; c += 256) {
    float v;
    if constexpr (sizeof(TIn) == 2) v = bf2f(r[c]); else v = r[c];
    if (v > mv || (v == mv && c < mi)) { mv = v; mi = c; }
  }
  sv[threadIdx.x] = mv;
  si[threadIdx.x] = mi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float ov = sv[threadIdx.x + s];
      const int oi = si[threadIdx.x + s];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { best[row] = si[0]; bestv[row] = sv[0]; }
}
// one thread per utterance: tokens[b][0..n) (collapsed, blank removed), align[b][u] = first frame of token u
__global__ void ctc_collapse_kernel(const int* __restrict__ best, const float* __restrict__ bestv, const int* __restrict__ in_len,
                                    int* __restrict__ tokens, int* __restrict__ align, int* __restrict__ out_len,
                                    float* __restrict__ score, int B, int T, int blank, int pad) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int n = in_len[b];
  int u = 0, prev = -1;
  float s = 0.f;
  for (int t = 0; t < n; ++t) {
    const int k = best[(long)b * T + t];
    s += bestv[(long)b * T + t];
    if (k != blank && k != prev) {
      tokens[(long)b * T + u] = k;
      if (align) align[(long)b * T + u] = t;
      ++u;
    }
    prev = k;
  }
  for (int t = u; t < T; ++t) tokens[(long)b * T + t] = pad;
  out_len[b] = u;
  score[b] = s;
}

// ---------------------------------------------------------------------------------------------
// Label-smoothed CE — espresso/criterions/label_smoothed_cross_entropy_v2.py:49-119.  logits [M][ld] (bf16 or fp32), int32
// targets.  smoothing: 0 uniform (eps_i = eps/(V-1): (1-eps-eps_i) nll + eps_i * -sum_v lprob_v), 1 unigram (prior[v] over
// the vocabulary: (1-eps) nll + eps * -sum_v prior_v lprob_v), 2 temporal (the previous / next two targets of the same
// sentence with weights 2:5:5:2, pad neighbours dropped, normalised: row m = b*U + u).  out_loss[0] += sum loss,
// out_loss[1] += sum nll; optional dlogits = scale * dloss/dlogits.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void lsce_kernel(const TIn* __restrict__ logits, long ld, const int* __restrict__ target,
                                                   float* __restrict__ out_loss, TOut* __restrict__ dlogits, long ld_out,
                                                   int V, int pad_idx, float eps, float scale, int smoothing,
                                                   const float* __restrict__ prior, int U) {
  __shared__ float sm[16];
  const long row = blockIdx.x;
  const TIn* x = logits + row * ld;
  const int tgt = target[row];
  TOut* dl = dlogits ? dlogits + row * ld_out : nullptr;
  if (tgt == pad_idx) {
    if (dl) for (int c = threadIdx.x; c < V; c += 256) { if constexpr (sizeof(TOut) == 2) dl[c] = 0; else dl[c] = 0.f; }
    return;
  }
  auto ld1 = [&](int c) -> float { if constexpr (sizeof(TIn) == 2) return bf2f(x[c]); else return x[c]; };
  float mx = -INFINITY, tot = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) {
    const float v = ld1(c);
    mx = fmaxf(mx, v);
    tot += smoothing == 1 ? prior[c] * v : v;  // sum_v w_v x_v for the dense weightings
  }
  mx = block_max(mx, sm);
  tot = block_sum(tot, sm);
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) s += expf(ld1(c) - mx);
  s = block_sum(s, sm);
  const float lse = mx + logf(s);
  const float nll = lse - ld1(tgt);
  // temporal neighbours (at most four non-zero weights)
  int nb[4] = {-1, -1, -1, -1};
  float nw[4] = {0.f, 0.f, 0.f, 0.f};
  float wsum = 1.f;
  float smooth, wn, we;
  if (smoothing == 2) {
    const int u = (int)(row % U);
    const int off[4] = {-2, -1, 1, 2};
    const float wt[4] = {2.f, 5.f, 5.f, 2.f};
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int uu = u + off[k];
      if (uu < 0 || uu >= U) continue;
      const int t = target[row + off[k]];
      if (t == pad_idx) continue;
      nb[k] = t;
      nw[k] = wt[k];
      z += wt[k];
    }
    smooth = 0.f;
    if (z > 0.f) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (nb[k] >= 0) { nw[k] /= z; smooth += nw[k] * (lse - ld1(nb[k])); }
    } else {
      wsum = 0.f;
    }
    wn = 1.f - eps;
    we = eps;
  } else if (smoothing == 1) {
    smooth = lse - tot;  // prior sums to one
    wn = 1.f - eps;
    we = eps;
  } else {
    smooth = (float)V * lse - tot;  // -sum_c lprob_c
    we = eps / (float)(V - 1);
    wn = 1.f - eps - we;
    wsum = (float)V;
  }
  if (threadIdx.x == 0) {
    atomicAdd(out_loss + 0, wn * nll + we * smooth);
    atomicAdd(out_loss + 1, nll);
  }
  if (dl) {
    // d/dz [wn (lse - z_t) + we (W lse - sum_v w_v z_v)] = (wn + we W) softmax - wn onehot_t - we w
    const float k = wn + we * wsum;
    for (int c = threadIdx.x; c < V; c += 256) {
      float g = expf(ld1(c) - lse) * k - (c == tgt ? wn : 0.f);
      if (smoothing == 0) g -= we;
      else if (smoothing == 1) g -= we * prior[c];
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (nb[q] == c) g -= we * nw[q];
      }
      g *= scale;
      if constexpr (sizeof(TOut) == 2) dl[c] = f2bf(g); else dl[c] = g;
    }
  }
}

}  // namespace

extern "C" int ea_log_softmax_f32(const float* in, long ld_in, float* out, long M, int V, hipStream_t stream) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL((log_softmax_kernel<float>), dim3((unsigned)M), dim3(256), 0, stream, in, ld_in, out, V);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_log_softmax_bf16(const void* in, long ld_in, float* out, long M, int V, hipStream_t stream) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL((log_softmax_kernel<bf16_t>), dim3((unsigned)M), dim3(256), 0, stream, (const bf16_t*)in, ld_in, out, V);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_ctc_greedy_decode(const void* x, long ld, int x_bf16, const int* in_len, int* best /*[B*T]*/,
                                    float* bestv /*[B*T]*/, int* tokens /*[B][T]*/, int* align /*[B][T] or NULL*/,
                                    int* out_len, float* score, int B, int T, int V, int blank, int pad, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if (x_bf16)
    hipLaunchKernelGGL((argmax_rows_kernel<bf16_t>), dim3((unsigned)((long)B * T)), dim3(256), 0, stream, (const bf16_t*)x, ld, V, best, bestv);
  else
    hipLaunchKernelGGL((argmax_rows_kernel<float>), dim3((unsigned)((long)B * T)), dim3(256), 0, stream, (const float*)x, ld, V, best, bestv);
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, best, bestv, in_len, tokens, align, out_len, score,
                     B, T, blank, pad);
  return EA_CHECK_LAUNCH();
}

extern "C" long ea_ctc_workspace_bytes(int B, int T, int Lmax) {
  return 3L * B * T * (2L * Lmax + 1) * (long)sizeof(float);
}

extern "C" int ea_ctc_loss(const float* lprobs, const int* targets, const int* in_len, const int* tgt_len,
                           float* nll /*[B]*/, void* workspace, int B, int T, int V, int Lmax, int blank,
                           hipStream_t stream) {
  if (B <= 0) return 0;
  if (T <= 0 || V <= 0 || Lmax < 0) return -2;
  const int Smax = 2 * Lmax + 1;
  const int SP = ((Smax + 63) / 64) * 64;
  if (2 * SP > 1024) return -3;  // Lmax <= 255 (max_target_positions 200 in the recipes)
  float* G = (float*)workspace;
  float* A = G + (long)B * T * Smax;
  float* Bt = A + (long)B * T * Smax;
  hipLaunchKernelGGL(ctc_gather_kernel, dim3(T, B), dim3(256), 0, stream, lprobs, targets, in_len, tgt_len, G, T, V,
                     Lmax, blank);
  hipLaunchKernelGGL(ctc_scan_kernel, dim3(B, 2), dim3(SP), (size_t)2 * (SP + 2) * sizeof(float), stream, G, targets,
                     in_len, tgt_len, A, Bt, nll, T, Lmax, blank, SP);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_ctc_grad(const float* lprobs, const void* workspace, const float* nll, const int* targets,
                           const int* in_len, const int* tgt_len, void* dlogits, long ld_out, int dlogits_bf16, int B,
                           int T, int V, int Lmax, int blank, float grad_scale, const float* grad_scale_dev,
                           int zero_infinity, hipStream_t stream) {
  if (B <= 0) return 0;
  const int Smax = 2 * Lmax + 1;
  const float* G = (const float*)workspace;
  const float* A = G + (long)B * T * Smax;
  const float* Bt = A + (long)B * T * Smax;
  if (dlogits_bf16)
    hipLaunchKernelGGL((ctc_grad_kernel<bf16_t>), dim3(T, B), dim3(256), (size_t)V * sizeof(float), stream, lprobs, G, A,
                       Bt, nll, targets, in_len, tgt_len, (bf16_t*)dlogits, ld_out, T, V, Lmax, blank, grad_scale,
                       grad_scale_dev, zero_infinity);
  else
    hipLaunchKernelGGL((ctc_grad_kernel<float>), dim3(T, B), dim3(256), (size_t)V * sizeof(float), stream, lprobs, G, A,
                       Bt, nll, targets, in_len, tgt_len, (float*)dlogits, ld_out, T, V, Lmax, blank, grad_scale,
                       grad_scale_dev, zero_infinity);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_label_smoothed_ce(const void* logits, long ld, int logits_bf16, const int* target,
                                    float* out_loss /*[2] zeroed*/, void* dlogits, long ld_out, int dlogits_bf16, long M,
                                    int V, int pad_idx, float eps, float grad_scale, int smoothing, const float* prior,
                                    int tgt_len, hipStream_t stream) {
  if (M <= 0) return 0;
  if (smoothing < 0 || smoothing > 2 || (smoothing == 1 && !prior) || (smoothing == 2 && (tgt_len <= 0 || M % tgt_len))) return -2;
  dim3 g((unsigned)M), blk(256);
#define EA_LSCE(TI, TO)                                                                                               \
  hipLaunchKernelGGL((lsce_kernel<TI, TO>), g, blk, 0, stream, (const TI*)logits, ld, target, out_loss, (TO*)dlogits, ld_out, V, \
                     pad_idx, eps, grad_scale, smoothing, prior, tgt_len)
  if (logits_bf16) { if (dlogits_bf16 || !dlogits) EA_LSCE(bf16_t, bf16_t); else EA_LSCE(bf16_t, float); }
  else { if (dlogits_bf16 && dlogits) EA_LSCE(float, bf16_t); else EA_LSCE(float, float); }
#undef EA_LSCE
  return EA_CHECK_LAUNCH();
}
