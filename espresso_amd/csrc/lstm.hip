// LSTM cell element-wise stages for gfx950 (HBM-bound; the matrix products x W_ih^T / h W_hh^T run on the MFMA GEMM).
//
// Reference: torch.nn.LSTMCell as wrapped by fairseq/models/lstm.py:LSTMCell and driven step by step in
// espresso/models/speech_lstm.py:846-893 (SpeechLSTMDecoder.extract_features: predictor of the transducer, LSTM language
// model, attention decoder).  Gate order of the packed 4H pre-activations is PyTorch's (i, f, g, o):
//   i = sigmoid(G[0:H])  f = sigmoid(G[H:2H])  g = tanh(G[2H:3H])  o = sigmoid(G[3H:4H])
//   c' = f * c + i * g          h' = o * tanh(c')
// Forward keeps the activated gates (fp32) for the backward pass; backward returns the gradient of the pre-activations
// as bf16 (it is the A/B operand of the dgrad / wgrad GEMMs) and the gradient flowing to c_{t-1}.
#include "common.h"
#include "espresso_amd.h"

namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
  // tanh(x) = 1 - 2 / (exp(2x) + 1), saturating cleanly at +-1
  const float e = __expf(2.f * x);
  return 1.f - 2.f / (e + 1.f);
}

__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ G, long ldg, const float* __restrict__ c_prev,
                                                            float* __restrict__ c_out, float* __restrict__ h_f32,
                                                            bf16_t* __restrict__ h_bf16, long ldh, float* __restrict__ act,
                                                            const uint8_t* __restrict__ keep_row, const float* __restrict__ h_prev_f32,
                                                            int B, int H, int frozen_out_zero) {
  const long n = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / H), u = (int)(i % H);
    const float* g = G + (long)b * ldg + u;
    const float gi = sigmoid_f(g[0]), gf = sigmoid_f(g[H]), gg = tanh_f(g[2 * H]), go = sigmoid_f(g[3 * H]);
    const float cp = c_prev ? c_prev[i] : 0.f;
    float c = gf * cp + gi * gg;
    float h = go * tanh_f(c);
    const bool frozen = keep_row && keep_row[b];
    if (frozen) {  // frozen row (finished hypothesis / padded step): state passes through unchanged
      c = cp;
      h = h_prev_f32 ? h_prev_f32[i] : 0.f;
    }
    c_out[i] = c;
    if (h_f32) h_f32[i] = h;
    // packed-sequence semantics (torch pad_packed_sequence, padding_value 0): padded steps emit zeros
    if (h_bf16) h_bf16[(long)b * ldh + u] = (frozen && frozen_out_zero) ? (bf16_t)0 : f2bf(h);
    if (act) {
      float* a = act + (long)b * 4 * H + u;
      a[0] = gi; a[H] = gf; a[2 * H] = gg; a[3 * H] = go;
    }
  }
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const bf16_t* __restrict__ dh_a, long ld_dha, const float* __restrict__ dh_b,
                                                            const float* __restrict__ dc_in, const float* __restrict__ act,
                                                            const float* __restrict__ c_prev, const float* __restrict__ c,
                                                            bf16_t* __restrict__ dG, long lddg, float* __restrict__ dc_prev, int B,
                                                            int H, const uint8_t* __restrict__ frozen) {
  const long n = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / H), u = (int)(i % H);
    float dh = 0.f;
    if (dh_a) dh += bf2f(dh_a[(long)b * ld_dha + u]);
    if (dh_b) dh += dh_b[i];
    const float* a = act + (long)b * 4 * H + u;
    const float gi = a[0], gf = a[H], gg = a[2 * H], go = a[3 * H];
    const float tc = tanh_f(c[i]);
    const float dc = (dc_in ? dc_in[i] : 0.f) + dh * go * (1.f - tc * tc);
    const float cp = c_prev ? c_prev[i] : 0.f;
    bf16_t* d = dG + (long)b * lddg + u;
    if (frozen && frozen[b]) {  // padded step: no dependence on the gates; the cell state passed through
      d[0] = d[H] = d[2 * H] = d[3 * H] = 0;
      dc_prev[i] = dc_in ? dc_in[i] : 0.f;
      continue;
    }
    d[0] = f2bf(dc * gg * gi * (1.f - gi));
    d[H] = f2bf(dc * cp * gf * (1.f - gf));
    d[2 * H] = f2bf(dc * gi * (1.f - gg * gg));
    d[3 * H] = f2bf(dh * tc * go * (1.f - go));
    dc_prev[i] = dc * gf;
  }
}

// rows of a [N][W] state (fp32 or bf16) gathered by parent index: out[n] = in[parent[n]]  (beam reorder of LSTM states,
// speech_lstm.py:981-999 reorder_incremental_state)
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ in, T* __restrict__ out, const int* __restrict__ parent,
                                                          int N, int W) {
  const long n = (long)N * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / W), cidx = (int)(i % W);
    out[i] = in[(long)parent[r] * W + cidx];
  }
}

// ---- Bahdanau (additive) attention of one decoder step — espresso/modules/speech_attention.py:38-87 -----------------
//   score[t][b] = sum_a nv[a] * tanh(qp[b][a] + key[t][b][a] + bias[a]),  nv = g * v / ||v||  (computed by the caller)
//   p[:, b] = softmax over t < len[b];  ctx[b] = sum_t p[t][b] * value[t][b]
// One 256-thread workgroup per batch row; the T scores live in LDS.
__global__ __launch_bounds__(256) void bahdanau_fwd_kernel(const bf16_t* __restrict__ qp, const bf16_t* __restrict__ key,
                                                           const bf16_t* __restrict__ value, const float* __restrict__ nv,
                                                           const float* __restrict__ bias, const int* __restrict__ len,
                                                           float* __restrict__ p_out, bf16_t* __restrict__ ctx, long ldc, int T, int B,
                                                           int A, int Cv, const int* __restrict__ kv_col, int Bkv) {
  extern __shared__ float sc[];  // [T]
  __shared__ float sm[16];
  const int b = blockIdx.x;
  const int kb = kv_col ? kv_col[b] : b;  // beam search: hypotheses of one sentence share its encoder keys / values
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = len ? min(len[kb], T) : T;
  for (int t = wave; t < T; t += 4) {
    float s = 0.f;
    if (t < L) {
      const bf16_t* k = key + ((long)t * Bkv + kb) * A;
      for (int a = lane; a < A; a += 64) s += nv[a] * tanh_f(bf2f(qp[(long)b * A + a]) + bf2f(k[a]) + (bias ? bias[a] : 0.f));
      s = wave_sum(s);
    } else {
      s = -INFINITY;
    }
    if (lane == 0) sc[t] = s;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = threadIdx.x; t < T; t += 256) mx = fmaxf(mx, sc[t]);
  mx = block_max(mx, sm);
  float sum = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    const float e = sc[t] == -INFINITY ? 0.f : __expf(sc[t] - mx);
    sc[t] = e;
    sum += e;
  }
  sum = block_sum(sum, sm);
  const float inv = 1.f / sum;
  for (int t = threadIdx.x; t < T; t += 256) {
    const float pr = sc[t] * inv;
    sc[t] = pr;
    p_out[(long)t * B + b] = pr;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Cv; c += 256) {
    float acc = 0.f;
    for (int t = 0; t < L; ++t) acc += sc[t] * bf2f(value[((long)t * Bkv + kb) * Cv + c]);
    ctx[(long)b * ldc + c] = f2bf(acc);
  }
}

// backward of one step: dctx [B][ldd] -> dqp (bf16 [B][A]); dkey_acc / dvalue_acc (fp32, +=); dnv_acc / dbias_acc (fp32, atomics)
__global__ __launch_bounds__(256) void bahdanau_bwd_kernel(const bf16_t* __restrict__ dctx, long ldd, const bf16_t* __restrict__ qp,
                                                           const bf16_t* __restrict__ key, const bf16_t* __restrict__ value,
                                                           const float* __restrict__ nv, const float* __restrict__ bias,
                                                           const int* __restrict__ len, const float* __restrict__ p,
                                                           bf16_t* __restrict__ dqp, float* __restrict__ dkey_acc,
                                                           float* __restrict__ dvalue_acc, float* __restrict__ dnv_acc,
                                                           float* __restrict__ dbias_acc, int T, int B, int A, int Cv) {
  extern __shared__ float sh[];  // ds[T] | red[4][A] (dq) | red2[4][A] (dnv)
  __shared__ float sm[16];
  float* ds = sh;
  float* rq = sh + T;
  float* rn = rq + 4 * A;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = len ? min(len[b], T) : T;
  // dp[t] = dctx . value[t]
  for (int t = wave; t < T; t += 4) {
    float s = 0.f;
    if (t < L) {
      const bf16_t* v = value + ((long)t * B + b) * Cv;
      for (int c = lane; c < Cv; c += 64) s += bf2f(dctx[(long)b * ldd + c]) * bf2f(v[c]);
      s = wave_sum(s);
    }
    if (lane == 0) ds[t] = s;
  }
  __syncthreads();
  float dot = 0.f;
  for (int t = threadIdx.x; t < L; t += 256) dot += p[(long)t * B + b] * ds[t];
  dot = block_sum(dot, sm);
  for (int t = threadIdx.x; t < T; t += 256) ds[t] = t < L ? p[(long)t * B + b] * (ds[t] - dot) : 0.f;
  __syncthreads();
  // dvalue += p[t] * dctx
  for (int c = threadIdx.x; c < Cv; c += 256) {
    const float g = bf2f(dctx[(long)b * ldd + c]);
    for (int t = 0; t < L; ++t) dvalue_acc[((long)t * B + b) * Cv + c] += p[(long)t * B + b] * g;
  }
  // through tanh: every lane owns columns a = lane, lane+64, ... ; the 4 wavefronts split t
  for (int a0 = 0; a0 < A; a0 += 64) {
    const int a = a0 + lane;
    float dq = 0.f, dn = 0.f;
    if (a < A) {
      const float q = bf2f(qp[(long)b * A + a]) + (bias ? bias[a] : 0.f);
      const float w = nv[a];
      for (int t = wave; t < L; t += 4) {
        const long ki = ((long)t * B + b) * A + a;
        const float th = tanh_f(q + bf2f(key[ki]));
        const float dpre = ds[t] * w * (1.f - th * th);
        dkey_acc[ki] += dpre;
        dq += dpre;
        dn += ds[t] * th;
      }
      rq[wave * A + a] = dq;
      rn[wave * A + a] = dn;
    }
  }
  __syncthreads();
  for (int a = threadIdx.x; a < A; a += 256) {
    const float dq = rq[a] + rq[A + a] + rq[2 * A + a] + rq[3 * A + a];
    const float dn = rn[a] + rn[A + a] + rn[2 * A + a] + rn[3 * A + a];
    dqp[(long)b * A + a] = f2bf(dq);
    atomicAdd(dnv_acc + a, dn);
    if (dbias_acc) atomicAdd(dbias_acc + a, dq);
  }
}

inline int lgrid(long n) {
  long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int ea_lstm_cell_fwd(const float* gates_pre, long ldg, const float* c_prev, float* c_out, float* h_out_f32,
                                void* h_out_bf16, long ldh, float* gates_act, const uint8_t* keep_row, const float* h_prev_f32,
                                int frozen_out_zero, int B, int H, hipStream_t stream) {
  if (B <= 0 || H <= 0) return 0;
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(lgrid((long)B * H)), dim3(256), 0, stream, gates_pre, ldg, c_prev, c_out, h_out_f32,
                     (bf16_t*)h_out_bf16, ldh, gates_act, keep_row, h_prev_f32, B, H, frozen_out_zero);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_lstm_cell_bwd(const void* dh_bf16, long ld_dh, const float* dh_f32, const float* dc_in, const float* gates_act,
                                const float* c_prev, const float* c, void* dgates, long lddg, float* dc_prev,
                                const uint8_t* frozen, int B, int H, hipStream_t stream) {
  if (B <= 0 || H <= 0) return 0;
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(lgrid((long)B * H)), dim3(256), 0, stream, (const bf16_t*)dh_bf16, ld_dh, dh_f32, dc_in,
                     gates_act, c_prev, c, (bf16_t*)dgates, lddg, dc_prev, B, H, frozen);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_gather_rows(const void* in, void* out, const int* parent, int N, int W, int elem_bytes, hipStream_t stream) {
  if (N <= 0 || W <= 0) return 0;
  if (elem_bytes == 4)
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(lgrid((long)N * W)), dim3(256), 0, stream, (const float*)in, (float*)out, parent, N, W);
  else if (elem_bytes == 2)
    hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, dim3(lgrid((long)N * W)), dim3(256), 0, stream, (const bf16_t*)in, (bf16_t*)out, parent, N, W);
  else
    return -2;
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_bahdanau_fwd(const void* qp, const void* key, const void* value, const float* nv, const float* bias, const int* len,
                               float* p_out, void* ctx, long ldc, int T, int B, int A, int Cv, const int* kv_col, int Bkv,
                               hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  if ((size_t)T * sizeof(float) > 48 * 1024) return -2;
  if (!kv_col) Bkv = B;
  hipLaunchKernelGGL(bahdanau_fwd_kernel, dim3(B), dim3(256), (size_t)T * sizeof(float), stream, (const bf16_t*)qp, (const bf16_t*)key,
                     (const bf16_t*)value, nv, bias, len, p_out, (bf16_t*)ctx, ldc, T, B, A, Cv, kv_col, Bkv);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_bahdanau_bwd(const void* dctx, long ldd, const void* qp, const void* key, const void* value, const float* nv,
                               const float* bias, const int* len, const float* p, void* dqp, float* dkey_acc, float* dvalue_acc,
                               float* dnv_acc, float* dbias_acc, int T, int B, int A, int Cv, hipStream_t stream) {
  if (B <= 0 || T <= 0) return 0;
  const size_t lds = ((size_t)T + 8 * (size_t)A) * sizeof(float);
  if (lds > 60 * 1024) return -2;
  hipLaunchKernelGGL(bahdanau_bwd_kernel, dim3(B), dim3(256), lds, stream, (const bf16_t*)dctx, ldd, (const bf16_t*)qp, (const bf16_t*)key,
                     (const bf16_t*)value, nv, bias, len, p, (bf16_t*)dqp, dkey_acc, dvalue_acc, dnv_acc, dbias_acc, T, B, A, Cv);
  return EA_CHECK_LAUNCH();
}
