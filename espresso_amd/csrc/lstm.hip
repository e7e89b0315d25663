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
                                                            int B, int H) {
  const long n = (long)B * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / H), u = (int)(i % H);
    const float* g = G + (long)b * ldg + u;
    const float gi = sigmoid_f(g[0]), gf = sigmoid_f(g[H]), gg = tanh_f(g[2 * H]), go = sigmoid_f(g[3 * H]);
    const float cp = c_prev ? c_prev[i] : 0.f;
    float c = gf * cp + gi * gg;
    float h = go * tanh_f(c);
    if (keep_row && keep_row[b]) {  // frozen row (finished hypothesis / padded step): state passes through unchanged
      c = cp;
      h = h_prev_f32 ? h_prev_f32[i] : 0.f;
    }
    c_out[i] = c;
    if (h_f32) h_f32[i] = h;
    if (h_bf16) h_bf16[(long)b * ldh + u] = f2bf(h);
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
                                                            int H) {
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

inline int lgrid(long n) {
  long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int ea_lstm_cell_fwd(const float* gates_pre, long ldg, const float* c_prev, float* c_out, float* h_out_f32,
                                void* h_out_bf16, long ldh, float* gates_act, const uint8_t* keep_row, const float* h_prev_f32,
                                int B, int H, hipStream_t stream) {
  if (B <= 0 || H <= 0) return 0;
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(lgrid((long)B * H)), dim3(256), 0, stream, gates_pre, ldg, c_prev, c_out, h_out_f32,
                     (bf16_t*)h_out_bf16, ldh, gates_act, keep_row, h_prev_f32, B, H);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_lstm_cell_bwd(const void* dh_bf16, long ld_dh, const float* dh_f32, const float* dc_in, const float* gates_act,
                                const float* c_prev, const float* c, void* dgates, long lddg, float* dc_prev, int B, int H,
                                hipStream_t stream) {
  if (B <= 0 || H <= 0) return 0;
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(lgrid((long)B * H)), dim3(256), 0, stream, (const bf16_t*)dh_bf16, ld_dh, dh_f32, dc_in,
                     gates_act, c_prev, c, (bf16_t*)dgates, lddg, dc_prev, B, H);
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
