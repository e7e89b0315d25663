// Flat-buffer optimizer kernels for gfx950: gradient L2 norm, clip coefficient, fused Adam.
// Replaces fairseq/utils.py:347-397 (clip_grad_norm_, apex multi_tensor_l2norm when present),
// fairseq/optim/fairseq_optimizer.py multiply_grads, and fairseq/optim/adam.py:215-240 /
// fused_adam.py:255-370 (per-parameter Adam update).  Parameters, gradients and both moments live
// in ONE contiguous fp32 buffer each (sized for HBM3E: ~1.3 GB for the 80 M-parameter Conformer), so
// every step is a single streaming pass at HBM bandwidth: read p,g,m,v + write p,m,v,bf16(p) and
// the gradient buffer is re-zeroed in the same pass.  The clip coefficient stays on the device
// (no host sync between backward and the update).
#include "common.h"
#include "espresso_amd.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  __shared__ float sm[16];
  float s = 0.f;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (long j = i; j < n; ++j) s += g[j] * g[j];
    }
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) atomicAdd(out, s);
}

// coef[0] = pre_scale * min(1, max_norm / (norm + 1e-6)),  coef[1] = norm = sqrt(sumsq) * pre_scale
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float pre_scale, const float* __restrict__ denom,
                                 float max_norm, float* __restrict__ coef) {
  if (denom) pre_scale = pre_scale / denom[0];
  const float norm = sqrtf(sumsq[0]) * pre_scale;
  float c = 1.f;
  if (max_norm > 0.f) c = fminf(1.f, max_norm / (norm + 1e-6f));
  coef[0] = pre_scale * c;
  coef[1] = norm;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ p_bf16, long n,
                                                   const float* __restrict__ coef, float lr, float beta1, float beta2,
                                                   float eps, float weight_decay, float step_size, int zero_grad) {
  const float gs = coef ? coef[0] : 1.f;
  const bool skip = !(gs == gs) || fabsf(gs) == INFINITY;  // non-finite norm: leave everything untouched
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    const int cnt = (int)((n - i) < 4 ? (n - i) : 4);
    float pv[4], gv[4], mv[4], vv[4];
    if (cnt == 4) {
      // (non-temporal: 2.7 GB stream through once per step — master weights, both moments, the gradient — and should not push the
      // 157 MB of bf16 weights written below, which the next forward pass reads, out of the 256 MB last-level cache)
      const f32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p + i));
      const f32x4_t b = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(g + i));
      const f32x4_t c = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(m + i));
      const f32x4_t d = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(v + i));
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w;
      gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w;
      vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
      for (int e = 0; e < 4; ++e) {
        const bool ok = e < cnt;
        pv[e] = ok ? p[i + e] : 0.f; gv[e] = ok ? g[i + e] : 0.f; mv[e] = ok ? m[i + e] : 0.f; vv[e] = ok ? v[i + e] : 0.f;
      }
    }
    if (!skip) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gg = gv[e] * gs;
        mv[e] = mv[e] * beta1 + (1.f - beta1) * gg;
        vv[e] = vv[e] * beta2 + (1.f - beta2) * gg * gg;
        const float denom = sqrtf(vv[e]) + eps;
        if (weight_decay != 0.f) pv[e] += pv[e] * (-weight_decay * lr);
        pv[e] += -step_size * (mv[e] / denom);
      }
    }
    if (cnt == 4) {
      __builtin_nontemporal_store((f32x4_t){pv[0], pv[1], pv[2], pv[3]}, reinterpret_cast<f32x4_t*>(p + i));
      __builtin_nontemporal_store((f32x4_t){mv[0], mv[1], mv[2], mv[3]}, reinterpret_cast<f32x4_t*>(m + i));
      __builtin_nontemporal_store((f32x4_t){vv[0], vv[1], vv[2], vv[3]}, reinterpret_cast<f32x4_t*>(v + i));
      if (zero_grad) __builtin_nontemporal_store((f32x4_t){0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f32x4_t*>(g + i));
      if (p_bf16) *reinterpret_cast<uint2*>(p_bf16 + i) = make_uint2(pack_bf2(pv[0], pv[1]), pack_bf2(pv[2], pv[3]));
    } else {
      for (int e = 0; e < cnt; ++e) {
        p[i + e] = pv[e]; m[i + e] = mv[e]; v[i + e] = vv[e];
        if (zero_grad) g[i + e] = 0.f;
        if (p_bf16) p_bf16[i + e] = f2bf(pv[e]);
      }
    }
  }
}

}  // namespace

extern "C" int ea_grad_sumsq(const float* g, long n, float* out /*zeroed by caller*/, hipStream_t stream) {
  if (n <= 0) return 0;
  long blocks = (n + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g, n, out);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_clip_coef(const float* sumsq, float pre_scale, const float* denom_dev, float max_norm,
                            float* coef /*[2]*/, hipStream_t stream) {
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, stream, sumsq, pre_scale, denom_dev, max_norm, coef);
  return EA_CHECK_LAUNCH();
}
extern "C" int ea_adam_step(float* p, float* g, float* m, float* v, void* p_bf16, long n, const float* coef,
                            float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                            int zero_grad, hipStream_t stream) {
  if (n <= 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
  long blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, n, coef, lr,
                     beta1, beta2, eps, weight_decay, step_size, zero_grad);
  return EA_CHECK_LAUNCH();
}
