// Incremental-decoding kernels for batched beam search on gfx950 (all HBM-bound).
//
// Reference path: fairseq/sequence_generator.py:355-609 (per-step loop), fairseq/search.py:103-144 (BeamSearch.step:
// add previous scores, topk(2*beam) over beam*V), fairseq/modules/multihead_attention.py:716-760 (KV cache append),
// :878-897 (beam-deduplicated encoder K/V), :964-989 (reorder_incremental_state = index_select of every cache).
//
//  ea_decode_attention   one query per hypothesis against a cached K/V (self-attention over the L tokens decoded so
//                        far, or cross-attention over the S encoder frames of the hypothesis' SENTENCE — encoder K/V are
//                        stored once per sentence and addressed through `kv_row`, the beam-aware dedup of the reference).
//  ea_kv_append_reorder  fuses "reorder caches by the surviving beams" with "append this step's K/V": new_cache[n] =
//                        concat(old_cache[parent[n]][0:L], kv_new[n]) — one read + one write of the cache per step.
//  ea_beam_mask_rows     NaN -> -inf, never-pad, unk penalty, max-len / min-len / eos_factor rules (sequence_generator.py:395-424)
//  ea_beam_topk          per sentence: top-k over beam*V of (lprobs + cumulative score), k <= 128 — search.py:117-141
#include "common.h"
#include "espresso_amd.h"

namespace {

// q: [N][C] bf16 (already scaled); K,V: row r = kv_row[n] (or n), layout [rows][Lmax][ldkv] with head h at column
// koff/voff + h*dh; len[r] valid keys.  out: [N][C] bf16.  One wavefront per (n, h); dh == 64 (one lane per channel)
// or dh <= 64.
__global__ __launch_bounds__(256) void decode_attention_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ K,
                                                               const bf16_t* __restrict__ V, const int* __restrict__ kv_row,
                                                               const int* __restrict__ len, bf16_t* __restrict__ out, int N, int H,
                                                               int dh, long ldq, long row_stride, long ldkv, int koff, int voff,
                                                               int fixed_len) {
  extern __shared__ float sprob[];  // [4 waves][Lpad]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave;
  if (gw >= N * H) return;
  const int n = gw / H, h = gw % H;
  const int r = kv_row ? kv_row[n] : n;
  const int L = len ? len[r] : fixed_len;
  float* pr = sprob + (long)wave * ((fixed_len + 63) / 64 * 64);
  const float qd = lane < dh ? bf2f(q[(long)n * ldq + h * dh + lane]) : 0.f;
  const bf16_t* Kb = K + (long)r * row_stride + koff + h * dh;
  const bf16_t* Vb = V + (long)r * row_stride + voff + h * dh;
  // scores: lanes cooperate on one key at a time (64 channels -> butterfly sum); keys are walked 4 at a time for ILP
  float mx = -INFINITY;
  for (int j = 0; j < L; ++j) {
    float p = lane < dh ? qd * bf2f(Kb[(long)j * ldkv + lane]) : 0.f;
    p = wave_sum(p);
    if (lane == 0) pr[j] = p;
    mx = fmaxf(mx, p);
  }
  __builtin_amdgcn_wave_barrier();
  float sum = 0.f;
  for (int j = lane; j < L; j += 64) {
    const float e = __expf(pr[j] - mx);
    pr[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_wave_barrier();
  float acc = 0.f;
  if (lane < dh)
    for (int j = 0; j < L; ++j) acc += pr[j] * bf2f(Vb[(long)j * ldkv + lane]);
  if (lane < dh) out[(long)n * ldq + h * dh + lane] = f2bf(acc / sum);
}

// The same for head dims 16 / 32 / 64 with 16-byte aligned rows (every recipe): the kernel above moves ONE bf16 per lane and key
// and folds 64 lanes per key (six shuffles inside a serial loop over the keys: 183 us per call in the config-5 decode block, 59 %
// of its kernel time — round 5 trace).  Here a wavefront takes 16 keys per trip: lane = (key group l >> 2, channel quarter l & 3)
// loads DH / 4 channels of its key in one 8 / 16 / 32-byte piece (a key's row = 4 neighbouring lanes: whole 128-byte lines), two
// shuffle steps finish a score, and the P·V pass accumulates DH / 4 channels per lane over its keys (four shuffle steps at the end).
template <int DH>
__global__ __launch_bounds__(256) void decode_attention_vec_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ K,
                                                                   const bf16_t* __restrict__ V, const int* __restrict__ kv_row,
                                                                   const int* __restrict__ len, bf16_t* __restrict__ out, int N, int H,
                                                                   long ldq, long row_stride, long ldkv, int koff, int voff, int fixed_len) {
  constexpr int E = DH / 4;  // channels per lane
  extern __shared__ float sprob[];  // [4 waves][Lpad]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave;
  if (gw >= N * H) return;
  const int n = gw / H, h = gw % H;
  const int r = kv_row ? kv_row[n] : n;
  const int L = len ? len[r] : fixed_len;
  float* pr = sprob + (long)wave * ((fixed_len + 63) / 64 * 64);
  const int kg = lane >> 2, cq = lane & 3;
  auto load_e = [&](const bf16_t* p, float (&o)[E]) {
    if constexpr (E == 4) {
      const uint2 u = *reinterpret_cast<const uint2*>(p);
      o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
      o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
    } else {
#pragma unroll
      for (int c = 0; c < E / 8; ++c) {
        const uint4 u = *reinterpret_cast<const uint4*>(p + c * 8);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[c * 8 + 2 * e] = __uint_as_float(w[e] << 16);
          o[c * 8 + 2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
        }
      }
    }
  };
  float qv[E];
  load_e(q + (long)n * ldq + h * DH + cq * E, qv);
  const bf16_t* Kb = K + (long)r * row_stride + koff + h * DH + cq * E;
  const bf16_t* Vb = V + (long)r * row_stride + voff + h * DH + cq * E;
  float mx = -INFINITY;
  for (int j0 = 0; j0 < L; j0 += 32) {  // two key groups per trip: both loads are in flight before the first dot product
    const int ja = j0 + kg, jb = ja + 16;
    float ka[E], kb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) ka[e] = kb[e] = 0.f;
    if (ja < L) load_e(Kb + (long)ja * ldkv, ka);
    if (jb < L) load_e(Kb + (long)jb * ldkv, kb);
    float pa = 0.f, pb = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      pa += qv[e] * ka[e];
      pb += qv[e] * kb[e];
    }
    pa += __shfl_xor(pa, 1, 64);
    pb += __shfl_xor(pb, 1, 64);
    pa += __shfl_xor(pa, 2, 64);
    pb += __shfl_xor(pb, 2, 64);
    if (ja < L) {
      if (cq == 0) pr[ja] = pa;
      mx = fmaxf(mx, pa);
    }
    if (jb < L) {
      if (cq == 0) pr[jb] = pb;
      mx = fmaxf(mx, pb);
    }
  }
  mx = wave_max(mx);
  __builtin_amdgcn_wave_barrier();
  float sum = 0.f;
  for (int j = lane; j < L; j += 64) {
    const float e = __expf(pr[j] - mx);
    pr[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_wave_barrier();
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = 0.f;
#pragma unroll 4
  for (int j0 = 0; j0 < L; j0 += 16) {
    const int j = j0 + kg;
    if (j < L) {
      float vv[E];
      load_e(Vb + (long)j * ldkv, vv);
      const float pj = pr[j];
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] += pj * vv[e];
    }
  }
#pragma unroll
  for (int m = 4; m < 64; m <<= 1)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] += __shfl_xor(acc[e], m, 64);
  if (kg == 0) {
    const float inv = 1.f / sum;
    bf16_t* o = out + (long)n * ldq + h * DH + cq * E;
#pragma unroll
    for (int e = 0; e < E; e += 2) *reinterpret_cast<uint32_t*>(o + e) = pack_bf2(acc[e] * inv, acc[e + 1] * inv);
  }
}

// new_cache[n][0:L] = old_cache[parent[n]][0:L] ; new_cache[n][L] = kv_new[n]    (row = 2C bf16: k | v)
__global__ __launch_bounds__(256) void kv_append_reorder_kernel(const bf16_t* __restrict__ old_cache, bf16_t* __restrict__ new_cache,
                                                                const bf16_t* __restrict__ kv_new, const int* __restrict__ parent,
                                                                int N, int L, int Lmax, int W /*row width in elements*/) {
  const int n = blockIdx.y;
  const int p = parent ? parent[n] : n;
  const int nch = W >> 3;
  const long total = (long)(L + 1) * nch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i / nch), ch = (int)(i % nch);
    uint4 v;
    if (t < L) v = *reinterpret_cast<const uint4*>(old_cache + ((long)p * Lmax + t) * W + ch * 8);
    else v = *reinterpret_cast<const uint4*>(kv_new + (long)n * W + ch * 8);
    *reinterpret_cast<uint4*>(new_cache + ((long)n * Lmax + t) * W + ch * 8) = v;
  }
}

// in-place row rules on fp32 lprobs [N][V]
__global__ __launch_bounds__(256) void beam_mask_rows_kernel(float* __restrict__ lp, int V, int pad, int unk, int eos, float unk_penalty,
                                                             int only_eos, int forbid_eos, float eos_factor, int use_eos_factor) {
  __shared__ float sm[16];
  float* row = lp + (long)blockIdx.x * V;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < V; c += 256) {
    float v = row[c];
    if (v != v) v = -INFINITY;
    if (c == pad) v = -INFINITY;
    if (c == unk) v -= unk_penalty;
    if (only_eos && c != eos) v = -INFINITY;
    row[c] = v;
    mx = fmaxf(mx, v);
  }
  mx = block_max(mx, sm);
  if (threadIdx.x == 0) {
    float e = row[eos];
    if (!only_eos) {
      if (use_eos_factor && e < eos_factor * mx) e = -INFINITY;
      if (forbid_eos) e = -INFINITY;
    }
    row[eos] = e;
  }
}

// One block per sentence.  cand[s][0..k): k best of lp[(s*beam+b)][v] + prev[s*beam+b] over (b < nbeam_used, v), descending.
constexpr int KMAXC = 128;
__global__ __launch_bounds__(256) void beam_topk_kernel(const float* __restrict__ lp, const float* __restrict__ prev, int V, int beam,
                                                        int nbeam_used, int k, float* __restrict__ cand_score, int* __restrict__ cand_tok,
                                                        int* __restrict__ cand_beam) {
  __shared__ float sv[256];
  __shared__ int si[256];
  __shared__ int taken[KMAXC];
  const int s = blockIdx.x;
  const long total = (long)nbeam_used * V;
  const float* base = lp + (long)s * beam * V;
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY;
    int bi = -1;
    for (long i = threadIdx.x; i < total; i += 256) {
      const int b = (int)(i / V);
      float v = base[i] + (prev ? prev[s * beam + b] : 0.f);
      if (!(v > bv)) continue;
      bool used = false;
      for (int t = 0; t < r; ++t) used |= (taken[t] == (int)i);
      if (!used) { bv = v; bi = (int)i; }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (threadIdx.x < st) {
        const float ov = sv[threadIdx.x + st];
        const int oi = si[threadIdx.x + st];
        const float cv = sv[threadIdx.x];
        const int ci = si[threadIdx.x];
        // larger value wins; on ties the lower flat index (deterministic)
        if (oi >= 0 && (ci < 0 || ov > cv || (ov == cv && oi < ci))) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const int w = si[0];
      taken[r] = w;
      cand_score[(long)s * k + r] = w >= 0 ? sv[0] : -INFINITY;
      cand_tok[(long)s * k + r] = w >= 0 ? w % V : 0;
      cand_beam[(long)s * k + r] = w >= 0 ? w / V : 0;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int ea_decode_attention(const void* q, const void* K, const void* V, const int* kv_row, const int* len, void* out, int N,
                                   int H, int dh, long ldq, long row_stride, long ldkv, int koff, int voff, int max_len,
                                   hipStream_t stream) {
  if (N <= 0) return 0;
  if (dh > 64 || max_len <= 0) return -2;
  const size_t lds = (size_t)4 * ((max_len + 63) / 64 * 64) * sizeof(float);
  if (lds > 64 * 1024) return -3;
  static const int vec_on = [] { const char* e = getenv("EA_DECODE_ATTN_VEC"); return e ? atoi(e) : 1; }();  // (diagnostic A/B switch)
  const bool aligned = ((ldq | row_stride | ldkv | koff | voff) & 7) == 0 &&
                       ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (vec_on && aligned && (dh == 64 || dh == 32 || dh == 16)) {
    const dim3 grid((N * H + 3) / 4), block(256);
#define EA_DA(D) hipLaunchKernelGGL(decode_attention_vec_kernel<D>, grid, block, lds, stream, (const bf16_t*)q, (const bf16_t*)K, (const bf16_t*)V, \
                                    kv_row, len, (bf16_t*)out, N, H, ldq, row_stride, ldkv, koff, voff, max_len)
    if (dh == 64) EA_DA(64); else if (dh == 32) EA_DA(32); else EA_DA(16);
#undef EA_DA
    return EA_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(decode_attention_kernel, dim3((N * H + 3) / 4), dim3(256), lds, stream, (const bf16_t*)q, (const bf16_t*)K,
                     (const bf16_t*)V, kv_row, len, (bf16_t*)out, N, H, dh, ldq, row_stride, ldkv, koff, voff, max_len);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_kv_append_reorder(const void* old_cache, void* new_cache, const void* kv_new, const int* parent, int N, int L,
                                    int Lmax, int W, hipStream_t stream) {
  if (N <= 0) return 0;
  if (W % 8 || L + 1 > Lmax) return -2;
  int gx = (int)(((long)(L + 1) * (W / 8) + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(kv_append_reorder_kernel, dim3(gx, N), dim3(256), 0, stream, (const bf16_t*)old_cache, (bf16_t*)new_cache,
                     (const bf16_t*)kv_new, parent, N, L, Lmax, W);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_beam_mask_rows(float* lprobs, int N, int V, int pad, int unk, int eos, float unk_penalty, int only_eos,
                                 int forbid_eos, float eos_factor, int use_eos_factor, hipStream_t stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(beam_mask_rows_kernel, dim3(N), dim3(256), 0, stream, lprobs, V, pad, unk, eos, unk_penalty, only_eos, forbid_eos,
                     eos_factor, use_eos_factor);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_beam_topk(const float* lprobs, const float* prev_scores, int bsz, int beam, int nbeam_used, int V, int k,
                            float* cand_score, int* cand_tok, int* cand_beam, hipStream_t stream) {
  if (bsz <= 0) return 0;
  if (k > KMAXC || k <= 0) return -2;
  hipLaunchKernelGGL(beam_topk_kernel, dim3(bsz), dim3(256), 0, stream, lprobs, prev_scores, V, beam, nbeam_used, k, cand_score,
                     cand_tok, cand_beam);
  return EA_CHECK_LAUNCH();
}
