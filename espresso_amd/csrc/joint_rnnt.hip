// Transducer joint: vocabulary projection FUSED with the RNN-T loss for gfx950 (round 6) — the (B, T', U+1, V) logits never
// reach HBM.
//
// Reference: espresso/models/transformer/speech_transformer_transducer_base.py:276-299 (fc_out on relu(E + D)) feeding
// espresso/criterions/transducer_loss.py:130-140 (torchaudio.functional.rnnt_loss with fused_log_softmax).  The unfused path
// (csrc/rnnt.hip) writes the logits as bf16 [n][V] (708 MB at the recipe's 70 000 lattice nodes), reads them back for the
// log-sum-exp and once more for the gradient, each logit rounded to bf16 in between (a last bit of a logit of magnitude 4 - 8 is
// 1.5 - 3 % of exp()).  Here
//   forward   logits tile = Z W^T + bias on the 8-wavefront 256 x 256 GEMM of gemm_w8.hip; the epilogue keeps, per lattice node
//             and vocabulary tile, (max, sum of exp) in fp32 plus the two logits the recursion needs (blank, next label); a small
//             kernel folds the tiles into lse / log p(blank) / log p(label); alpha / beta by rnnt_scan_kernel (csrc/rnnt.hip).
//   backward  the same product again; the epilogue turns each fp32 logit into the loss gradient
//             exp(z - lse + occupancy) - [v = blank] c_blank - [v = label] c_label  (csrc/rnnt.hip, rnnt_grad_kernel) and stores
//             it as bf16 [n][pitch] (pad columns zero): the operand of the joint's data- and weight-gradient GEMMs.
// One more GEMM pass (0.36 TFLOP at the recipe size) instead of 2.1 GB of HBM traffic and two bf16 roundings of every logit.
//
// The k loop is gemm_w8_kernel<256, 256, 2, 4, 2, 1> (same ring, same fragment order, same accumulation order along k): the
// fp32 accumulators are bit-identical to what the unfused path rounds to bf16.
#include "common.h"
#include "espresso_amd.h"
#include "gemm_common.h"
#include "gemm_w8_common.h"

extern "C" int ea_rnnt_scan(const float* lpb, const float* lpy, const int* logit_lengths, const int* target_lengths, float* alpha,
                            float* beta, float* loss, int B, int T, int U1, hipStream_t stream);  // csrc/rnnt.hip

namespace {

constexpr int JBM = 256, JBN = 256, JWM = 2, JWN = 4, JNST = 2;
constexpr int JTM = JBM / JWM, JTN = JBN / JWN, JMI = JTM / 16, JNJ = JTN / 16;  // 128 x 64 per wavefront: 8 x 4 MFMA tiles

struct JointArgs {
  const bf16_t* Z;     // [M][K] lattice nodes x joint dim (k-contiguous)
  const bf16_t* W;     // [V][K] output layer (k-contiguous)
  const float* bias;   // [V] or null
  int M, V, K;
  int blank;
  // forward
  const int* ycol;     // [M] next label of the node, -1: none (u = U_b) or node outside the utterance's lattice
  float2* part;        // [M][tiles_n] (max, sum exp) of the node's logits inside one vocabulary tile
  float* lpb;          // [M] raw logit of blank   (the fold kernel subtracts lse)
  float* lpy;          // [M] raw logit of the next label
  // backward
  const float4* rowc;  // [M] (occupancy - lse, c_blank, c_label, label as int bits)
  bf16_t* dl;          // [M][ld] gradient of the logits, columns V .. ld - 1 zero
  long ld;
  float scale;
  const float* scale_dev;
};

enum { J_LSE = 0, J_GRAD = 1 };

// butterfly steps across the four 16-lane groups of a wavefront on the VALU (v_permlane16_swap / v_permlane32_swap, lane semantics
// pinned on hardware by tools/probes/permlane_probe.hip: swap(x, x) returns {rows 0,0,2,2 ; rows 1,1,3,3} resp. {lower half twice ;
// upper half twice}) — __shfl_xor goes through ds_bpermute: an LDS round trip per step, 32 of them per lane in this epilogue
__device__ __forceinline__ float xor16_max(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int KIND>
__global__ __launch_bounds__(512, 2) void joint_rnnt_kernel(const JointArgs a, const int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  constexpr int A_BYTES = JBM * 128, STAGE = (JBM + JBN) * 128;
  constexpr int NA = JBM / 64, NB = JBN / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / JWN, wn = wave % JWN;
  int tile = blockIdx.x;
  {  // XCD-aware order (gemm_w8_kernel, flags bit 0): the vocabulary tiles of a row block share Z's rows in ONE L2
    const int total = gridDim.x, xcd = tile & 7, q = total >> 3, r = total & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (tile >> 3);
  }
  const int tile_y = tile / tiles_n, tile_x = tile - tile_y * tiles_n;
  const int m0 = tile_y * JBM, n0 = tile_x * JBN;
  const int nk = a.K / BK;

  f32x4_t acc[JMI][JNJ];
#pragma unroll
  for (int i = 0; i < JMI; ++i)
#pragma unroll
    for (int j = 0; j < JNJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- everything the epilogue reads from memory is fetched HERE, before the k loop, into 6 KB of LDS behind the ring: one
  // 512-thread workgroup owns the CU, so a load issued in the epilogue is a round trip nothing overlaps (measured: label / constant
  // loads inside the row loop made the two kernels 1005 / 910 us per launch, against 776 us for the plain product with its 708 MB
  // store); held in registers across the k loop they cost 32 VGPRs and spilled.  The k loop's barriers publish the LDS writes.
  const int g = lane >> 4, r16 = lane & 15;
  const int row0 = wm * JTM, col0 = wn * JTN;
  const int nown = n0 + col0 + g * 4;
  float4* x_rowc = reinterpret_cast<float4*>(dsm + JNST * STAGE);            // [256] (J_GRAD)
  int* x_ycol = reinterpret_cast<int*>(dsm + JNST * STAGE + JBM * 16);       // [256] (J_LSE)
  float* x_bias = reinterpret_cast<float*>(dsm + JNST * STAGE + JBM * 20);   // [256]
  if (tid < JBM) {
    const int m = m0 + tid;
    if constexpr (KIND == J_LSE) x_ycol[tid] = m < a.M ? a.ycol[m] : -1;
    else x_rowc[tid] = m < a.M ? a.rowc[m] : make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));
  } else {
    const int c = n0 + tid - JBM;
    x_bias[tid - JBM] = (a.bias && c < a.V) ? a.bias[c] : 0.f;
  }
  float scale = a.scale;
  if (KIND == J_GRAD && a.scale_dev) scale *= a.scale_dev[0];

  const int src_chunk = (lane & 7) ^ (lane >> 3);
  const bf16_t* ap[NA];
  const bf16_t* bp[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) ap[i] = a.Z + (long)min(m0 + (wave + 8 * i) * 8 + (lane >> 3), a.M - 1) * a.K + src_chunk * 8;
#pragma unroll
  for (int i = 0; i < NB; ++i) bp[i] = a.W + (long)min(n0 + (wave + 8 * i) * 8 + (lane >> 3), a.V - 1) * a.K + src_chunk * 8;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue = [&](int stage, int kt) {
    char* base = dsm + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < NA; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(ap[i] + (long)kt * BK), (lptr_t)(base + i * 8192), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(bp[i] + (long)kt * BK), (lptr_t)(base + A_BYTES + i * 8192), 16, 0, 0);
  };
  uint32_t a_base[2], b_base[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_base[ks] = w8_off(wm * JTM + (lane & 15), ks * 4 + (lane >> 4));
    b_base[ks] = A_BYTES + w8_off(wn * JTN + (lane & 15), ks * 4 + (lane >> 4));
  }
#pragma unroll
  for (int s = 0; s < JNST - 1; ++s)
    if (s < nk) issue(s, s);
  int stage = 0, fill = JNST - 1;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + JNST - 2 < nk) w8_wait_vmcnt<(JNST - 2) * (NA + NB)>();
    else w8_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + JNST - 1 < nk) issue(fill, kt + JNST - 1);
    const char* st = dsm + stage * STAGE;
    constexpr int AG = 2, GPK = JMI / AG, NG = 2 * GPK;
    bf16x8_t bq[2][JNJ], aq[2][AG];
#pragma unroll
    for (int j = 0; j < JNJ; ++j) bq[0][j] = *reinterpret_cast<const bf16x8_t*>(st + b_base[0] + j * 2048);
#pragma unroll
    for (int q = 0; q < AG; ++q) aq[0][q] = *reinterpret_cast<const bf16x8_t*>(st + a_base[0] + q * 2048);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) {
        const int ks1 = (g + 1) / GPK, gi = (g + 1) % GPK;
        if (gi == 0) {
#pragma unroll
          for (int j = 0; j < JNJ; ++j) bq[ks1][j] = *reinterpret_cast<const bf16x8_t*>(st + b_base[ks1] + j * 2048);
        }
#pragma unroll
        for (int q = 0; q < AG; ++q) aq[(g + 1) & 1][q] = *reinterpret_cast<const bf16x8_t*>(st + a_base[ks1] + (gi * AG + q) * 2048);
      }
      const int ks = g / GPK, gi0 = g % GPK;
#pragma unroll
      for (int q = 0; q < AG; ++q)
#pragma unroll
        for (int j = 0; j < JNJ; ++j)
          acc[gi0 * AG + q][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, bq[ks][j]),
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, aq[g & 1][q]), acc[gi0 * AG + q][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, JNJ + AG, 0);
    w8_sched_groups<NG, GPK, AG, JNJ>();
    stage = stage + 1 == JNST ? 0 : stage + 1;
    fill = fill + 1 == JNST ? 0 : fill + 1;
  }

  // ---- epilogue from registers: acc[i][j][e] = logit[m0 + row0 + i*16 + (lane & 15)][n0 + col0 + j*16 + (lane >> 4)*4 + e] - bias ----
  __syncthreads();  // every wavefront is done with the ring (J_LSE reuses its first 8 KB); the constants written above are visible
  float bias4[JNJ][4];
#pragma unroll
  for (int j = 0; j < JNJ; ++j) {
    const float4 b = *reinterpret_cast<const float4*>(x_bias + col0 + j * 16 + g * 4);
    bias4[j][0] = b.x; bias4[j][1] = b.y; bias4[j][2] = b.z; bias4[j][3] = b.w;
  }
  // EDGE: the tile holds columns >= V (the last vocabulary tile only): everything else runs without a per-element column test
  const bool edge = n0 + JBN > a.V;
  const int slab0 = n0 + col0;                                             // first column of this wavefront's 64
  const bool blank_here = (unsigned)(a.blank - slab0) < (unsigned)JTN;     // (uniform per wavefront)
  if constexpr (KIND == J_LSE) {
    int ycs[JMI];
#pragma unroll
    for (int i = 0; i < JMI; ++i) ycs[i] = x_ycol[row0 + i * 16 + r16];
    float2* tab = reinterpret_cast<float2*>(dsm);  // [256 rows][4 column waves] partial (max, sum) table
#pragma unroll
    for (int i = 0; i < JMI; ++i) {
      const int m = m0 + row0 + i * 16 + r16;
      const int yc = ycs[i];
      float x[JNJ][4];
#pragma unroll
      for (int j = 0; j < JNJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[j][e] = acc[i][j][e] + bias4[j][e];
      // the two logits the recursion needs: only the wavefront whose 64 columns hold blank / the row's label looks for them
      if (m < a.M && (blank_here || (unsigned)(yc - slab0) < (unsigned)JTN)) {
#pragma unroll
        for (int j = 0; j < JNJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = nown + j * 16 + e;
            if (c == a.blank) a.lpb[m] = x[j][e];
            if (c == yc) a.lpy[m] = x[j][e];
          }
      }
      if (edge) {
#pragma unroll
        for (int j = 0; j < JNJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nown + j * 16 + e >= a.V) x[j][e] = -INFINITY;
      }
      float mx = x[0][0];
#pragma unroll
      for (int j = 0; j < JNJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, x[j][e]);
      mx = xor32_max(xor16_max(mx));  // over the four 16-lane groups that share the row
      const float ms = mx > -INFINITY ? mx : 0.f;  // (a slab wholly past V: every term below is exp(-inf) = 0)
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < JNJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += __expf(x[j][e] - ms);
      s = xor32_sum(xor16_sum(s));
      if (g == 0) tab[(row0 + i * 16 + r16) * JWN + wn] = make_float2(mx, s);
    }
    __syncthreads();
    if (tid < JBM && m0 + tid < a.M) {
      float mx = -INFINITY;
#pragma unroll
      for (int w = 0; w < JWN; ++w) mx = fmaxf(mx, tab[tid * JWN + w].x);
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < JWN; ++w) {
        const float2 p = tab[tid * JWN + w];
        if (p.x > -INFINITY) s += p.y * __expf(p.x - mx);
      }
      a.part[(long)(m0 + tid) * tiles_n + tile_x] = make_float2(mx, s);
    }
  } else {
    const int npair = n0 + col0 + (g & 1) * 16 + (g >> 1) * 8;
#pragma unroll
    for (int i = 0; i < JMI; ++i) {
      const int m = m0 + row0 + i * 16 + r16;
      const bool mv = m < a.M;
      const float4 rc = x_rowc[row0 + i * 16 + r16];
      const int y = __float_as_int(rc.w);
      float v[JNJ][4];
#pragma unroll
      for (int j = 0; j < JNJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = __expf(acc[i][j][e] + bias4[j][e] + rc.x);
      if (blank_here || (unsigned)(y - slab0) < (unsigned)JTN) {  // the two columns with a subtracted term: one slab in 80
#pragma unroll
        for (int j = 0; j < JNJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = nown + j * 16 + e;
            if (c == a.blank) v[j][e] -= rc.y;
            if (c == y) v[j][e] -= rc.z;
          }
      }
      if (edge) {
#pragma unroll
        for (int j = 0; j < JNJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nown + j * 16 + e >= a.V) v[j][e] = 0.f;
      }
#pragma unroll
      for (int jp = 0; jp < JNJ / 2; ++jp) {
        if (edge && n0 + col0 + jp * 32 >= (int)a.ld) continue;  // (uniform per wavefront: whole 32-column groups past the row pitch)
        const uint2 o0 = make_uint2(pack_bf2(v[2 * jp][0] * scale, v[2 * jp][1] * scale), pack_bf2(v[2 * jp][2] * scale, v[2 * jp][3] * scale));
        const uint2 o1 = make_uint2(pack_bf2(v[2 * jp + 1][0] * scale, v[2 * jp + 1][1] * scale),
                                    pack_bf2(v[2 * jp + 1][2] * scale, v[2 * jp + 1][3] * scale));
        const u32x4_t o = w8_swap_pair(o0, o1);
        if (mv) w8_store16(a.dl + (long)m * a.ld + npair + jp * 32, o, false);
      }
    }
  }
}

// ycol[node] = label the node may emit next, -1 when it has none (u = U_b) or lies outside its utterance's lattice
__global__ __launch_bounds__(256) void joint_label_kernel(const int* __restrict__ targets, const int* __restrict__ T_len,
                                                          const int* __restrict__ U_len, int* __restrict__ ycol, float* __restrict__ lpy,
                                                          int T, int U1, int Umax, long n) {
  const long node = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= n) return;
  const int u = (int)(node % U1);
  const int t = (int)((node / U1) % T);
  const int b = (int)(node / ((long)U1 * T));
  const bool has = t < T_len[b] && u < U_len[b];
  ycol[node] = has ? targets[(long)b * Umax + u] : -1;
  lpy[node] = -INFINITY;  // (nodes without a label keep it: the GEMM epilogue only writes the others)
}
// folds the vocabulary tiles of a node: lse = log sum_v exp(z_v); the raw logits of blank / label become log-probabilities
__global__ __launch_bounds__(256) void joint_fold_kernel(const float2* __restrict__ part, float* __restrict__ lse, float* __restrict__ lpb,
                                                         float* __restrict__ lpy, int tiles_n, long n) {
  const long node = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= n) return;
  const float2* p = part + node * tiles_n;
  float mx = -INFINITY;
  for (int i = 0; i < tiles_n; ++i) mx = fmaxf(mx, p[i].x);
  float s = 0.f;
  for (int i = 0; i < tiles_n; ++i)
    if (p[i].x > -INFINITY) s += p[i].y * expf(p[i].x - mx);
  const float l = mx + logf(s);
  lse[node] = l;
  lpb[node] -= l;
  lpy[node] -= l;  // (-inf stays -inf)
}
// per-node constants of the gradient epilogue (the arithmetic of rnnt_grad_kernel's preamble, csrc/rnnt.hip)
__global__ __launch_bounds__(256) void joint_coef_kernel(const int* __restrict__ targets, const int* __restrict__ T_len,
                                                         const int* __restrict__ U_len, const float* __restrict__ lse,
                                                         const float* __restrict__ lpb, const float* __restrict__ lpy,
                                                         const float* __restrict__ alpha, const float* __restrict__ beta,
                                                         const float* __restrict__ loss, float4* __restrict__ rowc, int T, int U1,
                                                         int Umax, long n) {
  const long node = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= n) return;
  const int u = (int)(node % U1);
  const int t = (int)((node / U1) % T);
  const int b = (int)(node / ((long)U1 * T));
  const int Tb = T_len[b], Ub = U_len[b];
  const float L = loss[b];
  if (t >= Tb || u > Ub || !(L < INFINITY)) {
    rowc[node] = make_float4(-INFINITY, 0.f, 0.f, __int_as_float(-1));  // exp(z - inf) = 0: a zero gradient row
    return;
  }
  const float al = alpha[node];
  const float occ = al + beta[node] + L;
  float cb, cy = 0.f;
  if (t == Tb - 1 && u == Ub) cb = expf(al + lpb[node] + L);
  else cb = t < Tb - 1 ? expf(al + lpb[node] + beta[node + U1] + L) : 0.f;
  int y = -1;
  if (u < Ub) {
    y = targets[(long)b * Umax + u];
    cy = expf(al + lpy[node] + beta[node + 1] + L);
  }
  rowc[node] = make_float4(occ - lse[node], cb, cy, __int_as_float(y));
}

inline long align256(long x) { return (x + 255) & ~255L; }
struct JointWs {
  float *lse, *lpb, *lpy, *alpha, *beta;
  float2* part;
  int* ycol;
  float4* rowc;
  long bytes;
};
JointWs joint_ws(void* base, long n, int tiles_n) {
  JointWs w;
  char* p = (char*)base;
  long off = 0;
  auto take = [&](long nbytes) { char* q = p ? p + off : nullptr; off += align256(nbytes); return q; };
  w.lse = (float*)take(n * 4); w.lpb = (float*)take(n * 4); w.lpy = (float*)take(n * 4);
  w.alpha = (float*)take(n * 4); w.beta = (float*)take(n * 4);
  w.part = (float2*)take(n * tiles_n * 8);
  w.ycol = (int*)take(n * 4);
  w.rowc = (float4*)take(n * 16);
  w.bytes = off;
  return w;
}
bool joint_shape_ok(const void* Z, const void* W, long n, int V, int J) {
  return n > 0 && V > 0 && J > 0 && J % BK == 0 && n * (long)J < (1L << 31) && (long)V * J < (1L << 31) &&
         ((reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
}
template <int KIND>
bool joint_launch(const JointArgs& a, hipStream_t stream) {
  constexpr int bytes = JNST * (JBM + JBN) * 128 + JBM * 24;  // ring + [rowc | ycol | bias] of the tile
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_rnnt_kernel<KIND>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!attr_ok) return false;
  const int tiles_n = (a.V + JBN - 1) / JBN, tiles_m = (a.M + JBM - 1) / JBM;
  hipLaunchKernelGGL((joint_rnnt_kernel<KIND>), dim3(tiles_n * tiles_m), dim3(512), bytes, stream, a, tiles_n);
  return true;
}

}  // namespace

extern "C" long ea_joint_rnnt_workspace_bytes(int B, int T, int U1, int V) {
  const long n = (long)B * T * U1;
  return joint_ws(nullptr, n, (V + JBN - 1) / JBN).bytes;
}

extern "C" int ea_joint_rnnt_loss(const void* Z, const void* W, const float* bias, const int* targets, const int* logit_lengths,
                                  const int* target_lengths, float* loss, void* workspace, int B, int T, int U1, int V, int J,
                                  int Umax, int blank, hipStream_t stream) {
  if (B <= 0) return 0;
  const long n = (long)B * T * U1;
  if (T <= 0 || U1 <= 0 || U1 > 512 || blank < 0 || blank >= V || !joint_shape_ok(Z, W, n, V, J)) return -2;
  const int tiles_n = (V + JBN - 1) / JBN;
  const JointWs w = joint_ws(workspace, n, tiles_n);
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(joint_label_kernel, dim3(nb), dim3(256), 0, stream, targets, logit_lengths, target_lengths, w.ycol, w.lpy, T, U1, Umax, n);
  JointArgs a{};
  a.Z = (const bf16_t*)Z; a.W = (const bf16_t*)W; a.bias = bias; a.M = (int)n; a.V = V; a.K = J; a.blank = blank;
  a.ycol = w.ycol; a.part = w.part; a.lpb = w.lpb; a.lpy = w.lpy;
  if (!joint_launch<J_LSE>(a, stream)) return -1;
  hipLaunchKernelGGL(joint_fold_kernel, dim3(nb), dim3(256), 0, stream, (const float2*)w.part, w.lse, w.lpb, w.lpy, tiles_n, n);
  if (EA_CHECK_LAUNCH() != 0) return -1;
  return ea_rnnt_scan(w.lpb, w.lpy, logit_lengths, target_lengths, w.alpha, w.beta, loss, B, T, U1, stream);
}

extern "C" int ea_joint_rnnt_grad(const void* Z, const void* W, const float* bias, const int* targets, const int* logit_lengths,
                                  const int* target_lengths, const float* loss, void* workspace, void* dl, long ld, int B, int T,
                                  int U1, int V, int J, int Umax, int blank, float grad_scale, const float* grad_scale_dev,
                                  hipStream_t stream) {
  if (B <= 0) return 0;
  const long n = (long)B * T * U1;
  if (!joint_shape_ok(Z, W, n, V, J) || ld < V || ld % 32 != 0 || (reinterpret_cast<uintptr_t>(dl) & 15) != 0 || n * ld >= (1L << 40)) return -2;
  const int tiles_n = (V + JBN - 1) / JBN;
  if ((long)tiles_n * JBN < ld) return -2;  // (the pad columns up to the pitch are written by the last vocabulary tile)
  const JointWs w = joint_ws(workspace, n, tiles_n);
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(joint_coef_kernel, dim3(nb), dim3(256), 0, stream, targets, logit_lengths, target_lengths, (const float*)w.lse,
                     (const float*)w.lpb, (const float*)w.lpy, (const float*)w.alpha, (const float*)w.beta, loss, w.rowc, T, U1, Umax, n);
  JointArgs a{};
  a.Z = (const bf16_t*)Z; a.W = (const bf16_t*)W; a.bias = bias; a.M = (int)n; a.V = V; a.K = J; a.blank = blank;
  a.rowc = w.rowc; a.dl = (bf16_t*)dl; a.ld = ld; a.scale = grad_scale; a.scale_dev = grad_scale_dev;
  if (!joint_launch<J_GRAD>(a, stream)) return -1;
  return EA_CHECK_LAUNCH();
}
