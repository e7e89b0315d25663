// Fused multi-head attention for gfx950: scores, (relative-position skew), masking, softmax, attention dropout and
// P·V in ONE kernel — the (B·H, T, S) score tensors of the reference never touch HBM.
//
// Reference semantics: fairseq/modules/multihead_attention.py:679-688 (q + pos_bias_u / pos_bias_v, scaling),
// :788-831 (content logits q_u k^T plus positional logits q_v p^T read through the as_strided "skew"
// pos[i][j] = raw[i][(T-1) - i + j]), :835-867 (key-padding -inf, fp32 softmax), :874 (dropout on the probabilities),
// :884-907 (probabilities x values).  Also serves the plain (absolute-position) encoder self-attention, the causal
// decoder self-attention and the encoder-decoder cross-attention of espresso/models/transformer/ (qv = NULL).
//
// Layout of the work (dh = 64):
//   grid = ceil(T/64) x (H*B) workgroups of 4 wavefronts; wavefront w owns 16 query rows and walks the keys 64 at a time
//   (online softmax).  Everything is computed TRANSPOSED so that a lane always owns query column i = lane&15:
//     S^T[j][i]  = sum_d K[j][d] Qu[i][d]      mfma 16x16x32, A = K tile from LDS, B = Qu fragment held in registers
//     BD^T[c][i] = sum_d PP[c][d] Qv[i][d]     same, A = window of the projected positional table from LDS; the 16 query
//                                               rows of a wavefront and 64 keys only touch 79 consecutive relative
//                                               positions (5 MFMA tiles instead of the full 2T-1 wide product)
//     skew       : S^T[j][i] += BD^T[15 - i_w + j_l][i_w]   (per-wavefront LDS bounce; an index transform, no data
//                                               movement in HBM)
//     softmax    : column statistics are in-lane reductions over 16 registers + 2 cross-lane steps (xor 16, 32)
//     O^T[d][i]  = sum_j V^T[d][j] P^T[j][i]   mfma 16x16x16: the C-layout of S^T (4 consecutive j per lane) IS the
//                                               B-operand layout, so the probabilities never leave the registers
//   The per-row logsumexp is written for the backward pass, which recomputes the probabilities from it.
// XCD-aware launch: the linear workgroup id is remapped so that all query tiles of one (head, sentence) run on the same
// XCD and share its L2 copy of K / V / PP.
#include "common.h"
#include "espresso_amd.h"

namespace {

typedef short bf16x4_t __attribute__((ext_vector_type(4)));

constexpr int DH = 64;           // head dim
constexpr int TQ = 64, TK = 64;  // query rows per workgroup, keys per step
constexpr int ROWB = DH * 2;     // 128-byte LDS rows
constexpr int BDP = 18;          // pitch (floats) of the per-wavefront BD^T bounce buffer [80][BDP]

// k-contiguous [row][64] bf16 tile: 16-byte chunk c of row r at chunk c ^ (r & 7)   (conflict-free ds_read_b128)
__device__ __forceinline__ uint32_t off16(int row, int chunk) { return (uint32_t)(row * ROWB + ((chunk ^ (row & 7)) << 4)); }
// transposed [d][64 j] tile read 8 bytes at a time: 8-byte chunk c of row d at chunk c ^ (d & 15)
__device__ __forceinline__ uint32_t off8(int row, int chunk) { return (uint32_t)(row * ROWB + ((chunk ^ (row & 15)) << 3)); }

struct FlashFwdArgs {
  const bf16_t* qu; const bf16_t* qv; long ldq;
  const bf16_t* k; const bf16_t* v; long ldkv;
  const bf16_t* pp; long ldpp;
  const int* klen;
  bf16_t* out; long ldo;
  float* lse;
  int H, B, T, S, causal, nq;
  uint64_t seed; uint32_t thr; float inv_keep;
};

// bijective XCD-aware remap of a 1-D grid (8 XCDs, round-robin dispatch): consecutive virtual ids stay on one XCD
__device__ __forceinline__ int xcd_remap(int id, int total) {
  const int xcd = id & 7, q = total >> 3, r = total & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// stage rows [row0, row0+nrows) (nrows multiple of 32) of a k-contiguous source into `s`; rows outside [0, limit) are zero
__device__ __forceinline__ void stage_rows(char* s, const bf16_t* src, long ld, int row0, int limit, int nrows, int tid) {
  const int c = tid & 7;
  for (int rr = tid >> 3; rr < nrows; rr += 32) {
    const int g = row0 + rr;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (g >= 0 && g < limit) val = *reinterpret_cast<const uint4*>(src + (long)g * ld + c * 8);
    *reinterpret_cast<uint4*>(s + off16(rr, c)) = val;
  }
}
// stage V rows [j0, j0+64) transposed: s[d][j]; threads 0..127 each own a 4(j) x 8(d) block
__device__ __forceinline__ void stage_vt(char* s, const bf16_t* src, long ld, int j0, int limit, int tid) {
  if (tid >= 128) return;
  const int jq = tid >> 3, dc = tid & 7;
  uint32_t w[4][4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int g = j0 + jq * 4 + jj;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (g < limit) val = *reinterpret_cast<const uint4*>(src + (long)g * ld + dc * 8);
    w[jj][0] = val.x; w[jj][1] = val.y; w[jj][2] = val.z; w[jj][3] = val.w;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint2 lo, hi;
    lo.x = (w[0][q] & 0xffffu) | (w[1][q] << 16);
    lo.y = (w[2][q] & 0xffffu) | (w[3][q] << 16);
    hi.x = (w[0][q] >> 16) | (w[1][q] & 0xffff0000u);
    hi.y = (w[2][q] >> 16) | (w[3][q] & 0xffff0000u);
    const int d = dc * 8 + 2 * q;
    *reinterpret_cast<uint2*>(s + off8(d, jq)) = lo;
    *reinterpret_cast<uint2*>(s + off8(d + 1, jq)) = hi;
  }
}

__device__ __forceinline__ f32x4_t mfma32(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a),
                                                 __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma16(bf16x4_t a, bf16x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

template <bool RELPOS>
__global__ __launch_bounds__(256, 2) void flash_fwd_kernel(const FlashFwdArgs a) {
  __shared__ __attribute__((aligned(16))) char sK[TK * ROWB];        // [j][d]
  __shared__ __attribute__((aligned(16))) char sVt[DH * ROWB];       // [d][j]
  __shared__ __attribute__((aligned(16))) char sPP[RELPOS ? 128 * ROWB : 16];  // [window row][d]
  __shared__ float sBD[RELPOS ? 4 * 80 * BDP : 1];                   // per wavefront [c'][i_w]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g4 = lane >> 4;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int z = vid / a.nq, qt = vid % a.nq;
  const int h = z / a.B, b = z % a.B;
  const int T = a.T, S = a.S;
  const int i0 = qt * TQ;
  const int i = i0 + 16 * w + li;  // this lane's query row
  const int kl = a.klen ? min(a.klen[b], S) : S;

  // query fragments (B operands): row i, d = ks*32 + g4*8 .. +8
  bf16x8_t qu[2], qv[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 u = make_uint4(0, 0, 0, 0), v = make_uint4(0, 0, 0, 0);
    if (i < T) {
      const long o = ((long)b * T + i) * a.ldq + h * DH + ks * 32 + g4 * 8;
      u = *reinterpret_cast<const uint4*>(a.qu + o);
      if (RELPOS) v = *reinterpret_cast<const uint4*>(a.qv + o);
    }
    qu[ks] = __builtin_bit_cast(bf16x8_t, u);
    qv[ks] = __builtin_bit_cast(bf16x8_t, v);
  }

  f32x4_t acc_o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc_o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const bf16_t* Kb = a.k + (long)b * S * a.ldkv + h * DH;
  const bf16_t* Vb = a.v + (long)b * S * a.ldkv + h * DH;
  const bf16_t* PPb = RELPOS ? a.pp + h * DH : nullptr;
  const int R = 2 * T - 1;
  float* bd = sBD + (RELPOS ? w * 80 * BDP : 0);
  const int c0w = 48 - 16 * w;  // first window row of this wavefront's 80-row band

  int jend = kl;  // keys >= kl are masked for every row
  if (a.causal) jend = min(jend, i0 + TQ + (S - T));
  for (int j0 = 0; j0 < jend; j0 += TK) {
    __syncthreads();
    stage_rows(sK, Kb, a.ldkv, j0, S, TK, tid);
    stage_vt(sVt, Vb, a.ldkv, j0, S, tid);
    if (RELPOS) stage_rows(sPP, PPb, a.ldpp, (T - 1) - (i0 + TQ - 1) + j0, R, 128, tid);
    __syncthreads();

    // content scores (transposed): acc_s[jt][r] = S[i][j0 + jt*16 + g4*4 + r]
    f32x4_t acc_s[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc_s[jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + off16(jt * 16 + li, ks * 4 + g4));
        acc_s[jt] = mfma32(kf, qu[ks], acc_s[jt]);
      }
    if (RELPOS) {
      // positional band: bd[c'][i_w] = Qv[i] . PP[rbase + c0w + c'],  c' in [0, 80)
#pragma unroll
      for (int ct = 0; ct < 5; ++ct) {
        f32x4_t t = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(sPP + off16(c0w + ct * 16 + li, ks * 4 + g4));
          t = mfma32(pf, qv[ks], t);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bd[(ct * 16 + g4 * 4 + r) * BDP + li] = t[r];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc_s[jt][r] += bd[(15 - li + jt * 16 + g4 * 4 + r) * BDP + li];
    }
    // masking + online softmax (column i lives in the 4 lanes {li, li+16, li+32, li+48})
    float tmax = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + jt * 16 + g4 * 4 + r;
        float s = acc_s[jt][r];
        if (j >= kl || (a.causal && j > i + (S - T))) s = -INFINITY;
        acc_s[jt][r] = s;
        tmax = fmaxf(tmax, s);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const bool dead = (m_new == -INFINITY);
    const float alpha = dead ? 1.f : __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = dead ? 0.f : __expf(acc_s[jt][r] - m_new);
        acc_s[jt][r] = p;
        psum += p;
      }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_o[dt][r] *= alpha;

    // O^T += V^T P^T  (dropout applied to the probabilities that multiply V, not to the normaliser)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = acc_s[jt][r];
        if (a.thr) {
          const int j = j0 + jt * 16 + g4 * 4 + r;
          p[r] *= ea_keep(a.seed, ((uint64_t)z * T + (uint64_t)i) * (uint64_t)S + (uint64_t)j, a.thr, a.inv_keep);
        }
      }
      uint2 pk;
      pk.x = pack_bf2(p[0], p[1]);
      pk.y = pack_bf2(p[2], p[3]);
      const bf16x4_t pb = __builtin_bit_cast(bf16x4_t, pk);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x4_t vf = *reinterpret_cast<const bf16x4_t*>(sVt + off8(dt * 16 + li, jt * 4 + g4));
        acc_o[dt] = mfma16(vf, pb, acc_o[dt]);
      }
    }
  }

  if (i < T) {
    const float inv = 1.f / l_run;  // all-masked rows: 0/0 = NaN, exactly like the reference softmax
    bf16_t* o = a.out + ((long)b * T + i) * a.ldo + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 pk;
      pk.x = pack_bf2(acc_o[dt][0] * inv, acc_o[dt][1] * inv);
      pk.y = pack_bf2(acc_o[dt][2] * inv, acc_o[dt][3] * inv);
      *reinterpret_cast<uint2*>(o + dt * 16 + g4 * 4) = pk;
    }
    if (a.lse && g4 == 0) a.lse[(long)z * T + i] = m_run + __logf(l_run);
  }
}

}  // namespace

extern "C" int ea_flash_attention_supported(int dh, int T, int S, int relpos) {
  return dh == DH && T > 0 && S > 0 && (!relpos || T == S);
}

extern "C" int ea_flash_attention_fwd(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldkv,
                                      const void* pp, long ldpp, const int* key_len, void* out, long ldo, float* lse, int H,
                                      int B, int T, int S, int dh, int causal, uint64_t drop_seed, uint32_t drop_thr,
                                      float drop_scale, hipStream_t stream) {
  if (H <= 0 || B <= 0 || T <= 0) return 0;
  const bool relpos = qv != nullptr;
  if (!ea_flash_attention_supported(dh, T, S, relpos) || (relpos && !pp)) return -2;
  if ((ldq | ldkv | ldo) % 8 || (relpos && ldpp % 8)) return -2;
  if ((((uintptr_t)qu | (uintptr_t)qv | (uintptr_t)k | (uintptr_t)v | (uintptr_t)pp) & 15) || ((uintptr_t)out & 7)) return -2;
  FlashFwdArgs a;
  a.qu = (const bf16_t*)qu; a.qv = (const bf16_t*)qv; a.ldq = ldq;
  a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ldkv = ldkv;
  a.pp = (const bf16_t*)pp; a.ldpp = ldpp;
  a.klen = key_len;
  a.out = (bf16_t*)out; a.ldo = ldo;
  a.lse = lse;
  a.H = H; a.B = B; a.T = T; a.S = S; a.causal = causal; a.nq = (T + TQ - 1) / TQ;
  a.seed = drop_seed; a.thr = drop_thr; a.inv_keep = drop_scale;
  const dim3 grid((unsigned)(a.nq * H * B));
  if (relpos) hipLaunchKernelGGL(flash_fwd_kernel<true>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(flash_fwd_kernel<false>, grid, dim3(256), 0, stream, a);
  return EA_CHECK_LAUNCH();
}
