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
#include "flash_internal.h"

namespace {

typedef short bf16x4_t __attribute__((ext_vector_type(4)));

constexpr int DH = 64;           // head dim
constexpr int TQ = 64, TK = 64;  // query rows per workgroup, keys per step
constexpr int ROWB = DH * 2;     // 128-byte LDS rows
constexpr int BDP = 18;          // pitch (floats) of the per-wavefront BD^T bounce buffer [80][BDP]

// k-contiguous [row][64] bf16 tile: 16-byte chunk c of row r at chunk c ^ (r & 7)   (conflict-free ds_read_b128)
__device__ __forceinline__ uint32_t off16(int row, int chunk) { return (uint32_t)(row * ROWB + ((chunk ^ (row & 7)) << 4)); }




// bijective XCD-aware remap of a 1-D grid (8 XCDs, round-robin dispatch): consecutive virtual ids stay on one XCD
__device__ __forceinline__ int xcd_remap(int id, int total) {
  const int xcd = id & 7, q = total >> 3, r = total & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// ---- register-staged tile movers: global -> registers (issued one tile ahead) -> LDS ------------------------
// k-contiguous image: a thread owns NP 16-byte chunks (rows (tid>>3) + 32p, chunk tid&7); rows outside [0, limit) read 0
template <int NP>
struct RowRegs { uint4 v[NP]; };
template <int NP>
__device__ __forceinline__ void load_rows(RowRegs<NP>& r, const bf16_t* src, long ld, int row0, int limit, int tid) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int g = row0 + (tid >> 3) + 32 * p;
    r.v[p] = make_uint4(0, 0, 0, 0);
    if (g >= 0 && g < limit) r.v[p] = *reinterpret_cast<const uint4*>(src + (long)g * ld + (tid & 7) * 8);
  }
}
template <int NP>
__device__ __forceinline__ void store_rows(char* s, const RowRegs<NP>& r, int tid) {
#pragma unroll
  for (int p = 0; p < NP; ++p) *reinterpret_cast<uint4*>(s + off16((tid >> 3) + 32 * p, tid & 7)) = r.v[p];
}
// 4(row) x 8(d) block owned by thread t: rows row0 + (t>>3)*4 + jj, d chunk t&7.  Feeds the k-contiguous image and/or
// the transposed image s[d][row] (8-byte chunk c of row d at c ^ (d & 15); NR rows -> 2*NR-byte LDS rows).
struct BlkRegs { uint4 v[4]; };
__device__ __forceinline__ void load_blk(BlkRegs& r, const bf16_t* src, long ld, int row0, int limit, int t) {
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int g = row0 + (t >> 3) * 4 + jj;
    r.v[jj] = make_uint4(0, 0, 0, 0);
    if (g >= 0 && g < limit) r.v[jj] = *reinterpret_cast<const uint4*>(src + (long)g * ld + (t & 7) * 8);
  }
}
__device__ __forceinline__ void store_blk_rows(char* s, const BlkRegs& r, int t) {
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<uint4*>(s + off16((t >> 3) * 4 + jj, t & 7)) = r.v[jj];
}
template <int NR>
__device__ __forceinline__ void store_blk_t(char* s, const BlkRegs& r, int t) {
  const int rq = t >> 3, dc = t & 7;
  const uint32_t w[4][4] = {{r.v[0].x, r.v[0].y, r.v[0].z, r.v[0].w}, {r.v[1].x, r.v[1].y, r.v[1].z, r.v[1].w},
                            {r.v[2].x, r.v[2].y, r.v[2].z, r.v[2].w}, {r.v[3].x, r.v[3].y, r.v[3].z, r.v[3].w}};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint2 lo, hi;
    lo.x = (w[0][q] & 0xffffu) | (w[1][q] << 16);
    lo.y = (w[2][q] & 0xffffu) | (w[3][q] << 16);
    hi.x = (w[0][q] >> 16) | (w[1][q] & 0xffff0000u);
    hi.y = (w[2][q] >> 16) | (w[3][q] & 0xffff0000u);
    const int d = dc * 8 + 2 * q;
    *reinterpret_cast<uint2*>(s + d * (NR * 2) + ((rq ^ (d & 15)) << 3)) = lo;
    *reinterpret_cast<uint2*>(s + (d + 1) * (NR * 2) + ((rq ^ ((d + 1) & 15)) << 3)) = hi;
  }
}
template <int NR>
__device__ __forceinline__ bf16x4_t read_t(const char* s, int d, int chunk) {
  return *reinterpret_cast<const bf16x4_t*>(s + d * (NR * 2) + ((chunk ^ (d & 15)) << 3));
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
  RowRegs<2> rK;
  BlkRegs rV;
  RowRegs<4> rP;
  auto load_tile = [&](int j0) {
    load_rows<2>(rK, Kb, a.ldkv, j0, S, tid);
    if (tid < 128) load_blk(rV, Vb, a.ldkv, j0, S, tid);
    if (RELPOS) load_rows<4>(rP, PPb, a.ldpp, (T - 1) - (i0 + TQ - 1) + j0, R, tid);
  };
  if (jend > 0) load_tile(0);
  for (int j0 = 0; j0 < jend; j0 += TK) {
    __syncthreads();
    store_rows<2>(sK, rK, tid);
    if (tid < 128) store_blk_t<64>(sVt, rV, tid);
    if (RELPOS) store_rows<4>(sPP, rP, tid);
    __syncthreads();
    if (j0 + TK < jend) load_tile(j0 + TK);  // next tile's global loads fly under this tile's MFMA / softmax

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
        acc_o[dt] = mfma16(read_t<64>(sVt, dt * 16 + li, jt * 4 + g4), pb, acc_o[dt]);
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

// ===========================================================================================================
// Backward.  P is recomputed from the saved row logsumexp; with D_i = sum_d dO[i][d] O[i][d] (= sum_j P dP, dropout
// included) the softmax backward is dS = P * (dP - D_i), no row reductions needed.
//   Q kernel  (grid: query tiles)  -> t1 = scaling * dS K, t2 = scaling * dBD PP (gradients of q+u and q+v), dBD in
//                                     the un-skewed (T x 2T-1) layout for the pos_proj weight gradient, and D.
//   KV kernel (grid: key tiles)    -> dK = dS^T Qu, dV = Pd^T dO accumulated in registers over the query tiles.
// Both recompute the scores on MFMA (cheap next to moving T x S tensors through HBM).

__device__ __forceinline__ bf16x4_t pack4(float a, float b, float c, float d) {
  uint2 pk;
  pk.x = pack_bf2(a, b);
  pk.y = pack_bf2(c, d);
  return __builtin_bit_cast(bf16x4_t, pk);
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


template <bool RELPOS>
__global__ __launch_bounds__(256, 2) void flash_bwd_q_kernel(const FlashBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char sK[TK * ROWB];                   // [j][d]
  __shared__ __attribute__((aligned(16))) char sKt[DH * ROWB];                  // [d][j]
  __shared__ __attribute__((aligned(16))) char sV[TK * ROWB];                   // [j][d]
  __shared__ __attribute__((aligned(16))) char sPP[RELPOS ? 128 * ROWB : 16];   // [window row][d]
  __shared__ __attribute__((aligned(16))) char sPPt[RELPOS ? DH * 256 : 16];    // [d][window row]
  __shared__ float sBD[RELPOS ? 4 * 80 * BDP : 1];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g4 = lane >> 4;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int z = vid / a.nq, qt = vid % a.nq;
  const int h = z / a.B, b = z % a.B;
  const int T = a.T, S = a.S;
  const int i0 = qt * TQ;
  const int i = i0 + 16 * w + li;
  const int kl = a.klen ? min(a.klen[b], S) : S;

  bf16x8_t qu[2], qv[2], dO[2];
  float Di = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 u = make_uint4(0, 0, 0, 0), v = u, g = u, o = u;
    if (i < T) {
      const long oq = ((long)b * T + i) * a.ldq + h * DH + ks * 32 + g4 * 8;
      u = *reinterpret_cast<const uint4*>(a.qu + oq);
      if (RELPOS) v = *reinterpret_cast<const uint4*>(a.qv + oq);
      const long oo = ((long)b * T + i) * a.ldo + h * DH + ks * 32 + g4 * 8;
      g = *reinterpret_cast<const uint4*>(a.dout + oo);
      o = *reinterpret_cast<const uint4*>(a.out + oo);
    }
    qu[ks] = __builtin_bit_cast(bf16x8_t, u);
    qv[ks] = __builtin_bit_cast(bf16x8_t, v);
    dO[ks] = __builtin_bit_cast(bf16x8_t, g);
    const uint32_t wg[4] = {g.x, g.y, g.z, g.w}, wo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      Di += __uint_as_float(wg[e] << 16) * __uint_as_float(wo[e] << 16) +
            __uint_as_float(wg[e] & 0xffff0000u) * __uint_as_float(wo[e] & 0xffff0000u);
  }
  Di += __shfl_xor(Di, 16, 64);
  Di += __shfl_xor(Di, 32, 64);
  const float lse_i = (i < T) ? a.lse[(long)z * T + i] : 0.f;
  if (i < T && g4 == 0 && a.D) a.D[(long)z * T + i] = Di;

  f32x4_t acc_t1[4], acc_t2[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    acc_t1[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    acc_t2[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }

  const bf16_t* Kb = a.k + (long)b * S * a.ldkv + h * DH;
  const bf16_t* Vb = a.v + (long)b * S * a.ldkv + h * DH;
  const bf16_t* PPb = RELPOS ? a.pp + h * DH : nullptr;
  const int R = 2 * T - 1;
  float* bd = sBD + (RELPOS ? w * 80 * BDP : 0);
  const int c0w = 48 - 16 * w;

  int jend = kl;
  if (a.causal) jend = min(jend, i0 + TQ + (S - T));
  int jcov = 0;
  BlkRegs rK, rP;
  RowRegs<2> rV;
  auto load_tile = [&](int j0) {
    if (tid < 128) load_blk(rK, Kb, a.ldkv, j0, S, tid);
    load_rows<2>(rV, Vb, a.ldkv, j0, S, tid);
    if (RELPOS) load_blk(rP, PPb, a.ldpp, (T - 1) - (i0 + TQ - 1) + j0, R, tid);
  };
  if (jend > 0) load_tile(0);
  for (int j0 = 0; j0 < jend; j0 += TK) {
    jcov = min(S, j0 + TK);
    __syncthreads();
    if (tid < 128) {
      store_blk_rows(sK, rK, tid);
      store_blk_t<64>(sKt, rK, tid);
    }
    store_rows<2>(sV, rV, tid);
    if (RELPOS) {
      store_blk_rows(sPP, rP, tid);
      store_blk_t<128>(sPPt, rP, tid);
    }
    __syncthreads();
    if (j0 + TK < jend) load_tile(j0 + TK);

    f32x4_t acc_s[4], acc_dp[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      acc_s[jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      acc_dp[jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + off16(jt * 16 + li, ks * 4 + g4));
        acc_s[jt] = mfma32(kf, qu[ks], acc_s[jt]);
        const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(sV + off16(jt * 16 + li, ks * 4 + g4));
        acc_dp[jt] = mfma32(vf, dO[ks], acc_dp[jt]);
      }
    if (RELPOS) {
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
      wave_lds_sync();
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc_s[jt][r] += bd[(15 - li + jt * 16 + g4 * 4 + r) * BDP + li];
      wave_lds_sync();
      // clear the band: it now receives dS in the un-skewed layout (entries outside this key tile stay zero)
#pragma unroll
      for (int q = 0; q < 20; ++q) bd[((q * 4 + g4) * BDP) + li] = 0.f;
      wave_lds_sync();
    }
    // dS = P * (dP - D)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + jt * 16 + g4 * 4 + r;
        const bool masked = j >= kl || (a.causal && j > i + (S - T)) || i >= T;
        const float p = masked ? 0.f : __expf(acc_s[jt][r] - lse_i);
        float dp = acc_dp[jt][r];
        if (a.thr) dp *= ea_keep(a.seed, ((uint64_t)z * T + (uint64_t)i) * (uint64_t)S + (uint64_t)j, a.thr, a.inv_keep);
        const float ds = p * (dp - Di);
        acc_s[jt][r] = ds;
        if (RELPOS) bd[(15 - li + jt * 16 + g4 * 4 + r) * BDP + li] = ds;
      }
    // t1^T[d][i] += K^T[d][j] dS^T[j][i]
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const bf16x4_t db = pack4(acc_s[jt][0], acc_s[jt][1], acc_s[jt][2], acc_s[jt][3]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc_t1[dt] = mfma16(read_t<64>(sKt, dt * 16 + li, jt * 4 + g4), db, acc_t1[dt]);
    }
    if (RELPOS) {
      wave_lds_sync();
      // t2^T[d][i] += PP^T[d][c] dBD^T[c][i] over this wavefront's 80-position band
#pragma unroll
      for (int ct = 0; ct < 5; ++ct) {
        const int cb = (ct * 16 + g4 * 4) * BDP + li;
        const bf16x4_t db = pack4(bd[cb], bd[cb + BDP], bd[cb + 2 * BDP], bd[cb + 3 * BDP]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          acc_t2[dt] = mfma16(read_t<128>(sPPt, dt * 16 + li, (c0w + ct * 16) / 4 + g4), db, acc_t2[dt]);
      }
      // dBD[z][row][T-1-row + j] = dS[row][j]: one 128-byte row segment per store instruction
      const int j = j0 + lane;
      if (j < S) {
        const int row_w = i0 + 16 * w, nrow = T - row_w;  // wavefront-uniform
        bf16_t* dp = a.dBD + ((long)z * T + row_w) * a.ld_bd + (T - 1 - row_w) + j;
        const float* bs = bd + (15 + lane) * BDP;
#pragma unroll
        for (int iw = 0; iw < 16; ++iw) {
          if (iw < nrow) dp[(long)iw * (a.ld_bd - 1)] = f2bf(bs[iw * (1 - BDP)]);
        }
      }
    }
  }

  if (i < T) {
    bf16_t* o1 = a.t1 + ((long)b * T + i) * a.ldt + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 pk;
      pk.x = pack_bf2(acc_t1[dt][0] * a.scaling, acc_t1[dt][1] * a.scaling);
      pk.y = pack_bf2(acc_t1[dt][2] * a.scaling, acc_t1[dt][3] * a.scaling);
      *reinterpret_cast<uint2*>(o1 + dt * 16 + g4 * 4) = pk;
    }
    if (RELPOS) {
      bf16_t* o2 = a.t2 + ((long)b * T + i) * a.ldt + h * DH;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 pk;
        pk.x = pack_bf2(acc_t2[dt][0] * a.scaling, acc_t2[dt][1] * a.scaling);
        pk.y = pack_bf2(acc_t2[dt][2] * a.scaling, acc_t2[dt][3] * a.scaling);
        *reinterpret_cast<uint2*>(o2 + dt * 16 + g4 * 4) = pk;
      }
      if (a.dq) {
        bf16_t* oq = a.dq + ((long)b * T + i) * a.lddq + h * DH;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 pk;
          pk.x = pack_bf2((acc_t1[dt][0] + acc_t2[dt][0]) * a.scaling, (acc_t1[dt][1] + acc_t2[dt][1]) * a.scaling);
          pk.y = pack_bf2((acc_t1[dt][2] + acc_t2[dt][2]) * a.scaling, (acc_t1[dt][3] + acc_t2[dt][3]) * a.scaling);
          *reinterpret_cast<uint2*>(oq + dt * 16 + g4 * 4) = pk;
        }
      }
    }
  }
  if (RELPOS && !a.dbd_prezeroed) {
    // columns of dBD no (row, key) pair of this workgroup wrote: r < T-1-row or r >= T-1-row + jcov.  16-byte stores
    // for the 8-element chunks that are entirely outside the band, element stores for the two boundary chunks.
    for (int iw = 0; iw < 16; ++iw) {
      const int row = i0 + 16 * w + iw;
      if (row >= T) break;
      const int lo = T - 1 - row, hi = lo + jcov;
      bf16_t* dr = a.dBD + ((long)z * T + row) * a.ld_bd;
      for (int c = lane; c * 8 < a.ld_bd; c += 64) {
        const int e0 = c * 8;
        if (e0 + 8 <= lo || e0 >= hi) {
          *reinterpret_cast<uint4*>(dr + e0) = make_uint4(0, 0, 0, 0);
        } else if (e0 < lo || e0 + 8 > hi) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e0 + e < lo || e0 + e >= hi) dr[e0 + e] = 0;
        }
      }
    }
  }
}

// KV kernel: workgroup = (z, 64 keys), wavefront = 16 keys (a lane owns key column j = lane&15), loop over query tiles.
template <bool RELPOS>
__global__ __launch_bounds__(256, 2) void flash_bwd_kv_kernel(const FlashBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char sQu[TQ * ROWB];                  // [i][d]
  __shared__ __attribute__((aligned(16))) char sQut[DH * ROWB];                 // [d][i]
  __shared__ __attribute__((aligned(16))) char sQv[RELPOS ? TQ * ROWB : 16];    // [i][d]
  __shared__ __attribute__((aligned(16))) char sdO[TQ * ROWB];                  // [i][d]
  __shared__ __attribute__((aligned(16))) char sdOt[DH * ROWB];                 // [d][i]
  __shared__ __attribute__((aligned(16))) char sPP[RELPOS ? 128 * ROWB : 16];   // [window row][d]
  __shared__ float sBD[RELPOS ? 4 * 16 * 33 : 1];                               // per wavefront [i_l][c'] (pitch 33)
  __shared__ float sLse[TQ], sD[TQ];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 15, g4 = lane >> 4;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int z = vid / a.nk, kt = vid % a.nk;
  const int h = z / a.B, b = z % a.B;
  const int T = a.T, S = a.S;
  const int j0 = kt * TK;
  const int j = j0 + 16 * w + li;  // this lane's key
  const int kl = a.klen ? min(a.klen[b], S) : S;

  bf16x8_t kf[2], vf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    uint4 u = make_uint4(0, 0, 0, 0), v = u;
    if (j < S) {
      const long o = ((long)b * S + j) * a.ldkv + h * DH + ks * 32 + g4 * 8;
      u = *reinterpret_cast<const uint4*>(a.k + o);
      v = *reinterpret_cast<const uint4*>(a.v + o);
    }
    kf[ks] = __builtin_bit_cast(bf16x8_t, u);
    vf[ks] = __builtin_bit_cast(bf16x8_t, v);
  }
  f32x4_t acc_dk[4], acc_dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    acc_dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    acc_dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const bf16_t* Qub = a.qu + (long)b * T * a.ldq + h * DH;
  const bf16_t* Qvb = RELPOS ? a.qv + (long)b * T * a.ldq + h * DH : nullptr;
  const bf16_t* dOb = a.dout + (long)b * T * a.ldo + h * DH;
  const bf16_t* PPb = RELPOS ? a.pp + h * DH : nullptr;
  const int R = 2 * T - 1;
  float* bd = sBD + (RELPOS ? w * 16 * 33 : 0);

  int ibeg = 0;
  if (a.causal) ibeg = max(0, (j0 - (S - T)) / TQ * TQ);  // rows i with j0 > i + (S-T) + 63 see none of these keys
  if (j0 >= kl) ibeg = T;                                    // padded keys: zero gradient
  BlkRegs rQ;  // threads 0..127: Qu block ; threads 128..255: dO block
  RowRegs<2> rQv;
  RowRegs<4> rP;
  float rLse = 0.f, rD = 0.f;
  auto load_tile = [&](int i0) {
    if (tid < 128) load_blk(rQ, Qub, a.ldq, i0, T, tid);
    else load_blk(rQ, dOb, a.ldo, i0, T, tid - 128);
    if (RELPOS) {
      load_rows<2>(rQv, Qvb, a.ldq, i0, T, tid);
      load_rows<4>(rP, PPb, a.ldpp, (T - 1) - (i0 + TQ - 1) + j0, R, tid);
    }
    if (tid < TQ) {
      const int ii = i0 + tid;
      rLse = ii < T ? a.lse[(long)z * T + ii] : 0.f;
      rD = ii < T ? a.D[(long)z * T + ii] : 0.f;
    }
  };
  if (ibeg < T) load_tile(ibeg);
  for (int i0 = ibeg; i0 < T; i0 += TQ) {
    __syncthreads();
    if (tid < 128) {
      store_blk_rows(sQu, rQ, tid);
      store_blk_t<64>(sQut, rQ, tid);
    } else {
      store_blk_rows(sdO, rQ, tid - 128);
      store_blk_t<64>(sdOt, rQ, tid - 128);
    }
    if (RELPOS) {
      store_rows<2>(sQv, rQv, tid);
      store_rows<4>(sPP, rP, tid);
    }
    if (tid < TQ) {
      sLse[tid] = rLse;
      sD[tid] = rD;
    }
    __syncthreads();
    if (i0 + TQ < T) load_tile(i0 + TQ);

#pragma unroll
    for (int it = 0; it < 4; ++it) {
      // S[i][j], dPd[i][j] for rows i = i0 + it*16 + g4*4 + r, column j
      f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(sQu + off16(it * 16 + li, ks * 4 + g4));
        s = mfma32(qf, kf[ks], s);
        const bf16x8_t gf = *reinterpret_cast<const bf16x8_t*>(sdO + off16(it * 16 + li, ks * 4 + g4));
        dp = mfma32(gf, vf[ks], dp);
      }
      if (RELPOS) {
        // BD_sub[i_l][c'] = Qv[i] . PP[row0 + cbase + c'],  c' = 15 - i_l + j_l in [0, 31)
        const int cbase = 48 - 16 * it + 16 * w;
        wave_lds_sync();
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          f32x4_t t = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t qf = *reinterpret_cast<const bf16x8_t*>(sQv + off16(it * 16 + li, ks * 4 + g4));
            const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(sPP + off16(cbase + ct * 16 + li, ks * 4 + g4));
            t = mfma32(qf, pf, t);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) bd[(g4 * 4 + r) * 33 + ct * 16 + li] = t[r];
        }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += bd[(g4 * 4 + r) * 33 + 15 - (g4 * 4 + r) + li];
      }
      float ps[4], dsv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = it * 16 + g4 * 4 + r;
        const int i = i0 + il;
        const bool masked = j >= kl || i >= T || (a.causal && j > i + (S - T));
        const float p = masked ? 0.f : __expf(s[r] - sLse[il]);
        float keep = 1.f;
        if (a.thr) keep = ea_keep(a.seed, ((uint64_t)z * T + (uint64_t)i) * (uint64_t)S + (uint64_t)j, a.thr, a.inv_keep);
        ps[r] = p * keep;                       // Pd
        dsv[r] = p * (dp[r] * keep - sD[il]);   // dS
      }
      const bf16x4_t pb = pack4(ps[0], ps[1], ps[2], ps[3]);
      const bf16x4_t db = pack4(dsv[0], dsv[1], dsv[2], dsv[3]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        acc_dv[dt] = mfma16(read_t<64>(sdOt, dt * 16 + li, it * 4 + g4), pb, acc_dv[dt]);
        acc_dk[dt] = mfma16(read_t<64>(sQut, dt * 16 + li, it * 4 + g4), db, acc_dk[dt]);
      }
    }
  }
  if (j < S) {
    bf16_t* ok = a.dk + ((long)b * S + j) * a.lddkv + h * DH;
    bf16_t* ov = a.dv + ((long)b * S + j) * a.lddkv + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 pk;
      pk.x = pack_bf2(acc_dk[dt][0], acc_dk[dt][1]);
      pk.y = pack_bf2(acc_dk[dt][2], acc_dk[dt][3]);
      *reinterpret_cast<uint2*>(ok + dt * 16 + g4 * 4) = pk;
      pk.x = pack_bf2(acc_dv[dt][0], acc_dv[dt][1]);
      pk.y = pack_bf2(acc_dv[dt][2], acc_dv[dt][3]);
      *reinterpret_cast<uint2*>(ov + dt * 16 + g4 * 4) = pk;
    }
  }
}

}  // namespace

extern "C" int ea_flash_attention_supported(int dh, int T, int S, int relpos) {
  return dh == DH && T > 0 && S > 0 && (!relpos || T == S);
}

extern "C" long ea_flash_keep_bits_bytes(int H, int B, int T) {
  const long nkt = (T + TK - 1) / TK;
  return (long)H * B * nkt * (nkt * 64) * 4 * (long)sizeof(uint16_t);
}

extern "C" int ea_flash_keep_bits(void* keep_bits, int H, int B, int T, uint64_t drop_seed, uint32_t drop_thr, hipStream_t stream) {
  if (H <= 0 || B <= 0 || T <= 0 || !keep_bits) return keep_bits ? 0 : -2;
  return ea_rp_keep_bits((uint16_t*)keep_bits, H, B, T, drop_seed, drop_thr, stream);
}

extern "C" int ea_flash_attention_fwd(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldkv,
                                      const void* pp, long ldpp, const int* key_len, void* out, long ldo, float* lse, int H,
                                      int B, int T, int S, int dh, int causal, uint64_t drop_seed, uint32_t drop_thr,
                                      float drop_scale, void* keep_bits, hipStream_t stream) {
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
  a.H = H; a.B = B; a.T = T; a.S = S; a.causal = causal & 1; a.nq = (T + TQ - 1) / TQ;
  a.seed = drop_seed; a.thr = drop_thr; a.inv_keep = drop_scale;
  a.bits = (const uint16_t*)keep_bits;
  if (ea_rp_eligible(relpos, T, S, causal, drop_thr, keep_bits)) {
    // encoder hot path (flash_relpos.hip); the keep decisions are evaluated once, here, unless the caller already did
    if (drop_thr && !(causal & 4)) {
      const int rc = ea_rp_keep_bits((uint16_t*)keep_bits, H, B, T, drop_seed, drop_thr, stream);
      if (rc) return rc;
    }
    return ea_rp_fwd(a, stream);
  }
  const dim3 grid((unsigned)(a.nq * H * B));
  if (relpos) hipLaunchKernelGGL(flash_fwd_kernel<true>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(flash_fwd_kernel<false>, grid, dim3(256), 0, stream, a);
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_flash_attention_bwd(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldkv,
                                      const void* pp, long ldpp, const int* key_len, const void* out, const void* dout, long ldo,
                                      const float* lse, float* D, void* t1, void* t2, long ldt, void* dBD, int ld_bd, void* dk,
                                      void* dv, long lddkv, int H, int B, int T, int S, int dh, int causal, float scaling,
                                      uint64_t drop_seed, uint32_t drop_thr, float drop_scale, const void* keep_bits,
                                      void* dq, long lddq, hipStream_t stream) {
  if (H <= 0 || B <= 0 || T <= 0) return 0;
  const bool relpos = qv != nullptr;
  if (!ea_flash_attention_supported(dh, T, S, relpos)) return -2;
  if (relpos && (!pp || !t2 || !dBD || ld_bd < 2 * T - 1 || ld_bd % 8 || ((uintptr_t)dBD & 15))) return -2;
  if ((ldq | ldkv | ldo | ldt | lddkv) % 8 || (relpos && ldpp % 8)) return -2;
  if (((uintptr_t)qu | (uintptr_t)qv | (uintptr_t)k | (uintptr_t)v | (uintptr_t)pp | (uintptr_t)out | (uintptr_t)dout) & 15) return -2;
  if (((uintptr_t)t1 | (uintptr_t)t2 | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)dq) & 7) return -2;
  if (dq && (!relpos || lddq % 4)) return -2;  // without the positional term t1 IS the query gradient
  FlashBwdArgs a;
  a.qu = (const bf16_t*)qu; a.qv = (const bf16_t*)qv; a.ldq = ldq;
  a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.ldkv = ldkv;
  a.pp = (const bf16_t*)pp; a.ldpp = ldpp;
  a.klen = key_len;
  a.out = (const bf16_t*)out; a.dout = (const bf16_t*)dout; a.ldo = ldo;
  a.lse = lse; a.D = D;
  a.t1 = (bf16_t*)t1; a.t2 = (bf16_t*)t2; a.ldt = ldt;
  a.dq = (bf16_t*)dq; a.lddq = lddq;
  a.dBD = (bf16_t*)dBD; a.ld_bd = ld_bd;
  a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.lddkv = lddkv;
  a.H = H; a.B = B; a.T = T; a.S = S; a.causal = causal & 1; a.dbd_prezeroed = (causal >> 1) & 1;
  a.nq = (T + TQ - 1) / TQ; a.nk = (S + TK - 1) / TK;
  a.scaling = scaling;
  a.seed = drop_seed; a.thr = drop_thr; a.inv_keep = drop_scale;
  a.bits = (const uint16_t*)keep_bits;
  if (ea_rp_eligible(relpos, T, S, causal, drop_thr, keep_bits)) return ea_rp_bwd(a, stream);
  const dim3 gq((unsigned)(a.nq * H * B)), gk((unsigned)(a.nk * H * B));
  if (relpos) {
    hipLaunchKernelGGL(flash_bwd_q_kernel<true>, gq, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(flash_bwd_kv_kernel<true>, gk, dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL(flash_bwd_q_kernel<false>, gq, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(flash_bwd_kv_kernel<false>, gk, dim3(256), 0, stream, a);
  }
  return EA_CHECK_LAUNCH();
}
