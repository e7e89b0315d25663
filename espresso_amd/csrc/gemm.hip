// bf16 MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
//   C[z][m][n] = epilogue( alpha * sum_k A(z; m,k) * B(z; n,k) )
//
// Every dense product on the ASR hot path goes through this kernel: the Linear layers of the
// Conformer/Transformer blocks (reference: fairseq/modules/conformer_layer.py:134-146 FFN,
// fairseq/modules/multihead_attention.py:650-688 q/k/v/out/pos projections, pointwise convs
// conformer_layer.py:79-101), fc0/fc_out (espresso/models/transformer/speech_transformer_encoder.py:341-343,
// speech_transformer_encoder_model.py:207-208), the attention products QK^T / Q P^T / P V
// (multihead_attention.py:788-831, 884-907) and the im2col form of the conv2d sub-sampler
// (espresso/modules/speech_convolutions.py:78-102) — forward, dgrad and wgrad.
//
// Design (CDNA4): 128x128 output tile per 256-thread workgroup (4 waves, each 64x64 = 4x4 MFMA
// 16x16x32 bf16 tiles, fp32 accumulators in AGPR/VGPR), BK = 64.  Operands are staged
// global -> registers -> LDS; the next K-tile's global loads are issued before the MFMAs of the
// current one so HBM latency hides under the matrix pipe.  The LDS image is always
// [row][k] (k contiguous, 128 B rows) with a 16-B-chunk XOR swizzle (chunk ^= row & 7) so the
// ds_read_b128 fragment reads are bank-conflict free.  An operand may be stored "k-strided"
// in memory (element (row,k) at ptr[k*ld + row]); it is then transposed in registers while
// being staged (4x8 block per thread -> eight ds_write_b64), which is what lets dgrad / wgrad /
// P·V run without materialising transposed copies in HBM.
#include <cstdio>
#include <type_traits>
#include "common.h"
#include "espresso_amd.h"
#include "gemm_common.h"
#include "gemm_epilogue.h"

// 8-wave large-tile kernels (gemm_w8.hip): takes the launch (returns 1) or leaves it to the kernels below (0)
int ea_gemm_w8_try(const EaGemmParams& q, int nt_flag, hipStream_t stream, int* cfg_out);
int ea_wgrad_w8_try(const EaWgradGroup& g, hipStream_t stream, int* grid_out, EaWgradGroup* rest);  // wgrad_w8.hip
// Hint from the layer runtime (engine.hip): the launches that follow run NEXT TO side-stream work (the backward pass: grouped weight
// gradients of 2 x 64 KB of LDS per CU).  A one-workgroup-per-CU kernel with 128 - 144 KB of LDS cannot share a CU with them: it
// waits for both to drain and then keeps them out, so the 8-wave kernels, 10 - 15 % faster alone, lose in that half of the step
// (round 5, same box: 14.18 ms per step with them everywhere, 13.91 without).  Thread-local: one host thread drives one stream.
static thread_local int g_gemm_corun = 0;
void ea_gemm_corun_hint(int on) { g_gemm_corun = on; }
// 8-wave kernels for launches that share the device with side-queue work too (the backward's data-gradient chain): round 5 measured
// them slower there, beside a chip-filling 4-wave weight-gradient launch; beside the 116-workgroup 8-wave one (round 6) they win:
// 12.47 -> 12.35 ms per step (profiles/r06_side_kernel_grids_ab.txt).  EA_GEMM_W8_CORUN=0 restores the 4-wave choice.
static const int g_w8_corun = [] { const char* e = getenv("EA_GEMM_W8_CORUN"); return e ? atoi(e) : 1; }();

namespace {

constexpr int BM = 128, BN = 128;

// ---- global -> register staging -------------------------------------------------------------
// K-contiguous operand: element (row,k) at p[row*ld + k]. Thread owns 4 chunks of 8 k.
__device__ __forceinline__ void load_kc(const bf16_t* __restrict__ p, long ld, int rows, int K,
                                        int row0, int k0, int tid, uint4 (&r)[4]) {
  const int c = tid & 7;
  const int gk = k0 + c * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + (tid >> 3) + 32 * i;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < rows && gk < K) {
      const bf16_t* q = p + (long)row * ld + gk;
      if (gk + 8 <= K && (((uintptr_t)q) & 15) == 0) {
        v = *reinterpret_cast<const uint4*>(q);
      } else {
        bf16_t t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (gk + e < K) ? q[e] : (bf16_t)0;
        v.x = t[0] | ((uint32_t)t[1] << 16);
        v.y = t[2] | ((uint32_t)t[3] << 16);
        v.z = t[4] | ((uint32_t)t[5] << 16);
        v.w = t[6] | ((uint32_t)t[7] << 16);
      }
    }
    r[i] = v;
  }
}
__device__ __forceinline__ void store_kc(char* __restrict__ s, int tid, const uint4 (&r)[4]) {
  const int c = tid & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 3) + 32 * i;
    *reinterpret_cast<uint4*>(s + lds_off(row, c)) = r[i];
  }
}
// K-strided operand: element (row,k) at p[k*ld + row]. Thread owns a 4(k) x 8(row) block.
__device__ __forceinline__ void load_ks(const bf16_t* __restrict__ p, long ld, int rows, int K,
                                        int row0, int k0, int tid, uint4 (&r)[4]) {
  const int grow = row0 + (tid & 15) * 8;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gk = k0 + (tid >> 4) * 4 + j;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gk < K && grow < rows) {
      const bf16_t* q = p + (long)gk * ld + grow;
      if (grow + 8 <= rows && (((uintptr_t)q) & 15) == 0) {
        v = *reinterpret_cast<const uint4*>(q);
      } else {
        bf16_t t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (grow + e < rows) ? q[e] : (bf16_t)0;
        v.x = t[0] | ((uint32_t)t[1] << 16);
        v.y = t[2] | ((uint32_t)t[3] << 16);
        v.z = t[4] | ((uint32_t)t[5] << 16);
        v.w = t[6] | ((uint32_t)t[7] << 16);
      }
    }
    r[j] = v;
  }
}
__device__ __forceinline__ void store_ks(char* __restrict__ s, int tid, const uint4 (&r)[4]) {
  const int kk0 = (tid >> 4) * 4;
  const int r0 = (tid & 15) * 8;
  const uint32_t w0[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
  const uint32_t w1[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
  const uint32_t w2[4] = {r[2].x, r[2].y, r[2].z, r[2].w};
  const uint32_t w3[4] = {r[3].x, r[3].y, r[3].z, r[3].w};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    // even row 2d : low halves ; odd row 2d+1 : high halves
    uint2 lo, hi;
    lo.x = (w0[d] & 0xffffu) | (w1[d] << 16);
    lo.y = (w2[d] & 0xffffu) | (w3[d] << 16);
    hi.x = (w0[d] >> 16) | (w1[d] & 0xffff0000u);
    hi.y = (w2[d] >> 16) | (w3[d] & 0xffff0000u);
    const int ra = r0 + 2 * d, rb = ra + 1;
    *reinterpret_cast<uint2*>(s + lds_off(ra, kk0 >> 3) + (kk0 & 4) * 2) = lo;
    *reinterpret_cast<uint2*>(s + lds_off(rb, kk0 >> 3) + (kk0 & 4) * 2) = hi;
  }
}

// k-strided 4(k) x 8(row) block at an explicit (row, k) position (guarded)
__device__ __forceinline__ void load_ks_at(const bf16_t* __restrict__ p, long ld, int rows, int K, int grow, int gk0,
                                           uint4 (&r)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gk = gk0 + j;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gk < K && grow < rows) {
      const bf16_t* q = p + (long)gk * ld + grow;
      if (grow + 8 <= rows && (((uintptr_t)q) & 15) == 0) {
        v = *reinterpret_cast<const uint4*>(q);
      } else {
        bf16_t t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (grow + e < rows) ? q[e] : (bf16_t)0;
        v.x = t[0] | ((uint32_t)t[1] << 16);
        v.y = t[2] | ((uint32_t)t[3] << 16);
        v.z = t[4] | ((uint32_t)t[5] << 16);
        v.w = t[6] | ((uint32_t)t[7] << 16);
      }
    }
    r[j] = v;
  }
}
__device__ __forceinline__ void store_ks_at(char* __restrict__ s, int r0, int kk0, const uint4 (&r)[4]) {
  const uint32_t w0[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
  const uint32_t w1[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
  const uint32_t w2[4] = {r[2].x, r[2].y, r[2].z, r[2].w};
  const uint32_t w3[4] = {r[3].x, r[3].y, r[3].z, r[3].w};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint2 lo, hi;
    lo.x = (w0[d] & 0xffffu) | (w1[d] << 16);
    lo.y = (w2[d] & 0xffffu) | (w3[d] << 16);
    hi.x = (w0[d] >> 16) | (w1[d] & 0xffff0000u);
    hi.y = (w2[d] >> 16) | (w3[d] & 0xffff0000u);
    const int ra = r0 + 2 * d, rb = ra + 1;
    *reinterpret_cast<uint2*>(s + lds_off(ra, kk0 >> 3) + (kk0 & 4) * 2) = lo;
    *reinterpret_cast<uint2*>(s + lds_off(rb, kk0 >> 3) + (kk0 & 4) * 2) = hi;
  }
}
// guarded k-contiguous load of NI row groups (32 rows each)
template <int NI>
__device__ __forceinline__ void load_kc_n(const bf16_t* __restrict__ p, long ld, int rows, int K, int row0, int k0, int tid,
                                          uint4 (&r)[4]) {
  const int c = tid & 7;
  const int gk = k0 + c * 8;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = row0 + (tid >> 3) + 32 * i;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < rows && gk < K) {
      const bf16_t* q = p + (long)row * ld + gk;
      if (gk + 8 <= K && (((uintptr_t)q) & 15) == 0) {
        v = *reinterpret_cast<const uint4*>(q);
      } else {
        bf16_t t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (gk + e < K) ? q[e] : (bf16_t)0;
        v.x = t[0] | ((uint32_t)t[1] << 16);
        v.y = t[2] | ((uint32_t)t[3] << 16);
        v.z = t[4] | ((uint32_t)t[5] << 16);
        v.w = t[6] | ((uint32_t)t[7] << 16);
      }
    }
    r[i] = v;
  }
}

// Fast paths: whole tile in bounds and 16-byte aligned -> unconditional vector loads from per-thread base
// pointers that only advance along k (no per-load bounds / alignment logic in the K loop).
__device__ __forceinline__ void load_kc_fast(const bf16_t* __restrict__ base, long ld, int k0, uint4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const uint4*>(base + (long)(32 * i) * ld + k0);
}
__device__ __forceinline__ void load_ks_fast(const bf16_t* __restrict__ base, long ld, int k0, uint4 (&r)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const uint4*>(base + (long)(k0 + j) * ld);
}

// BM_ = 128: 2x2 wavefronts of 64x64 (acc 4x4 MFMA tiles).  BM_ = 64: 1x4 wavefronts of 64x32 (acc 4x2) — twice
// as many workgroups for GEMMs whose output has too few 128x128 tiles to fill 256 CUs (N = 512 projections).
// ---- tile epilogue shared by the GEMM kernels: accumulators -> fp32 LDS tile (64 rows at a time) -> coalesced 16-byte stores ----
// acc[i][j][r] = C[m0 + wm*64 + i*16 + (lane>>4)*4 + r][n0 + wcol + j*16 + (lane&15)]; sC = 32 KiB of LDS no wave still reads
// operands from once it has passed the first barrier in here.
template <int BM_, bool FAST>
__device__ __forceinline__ void gemm_tile_epilogue(const EaGemmParams& p, const f32x4_t (&acc)[4][BM_ == 128 ? 4 : 2], float* sC, int m0,
                                                   int n0, int z, int ks_id, int zhi, int zlo, long coff, int xcd_swizzle) {
  constexpr int NJ = BM_ == 128 ? 4 : 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = BM_ == 128 ? (wave >> 1) : 0;
  const int wcol = BM_ == 128 ? (wave & 1) * 64 : wave * 32;
  const bool vec_ok = (p.ldc & 7) == 0 && (((uintptr_t)p.C) & 15) == 0 && ((p.sC_hi | p.sC_lo) & 7) == 0;
  float bias8[8];
  load_bias8(p, n0 + (tid & 15) * 8, bias8);
  float posu8[8], posv8[8];
  load_pos8(p, n0 + (tid & 15) * 8, posu8, posv8);
  // FAST: the residual (else the auxiliary) rows of the four passes are fetched here, in one round trip that overlaps the LDS
  // bounce, and rotate through the rolled pass loop (probe tools/probes/gemm_timing.hip: the epilogue of a 64 x 128 tile took
  // 3.0 us plain and 5.0 us with a residual — one dependent L2 / MALL round trip per pass)
  const int pre_kind = !FAST ? 0 : p.resid ? 1 : p.aux ? 2 : 0;
  const bf16_t* pre_base = pre_kind == 1 ? reinterpret_cast<const bf16_t*>(p.resid) : reinterpret_cast<const bf16_t*>(p.aux);
  const long pre_ld = pre_kind == 1 ? p.ldr : p.ldaux;
#pragma unroll
  for (int half = 0; half < BM_ / 64; ++half) {
    uint4 pre0 = uint4{0, 0, 0, 0}, pre1 = pre0, pre2 = pre0, pre3 = pre0;
    if (FAST && pre_kind) {
      const bf16_t* q = pre_base + n0 + (tid & 15) * 8;
      const int mr = m0 + half * 64 + (tid >> 4), ml = p.M - 1;
      pre0 = *reinterpret_cast<const uint4*>(q + (long)min(mr, ml) * pre_ld);
      pre1 = *reinterpret_cast<const uint4*>(q + (long)min(mr + 16, ml) * pre_ld);
      pre2 = *reinterpret_cast<const uint4*>(q + (long)min(mr + 32, ml) * pre_ld);
      pre3 = *reinterpret_cast<const uint4*>(q + (long)min(mr + 48, ml) * pre_ld);
    }
    __syncthreads();
    if (half == 0) EA_STAMP(5);
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sC[(i * 16 + (lane >> 4) * 4 + r) * BN + wcol + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    __syncthreads();
    if (half == 0) EA_STAMP(6);
    if (m0 + half * 64 < p.M) {
#pragma unroll 1  // (a real loop: four inlined copies of the epilogue were most of the kernel's code)
      for (int pass = 0; pass < 4; ++pass) {
        const int rl = pass * 16 + (tid >> 4);
        const int m = m0 + half * 64 + rl;
        const int n = n0 + (tid & 15) * 8;
        if (m < p.M && (FAST || n < p.N)) {
          float v[8];
          const float4 x0 = *reinterpret_cast<const float4*>(sC + rl * BN + (tid & 15) * 8);
          const float4 x1 = *reinterpret_cast<const float4*>(sC + rl * BN + (tid & 15) * 8 + 4);
          v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
          epilogue_chunk<FAST>(p, z, ks_id, zhi, zlo, coff, m, n, v, bias8, posu8, posv8, vec_ok, (xcd_swizzle & 2) != 0, pre_kind, pre0);
        }
        pre0 = pre1; pre1 = pre2; pre2 = pre3;
      }
    }
  }
}

template <bool A_KS, bool B_KS, int BM_, bool FAST = false>
__global__ __launch_bounds__(256, 3) void gemm_bf16_kernel(const EaGemmParams p, const int xcd_swizzle) {
  constexpr int NJ = BM_ == 128 ? 4 : 2;  // n-tiles per wave
  __shared__ __attribute__((aligned(16))) char smem[2 * BM * ROW_BYTES];  // A tile (<=128 rows) + B tile ; reused as fp32 C tile
  char* sA = smem;
  char* sB = smem + BM * ROW_BYTES;
  EA_STAMP(0);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = BM_ == 128 ? (wave >> 1) : 0;
  const int wcol = BM_ == 128 ? (wave & 1) * 64 : wave * 32;  // first column of this wave's sub-tile
  int tile_x = blockIdx.x, tile_y = blockIdx.y, tile_z = blockIdx.z;
  if (xcd_swizzle & 1) {
    // workgroups are dealt round-robin to the 8 XCDs in dispatch order (x fastest, then y, z); give every XCD one contiguous
    // range of the (z, y, x) tile order instead: the n-tiles of a row block and the row blocks of a batch item / k-chunk
    // then share their operand rows in ONE L2 (bijective for any grid size)
    const int gx = gridDim.x, gxy = gridDim.x * gridDim.y, total = gxy * gridDim.z;
    const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gx + blockIdx.x;
    const int xcd = lin & 7, q = total >> 3, r = total & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    tile_z = FAST ? 0 : v / gxy;  // (FAST launches have gridDim.z == 1: no batch / split-K index arithmetic at all)
    const int rem = v - tile_z * gxy;
    tile_y = rem / gx;
    tile_x = rem - tile_y * gx;
  }
  const int m0 = tile_y * BM_, n0 = tile_x * BN;
  const int z = FAST ? 0 : tile_z / p.splitk;
  const int ks_id = FAST ? 0 : tile_z % p.splitk;
  const int zhi = FAST ? 0 : z / p.zdiv, zlo = FAST ? 0 : z % p.zdiv;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (long)zhi * p.sA_hi + (long)zlo * p.sA_lo;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + (long)zhi * p.sB_hi + (long)zlo * p.sB_lo;
  const long coff = (long)zhi * p.sC_hi + (long)zlo * p.sC_lo;

  const int kbeg = ks_id * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  f32x4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const bool a_ok = (m0 + BM_ <= p.M) && (p.lda & 7) == 0 && ((uintptr_t)A & 15) == 0 && (!A_KS || (m0 & 7) == 0);
  const bool b_ok = (n0 + BN <= p.N) && (p.ldb & 7) == 0 && ((uintptr_t)B & 15) == 0 && (!B_KS || (n0 & 7) == 0);
  // thread -> staging role.  A tile with 64 rows: k-contiguous uses i < 2 ; k-strided uses threads 0..127 only.
  const int a_kq = BM_ == 128 ? (tid >> 4) : (tid >> 3);       // k-quad of the k-strided A block
  const int a_rc = BM_ == 128 ? (tid & 15) : (tid & 7);        // 8-row chunk of the k-strided A block
  const bool a_ks_active = BM_ == 128 || tid < 128;
  const bf16_t* a_base = A_KS ? A + (long)(a_kq * 4) * p.lda + m0 + a_rc * 8 : A + (long)(m0 + (tid >> 3)) * p.lda + (tid & 7) * 8;
  const bf16_t* b_base = B_KS ? B + (long)((tid >> 4) * 4) * p.ldb + n0 + (tid & 15) * 8
                              : B + (long)(n0 + (tid >> 3)) * p.ldb + (tid & 7) * 8;
  auto loadA = [&](int k0, uint4 (&ra)[4]) {
    if (A_KS) {
      if (!a_ks_active) return;
      if (a_ok && k0 + BK <= kend) load_ks_fast(a_base, p.lda, k0, ra);
      else load_ks_at(A, p.lda, p.M, kend, m0 + a_rc * 8, k0 + a_kq * 4, ra);
    } else {
      if (a_ok && k0 + BK <= kend) {
#pragma unroll
        for (int i = 0; i < BM_ / 32; ++i) ra[i] = *reinterpret_cast<const uint4*>(a_base + (long)(32 * i) * p.lda + k0);
      } else {
        load_kc_n<BM_ / 32>(A, p.lda, p.M, kend, m0, k0, tid, ra);
      }
    }
  };
  auto loadB = [&](int k0, uint4 (&rb)[4]) {
    if (b_ok && k0 + BK <= kend) { if (B_KS) load_ks_fast(b_base, p.ldb, k0, rb); else load_kc_fast(b_base, p.ldb, k0, rb); }
    else { if (B_KS) load_ks(B, p.ldb, p.N, kend, n0, k0, tid, rb); else load_kc(B, p.ldb, p.N, kend, n0, k0, tid, rb); }
  };
  auto storeA = [&](const uint4 (&ra)[4]) {
    if (A_KS) {
      if (a_ks_active) store_ks_at(sA, a_rc * 8, a_kq * 4, ra);
    } else {
#pragma unroll
      for (int i = 0; i < BM_ / 32; ++i) *reinterpret_cast<uint4*>(sA + lds_off((tid >> 3) + 32 * i, tid & 7)) = ra[i];
    }
  };
  if (nk > 0) {
    loadA(kbeg, ra);
    loadB(kbeg, rb);
  }
  uint32_t a_off[2][4], b_off[2][NJ];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[ks][i] = lds_off(wm * 64 + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
    for (int j = 0; j < NJ; ++j) b_off[ks][j] = lds_off(wcol + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
  }

  auto storeB = [&](const uint4 (&rb)[4]) { if (B_KS) store_ks(sB, tid, rb); else store_kc(sB, tid, rb); };
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[4], bfr[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sA + a_off[ks][i]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(sB + b_off[ks][j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, af[i]),
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, bfr[j]), acc[i][j], 0, 0, 0);
    }
  };
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    storeA(ra);
    storeB(rb);
    __syncthreads();
    if (kt == 0) EA_STAMP(1);
    if (kt + 1 < nk) { loadA(kbeg + (kt + 1) * BK, ra); loadB(kbeg + (kt + 1) * BK, rb); }
    compute();
  }
  EA_STAMP(2);

  // ---- epilogue: accumulators -> fp32 LDS tile (64 rows at a time) -> coalesced 16/32-byte stores ----
  // acc[i][j][r] = C[m0 + wm*64 + i*16 + (lane>>4)*4 + r][n0 + wcol + j*16 + (lane&15)]
  float* sC = reinterpret_cast<float*>(smem);  // [64][128] fp32 = 32 KiB
  gemm_tile_epilogue<BM_, FAST>(p, acc, sC, m0, n0, z, ks_id, zhi, zlo, coff, xcd_swizzle);
  EA_STAMP(3);
}

// ---- direct-to-LDS variant (both operands k-contiguous) ------------------------------------------------------------
// Same tiles, fragment layout, swizzle and epilogue as gemm_bf16_kernel, but the operand tiles go global -> LDS with
// `global_load_lds_dwordx4` (no staging registers, no ds_write pass) into a ring of NST stages, with NST-1 k-tiles in
// flight and ONE raw barrier per k-tile.  A wave instruction moves 64 x 16 B = 8 LDS rows: lane l lands at row l>>3,
// 16-byte slot l&7, so the XOR swizzle is applied to the SOURCE address (lane fetches chunk slot ^ (r&7) ^ ((r>>4)&7)).
// Rows past M / N are clamped to the last valid row (their outputs are never stored).  Requires K % 64 == 0, leading
// dimensions % 8 == 0 and 16-byte aligned bases (checked on the host; other launches use gemm_bf16_kernel).
template <int BM_, int NST, bool FAST = false>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(const EaGemmParams p, const int xcd_swizzle) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];  // the ONLY LDS object (ring, then the fp32 C tile)
  constexpr int NJ = BM_ == 128 ? 4 : 2;
  constexpr int A_BYTES = BM_ * ROW_BYTES;
  constexpr int STAGE = A_BYTES + BN * ROW_BYTES;
  constexpr int NA = BM_ / 32, NB = BN / 32;  // glds instructions per wave and k-tile
  EA_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = BM_ == 128 ? (wave >> 1) : 0;
  const int wcol = BM_ == 128 ? (wave & 1) * 64 : wave * 32;
  int tile_x = blockIdx.x, tile_y = blockIdx.y, tile_z = blockIdx.z;
  if (xcd_swizzle & 1) {
    // workgroups are dealt round-robin to the 8 XCDs in dispatch order (x fastest, then y, z); give every XCD one contiguous
    // range of the (z, y, x) tile order instead: the n-tiles of a row block and the row blocks of a batch item / k-chunk
    // then share their operand rows in ONE L2 (bijective for any grid size)
    const int gx = gridDim.x, gxy = gridDim.x * gridDim.y, total = gxy * gridDim.z;
    const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gx + blockIdx.x;
    const int xcd = lin & 7, q = total >> 3, r = total & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    tile_z = FAST ? 0 : v / gxy;  // (FAST launches have gridDim.z == 1: no batch / split-K index arithmetic at all)
    const int rem = v - tile_z * gxy;
    tile_y = rem / gx;
    tile_x = rem - tile_y * gx;
  }
  const int m0 = tile_y * BM_, n0 = tile_x * BN;
  const int z = FAST ? 0 : tile_z / p.splitk;
  const int ks_id = FAST ? 0 : tile_z % p.splitk;
  const int zhi = FAST ? 0 : z / p.zdiv, zlo = FAST ? 0 : z % p.zdiv;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (long)zhi * p.sA_hi + (long)zlo * p.sA_lo;
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + (long)zhi * p.sB_hi + (long)zlo * p.sB_lo;
  const long coff = (long)zhi * p.sC_hi + (long)zlo * p.sC_lo;
  const int kbeg = ks_id * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = kend > kbeg ? (kend - kbeg) / BK : 0;

  f32x4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // per-lane source pointers (advance along k only)
  const bf16_t* ap[NA];
  const bf16_t* bp[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = (wave + 4 * i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7) ^ ((r >> 4) & 7);
    ap[i] = A + (long)min(m0 + r, p.M - 1) * p.lda + kbeg + c * 8;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = (wave + 4 * i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7) ^ ((r >> 4) & 7);
    bp[i] = B + (long)min(n0 + r, p.N - 1) * p.ldb + kbeg + c * 8;
  }
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue = [&](int stage, int kt) {
    char* base = dsm + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(ap[i] + (long)kt * BK), (lptr_t)(base + i * 4096), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(bp[i] + (long)kt * BK), (lptr_t)(base + A_BYTES + i * 4096), 16, 0, 0);
  };
  uint32_t a_off[2][4], b_off[2][NJ];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[ks][i] = lds_off(wm * 64 + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
    for (int j = 0; j < NJ; ++j) b_off[ks][j] = A_BYTES + lds_off(wcol + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
  }

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s, s);
  int stage = 0, fill = NST - 1;  // stage holding tile kt ; stage that tile kt+NST-1 goes to
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most NST-2 younger tiles (NA+NB instructions each) are still outstanding
    if (kt + NST - 2 < nk) {
      if (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if ((NST - 2) * (NA + NB) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if ((NST - 2) * (NA + NB) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if ((NST - 2) * (NA + NB) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if ((NST - 2) * (NA + NB) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // every wave's part of tile kt is in LDS; everyone is done reading tile kt-1's stage
    if (kt == 0) EA_STAMP(1);
    if (kt + NST - 1 < nk) issue(fill, kt + NST - 1);
    const char* st = dsm + stage * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[4], bfr[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(st + a_off[ks][i]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(st + b_off[ks][j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, af[i]),
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, bfr[j]), acc[i][j], 0, 0, 0);
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
    fill = fill + 1 == NST ? 0 : fill + 1;
  }
  EA_STAMP(2);

  float* sC = reinterpret_cast<float*>(dsm);  // [64][128] fp32 = 32 KiB
  gemm_tile_epilogue<BM_, FAST>(p, acc, sC, m0, n0, z, ks_id, zhi, zlo, coff, xcd_swizzle);
  EA_STAMP(3);
}

// C[z][m][n] (+)= sum_s W[s][z][m][n]      (VEC: N % 4 == 0 and 16-byte aligned C rows -> float4 per thread)
template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const EaGemmParams p) {
  constexpr int W_ = VEC ? 4 : 1;
  const long per = (long)p.batch * p.M * p.N;
  const long stride = (long)gridDim.x * blockDim.x * W_;
  const float* W = reinterpret_cast<const float*>(p.workspace);
  float* C = reinterpret_cast<float*>(p.C);
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * W_; i < per; i += stride) {
    const int n = (int)(i % p.N);
    const long t = i / p.N;
    const int m = (int)(t % p.M);
    const int z = (int)(t / p.M);
    const long co = (long)(z / p.zdiv) * p.sC_hi + (long)(z % p.zdiv) * p.sC_lo + (long)m * p.ldc + n;
    if constexpr (VEC) {
      float4 a = *reinterpret_cast<const float4*>(W + i);
      for (int s = 1; s < p.splitk; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(W + (long)s * per + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      if (p.accumulate) {
        const float4 c = *reinterpret_cast<const float4*>(C + co);
        a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
      }
      *reinterpret_cast<float4*>(C + co) = a;
    } else {
      float a = 0.f;
      for (int s = 0; s < p.splitk; ++s) a += W[(long)s * per + i];
      C[co] = p.accumulate ? C[co] + a : a;
    }
  }
}


// ---- grouped weight-gradient GEMM ----------------------------------------------------------------------------------------
// One launch computes, for every problem i of a group,   dW_i[n][k] += sum_m dy_i[m][n] * x_i[m][k]   and
// db_i[n] += sum_m dy_i[m][n]: all weight (and bias) gradients of one encoder layer's backward in a single grid.
// Why grouped: each of these products has a tiny output (512x512 ... 2048x512 = 16..64 tiles of 128x128) and a long
// reduction (M = all frames of the batch, ~6000), so launched one by one they need split-K + a reduce pass + a column-sum
// launch each (round 1: ~30 launches and ~9 ms of side-stream kernel time per step).  Together the ~10 problems of a layer
// are 370 (128-row) / 740 (64-row) tiles: the grid fills the chip with NO split, every workgroup owns its output tile (plain
// fp32 read-modify-write, no atomics, no workspace) and walks the whole reduction.  Both operands are "k-strided" (the
// reduction index m is the slow axis of dy / x): staged global -> registers -> LDS with the 4(k) x 8(row) register transpose of
// gemm_bf16_kernel.  The bias gradient rides along: the workgroups of tile column 0 add up the dy values that pass through
// their staging registers.
struct WgradTable {
  int start[EA_WGRAD_MAX + 1];  // first workgroup of problem i ; start[count] = grid size
  int tiles_x[EA_WGRAD_MAX];    // column tiles (over K) of problem i
};

template <int BM_>
__global__ __launch_bounds__(256) void wgrad_group_kernel(const EaWgradGroup g, const WgradTable tb) {
  constexpr int NJ = BM_ == 128 ? 4 : 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * BM * ROW_BYTES];
  char* sA = smem;
  char* sB = smem + BM * ROW_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = BM_ == 128 ? (wave >> 1) : 0;
  const int wcol = BM_ == 128 ? (wave & 1) * 64 : wave * 32;

  int pi = 0;
  // workgroups are dealt round-robin to the 8 XCDs: hand every XCD one contiguous range of the (problem, row block, column
  // block) tile list instead, so that the tiles that walk the same dy / x columns share ONE L2 (PMC: 601 MB of L2 fills per
  // launch against 207 MB of operands when neighbouring tiles sat on 8 different XCDs)
  const int total = gridDim.x, xcd = blockIdx.x & 7, xq = total >> 3, xr = total & 7;
  const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  while (pi + 1 < g.count && bid >= tb.start[pi + 1]) ++pi;
  const EaWgradProblem P = g.p[pi];
  const int local = bid - tb.start[pi];
  const int tx = tb.tiles_x[pi];
  const int tile_y = local / tx, tile_x = local - tile_y * tx;
  const int R = P.N, Cn = P.K, Kr = P.M;  // output rows (dy columns), output columns (x columns), reduction length
  const int m0 = tile_y * BM_, n0 = tile_x * BN;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(P.dy);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(P.x);
  const long lda = P.ld_dy, ldb = P.ld_x;

  f32x4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  const int nk = (Kr + BK - 1) / BK;
  const bool a_ok = (m0 + BM_ <= R) && (lda & 7) == 0 && ((uintptr_t)A & 15) == 0;
  const bool b_ok = (n0 + BN <= Cn) && (ldb & 7) == 0 && ((uintptr_t)B & 15) == 0;
  const int a_kq = BM_ == 128 ? (tid >> 4) : (tid >> 3);
  const int a_rc = BM_ == 128 ? (tid & 15) : (tid & 7);
  const bool a_active = BM_ == 128 || tid < 128;
  const bf16_t* a_base = A + (long)(a_kq * 4) * lda + m0 + a_rc * 8;
  const bf16_t* b_base = B + (long)((tid >> 4) * 4) * ldb + n0 + (tid & 15) * 8;
  const bool do_bias = P.dbias != nullptr && tile_x == 0;
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

  auto loadA = [&](int k0) {
    if (!a_active) return;
    if (a_ok && k0 + BK <= Kr) load_ks_fast(a_base, lda, k0, ra);
    else load_ks_at(A, lda, R, Kr, m0 + a_rc * 8, k0 + a_kq * 4, ra);
  };
  auto loadB = [&](int k0) {
    if (b_ok && k0 + BK <= Kr) load_ks_fast(b_base, ldb, k0, rb);
    else load_ks(B, ldb, Cn, Kr, n0, k0, tid, rb);
  };
  if (nk > 0) { loadA(0); loadB(0); }
  uint32_t a_off[2][4], b_off[2][NJ];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[ks][i] = lds_off(wm * 64 + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
    for (int j = 0; j < NJ; ++j) b_off[ks][j] = lds_off(wcol + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
  }
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    if (a_active) {
      if (do_bias) {  // column sums of dy: the 4 (k) x 8 (row) block this thread is about to stage
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t w[4] = {ra[j].x, ra[j].y, ra[j].z, ra[j].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bsum[2 * e] += __uint_as_float(w[e] << 16);
            bsum[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
          }
        }
      }
      store_ks_at(sA, a_rc * 8, a_kq * 4, ra);
    }
    store_ks(sB, tid, rb);
    __syncthreads();
    if (kt + 1 < nk) { loadA((kt + 1) * BK); loadB((kt + 1) * BK); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[4], bfr[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sA + a_off[ks][i]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(sB + b_off[ks][j]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, af[i]),
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, bfr[j]), acc[i][j], 0, 0, 0);
    }
  }

  float* sC = reinterpret_cast<float*>(smem);
  if (do_bias) {  // fold the 16 k-quads that share a row: partial[kq][row] in LDS, one thread per row finishes
    __syncthreads();
    if (a_active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sC[a_kq * BM_ + a_rc * 8 + e] = bsum[e];
    }
    __syncthreads();
    if (tid < BM_ && m0 + tid < R) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) s += sC[q * BM_ + tid];
      P.dbias[m0 + tid] += s;
    }
  }
  // accumulators -> fp32 LDS tile (64 rows at a time) -> dW += tile (each output element belongs to exactly one workgroup)
  const bool vec_ok = (P.ldw & 3) == 0 && (((uintptr_t)P.dW) & 15) == 0;
#pragma unroll
  for (int half = 0; half < BM_ / 64; ++half) {
    __syncthreads();
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sC[(i * 16 + (lane >> 4) * 4 + r) * BN + wcol + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int rl = pass * 8 + (tid >> 5);
      const int m = m0 + half * 64 + rl;
      const int n = n0 + (tid & 31) * 4;
      if (m < R && n < Cn) {
        const float4 x = *reinterpret_cast<const float4*>(sC + rl * BN + (tid & 31) * 4);
        float* C = P.dW + (long)m * P.ldw + n;
        if (vec_ok && n + 4 <= Cn) {
          float4 c = *reinterpret_cast<const float4*>(C);
          c.x += x.x; c.y += x.y; c.z += x.z; c.w += x.w;
          *reinterpret_cast<float4*>(C) = c;
        } else {
          const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) if (n + e < Cn) C[e] += v[e];
        }
      }
    }
  }
}


// ---- grouped weight gradients, direct-to-LDS + transposing reads ----------------------------------------------------------
// Same contract as wgrad_group_kernel for problems whose tiles are whole (N % BM_ == 0, K % 128 == 0, 16-byte aligned rows).
// Both operand tiles are [64 reduction rows][BM_ or 128 columns] AS THEY LIE IN MEMORY: global_load_lds moves them (no staging
// registers, no register transpose, no ds_write), and the MFMA fragments — 8 consecutive reduction indices for one column —
// come out of ds_read_b64_tr_b16 (lane (g, j) passes the address of [8g + (j>>2)][c + 4(j&3)] and receives
// [8g + 0..3][c + j]; a second read 4 rows down completes the fragment).  A wave instruction of the load fills 1 KB = 4 (8)
// image rows; the 16-byte slot s of row r holds source slot s ^ f(r), f chosen so that the 8 rows x 32 bytes a half-wave
// reads hit 64 different banks (lane model + bank check: tools/emu_wgrad_tr.py, tests/test_kernel_models.py).  Rows past M
// are fetched from a zero page.  The bias gradient is one more MFMA per A fragment against a fragment of ones.
__device__ __attribute__((aligned(256))) unsigned char g_wgrad_zero_page[256];

__device__ __forceinline__ int wg_f256(int r) { return ((r & 3) | ((r >> 1) & 4)) << 1; }
__device__ __forceinline__ int wg_f128(int r) { return (((r >> 1) & 1) | ((r >> 2) & 2)) << 1; }
// transposing reads from inline asm with their own wait (through the builtin hipcc drains vmcnt — the prefetch — first)
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void wg_trr16(const uint32_t (&ad)[4], uint2 (&o)[16]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %16 offset:%20\n\tds_read_b64_tr_b16 %1, %16 offset:%21\n\t"
      "ds_read_b64_tr_b16 %2, %16 offset:%22\n\tds_read_b64_tr_b16 %3, %16 offset:%23\n\t"
      "ds_read_b64_tr_b16 %4, %17 offset:%20\n\tds_read_b64_tr_b16 %5, %17 offset:%21\n\t"
      "ds_read_b64_tr_b16 %6, %17 offset:%22\n\tds_read_b64_tr_b16 %7, %17 offset:%23\n\t"
      "ds_read_b64_tr_b16 %8, %18 offset:%20\n\tds_read_b64_tr_b16 %9, %18 offset:%21\n\t"
      "ds_read_b64_tr_b16 %10, %18 offset:%22\n\tds_read_b64_tr_b16 %11, %18 offset:%23\n\t"
      "ds_read_b64_tr_b16 %12, %19 offset:%20\n\tds_read_b64_tr_b16 %13, %19 offset:%21\n\t"
      "ds_read_b64_tr_b16 %14, %19 offset:%22\n\tds_read_b64_tr_b16 %15, %19 offset:%23\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]),
        "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15])
      : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "i"(O0), "i"(O1), "i"(O2), "i"(O3)
      : "memory");
}
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void wg_trr8(const uint32_t (&ad)[2], uint2 (&o)[8]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%10\n\tds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
      "ds_read_b64_tr_b16 %2, %8 offset:%12\n\tds_read_b64_tr_b16 %3, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %4, %9 offset:%10\n\tds_read_b64_tr_b16 %5, %9 offset:%11\n\t"
      "ds_read_b64_tr_b16 %6, %9 offset:%12\n\tds_read_b64_tr_b16 %7, %9 offset:%13\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(ad[0]), "v"(ad[1]), "i"(O0), "i"(O1), "i"(O2), "i"(O3)
      : "memory");
}
__device__ __forceinline__ bf16x8_t wg_cat(uint2 lo, uint2 hi) {
  const uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return __builtin_bit_cast(bf16x8_t, u);
}

template <int BM_, int NST>
__global__ __launch_bounds__(256, BM_ == 64 ? 3 : 2) void wgrad_group_tr_kernel(const EaWgradGroup g, const WgradTable tb) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];  // the ONLY LDS object (ring, then the fp32 output tile)
  constexpr int NJ = BM_ == 128 ? 4 : 2;
  constexpr int PA = BM_ * 2, PB = 256;                 // image row pitches (bytes)
  constexpr int A_BYTES = 64 * PA, STAGE = A_BYTES + 64 * PB;
  constexpr int NA = BM_ / 32, NB = 4;                  // load instructions per wave and stage
  constexpr int SPR_A = PA / 16, RPI_A = 1024 / PA;     // 16-byte slots per A row, A rows per load instruction
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = BM_ == 128 ? (wave >> 1) : 0;
  const int wcol = BM_ == 128 ? (wave & 1) * 64 : wave * 32;

  int pi = 0;
  const int total = gridDim.x, xcd = blockIdx.x & 7, xq = total >> 3, xr = total & 7;
  const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);  // see wgrad_group_kernel
  while (pi + 1 < g.count && bid >= tb.start[pi + 1]) ++pi;
  const EaWgradProblem P = g.p[pi];
  const int local = bid - tb.start[pi];
  const int tx = tb.tiles_x[pi];
  const int tile_y = local / tx, tile_x = local - tile_y * tx;
  const int R = P.N, Cn = P.K, Kr = P.M;
  const int m0 = tile_y * BM_, n0 = tile_x * BN;
  const long lda = P.ld_dy, ldb = P.ld_x;
  const int nk = (Kr + BK - 1) / BK;

  f32x4_t acc[4][NJ], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = P.dbias != nullptr && tile_x == 0;

  // per-lane source pointers of this wave's load instructions (advance by 64 rows per stage)
  const bf16_t* ap[NA];
  const bf16_t* bp[NB];
  int arow[NA], brow[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (wave + 4 * i) * RPI_A + lane / SPR_A, slot = lane % SPR_A;
    arow[i] = row;
    ap[i] = reinterpret_cast<const bf16_t*>(P.dy) + (long)row * lda + m0 + 8 * (slot ^ (BM_ == 128 ? wg_f256(row) : wg_f128(row)));
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (wave + 4 * i) * 4 + (lane >> 4), slot = lane & 15;
    brow[i] = row;
    bp[i] = reinterpret_cast<const bf16_t*>(P.x) + (long)row * ldb + n0 + 8 * (slot ^ wg_f256(row));
  }
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_wgrad_zero_page) + (lane & 15) * 8;
  auto issue = [&](int stage, int kt) {
    char* base = dsm + stage * STAGE + wave * 1024;
    const long roff = (long)kt * BK;
    if ((kt + 1) * BK <= Kr) {
#pragma unroll
      for (int i = 0; i < NA; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(ap[i] + roff * lda), (lptr_t)(base + i * 4096), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(bp[i] + roff * ldb), (lptr_t)(base + A_BYTES + i * 4096), 16, 0, 0);
    } else {  // last, partial stage: rows past M contribute zeros
#pragma unroll
      for (int i = 0; i < NA; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(kt * BK + arow[i] < Kr ? ap[i] + roff * lda : zero), (lptr_t)(base + i * 4096), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(kt * BK + brow[i] < Kr ? bp[i] + roff * ldb : zero), (lptr_t)(base + A_BYTES + i * 4096), 16,
                                         0, 0);
    }
  };
  // fragment addresses inside a stage (the swizzle term does not depend on the +4-row / +32-row immediates)
  const int g4 = lane >> 4, lj = lane & 15, le = lj >> 2, lq = lj & 3, row0 = 8 * g4 + le;
  const uint32_t lds0 = (uint32_t)(uintptr_t)dsm;
  uint32_t adA[4], adB[NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    adA[i] = lds0 + row0 * PA + (((((wm * 64 + 16 * i) >> 3) + (lq >> 1)) ^ (BM_ == 128 ? wg_f256(row0) : wg_f128(row0))) << 4) + (lq & 1) * 8;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    adB[j] = lds0 + A_BYTES + row0 * PB + (((((wcol + 16 * j) >> 3) + (lq >> 1)) ^ wg_f256(row0)) << 4) + (lq & 1) * 8;
  const uint4 ones_u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_u);

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s, s);
  // the k-loop exists twice: with and without the bias MFMAs (a per-MFMA-pair branch on the workgroup-uniform `do_bias` broke
  // the back-to-back MFMA issue of every tile, and only the tiles of column 0 ever take it)
  auto kloop = [&](auto bias_tag) {
  constexpr bool DO_BIAS = decltype(bias_tag)::value;
  int stage = 0, fill = NST - 1;
  for (int kt = 0; kt < nk; ++kt) {
    if (NST > 2 && kt + NST - 2 < nk) {
      if ((NST - 2) * (NA + NB) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if ((NST - 2) * (NA + NB) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // tile kt is in LDS for every wave; everyone is done reading the stage that is refilled next
    if (kt + NST - 1 < nk) issue(fill, kt + NST - 1);
    const uint32_t so = (uint32_t)(stage * STAGE);
    uint32_t a4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] = adA[i] + so;
    uint2 oa[16];
    wg_trr16<0, 4 * PA, 32 * PA, 36 * PA>(a4, oa);
    bf16x8_t bfr[2][NJ];
    if constexpr (BM_ == 128) {
      uint32_t b4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b4[j] = adB[j] + so;
      uint2 ob[16];
      wg_trr16<0, 4 * PB, 32 * PB, 36 * PB>(b4, ob);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[ks][j] = wg_cat(ob[j * 4 + 2 * ks], ob[j * 4 + 2 * ks + 1]);
    } else {
      uint32_t b2[2] = {adB[0] + so, adB[1] + so};
      uint2 ob[8];
      wg_trr8<0, 4 * PB, 32 * PB, 36 * PB>(b2, ob);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[ks][j] = wg_cat(ob[j * 4 + 2 * ks], ob[j * 4 + 2 * ks + 1]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8_t af = wg_cat(oa[i * 4 + 2 * ks], oa[i * 4 + 2 * ks + 1]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, af),
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, bfr[ks][j]), acc[i][j], 0, 0, 0);
        if constexpr (DO_BIAS)
          accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, af),
                                                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, ones), accb[i], 0, 0,
                                                            0);
      }
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
    fill = fill + 1 == NST ? 0 : fill + 1;
  }
  };
  if (do_bias) kloop(std::true_type{});
  else kloop(std::false_type{});

  if (do_bias && lj == 0 && wcol == 0) {  // every output row of the tile sits in column 0 of exactly one wave's bias accumulators
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = m0 + wm * 64 + i * 16 + g4 * 4 + r;
        if (n < R) P.dbias[n] += accb[i][r];
      }
  }
  float* sC = reinterpret_cast<float*>(dsm);
  const bool vec_ok = (P.ldw & 3) == 0 && (((uintptr_t)P.dW) & 15) == 0;
#pragma unroll
  for (int half = 0; half < BM_ / 64; ++half) {
    __syncthreads();
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sC[(i * 16 + g4 * 4 + r) * BN + wcol + j * 16 + lj] = acc[i][j][r];
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int rl = pass * 8 + (tid >> 5);
      const int m = m0 + half * 64 + rl;
      const int n = n0 + (tid & 31) * 4;
      if (m < R && n < Cn) {
        const float4 x = *reinterpret_cast<const float4*>(sC + rl * BN + (tid & 31) * 4);
        float* C = P.dW + (long)m * P.ldw + n;
        if (vec_ok && n + 4 <= Cn) {
          float4 c = *reinterpret_cast<const float4*>(C);
          c.x += x.x; c.y += x.y; c.z += x.z; c.w += x.w;
          *reinterpret_cast<float4*>(C) = c;
        } else {
          const float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) if (n + e < Cn) C[e] += v[e];
        }
      }
    }
  }
}

}  // namespace

// ---- optional live profiling of the dominant kernel (bench.py roofline): HIP events around every launch on
// the launch stream; flops = 2*M*N*K*batch per launch.
#include <vector>
struct GemmProf { hipEvent_t e0, e1; double flops, bytes; int M, N, K, batch, a_ks, b_ks, splitk, bm64, epi; };
static bool g_prof_on = false;
static std::vector<GemmProf> g_prof;

extern "C" int ea_gemm_profile_enable(int on) {
  g_prof_on = on != 0;
  if (on) {
    for (auto& r : g_prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    g_prof.clear();
  }
  return 0;
}
// total_ms / total_flops over the launches recorded since enable(1); returns the number of launches
extern "C" long ea_gemm_profile_read(double* total_ms, double* total_flops) {
  double ms = 0.0, fl = 0.0;
  for (auto& r : g_prof) {
    hipEventSynchronize(r.e1);
    float t = 0.f;
    hipEventElapsedTime(&t, r.e0, r.e1);
    ms += t;
    fl += r.flops;
  }
  *total_ms = ms;
  *total_flops = fl;
  return (long)g_prof.size();
}

// algorithmic bytes of the recorded launches: every operand read once, every output written once (A + B + C, plus the
// residual / auxiliary / second-output tensors of the fused epilogues; a split-K launch writes its fp32 slabs; a grouped
// weight-gradient launch reads dy and x once and read-modify-writes dW) — the denominator of bench.py's traffic ratio
extern "C" int ea_gemm_profile_bytes(double* total_bytes) {
  double b = 0.0;
  for (auto& r : g_prof) b += r.bytes;
  *total_bytes = b;
  return 0;
}

// one text line per recorded launch: M N K batch a_kstrided b_kstrided splitk bm64 epilogue-bits ms
extern "C" long ea_gemm_profile_dump(const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  for (auto& r : g_prof) {
    hipEventSynchronize(r.e1);
    float t = 0.f;
    hipEventElapsedTime(&t, r.e0, r.e1);
    fprintf(f, "%d %d %d %d %d %d %d %d %d %.6f\n", r.M, r.N, r.K, r.batch, r.a_ks, r.b_ks, r.splitk, r.bm64, r.epi, t);
  }
  fclose(f);
  return (long)g_prof.size();
}

static long g_nt_store_min_bytes = 0;  // > 0: bf16 outputs of at least this size are written with non-temporal stores (measured neutral on the bench step: off)
extern "C" long ea_set_gemm_nt_store_min_bytes(long bytes) {
  const long old = g_nt_store_min_bytes;
  g_nt_store_min_bytes = bytes;
  return old;
}
static inline int nt_flag(const EaGemmParams& q) {
  return (g_nt_store_min_bytes > 0 && !q.c_f32 && q.splitk <= 1 && 2L * q.M * q.N * q.batch >= g_nt_store_min_bytes) ? 2 : 0;
}
static int g_xcd_swizzle = 3;  // bit 0: direct-to-LDS kernel, bit 1: register-staged kernel; gated on the grid shape at the launch sites
extern "C" int ea_set_gemm_xcd_swizzle(int mask) {
  const int old = g_xcd_swizzle;
  g_xcd_swizzle = mask;
  return old;
}
static int g_gemm_variant = 0;  // 0: automatic tile height, 1: always 128-row tiles, 2: always 64-row tiles
extern "C" int ea_set_gemm_variant(int v) {
  const int old = g_gemm_variant;
  g_gemm_variant = v;
  return old;
}

// direct-to-LDS ring kernel for launches with both operands k-contiguous: 0 off, 1 automatic ring depth (3 stages when at most
// two workgroups land on a CU — long-K, few-tile launches such as the N = 512 projections, where a deeper ring replaces the
// latency hiding of co-resident workgroups: 41 -> 33 us in the 12-layer FFN chain — else 2), 2..4 forced depth
static int g_gemm_glds = [] { const char* e = getenv("EA_GEMM_GLDS"); return e ? atoi(e) : 1; }();  // (env: diagnostic override)
extern "C" int ea_set_gemm_glds(int stages) {
  const int old = g_gemm_glds;
  g_gemm_glds = stages;
  return old;
}
// every operand of the epilogue whole and vector-accessible: the lean instantiation (see epilogue_chunk<FAST>)
static int g_fast_epi = [] { const char* e = getenv("EA_GEMM_FAST_EPI"); return e ? atoi(e) : 1; }();  // (diagnostic A/B switch)
static bool fast_epilogue_ok(const EaGemmParams& q) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return g_fast_epi && !q.c_f32 && q.splitk == 1 && q.batch == 1 && !q.resid_f32 && q.N % 128 == 0 && (q.ldc & 7) == 0 && al(q.C) &&
         ((q.sC_hi | q.sC_lo) & 7) == 0 && (!q.bias || al(q.bias)) &&
         (!q.aux || ((q.ldaux & 7) == 0 && al(q.aux) && ((q.sX_hi | q.sX_lo) & 7) == 0)) &&
         (!q.resid || ((q.ldr & 7) == 0 && al(q.resid) && ((q.sR_hi | q.sR_lo) & 7) == 0)) &&
         (!q.C2 || ((q.ldc2 & 7) == 0 && al(q.C2))) &&
         (!q.q_u || ((q.ld_q & 7) == 0 && al(q.q_u) && al(q.q_v) && al(q.pos_u) && al(q.pos_v)));
}
// the 8-wave kernels' register epilogue also takes a plain (bias / activation / dropout only) product whose N is not a multiple of 128
// when the row pitch covers the last 32-column store group (the transducer joint's vocabulary projection: N = 5004, pitch 5056):
// the columns N .. pitch - 1 of a row are padding, whatever lands there is never read
static bool w8_ragged_ok(const EaGemmParams& q) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return g_fast_epi && !q.c_f32 && q.splitk == 1 && q.batch == 1 && !q.resid && !q.aux && !q.C2 && !q.q_u && q.N % 4 == 0 &&
         (q.ldc & 7) == 0 && q.ldc >= (q.N + 31) / 32 * 32 && al(q.C) && (!q.bias || al(q.bias));
}
template <int BM_, int NST>
static bool launch_glds(dim3 grid, hipStream_t stream, const EaGemmParams& q, int sw) {
  constexpr int bytes = NST * (BM_ + BN) * ROW_BYTES;
  static bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<BM_, NST>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<BM_, NST, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!attr_ok) return false;
  if (fast_epilogue_ok(q)) hipLaunchKernelGGL((gemm_glds_kernel<BM_, NST, true>), grid, dim3(256), bytes, stream, q, sw);
  else hipLaunchKernelGGL((gemm_glds_kernel<BM_, NST>), grid, dim3(256), bytes, stream, q, sw);
  return true;
}
static bool glds_eligible(const EaGemmParams& q) {
  return !q.a_kstrided && !q.b_kstrided && q.K % BK == 0 && q.kchunk % BK == 0 && (q.lda & 7) == 0 && (q.ldb & 7) == 0 &&
         ((q.sA_hi | q.sA_lo | q.sB_hi | q.sB_lo) & 7) == 0 && ((reinterpret_cast<uintptr_t>(q.A) | reinterpret_cast<uintptr_t>(q.B)) & 15) == 0;
}

template <bool A_KS, bool B_KS>
static void launch_gemm(dim3 grid, bool bm64, hipStream_t stream, const EaGemmParams& q) {
  // the remap helps when several n-tiles share a row block and the grid spans many row blocks (not for batched / split launches,
  // whose z index already separates the operands)
  const int sw = ((g_xcd_swizzle & 2) && grid.x <= 16 && (long)grid.x * grid.y * grid.z >= 64 ? 1 : 0) | nt_flag(q);
  if (!A_KS && !B_KS && fast_epilogue_ok(q)) {  // (the k-strided launches of the hot path are fp32 weight gradients: never FAST)
    if (bm64) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, 64, true>), grid, dim3(256), 0, stream, q, sw);
    else hipLaunchKernelGGL((gemm_bf16_kernel<false, false, 128, true>), grid, dim3(256), 0, stream, q, sw);
    return;
  }
  if (bm64) hipLaunchKernelGGL((gemm_bf16_kernel<A_KS, B_KS, 64>), grid, dim3(256), 0, stream, q, sw);
  else hipLaunchKernelGGL((gemm_bf16_kernel<A_KS, B_KS, 128>), grid, dim3(256), 0, stream, q, sw);
}

extern "C" long ea_gemm_splitk_workspace_bytes(int M, int N, int batch, int splitk) {
  return splitk > 1 ? (long)splitk * batch * M * N * (long)sizeof(float) : 0;
}

extern "C" int ea_gemm_bf16(const EaGemmParams* pp, hipStream_t stream) {
  const EaGemmParams& p = *pp;
  if (p.M <= 0 || p.N <= 0 || p.batch <= 0) return 0;
  if (p.K <= 0 || p.zdiv <= 0) return -2;
  EaGemmParams q = p;
  if (q.splitk < 1) q.splitk = 1;
  if (q.q_u) {
    if (q.qsplit_n <= 0 || q.qsplit_n % 128 || q.qsplit_n > q.N || q.N % 8 || q.batch != 1 || q.splitk > 1 || q.c_f32 || q.ld_q < q.qsplit_n)
      return -4;
  } else {
    q.qsplit_n = 0;
  }
  if (q.splitk > 1) {
    // partial sums are combined with fp32 atomics: only the plain accumulate epilogue is legal
    if (!q.c_f32 || q.bias || q.resid || q.aux || q.C2 || q.act != EA_ACT_NONE || q.drop_thr || !q.workspace) return -4;
    int chunk = (q.K + q.splitk - 1) / q.splitk;
    chunk = (chunk + BK - 1) / BK * BK;
    q.kchunk = chunk;
    q.splitk = (q.K + chunk - 1) / chunk;
  } else {
    q.kchunk = q.K;
  }
  // tile-height choice: 64-row tiles when 128-row tiles would leave the 256 CUs (x3 resident workgroups) under-filled
  const long tiles128 = (long)((q.N + BN - 1) / BN) * ((q.M + BM - 1) / BM) * q.batch * q.splitk;
  static const long bm_thr = [] { const char* e = getenv("EA_GEMM_BM_THR"); return e ? atol(e) : 1536L; }();  // (diagnostic override)
  const bool bm64 = g_gemm_variant == 2 ? true : (g_gemm_variant == 1 ? false : (tiles128 < bm_thr));
  const int bm = bm64 ? 64 : BM;
  dim3 grid((q.N + BN - 1) / BN, (q.M + bm - 1) / bm, q.batch * q.splitk), block(256);
  GemmProf pr;
  pr.bm64 = bm64;
  if (g_prof_on) {
    hipEventCreate(&pr.e0);
    hipEventCreate(&pr.e1);
    pr.flops = 2.0 * q.M * q.N * (double)q.K * q.batch;
    pr.M = q.M; pr.N = q.N; pr.K = q.K; pr.batch = q.batch; pr.a_ks = q.a_kstrided; pr.b_ks = q.b_kstrided;
    pr.splitk = q.splitk;
    pr.epi = (q.bias ? 1 : 0) | (q.resid ? 2 : 0) | (q.aux ? 4 : 0) | (q.C2 ? 8 : 0) | (q.act != EA_ACT_NONE ? 16 : 0) |
             (q.drop_thr ? 32 : 0) | (q.c_f32 ? 64 : 0);
    const double mn = (double)q.M * q.N * q.batch;
    pr.bytes = 2.0 * q.batch * ((double)q.M * q.K + (double)q.N * q.K) + mn * (q.splitk > 1 ? 4.0 * q.splitk : (q.c_f32 ? 4.0 : 2.0)) +
               (q.resid ? 2.0 * mn : 0.0) + (q.aux ? 2.0 * mn : 0.0) + (q.C2 ? 2.0 * mn : 0.0);
    hipEventRecord(pr.e0, stream);
  }
  bool done = false;
  const bool kc_ok = glds_eligible(q);
  if (kc_ok && (fast_epilogue_ok(q) || w8_ragged_ok(q)) && (!g_gemm_corun || g_w8_corun)) {  // one 512-thread workgroup per CU on a large tile when the grid fits the chip
    int cfg = 0;
    if (ea_gemm_w8_try(q, nt_flag(q), stream, &cfg)) {
      done = true;
      pr.bm64 = 100 + cfg;
    }
  }
  if (!done && g_gemm_glds && kc_ok) {
    // XCD-aware tile order when the whole B operand fits every XCD's L2 next to the streamed A rows (few n-tiles): measured
    // L2 hit rate 58 -> 82 % and -10..-20 % time on the N = 512 projections; slower for square problems (B no longer stationary)
    const int sw = (((g_xcd_swizzle & 1) && grid.x <= 16 && (long)grid.x * grid.y * grid.z >= 64 && (grid.x > 1 || grid.z > 1)) ? 1 : 0) | nt_flag(q);
    int nst = g_gemm_glds;
    if (nst == 1) {
      // many workgroups and a short reduction (the K = 512 projections: 8 k-tiles, 6 workgroups per CU): co-resident
      // workgroups already hide the latency and the ring's two-tile prologue costs more than it saves -> register-staged kernel
      const long blocks = (long)grid.x * grid.y * grid.z;
      const int nkt = q.kchunk / BK;
      nst = (blocks <= 512 && nkt >= 4) ? 3 : (nkt >= 16 ? 2 : 0);
    }
    if (nst == 0) {
    } else if (nst == 2) done = bm64 ? launch_glds<64, 2>(grid, stream, q, sw) : launch_glds<128, 2>(grid, stream, q, sw);
    else if (nst == 4) done = bm64 ? launch_glds<64, 4>(grid, stream, q, sw) : launch_glds<128, 4>(grid, stream, q, sw);
    else done = bm64 ? launch_glds<64, 3>(grid, stream, q, sw) : launch_glds<128, 3>(grid, stream, q, sw);
  }
  if (done) {
  } else if (p.a_kstrided) {
    if (p.b_kstrided) launch_gemm<true, true>(grid, bm64, stream, q); else launch_gemm<true, false>(grid, bm64, stream, q);
  } else {
    if (p.b_kstrided) launch_gemm<false, true>(grid, bm64, stream, q); else launch_gemm<false, false>(grid, bm64, stream, q);
  }
  if (g_prof_on) {  // the split-K reduce pass is a separate (HBM-bound) kernel and is not counted
    hipEventRecord(pr.e1, stream);
    g_prof.push_back(pr);
  }
  if (q.splitk > 1) {
    const long per = (long)q.batch * q.M * q.N;
    const bool vec = (q.N % 4 == 0) && (q.ldc % 4 == 0) && (q.sC_hi % 4 == 0) && (q.sC_lo % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(q.C) & 15) == 0) && ((reinterpret_cast<uintptr_t>(q.workspace) & 15) == 0);
    long blocks = (per / (vec ? 4 : 1) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (vec) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, q);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, q);
  }
  return EA_CHECK_LAUNCH();
}

static int g_wgrad_tr = [] { const char* e = getenv("EA_WGRAD_TR"); return e ? atoi(e) : 1; }();
extern "C" int ea_set_wgrad_transposing_reads(int on) {
  const int old = g_wgrad_tr;
  g_wgrad_tr = on;
  return old;
}
// All weight / bias gradients of one layer in one launch (see wgrad_group_kernel).
extern "C" int ea_wgrad_group(const EaWgradGroup* gp, hipStream_t stream) {
  const EaWgradGroup& g = *gp;
  if (g.count <= 0) return 0;
  if (g.count > EA_WGRAD_MAX) return -2;
  long tiles128 = 0;
  for (int i = 0; i < g.count; ++i) {
    const EaWgradProblem& p = g.p[i];
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.dy || !p.x || !p.dW) return -2;
    tiles128 += (long)((p.N + 127) / 128) * ((p.K + BN - 1) / BN);
  }
  // Tile height: 64-row tiles when 128-row tiles would leave CUs idle (each tile walks the whole reduction): fewer than ~1.25
  // 128-row tiles per CU.  Round 4, same box, same minute: the Conformer layer's group (380 tiles of 128 rows) with 128-row tiles
  // 13.79 vs 13.96 ms per step, GEMM-family time 9.0 vs 10.0 ms, live roofline 0.132 vs 0.118 (a 128-row tile sends a third less
  // through the CU's LDS port per flop, which is what the co-running main-stream GEMMs are short of); the Transformer layer's
  // group of the enc-dec recipe (192 tiles) 13.10 vs 12.94 ms the other way round.  (Round 3 measured the same switch at
  // +-0.2 ms across boxes with the old forward kernels and left it off.)
  static const int bm_env = [] { const char* e = getenv("EA_WGRAD_BM"); return e ? atoi(e) : 0; }();  // (diagnostic override)
  static const long wgrad_bm_thr = [] { const char* e = getenv("EA_WGRAD_BM_THR"); return e ? atol(e) : 320L; }();  // (128-row tiles from here)
  auto aligned_for = [&](int bm) {
    // (a ragged last tile may read the columns up to the next tile boundary when the row pitch covers them: those products only
    // reach output rows / columns >= N / K, which are never stored)
    for (int i = 0; i < g.count; ++i) {
      const EaWgradProblem& p = g.p[i];
      if (!((p.N + bm - 1) / bm * bm <= p.ld_dy && (p.K + BN - 1) / BN * BN <= p.ld_x && (p.ld_dy & 7) == 0 && (p.ld_x & 7) == 0 &&
            ((reinterpret_cast<uintptr_t>(p.dy) | reinterpret_cast<uintptr_t>(p.x)) & 15) == 0))
        return false;
    }
    return true;
  };
  bool tr_ok = g_wgrad_tr != 0;
  bool bm64 = g_gemm_variant == 2 ? true : (g_gemm_variant == 1 ? false : (tiles128 < wgrad_bm_thr));
  if (tr_ok) {
    // the direct-to-LDS kernel needs whole tiles inside the row pitch: the preferred height first, then the other one (the
    // transducer's output layer, pitch 5056 = 79 x 64, has no whole 128-row tiling: it must not drop to the register-staged
    // kernel — 3.2 ms against 2.7 ms per batch — just because its group is large enough to prefer 128 rows)
    bool tr64 = bm_env == 64 ? true : bm_env == 128 ? false : bm64;
    tr_ok = aligned_for(tr64 ? 64 : 128);
    if (!tr_ok && bm_env == 0) {
      tr64 = !tr64;
      tr_ok = aligned_for(tr64 ? 64 : 128);
    }
    if (tr_ok) bm64 = tr64;
  }
  const int bm = bm64 ? 64 : 128;
  WgradTable tb;
  int total = 0;
  for (int i = 0; i < g.count; ++i) {
    const EaWgradProblem& p = g.p[i];
    tb.start[i] = total;
    tb.tiles_x[i] = (p.K + BN - 1) / BN;
    total += ((p.N + bm - 1) / bm) * tb.tiles_x[i];
  }
  for (int i = g.count; i <= EA_WGRAD_MAX; ++i) tb.start[i] = total;
  for (int i = g.count; i < EA_WGRAD_MAX; ++i) tb.tiles_x[i] = 1;
  GemmProf pr;
  if (g_prof_on) {  // roofline accounting: one record per grouped launch (M = workgroups, N = problems, K = reduction length)
    hipEventCreate(&pr.e0);
    hipEventCreate(&pr.e1);
    pr.flops = 0.0;
    pr.bytes = 0.0;
    for (int i = 0; i < g.count; ++i) {
      pr.flops += 2.0 * g.p[i].M * (double)g.p[i].N * g.p[i].K;
      pr.bytes += 2.0 * g.p[i].M * ((double)g.p[i].N + g.p[i].K) + 8.0 * g.p[i].N * (double)g.p[i].K;
    }
    pr.M = total; pr.N = g.count; pr.K = g.p[0].M; pr.batch = 1; pr.a_ks = 1; pr.b_ks = 1; pr.splitk = 1; pr.bm64 = bm64; pr.epi = 128;
    hipEventRecord(pr.e0, stream);
  }
  // long reductions over chip-filling 256 x 256 tile grids (the transducer joint's slabs): the 8-wave kernel of wgrad_w8.hip
  EaWgradGroup rest;
  rest.count = 0;
  if (const int took = ea_wgrad_w8_try(g, stream, nullptr, &rest)) {
    if (g_prof_on) {
      for (int i = 0; i < rest.count; ++i) {  // (took == 2: the thin problems are accounted by the launch that does them)
        pr.flops -= 2.0 * rest.p[i].M * (double)rest.p[i].N * rest.p[i].K;
        pr.bytes -= 2.0 * rest.p[i].M * ((double)rest.p[i].N + rest.p[i].K) + 8.0 * rest.p[i].N * (double)rest.p[i].K;
      }
      pr.bm64 = 108;
      hipEventRecord(pr.e1, stream);
      g_prof.push_back(pr);
    }
    const int rc = EA_CHECK_LAUNCH();
    if (took == 2 && rc == 0) return ea_wgrad_group(&rest, stream);  // (thin problems only: the 8-wave kernel declines them)
    return rc;
  }
  // rows 16-byte aligned and whole tiles readable everywhere: direct-to-LDS kernel with transposing fragment reads
  if (tr_ok) {
    constexpr int lds64 = 2 * (64 * 128 + 64 * 256), lds128 = 2 * (64 * 256 + 64 * 256);
    static const bool attr_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_tr_kernel<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_tr_kernel<128, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds128) == hipSuccess;
    if (!attr_ok) tr_ok = false;
    else if (bm64) hipLaunchKernelGGL((wgrad_group_tr_kernel<64, 2>), dim3(total), dim3(256), lds64, stream, g, tb);
    else hipLaunchKernelGGL((wgrad_group_tr_kernel<128, 2>), dim3(total), dim3(256), lds128, stream, g, tb);
  }
  if (!tr_ok) {
    if (bm64) hipLaunchKernelGGL((wgrad_group_kernel<64>), dim3(total), dim3(256), 0, stream, g, tb);
    else hipLaunchKernelGGL((wgrad_group_kernel<128>), dim3(total), dim3(256), 0, stream, g, tb);
  }
  if (g_prof_on) {
    hipEventRecord(pr.e1, stream);
    g_prof.push_back(pr);
  }
  return EA_CHECK_LAUNCH();
}
