// 8-wavefront large-tile bf16 GEMM for gfx950 (round 5): the forward / data-gradient products of the encoder layers
// (fairseq/modules/conformer_layer.py:134-146 FFN, :79-101 pointwise convolutions; fairseq/modules/multihead_attention.py:650-688
// projections) when both operands are k-contiguous and the epilogue is the lean bf16 one (gemm_epilogue.h, FAST).
//
// Why a second kernel family.  The 4-wave kernels of gemm.hip move 24 KB (64 x 128 x 64 tile) through the CU's L2 -> LDS path per
// 1.05 MFLOP: at the ~50 B/clk a CU sustains from its L2 that is ~490 cycles per k-tile against 256 MFMA cycles per SIMD — exactly
// the 4 - 5 k-tiles per microsecond per CU the round-4 workgroup probe measured, whatever the loader or the residency.  The cure is
// arithmetic intensity per CU, not another loader: ONE 512-thread workgroup per CU owning a 256 x 256 (64 KB per k-tile for 8.4
// MFLOP: 31 B/clk at full MFMA rate) or 128 x 128 tile (N = 512 products: 196 tiles, one per CU, ring of four stages so that
// 96 KB are in flight per CU), two wavefronts per SIMD so that one wavefront's LDS reads hide under the other's MFMAs.
//
// Same products, same rounding points, same accumulation order along k (BK = 64, two 16x16x32 MFMAs per accumulator and k-tile)
// as the 4-wave kernels: outputs are bit-identical to theirs (checked by tests/test_gpu_parity.py).
//
// Where it is used (ea_gemm_w8_try; measurements in DESIGN.md 3.1b and profiles/r05_*): launches whose tile grid is ONE dispatch
// round on the chip (128 < tiles <= 256: the encoder layers' forward products at M ~ 6 200 rows) while nothing else runs GEMMs
// beside them (the layer runtime's co-run hint: backward keeps the 4-wave kernels, whose 24 - 72 KB workgroups still find room
// next to a weight-gradient launch), and launches of >= 1 024 tiles with N >= 1 024 (the transducer joint's vocabulary projection,
// also with a ragged N inside a padded row pitch).  8 - 24 % faster than the 4-wave kernels in isolation, 1.3 % of the update step:
// the step's GEMMs are bound by cold first tiles, the per-CU L2 -> LDS rate and lock-step epilogue writes, not by the k loop.
#include "common.h"
#include "espresso_amd.h"
#include "gemm_common.h"
#include "gemm_epilogue.h"
#include "gemm_w8_common.h"

// 0 = never, 1 = automatic (default), 2 .. 5 = forced tile configuration (diagnostic): 2 = 256x256 / 2 stages, 3 = 128x128 / 4 stages,
// 4 = 256x128 / 3 stages, 5 = 128x256 / 3 stages
static int g_gemm_w8 = [] { const char* e = getenv("EA_GEMM_W8"); return e ? atoi(e) : 1; }();
static long g_w8_many = [] { const char* e = getenv("EA_GEMM_W8_MANY"); return e ? atol(e) : 1024L; }();  // tiles from which "many rounds" applies

namespace {

// ---- epilogue from registers ------------------------------------------------------------------------------------------------
// The k loop multiplies with the operands swapped (weight fragment first), so that a lane holds FOUR CONSECUTIVE COLUMNS of one
// output row:  acc[i][j][r] = C[m0 + row0 + i*16 + (lane & 15)][n0 + col0 + j*16 + (lane >> 4)*4 + r]   (a x b = b x a exactly and
// the k order inside the MFMA does not depend on which operand is which: the sums are bit-identical to the un-swapped product).
// The fused epilogue is applied in that layout — no fp32 bounce through LDS, no barrier: 8 workgroup barriers, 256 KB of LDS
// writes + reads and sixteen serial passes per 256 x 256 tile took 9.5 us of a 22 us workgroup (probe) — and two neighbouring MFMA
// tiles exchange halves with v_permlane16_swap so that every lane stores (and loads the residual / auxiliary operand as) 16
// bytes: after swap(tile j, tile j+1) a lane of 16-lane group g holds columns (j + (g & 1)) * 16 + (g >> 1) * 8 .. + 8.
// The arithmetic is epilogue_chunk<FAST>'s, statement for statement, specialised at compile time by KIND.
enum { W8_PLAIN = 0, W8_QSPLIT = 1, W8_ACT2 = 2, W8_AUX = 3, W8_RESID = 4 };

// keep-scales of 4 consecutive elements: bit for bit ea_keep(seed, idx0 + e) (see ea_keep8)
__device__ __forceinline__ void ea_keep4(uint64_t seed, uint64_t idx0, uint32_t thr, float inv_keep, float (&k)[4]) {
  const uint32_t lo = (uint32_t)idx0;
  if (__builtin_expect(lo > 0xFFFFFFFBu, 0)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) k[e] = ea_keep(seed, idx0 + (uint64_t)e, thr, inv_keep);
    return;
  }
  const uint32_t hi = ((uint32_t)(idx0 >> 32) * 0x85EBCA77u) ^ (uint32_t)(seed >> 32) ^ ((uint32_t)seed * 0xC2B2AE3Du);
  const uint32_t x0 = lo * 0x9E3779B1u + (uint32_t)seed;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t x = x0 + (uint32_t)e * 0x9E3779B1u;
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= hi;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    k[e] = x >= thr ? inv_keep : 0.f;
  }
}
__device__ __forceinline__ void unpack4_bf16(uint32_t w0, uint32_t w1, float (&o)[4]) {
  o[0] = __uint_as_float(w0 << 16); o[1] = __uint_as_float(w0 & 0xffff0000u);
  o[2] = __uint_as_float(w1 << 16); o[3] = __uint_as_float(w1 & 0xffff0000u);
}
template <int KIND, int MI, int NJ>
__device__ __forceinline__ void w8_reg_epilogue(const EaGemmParams& p, const f32x4_t (&acc)[MI][NJ], int m0, int n0, int row0, int col0, bool nt) {
  static_assert(NJ % 2 == 0, "tiles are stored in pairs");
  const int lane = threadIdx.x & 63, g = lane >> 4, r16 = lane & 15;
  const int nown = n0 + col0 + g * 4;                         // + j*16: the 4 columns this lane holds of tile j
  const int npair = n0 + col0 + (g & 1) * 16 + (g >> 1) * 8;  // + jp*32: the 8 columns this lane stores of tile pair jp
  const bool has_drop = p.drop_thr != 0;
  float bias4[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && nown + j * 16 < p.N) b = *reinterpret_cast<const float4*>(p.bias + nown + j * 16);
    bias4[j][0] = b.x; bias4[j][1] = b.y; bias4[j][2] = b.z; bias4[j][3] = b.w;
  }
  // positional biases of the query columns (QSPLIT), loaded once like the bias
  float posu4[KIND == W8_QSPLIT ? NJ : 1][4], posv4[KIND == W8_QSPLIT ? NJ : 1][4];
  if (KIND == W8_QSPLIT) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float4 u = make_float4(0.f, 0.f, 0.f, 0.f), w = u;
      if (nown + j * 16 < p.qsplit_n) {
        if (p.pos_u) u = *reinterpret_cast<const float4*>(p.pos_u + nown + j * 16);
        if (p.pos_v) w = *reinterpret_cast<const float4*>(p.pos_v + nown + j * 16);
      }
      posu4[j][0] = u.x; posu4[j][1] = u.y; posu4[j][2] = u.z; posu4[j][3] = u.w;
      posv4[j][0] = w.x; posv4[j][1] = w.y; posv4[j][2] = w.z; posv4[j][3] = w.w;
    }
  }
  // residual / auxiliary operand: the 16-byte pieces of row group i + 1 are requested before row group i is finished (the loads
  // of a row group issued where they are used cost one memory round trip per 32 columns: 14 us for a 256 x 256 tile, probe)
  constexpr bool HAS_OP = KIND == W8_AUX || KIND == W8_RESID;
  const bf16_t* op_base = KIND == W8_AUX ? reinterpret_cast<const bf16_t*>(p.aux) : reinterpret_cast<const bf16_t*>(p.resid);
  const long op_ld = KIND == W8_AUX ? p.ldaux : p.ldr;
  uint4 opn[HAS_OP ? NJ / 2 : 1], opc[HAS_OP ? NJ / 2 : 1];
  auto fetch_ops = [&](int i) {
    const int mc = min(m0 + row0 + i * 16 + r16, p.M - 1);
#pragma unroll
    for (int jp = 0; jp < NJ / 2; ++jp) {
      const int nc = min(npair + jp * 32, p.N - 8);  // (an absent pair of tiles re-reads a valid address; its values are not used)
      opn[jp] = *reinterpret_cast<const uint4*>(op_base + (long)mc * op_ld + nc);
    }
  };
  if (HAS_OP) fetch_ops(0);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + row0 + i * 16 + r16;
    const bool mv = m < p.M;
    if (HAS_OP) {
#pragma unroll
      for (int jp = 0; jp < NJ / 2; ++jp) opc[jp] = opn[jp];
      if (i + 1 < MI) fetch_ops(i + 1);
    }
#pragma unroll
    for (int jp = 0; jp < NJ / 2; ++jp) {
      if (n0 + col0 + jp * 32 >= p.N) continue;  // (uniform per wavefront; a pair that straddles N stores into the row's padding: w8_ragged_ok)
      const int ns = npair + jp * 32;
      // operand of the epilogue in the stored layout -> swap -> this lane's 4 columns of tile 2jp (.x/.y of r0) and 2jp+1 (r1)
      float opnd[2][4];
      if (HAS_OP) {
        const uint4 L = opc[jp];
        const auto s0 = __builtin_amdgcn_permlane16_swap(L.x, L.z, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(L.y, L.w, false, false);
        unpack4_bf16(s0[0], s1[0], opnd[0]);
        unpack4_bf16(s0[1], s1[1], opnd[1]);
      }
      uint2 out[2], out2[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int j = 2 * jp + t;
        const int n = nown + j * 16;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * p.alpha + bias4[j][e];
        if (KIND == W8_QSPLIT && n0 + col0 + jp * 32 < p.qsplit_n) {
          // rel-pos query columns (whole 128-column groups): q rounded to bf16 as the plain epilogue would store it, then the two
          // biased, scaled copies the attention kernels read
          float u4[4], b4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float q = __uint_as_float((uint32_t)f2bf(v[e]) << 16);
            u4[e] = (q + posu4[KIND == W8_QSPLIT ? j : 0][e]) * p.qscale;
            b4[e] = (q + posv4[KIND == W8_QSPLIT ? j : 0][e]) * p.qscale;
          }
          out[t] = make_uint2(pack_bf2(u4[0], u4[1]), pack_bf2(u4[2], u4[3]));
          out2[t] = make_uint2(pack_bf2(b4[0], b4[1]), pack_bf2(b4[2], b4[3]));
          continue;
        }
        float keep4[4];
        if (has_drop) {
          ea_keep4(p.drop_seed, (uint64_t)m * (uint64_t)p.N + (uint64_t)n, p.drop_thr, p.drop_scale, keep4);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) keep4[e] = 1.f;
        }
        if (KIND == W8_AUX) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= keep4[e];
          if (p.act == EA_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= dsilu_f(opnd[t][e]);
          } else if (p.act == EA_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= opnd[t][e] > 0.f ? 1.f : 0.f;
          }
        } else {
          if (KIND == W8_ACT2) out2[t] = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));  // the pre-activation copy
          if (p.act == EA_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
          } else if (p.act == EA_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= keep4[e];
        }
        if (KIND != W8_ACT2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
          if (KIND == W8_RESID) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += opnd[t][e];
          }
        }
        out[t] = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
      }
      const u32x4_t o = w8_swap_pair(out[0], out[1]);
      if (KIND == W8_QSPLIT && n0 + col0 + jp * 32 < p.qsplit_n) {
        const u32x4_t o2 = w8_swap_pair(out2[0], out2[1]);
        if (mv) {
          w8_store16(reinterpret_cast<bf16_t*>(p.q_u) + (long)m * p.ld_q + ns, o, nt);
          if (p.q_v) w8_store16(reinterpret_cast<bf16_t*>(p.q_v) + (long)m * p.ld_q + ns, o2, nt);
        }
      } else if (KIND == W8_ACT2) {
        // the pre-activation copy is only read again by the backward pass: streamed past the caches (as epilogue_chunk does)
        const u32x4_t o2 = w8_swap_pair(out2[0], out2[1]);
        if (mv) {
          w8_store16(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + ns, o2, true);
          w8_store16(reinterpret_cast<bf16_t*>(p.C2) + (long)m * p.ldc2 + ns, o, nt);
        }
      } else if (mv) {
        w8_store16(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + ns, o, nt);
      }
    }
  }
}

// grid: 1-D, one workgroup per output tile (tiles_n fastest); flags bit 0: XCD-aware tile order, bit 1: non-temporal stores
template <int BM_, int BN_, int WM_, int WN_, int NST, int PIPE, int KIND>
__global__ __launch_bounds__(512, 2) void gemm_w8_kernel(const EaGemmParams p, const int tiles_n, const int flags) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];  // the ONLY LDS object: ring of operand stages
  constexpr int TM = BM_ / WM_, TN = BN_ / WN_, MI = TM / 16, NJ = TN / 16;
  constexpr int A_BYTES = BM_ * 128, STAGE = (BM_ + BN_) * 128;
  constexpr int NA = BM_ / 64, NB = BN_ / 64;  // load instructions per wavefront and stage (one instruction = 8 rows of 128 B)
  static_assert(WM_ * WN_ == 8 && NST >= 2, "eight wavefronts");
  EA_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN_, wn = wave % WN_;
  int tile = blockIdx.x;
  if (flags & 1) {
    // workgroups go round-robin to the 8 XCDs in dispatch order: give every XCD one contiguous range of the (row block, n-tile)
    // order instead, so that the n-tiles of a row block share their operand rows in ONE L2 (bijective for any grid size)
    const int total = gridDim.x, xcd = tile & 7, q = total >> 3, r = total & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (tile >> 3);
  }
  const int tile_y = tile / tiles_n, tile_x = tile - tile_y * tiles_n;
  const int m0 = tile_y * BM_, n0 = tile_x * BN_;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  const int nk = p.K / BK;

  f32x4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // per-lane source pointers: lane l of load instruction i fills row (wave + 8 i) * 8 + (l >> 3), slot l & 7, which holds k-chunk
  // slot ^ (row & 7); rows past M / N re-read the last valid row (their products are never stored)
  const int src_chunk = (lane & 7) ^ (lane >> 3);
  const bf16_t* ap[NA];
  const bf16_t* bp[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) ap[i] = A + (long)min(m0 + (wave + 8 * i) * 8 + (lane >> 3), p.M - 1) * p.lda + src_chunk * 8;
#pragma unroll
  for (int i = 0; i < NB; ++i) bp[i] = B + (long)min(n0 + (wave + 8 * i) * 8 + (lane >> 3), p.N - 1) * p.ldb + src_chunk * 8;
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
  // fragment base addresses inside a stage: row (lane & 15) of the wavefront's first 16-row group, k-chunk ks * 4 + (lane >> 4); the
  // other 16-row groups are +2048-byte immediates
  uint32_t a_base[2], b_base[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_base[ks] = w8_off(wm * TM + (lane & 15), ks * 4 + (lane >> 4));
    b_base[ks] = A_BYTES + w8_off(wn * TN + (lane & 15), ks * 4 + (lane >> 4));
  }

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nk) issue(s, s);
  int stage = 0, fill = NST - 1;  // stage holding tile kt ; stage that tile kt + NST - 1 goes to
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most NST - 2 younger tiles (NA + NB instructions each) are outstanding
    if (kt + NST - 2 < nk) w8_wait_vmcnt<(NST - 2) * (NA + NB)>();
    else w8_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // every wavefront's part of tile kt is in LDS; everyone is done reading tile kt - 1's stage
    if (kt == 0) EA_STAMP(1);
    if (kt == 4) EA_STAMP(7);
    if (kt + NST - 1 < nk) issue(fill, kt + NST - 1);
    const char* st = dsm + stage * STAGE;
    if constexpr (PIPE == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t af[MI], bfr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(st + b_base[ks] + j * 2048);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(st + a_base[ks] + i * 2048);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, bfr[j]),
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, af[i]), acc[i][j], 0, 0, 0);
      }
    } else {
      // Software pipeline inside the k-tile: the MFMAs are cut into groups of eight (AG A fragments x NJ B fragments); the
      // fragment reads of group g + 1 are issued before the MFMAs of group g (hipcc on its own reads two A fragments, waits for
      // them, issues eight MFMAs, reads the next two ...: every group then pays one LDS round trip in the open).  The order is
      // pinned with sched_group_barrier; the compiler still counts lgkmcnt itself.
      constexpr int AG = NJ >= 4 ? 2 : (MI >= 4 ? 4 : MI);
      constexpr int GPK = MI / AG, NG = 2 * GPK;
      bf16x8_t bq[2][NJ], aq[2][AG];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bq[0][j] = *reinterpret_cast<const bf16x8_t*>(st + b_base[0] + j * 2048);
#pragma unroll
      for (int a = 0; a < AG; ++a) aq[0][a] = *reinterpret_cast<const bf16x8_t*>(st + a_base[0] + a * 2048);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
          const int ks1 = (g + 1) / GPK, gi = (g + 1) % GPK;
          if (gi == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) bq[ks1][j] = *reinterpret_cast<const bf16x8_t*>(st + b_base[ks1] + j * 2048);
          }
#pragma unroll
          for (int a = 0; a < AG; ++a) aq[(g + 1) & 1][a] = *reinterpret_cast<const bf16x8_t*>(st + a_base[ks1] + (gi * AG + a) * 2048);
        }
        const int ks = g / GPK, gi0 = g % GPK;
#pragma unroll
        for (int a = 0; a < AG; ++a)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[gi0 * AG + a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, bq[ks][j]),
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, aq[g & 1][a]), acc[gi0 * AG + a][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, NJ + AG, 0);
      w8_sched_groups<NG, GPK, AG, NJ>();
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
    fill = fill + 1 == NST ? 0 : fill + 1;
  }
  EA_STAMP(2);
  w8_reg_epilogue<KIND, MI, NJ>(p, acc, m0, n0, wm * TM, wn * TN, (flags & 2) != 0);
  EA_STAMP(3);
}

template <int BM_, int BN_, int WM_, int WN_, int NST, int PIPE, int KIND>
bool w8_launch_kind(const EaGemmParams& q, int flags, hipStream_t stream) {
  constexpr int bytes = NST * (BM_ + BN_) * 128;
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w8_kernel<BM_, BN_, WM_, WN_, NST, PIPE, KIND>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!attr_ok) return false;
  const int tiles_n = (q.N + BN_ - 1) / BN_, tiles_m = (q.M + BM_ - 1) / BM_;
  hipLaunchKernelGGL((gemm_w8_kernel<BM_, BN_, WM_, WN_, NST, PIPE, KIND>), dim3(tiles_n * tiles_m), dim3(512), bytes, stream, q, tiles_n, flags);
  return true;
}
// which specialisation of the register epilogue a (FAST) launch takes; -1: a combination none of them covers
int w8_kind(const EaGemmParams& q) {
  if (q.q_u) {
    // (measured in the step, round 5: 35.0 us against 31.5 us for the 4-wave kernel — this launch runs next to the side stream's
    // keep-bits kernel; the specialisation stays for forced configurations)
    static const bool qs = [] { const char* e = getenv("EA_GEMM_W8_QSPLIT"); return e && e[0] == '1'; }();  // (A/B switch)
    if (g_gemm_w8 == 1 && !qs) return -1;
    return (q.aux || q.C2 || q.resid || q.drop_thr || q.act != EA_ACT_NONE || q.out_scale != 1.f) ? -1 : W8_QSPLIT;
  }
  if (q.C2) return (q.aux || q.resid) ? -1 : W8_ACT2;
  if (q.aux) return q.resid ? -1 : W8_AUX;
  return q.resid ? W8_RESID : W8_PLAIN;
}
template <int BM_, int BN_, int WM_, int WN_, int NST, int PIPE>
bool w8_launch(const EaGemmParams& q, int flags, hipStream_t stream) {
  switch (w8_kind(q)) {
    case W8_PLAIN: return w8_launch_kind<BM_, BN_, WM_, WN_, NST, PIPE, W8_PLAIN>(q, flags, stream);
    case W8_QSPLIT: return w8_launch_kind<BM_, BN_, WM_, WN_, NST, PIPE, W8_QSPLIT>(q, flags, stream);
    case W8_ACT2: return w8_launch_kind<BM_, BN_, WM_, WN_, NST, PIPE, W8_ACT2>(q, flags, stream);
    case W8_AUX: return w8_launch_kind<BM_, BN_, WM_, WN_, NST, PIPE, W8_AUX>(q, flags, stream);
    case W8_RESID: return w8_launch_kind<BM_, BN_, WM_, WN_, NST, PIPE, W8_RESID>(q, flags, stream);
    default: return false;
  }
}

}  // namespace

extern "C" int ea_set_gemm_w8(int mode) {
  const int old = g_gemm_w8;
  g_gemm_w8 = mode;
  return old;
}

// Called by ea_gemm_bf16 for launches that are glds-eligible (both operands k-contiguous, K % 64 == 0, aligned) and FAST (lean bf16
// epilogue, batch 1, no split-K).  Returns 1 when the launch was taken, 0 to fall through to the 4-wave kernels.  *cfg_out = the
// configuration used (for the profiling records).
int ea_gemm_w8_try(const EaGemmParams& q, int nt_flag, hipStream_t stream, int* cfg_out) {
  if (!g_gemm_w8) return 0;
  if ((long)q.M * q.lda >= (1L << 31) || (long)q.N * q.ldb >= (1L << 31)) return 0;
  int cfg = g_gemm_w8;
  if (cfg == 1) {
    // One 512-thread workgroup per CU: the largest tile whose grid is a single dispatch round that fills more than half the chip
    // (probe, M = 6240: 256 x 256 for N = 2048 / 1536 — 200 / 150 tiles; 256 x 128 for N = 1024 — 200; 128 x 128 with a ring of
    // four for the N = 512 products — 196).  Anything else stays with the 4-wave kernels, whose several workgroups per CU overlap
    // one tile's epilogue with another's k loop.
    const long rm256 = (q.M + 255) / 256, rm128 = (q.M + 127) / 128, cn256 = (q.N + 255) / 256, cn128 = (q.N + 127) / 128;
    auto fits = [](long tiles) { return tiles > 128 && tiles <= 256; };
    const bool whole = q.N % 128 == 0;  // (a ragged N is only taken in the many-round case below)
    if (whole && fits(rm256 * cn256)) cfg = 2;
    else if (whole && fits(rm256 * cn128)) cfg = 4;
    else if (whole && fits(rm128 * cn256)) cfg = 5;
    else if (whole && fits(rm128 * cn128)) cfg = 3;
    // Many dispatch rounds (the transducer joint's vocabulary projection at 70 000 lattice rows: 5 480 tiles of 256 x 256): the round quantisation and the lock-step epilogues of the single-round case wash out, what counts is
    // operand bytes per flop — forward 722 -> 489 us isolated (profiles/r05_joint_gemm_probe.txt); the data gradient (N = 512,
    // K = 5056) gains nothing from any 8-wave tile (417 -> 406 .. 460 us) and stays with the 4-wave kernel
    else if (rm256 * cn256 >= g_w8_many && q.N >= 1024) cfg = 2;
    else return 0;
  }
  const int flags = 1 | nt_flag;
  bool ok = false;
  switch (cfg) {
    case 2: ok = w8_launch<256, 256, 2, 4, 2, 1>(q, flags, stream); break;
    case 3: ok = w8_launch<128, 128, 2, 4, 4, 1>(q, flags, stream); break;
    case 4: ok = w8_launch<256, 128, 4, 2, 3, 1>(q, flags, stream); break;
    case 5: ok = w8_launch<128, 256, 2, 4, 3, 1>(q, flags, stream); break;
    default: return 0;
  }
  if (ok && cfg_out) *cfg_out = cfg;
  return ok ? 1 : 0;
}
