// Persistent bf16 MFMA GEMM for gfx950 (MI355X): both operands k-contiguous, K % 64 == 0 — every forward Linear / pointwise
// convolution and (through the k-contiguous weight copies) every data-gradient GEMM of the encoder layers
// (reference: fairseq/modules/conformer_layer.py:134-146, multihead_attention.py:650-688, the autograd of F.linear).
//
// Why a second kernel.  The hot path's GEMMs are M ~ 6000 x N = 512..2048 x K = 512..2048: 200-1500 tiles of 8-32 k-steps.
// Measured on the round-1 kernels (tools/bench_gemm_ksweep.py): 8-22 us of every launch is fixed cost — each workgroup pays a
// cold prologue (first operand tiles from L2/HBM), an epilogue that nothing overlaps and a drain of its stores before it
// retires, two or three rounds of that per CU — and inside the loop a 64x128 tile moves 24 KB into LDS per k-step, which the
// CU's load path (~35-40 B/clk) cannot feed faster than ~700 cycles.  This kernel:
//   * is PERSISTENT: one workgroup per CU (8 wavefronts) walks a static list of tiles; the k-steps of all its tiles form ONE
//     stream through a ring of NST LDS stages filled by global_load_lds (no staging registers), so the first operand tiles of
//     tile i+1 are already in flight while tile i runs its epilogue, and no store is waited for until the kernel ends;
//   * uses tiles up to 256x128 (each wavefront 64x64 = 4x4 MFMA 16x16x32 tiles): 48 KB of operands per k-step for twice the
//     math of two 64x128 tiles (36 KB less traffic per 128x... pair), one barrier per k-step, counted s_waitcnt vmcnt so NST-2
//     stages stay in flight across the barrier;
//   * writes the accumulators STRAIGHT from registers: the MFMA operands are swapped (C^T = B A^T), so a lane holds four
//     consecutive output columns of one row -> 8-byte bf16 / 16-byte fp32 accesses, no fp32 bounce through LDS (round 1: 29 %
//     of the LDS cycles were bank conflicts of that bounce) and the ring stays untouched during the epilogue;
//   * hands tiles to workgroups XCD-aware: the tile list is cut into 8 contiguous ranges, one per XCD (workgroup g runs on
//     XCD g % 8), so the workgroups sharing an L2 work on neighbouring tiles (same A row block / adjacent B panels).
// Same fused epilogues, dropout hash indices, split-K slabs and batch addressing as gemm_bf16_kernel (gemm.hip).
#include <hip/hip_runtime.h>

#include <utility>

#include "gemm_common.h"

namespace {

struct PkSched {
  int tiles_m, tiles_n;  // tile grid of one (batch item, k-chunk) slice
  int units;             // tiles_m * tiles_n * batch * splitk
  int nk;                // k-steps (of 64) per unit
  int per_x;             // workgroups per XCD (gridDim.x / 8)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// compile-time loop: the body is instantiated once per index, so the accumulator tiles it touches are addressed with constant
// indices (a loop the optimizer declines to unroll would index acc[][] at run time and push all 64 accumulators to scratch)
template <int... Is, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// Epilogue families (compile-time, chosen on the host from the parameter block): they only prune code, the arithmetic is that of
// gemm.hip's epilogue_chunk.
enum { EPI_GEN = 0, EPI_AUX = 1, EPI_C2 = 2, EPI_F32 = 3 };

// epilogue on 16 consecutive output columns (n .. n+15) of row m; v = raw accumulators
template <int KIND>
__device__ __forceinline__ void epilogue16(const EaGemmParams& p, int z, int ks_id, int zhi, int zlo, long coff, int m, int n,
                                           float (&v)[16], const float (&bias)[16]) {
  const int cnt = min(16, p.N - n);
  const bool full = cnt == 16;
  const bool has_drop = p.drop_thr != 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = v[e] * p.alpha + bias[e];
  const uint64_t didx = ((uint64_t)z * (uint64_t)p.M + (uint64_t)m) * (uint64_t)p.N + (uint64_t)n;
  const long co = coff + (long)m * p.ldc + n;
  auto ldf = [&](const float* q, float (&o)[16]) {
    if (full && ((((uintptr_t)q) & 15) == 0)) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 x = *reinterpret_cast<const float4*>(q + 4 * t);
        o[4 * t] = x.x; o[4 * t + 1] = x.y; o[4 * t + 2] = x.z; o[4 * t + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = e < cnt ? q[e] : 0.f;
    }
  };
  auto stf = [&](float* q, const float (&x)[16]) {
    if (full && ((((uintptr_t)q) & 15) == 0)) {
#pragma unroll
      for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(q + 4 * t) = make_float4(x[4 * t], x[4 * t + 1], x[4 * t + 2], x[4 * t + 3]);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) if (e < cnt) q[e] = x[e];
    }
  };
  auto ldh = [&](const bf16_t* q, float (&o)[16]) {
    if (full && ((((uintptr_t)q) & 15) == 0)) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint4 u = *reinterpret_cast<const uint4*>(q + 8 * t);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[8 * t + 2 * e] = __uint_as_float(w[e] << 16);
          o[8 * t + 2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) o[e] = e < cnt ? bf2f(q[e]) : 0.f;
    }
  };
  auto sth = [&](bf16_t* q, const float (&x)[16]) {
    if (full && ((((uintptr_t)q) & 15) == 0)) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint4 u;
        u.x = pack_bf2(x[8 * t], x[8 * t + 1]); u.y = pack_bf2(x[8 * t + 2], x[8 * t + 3]);
        u.z = pack_bf2(x[8 * t + 4], x[8 * t + 5]); u.w = pack_bf2(x[8 * t + 6], x[8 * t + 7]);
        *reinterpret_cast<uint4*>(q + 8 * t) = u;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) if (e < cnt) q[e] = f2bf(x[e]);
    }
  };
  if (KIND == EPI_F32 && p.splitk > 1) {  // fp32 partial slab [ks][batch][M][N] (dense), combined by splitk_reduce_kernel
    stf(reinterpret_cast<float*>(p.workspace) + (((long)ks_id * p.batch + z) * p.M + m) * (long)p.N + n, v);
    return;
  }
  if constexpr (KIND == EPI_AUX) {
    float zz[16];
    ldh(reinterpret_cast<const bf16_t*>(p.aux) + (long)zhi * p.sX_hi + (long)zlo * p.sX_lo + (long)m * p.ldaux + n, zz);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (has_drop) v[e] *= ea_keep(p.drop_seed, didx + e, p.drop_thr, p.drop_scale);
      v[e] *= apply_dact(zz[e], p.act);
    }
  } else {
    if constexpr (KIND == EPI_C2) {
      sth(reinterpret_cast<bf16_t*>(p.C) + co, v);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v[e] = apply_act(v[e], p.act);
        if (has_drop) v[e] *= ea_keep(p.drop_seed, didx + e, p.drop_thr, p.drop_scale);
      }
      sth(reinterpret_cast<bf16_t*>(p.C2) + coff + (long)m * p.ldc2 + n, v);
      return;
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v[e] = apply_act(v[e], p.act);
        if (has_drop) v[e] *= ea_keep(p.drop_seed, didx + e, p.drop_thr, p.drop_scale);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] *= p.out_scale;
  if (p.resid) {
    const long ro = (long)zhi * p.sR_hi + (long)zlo * p.sR_lo + (long)m * p.ldr + n;
    float rr[16];
    if (p.resid_f32) ldf(reinterpret_cast<const float*>(p.resid) + ro, rr);
    else ldh(reinterpret_cast<const bf16_t*>(p.resid) + ro, rr);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] += rr[e];
  }
  if constexpr (KIND == EPI_F32) {
    float* C = reinterpret_cast<float*>(p.C) + co;
    if (p.accumulate) {
      float cc[16];
      ldf(C, cc);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += cc[e];
    }
    stf(C, v);
  } else {
    sth(reinterpret_cast<bf16_t*>(p.C) + co, v);
  }
}

// WM x WN wavefronts, each (16 MI) x 64 outputs; ring of NST stages; MINW = workgroups per CU the register budget must allow.
//
// Schedule of one k-step s (frag0 / frag1 = the MFMA fragments of its two 32-deep halves), every wavefront:
//   issue ds_reads frag1(s)  |  MFMAs on frag0(s)  |  wait own global_load_lds of step s+1, BARRIER, issue step s+NST-1 into the
//   stage step s-1 used  |  issue ds_reads frag0(s+1)  |  MFMAs on frag1(s)
// so every block of 16 MFMAs runs with the LDS reads of the NEXT block in flight, there is one barrier per k-step, and the
// barrier does both jobs: step s+1 is complete in LDS (every wave waited for its own share first) and nobody still reads the
// stage that is refilled.  At a tile boundary the stream simply continues with the next tile's first step while the
// accumulators go out: one extra barrier (all waves are done reading the stage step s used), then each wave transposes its
// 64x64 block 16 rows at a time through a private 4 KB slab inside that free stage, so that a lane ends up with 16
// consecutive columns of one row: residual / aux loads and the stores are whole 128-byte lines (16-byte accesses) instead of
// 8-byte pieces at a row stride (measured: 16 such stores per lane cost ~9 us per 256x128 tile; store-issue bound).
template <int WM, int WN, int MI, int NJ, int NST, int MINW, int KIND>
__global__ __launch_bounds__(WM * WN * 64, MINW * WM * WN / 4) void gemm_pk_kernel(const EaGemmParams p, const PkSched sc) {
  static_assert(NJ == 4, "the epilogue transposes 64-column wave tiles");
  static_assert(NST >= 3, "the mid-step barrier schedule needs the refill target to differ from the stage being read next");
  constexpr int NW = WM * WN;
  constexpr int BM_ = WM * MI * 16, BN_ = WN * NJ * 16;
  constexpr int A_BYTES = BM_ * ROW_BYTES, STAGE = A_BYTES + BN_ * ROW_BYTES;
  constexpr int NA = BM_ / 8 / NW, NB = BN_ / 8 / NW;  // global_load_lds instructions per wavefront and k-step
  static_assert(NA * NW * 8 == BM_ && NB * NW * 8 == BN_, "tile rows must split evenly over the wavefronts");
  constexpr int SLAB_LD = 64;                   // floats per slab row; 16-byte chunk c of row r is stored at chunk c ^ r
  constexpr int SLAB_BYTES = 16 * SLAB_LD * 4;  // 4096 (the XOR makes the 8-lane ds_write_b128 groups — 8 rows, one column chunk — conflict-free)
  static_assert(NW * SLAB_BYTES <= STAGE, "epilogue slabs live inside one ring stage");
  extern __shared__ __attribute__((aligned(16))) char dsm[];  // the ONLY LDS object: the ring

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // ---- static schedule: units of XCD x are [cstart, cstart + csize); this workgroup takes slot, slot + per_x, ... ----
  const int g = blockIdx.x, xcd = g & 7, slot = g >> 3;
  const int q8 = sc.units >> 3, r8 = sc.units & 7;
  const int cstart = xcd * q8 + min(xcd, r8), csize = q8 + (xcd < r8 ? 1 : 0);
  const int my_units = slot < csize ? (csize - slot + sc.per_x - 1) / sc.per_x : 0;
  if (my_units == 0) return;
  const int tpn = sc.tiles_m * sc.tiles_n;

  auto decode = [&](int j, int& z, int& ks, int& m0, int& n0) {
    int u = cstart + slot + j * sc.per_x;
    const int zk = u / tpn;
    u -= zk * tpn;
    const int ty = u / sc.tiles_n, tx = u - ty * sc.tiles_n;
    z = zk / p.splitk;
    ks = zk - z * p.splitk;
    m0 = ty * BM_;
    n0 = tx * BN_;
  };
  auto steps_of = [&](int ks) { return (min(p.K, (ks + 1) * p.kchunk) - ks * p.kchunk) / BK; };  // (the last k-chunk may be short)

  // ---- producer: per-lane source pointers of the unit whose k-steps are being issued ----
  const bf16_t* ap[NA];
  const bf16_t* bp[NB];
  int pnk = 0;
  auto setup_src = [&](int j) {
    int z, ks, m0, n0;
    decode(j, z, ks, m0, n0);
    pnk = steps_of(ks);
    const int zhi = z / p.zdiv, zlo = z - zhi * p.zdiv;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (long)zhi * p.sA_hi + (long)zlo * p.sA_lo + (long)ks * p.kchunk;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + (long)zhi * p.sB_hi + (long)zlo * p.sB_lo + (long)ks * p.kchunk;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int r = (wave + NW * i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (r & 7) ^ ((r >> 4) & 7);
      ap[i] = A + (long)min(m0 + r, p.M - 1) * p.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = (wave + NW * i) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (r & 7) ^ ((r >> 4) & 7);
      bp[i] = B + (long)min(n0 + r, p.N - 1) * p.ldb + c * 8;
    }
  };
  int pu = 0, pk = 0, fill = 0;  // producer cursor (unit, k-step) and the ring stage the next step goes to
  auto produce = [&]() {
    char* base = dsm + fill * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(ap[i] + (long)pk * BK), (lptr_t)(base + i * (NW * 1024)), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(bp[i] + (long)pk * BK), (lptr_t)(base + A_BYTES + i * (NW * 1024)), 16, 0, 0);
    fill = fill + 1 == NST ? 0 : fill + 1;
    if (++pk == pnk) {
      pk = 0;
      ++pu;
      if (pu < my_units) setup_src(pu);
    }
  };

  uint32_t a_off[2][MI], b_off[2][NJ];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < MI; ++i) a_off[ks][i] = lds_off(wm * (MI * 16) + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
    for (int j = 0; j < NJ; ++j) b_off[ks][j] = A_BYTES + lds_off(wn * (NJ * 16) + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
  }
  bf16x8_t fa0[MI], fb0[NJ], fa1[MI], fb1[NJ];
  auto read_frags = [&](const char* st, int ks, bf16x8_t (&fa)[MI], bf16x8_t (&fb)[NJ]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(st + a_off[ks][i]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(st + b_off[ks][j]);
  };

  // ---- prologue: NST-1 steps in flight, step 0 complete in LDS, its first fragments on their way ----
  setup_src(0);
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (pu < my_units) produce();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int stage = 0;  // ring stage of the step being consumed
  read_frags(dsm, 0, fa0, fb0);

  for (int cu = 0; cu < my_units; ++cu) {
    int z, ks_id, m0, n0;
    decode(cu, z, ks_id, m0, n0);
    const int cnk = steps_of(ks_id);
    const bool last_unit = cu + 1 == my_units;
    f32x4_t acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < cnk; ++kt) {
      const char* st = dsm + stage * STAGE;
      const int nstage = stage + 1 == NST ? 0 : stage + 1;
      const bool more = !(last_unit && kt + 1 == cnk);  // a next step exists in this workgroup's stream
      read_frags(st, 1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      // operands swapped: D = Btile x Atile^T, so lane holds acc[i][j][r] = C[row i*16 + (lane&15)][col j*16 + (lane>>4)*4 + r]
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb0[j]),
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa0[i]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        // everything this wave has issued so far belongs to steps <= s+1 except (NST-3) younger steps: with NST = 3 that is all
        if constexpr (NST == 4) {
          if constexpr (NA + NB == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
          else if constexpr (NA + NB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else if constexpr (NA + NB == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
          else if constexpr (NA + NB == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // step s+1 complete in LDS; nobody reads the stage of step s-1 any more
        if (pu < my_units) produce();
        read_frags(dsm + nstage * STAGE, 0, fa0, fb0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb1[j]),
              __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa1[i]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      stage = nstage;
    }

    // ---- epilogue.  `stage` already names the NEXT step's stage; the one just consumed is (stage - 1) mod NST ----
    const int freed = stage == 0 ? NST - 1 : stage - 1;
    __builtin_amdgcn_s_barrier();  // every wave is done reading the freed stage (its refill is issued after the next k-step barrier)
    float* slab = reinterpret_cast<float*>(dsm + freed * STAGE + wave * SLAB_BYTES);
    const int zhi = z / p.zdiv, zlo = z - zhi * p.zdiv;
    const long coff = (long)zhi * p.sC_hi + (long)zlo * p.sC_lo;
    const int erow = lane >> 2, ecol = (lane & 3) * 16;  // after the transpose: lane owns 16 consecutive columns of one row
    const int n = n0 + wn * (NJ * 16) + ecol;
    float bias16[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bias16[e] = 0.f;
    if (p.bias && n < p.N) {
      if (n + 16 <= p.N && ((((uintptr_t)(p.bias + n)) & 15) == 0)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 x = *reinterpret_cast<const float4*>(p.bias + n + 4 * t);
          bias16[4 * t] = x.x; bias16[4 * t + 1] = x.y; bias16[4 * t + 2] = x.z; bias16[4 * t + 3] = x.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) if (n + e < p.N) bias16[e] = p.bias[n + e];
      }
    }
    static_for(std::make_integer_sequence<int, MI>{}, [&](auto t) {
      constexpr int i = decltype(t)::value;
      // acc[i][j][0..3] = row (lane&15), columns j*16 + (lane>>4)*4 .. +3 of this wave's 16 x 64 slice
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        *reinterpret_cast<f32x4_t*>(slab + (lane & 15) * SLAB_LD + (((j * 4 + (lane >> 4)) ^ (lane & 15)) << 2)) = acc[i][j];
      float v[16];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const f32x4_t x = *reinterpret_cast<const f32x4_t*>(slab + erow * SLAB_LD + ((((lane & 3) * 4 + t4) ^ erow) << 2));
        v[4 * t4] = x[0]; v[4 * t4 + 1] = x[1]; v[4 * t4 + 2] = x[2]; v[4 * t4 + 3] = x[3];
      }
      const int m = m0 + wm * (MI * 16) + i * 16 + erow;
      if (m < p.M && n < p.N) epilogue16<KIND>(p, z, ks_id, zhi, zlo, coff, m, n, v, bias16);
    });
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct PkConfig {
  int bm, bn, nst, waves, wgs_per_cu;
};
constexpr int PK_NCFG = 3;
const PkConfig g_cfgs[PK_NCFG] = {
    {256, 128, 3, 8, 1},  // 0: wavefronts 4 x 2, each 64 x 64 ; 144 KB ring
    {192, 128, 3, 8, 1},  // 1: wavefronts 4 x 2, each 48 x 64 ; 120 KB ring
    {128, 128, 3, 8, 1},  // 2: wavefronts 4 x 2, each 32 x 64 ;  96 KB ring
};

template <int WM, int WN, int MI, int NJ, int NST, int MINW, int KIND>
bool pk_launch_kind(const EaGemmParams& q, const PkSched& sc, int grid, hipStream_t stream) {
  constexpr int bytes = NST * (WM * MI * 16 + WN * NJ * 16) * ROW_BYTES;
  static bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pk_kernel<WM, WN, MI, NJ, NST, MINW, KIND>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!attr_ok) return false;
  hipLaunchKernelGGL((gemm_pk_kernel<WM, WN, MI, NJ, NST, MINW, KIND>), dim3(grid), dim3(WM * WN * 64), bytes, stream, q, sc);
  return true;
}
template <int WM, int WN, int MI, int NJ, int NST, int MINW>
bool pk_launch(const EaGemmParams& q, const PkSched& sc, int grid, hipStream_t stream) {
  if (q.c_f32) return pk_launch_kind<WM, WN, MI, NJ, NST, MINW, EPI_F32>(q, sc, grid, stream);
  if (q.aux) return pk_launch_kind<WM, WN, MI, NJ, NST, MINW, EPI_AUX>(q, sc, grid, stream);
  if (q.C2) return pk_launch_kind<WM, WN, MI, NJ, NST, MINW, EPI_C2>(q, sc, grid, stream);
  return pk_launch_kind<WM, WN, MI, NJ, NST, MINW, EPI_GEN>(q, sc, grid, stream);
}

int g_pk_mode = 0;  // 0 off (default: see DESIGN.md, measured slower inside the training step), 1 automatic configuration, 2 + c: configuration c forced

}  // namespace

extern "C" int ea_set_gemm_persistent(int mode) {
  const int old = g_pk_mode;
  g_pk_mode = mode;
  return old;
}

// Called by ea_gemm_bf16 (gemm.hip) for eligible launches (both operands k-contiguous, K and kchunk multiples of 64, 16-byte
// aligned rows).  q.splitk / q.kchunk are final.  Returns 1 when the launch was issued, 0 when the caller should use the
// non-persistent kernels (persistent path off, or nothing to gain: a single small tile).
int gemm_pk_try(const EaGemmParams& q, hipStream_t stream, int* cfg_used) {
  if (g_pk_mode == 0) return 0;
  if ((q.c_f32 && (q.aux || q.C2)) || (q.aux && q.C2)) return 0;  // epilogue combinations no caller uses: general kernels
  const int nk = q.kchunk / BK;
  const long slices = (long)q.batch * q.splitk;
  int best = -1;
  if (g_pk_mode >= 2) {
    best = g_pk_mode - 2;
    if (best >= PK_NCFG) return 0;
  } else {
    // cost model (cycles on the busiest CU), calibrated on tools/bench_gemm_ksweep.py: a tile's k-step is bound by the larger of
    // its MFMA time and its operand traffic into LDS; tiles of workgroups sharing a CU add up; the epilogue is not overlapped
    double best_cost = 0.0;
    for (int c = 0; c < PK_NCFG; ++c) {
      const PkConfig& k = g_cfgs[c];
      const long units = (long)((q.M + k.bm - 1) / k.bm) * ((q.N + k.bn - 1) / k.bn) * slices;
      const long slots = 256L * k.wgs_per_cu;
      const long rounds = (units + slots - 1) / slots;                        // tiles of the busiest workgroup
      const long on_cu = units >= slots ? k.wgs_per_cu : (units + 255) / 256;  // workgroups sharing the busiest CU
      const double mfma = (double)k.bm * k.bn * BK * 2.0 / (1024.0 * 4.0 * 0.62);  // cycles per k-step at ~62 % of the CU's MFMA rate
      const double ingest = (double)(k.bm + k.bn) * ROW_BYTES / 44.0;              // ~44 B/clk into LDS per CU
      const double step = mfma > ingest ? mfma : ingest;
      const double epi = (double)k.bm * k.bn * 0.02 + 1500.0;
      const double cost = (double)rounds * on_cu * (nk * step + epi) + 3000.0;
      if (best < 0 || cost < best_cost) { best = c; best_cost = cost; }
    }
  }
  const PkConfig& k = g_cfgs[best];
  PkSched sc;
  sc.tiles_m = (q.M + k.bm - 1) / k.bm;
  sc.tiles_n = (q.N + k.bn - 1) / k.bn;
  const long units = (long)sc.tiles_m * sc.tiles_n * slices;
  if (units > 0x7fffffffL) return 0;
  sc.units = (int)units;
  sc.nk = nk;
  long per_x = (256L * k.wgs_per_cu) / 8;
  if (per_x > (units + 7) / 8) per_x = (units + 7) / 8;
  if (per_x < 1) per_x = 1;
  sc.per_x = (int)per_x;
  const int grid = sc.per_x * 8;
  bool ok = false;
  switch (best) {
    case 0: ok = pk_launch<4, 2, 4, 4, 3, 1>(q, sc, grid, stream); break;
    case 1: ok = pk_launch<4, 2, 3, 4, 3, 1>(q, sc, grid, stream); break;
    case 2: ok = pk_launch<4, 2, 2, 4, 3, 1>(q, sc, grid, stream); break;
    default: break;
  }
  if (cfg_used) *cfg_used = best;
  return ok ? 1 : 0;
}
