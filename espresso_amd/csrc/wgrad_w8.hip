// 8-wavefront 256 x 256 weight-gradient kernel for gfx950 (round 5): the LONG-reduction members of the grouped weight-gradient
// contract of ea_wgrad_group (include/espresso_amd.h) — in practice the transducer joint's output layer
// (espresso/models/transformer/speech_transformer_transducer_base.py:276-299: dW [V][J] = dlogits^T Z over all B*T*(U+1) lattice
// nodes, V = 5004, J = 512, 70 000 rows per micro-batch, cut into row slabs by functional._joint_wgrad).
//
// Why: with 64 x 128 tiles (wgrad_group_tr_kernel<64>) that product re-reads its operands 8.5 GB per call — 1.78 ms, 8 % of the
// MFMA peak (profiles/r05_transducer_kernel_trace.txt).  A 256 x 256 tile re-reads 2.8 GB.  Same data movement as the 4-wave
// kernel (both operand tiles are [64 reduction rows][256 columns] AS THEY LIE IN MEMORY, moved by global_load_lds; MFMA fragments
// — 8 consecutive reduction indices of one column — come out of ds_read_b64_tr_b16), but ONE 512-thread workgroup per CU,
// a wavefront tile of 128 x 64 (128 accumulator registers), a two-stage ring of 64 KB stages with one barrier per stage, and
// fragment reads issued half a k-step ahead of the MFMAs that consume them.
//
// Products, accumulation order along the reduction (BK = 64, two 16x16x32 MFMAs per accumulator and stage) and the fp32 `+=` into dW
// are those of wgrad_group_tr_kernel: results are bit-identical to it (tests/test_gpu_parity.py).
#include <type_traits>

#include "common.h"
#include "espresso_amd.h"
#include "gemm_common.h"

// 0 = never, 1 = automatic (default: groups that fill the chip with 256 x 256 tiles and reduce over >= 4096 rows), 2 = every
// eligible group (tests, probes)
static int g_wgrad_w8 = [] { const char* e = getenv("EA_WGRAD_W8"); return e ? atoi(e) : 1; }();

namespace {

struct W8WgradTable {
  int start[EA_WGRAD_MAX + 1];  // first workgroup of problem i ; start[count] = grid size
  int tiles_x[EA_WGRAD_MAX];    // column tiles (over K) of problem i
};

__device__ __attribute__((aligned(256))) unsigned char g_wgrad_w8_zero_page[256];

// image rows are 512 B (256 bf16 columns); the 16-byte slot s of row r holds source slot s ^ f(r).  A half-wave of a transposing
// read touches rows {8g + e : g in a pair, e < 4} x 32 bytes: f gives those 8 rows 8 different 32-byte positions of a 256-byte
// bank window, and does not change under the +4-row / +32-row immediates (same function as the 256-byte-pitch image of gemm.hip)
__device__ __forceinline__ int wg8_f(int r) { return ((r & 3) | ((r >> 1) & 4)) << 1; }

// eight transposing reads (4 addresses x 2 row offsets), NOT waited for: o[2 a + h] = address a at offset h
template <int O0, int O1>
__device__ __forceinline__ void wg8_read8(const uint32_t (&ad)[4], uint2 (&o)[8]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\tds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %2, %9 offset:%12\n\tds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
      "ds_read_b64_tr_b16 %4, %10 offset:%12\n\tds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %6, %11 offset:%12\n\tds_read_b64_tr_b16 %7, %11 offset:%13"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "i"(O0), "i"(O1)
      : "memory");
}
// the wait for every outstanding LDS read; the registers of the reads it completes are tied to it so that no consumer is
// scheduled above it (the compiler does not count lgkmcnt for reads issued from inline asm)
__device__ __forceinline__ void wg8_wait8(uint2 (&o)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7])
               :
               : "memory");
}
__device__ __forceinline__ void wg8_wait16(uint2 (&o)[8], uint2 (&q)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "+v"(q[0]), "+v"(q[1]),
                 "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
               :
               : "memory");
}
__device__ __forceinline__ bf16x8_t wg8_cat(uint2 lo, uint2 hi) {
  const uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ bf16x8_t wg8_sel(bool c, bf16x8_t a, bf16x8_t b) {
  const uint4 x = __builtin_bit_cast(uint4, a), y = __builtin_bit_cast(uint4, b);
  return __builtin_bit_cast(bf16x8_t, make_uint4(c ? x.x : y.x, c ? x.y : y.y, c ? x.z : y.z, c ? x.w : y.w));
}
#define WG8_MFMA(A_, B_, C_)                                                                                              \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, A_),              \
                                          __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, B_), C_, 0, 0, 0)

// one 256 x 256 tile: `vb` = position of the tile in the launch's virtual 1-D grid of `total` tiles
__device__ __forceinline__ void wg8_tile(const EaWgradGroup& g, const W8WgradTable& tb, const int vb, const int total, char* dsm) {
  constexpr int PITCH = 512, A_BYTES = 64 * PITCH, STAGE = 2 * A_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wavefront tile: rows wm * 128 .. + 128 (columns of dy), columns wn * 64 .. + 64 (of x)

  int pi = 0;
  const int xcd = vb & 7, xq = total >> 3, xr = total & 7;
  // workgroups go round-robin to the 8 XCDs: give every XCD a contiguous range, so that the column tiles of a row block (same dy
  // columns) and neighbouring row blocks (same x rows) of one slab meet in ONE L2
  const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (vb >> 3);
  while (pi + 1 < g.count && bid >= tb.start[pi + 1]) ++pi;
  const EaWgradProblem P = g.p[pi];
  const int local = bid - tb.start[pi];
  const int tx = tb.tiles_x[pi];
  const int tile_y = local / tx, tile_x = local - tile_y * tx;
  const int R = P.N, Cn = P.K, Kr = P.M;
  const int m0 = tile_y * 256, n0 = tile_x * 256;
  const long lda = P.ld_dy, ldb = P.ld_x;
  const int nk = (Kr + BK - 1) / BK;

  f32x4_t acc[8][4], accb[2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = P.dbias != nullptr && tile_x == 0;

  // per-lane sources of this wavefront's load instructions: instruction i fills image rows lrow0 + 16 i, slot lane & 31 <- source
  // slot (lane & 31) ^ f(row).  32-bit byte offsets from the (uniform) operand bases.  A column past the row pitch re-reads the
  // row's last 16 bytes: such columns are >= N / >= K, their products only reach outputs that are never stored
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_wgrad_w8_zero_page) + (lane & 15) * 8;
  const char* abase = reinterpret_cast<const char*>(P.dy);
  const char* bbase = reinterpret_cast<const char*>(P.x);
  uint32_t aoff[4], boff[4];
  const int lrow0 = wave * 2 + (lane >> 5);
  const uint32_t astep = (uint32_t)lda * (BK * 2), bstep = (uint32_t)ldb * (BK * 2);  // bytes per stage of 64 reduction rows
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = lrow0 + 16 * i, slot = lane & 31;
    const int ca = min(m0 + 8 * (slot ^ wg8_f(row)), (int)lda - 8), cb = min(n0 + 8 * (slot ^ wg8_f(row)), (int)ldb - 8);
    aoff[i] = ((uint32_t)row * (uint32_t)lda + (uint32_t)ca) * 2u;
    boff[i] = ((uint32_t)row * (uint32_t)ldb + (uint32_t)cb) * 2u;
  }
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // loads of stage kt; the offsets walk forward one stage per call (stages are issued in order, each exactly once)
  auto issue = [&](int stage, int kt) {
    char* base = dsm + stage * STAGE + wave * 1024;
    if ((kt + 1) * BK <= Kr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(abase + aoff[i]), (lptr_t)(base + i * 8192), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(bbase + boff[i]), (lptr_t)(base + A_BYTES + i * 8192), 16, 0, 0);
    } else {  // last, partial stage: rows past M contribute zeros
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(kt * BK + lrow0 + 16 * i < Kr ? abase + aoff[i] : reinterpret_cast<const char*>(zero)),
                                         (lptr_t)(base + i * 8192), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(kt * BK + lrow0 + 16 * i < Kr ? bbase + boff[i] : reinterpret_cast<const char*>(zero)),
                                         (lptr_t)(base + A_BYTES + i * 8192), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      aoff[i] += astep;
      boff[i] += bstep;
    }
  };

  // fragment addresses inside a stage: lane (g4, lj) of a transposing read passes the address of [8 g4 + (lj >> 2)][c + 4 (lj & 3)]
  // and receives [8 g4 + 0..3][c + lj]; the read 4 rows down completes the 8 reduction indices; +32 rows is the second k-step
  const int g4 = lane >> 4, lj = lane & 15, le = lj >> 2, lq = lj & 3, row0 = 8 * g4 + le;
  const uint32_t lds0 = (uint32_t)(uintptr_t)dsm;
  uint32_t adAlo[4], adAhi[4], adB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    adAlo[i] = lds0 + row0 * PITCH + (((((wm * 128 + 16 * i) >> 3) + (lq >> 1)) ^ wg8_f(row0)) << 4) + (lq & 1) * 8;
    adAhi[i] = lds0 + row0 * PITCH + (((((wm * 128 + 64 + 16 * i) >> 3) + (lq >> 1)) ^ wg8_f(row0)) << 4) + (lq & 1) * 8;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    adB[j] = lds0 + A_BYTES + row0 * PITCH + (((((wn * 64 + 16 * j) >> 3) + (lq >> 1)) ^ wg8_f(row0)) << 4) + (lq & 1) * 8;
  const uint4 ones_u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_u);

  issue(0, 0);
  auto kloop = [&](auto bias_tag) {
    constexpr bool DO_BIAS = decltype(bias_tag)::value;
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // stage kt is in LDS for every wavefront; everyone is done reading the other stage
      if (kt + 1 < nk) issue(stage ^ 1, kt + 1);
      // (the fragment addresses themselves flip between the stages at the bottom of the loop: no per-stage copies)
      uint32_t(&bA)[4] = adB;
      uint32_t(&aLo)[4] = adAlo;
      uint32_t(&aHi)[4] = adAhi;
      // [B0, A0lo] wait | [A0hi] MFMA(A0lo) wait | [B1, A1lo] MFMA(A0hi) wait | [A1hi] MFMA(A1lo) wait | MFMA(A1hi)
      uint2 ob0[8], ob1[8], oal[8], oah[8];
      wg8_read8<0, 4 * PITCH>(bA, ob0);
      wg8_read8<0, 4 * PITCH>(aLo, oal);
      wg8_wait16(ob0, oal);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint2(&ob)[8] = ks == 0 ? ob0 : ob1;
        if (ks == 0) wg8_read8<0, 4 * PITCH>(aHi, oah);
        else wg8_read8<32 * PITCH, 36 * PITCH>(aHi, oah);
        __builtin_amdgcn_sched_barrier(0);  // (the reads stay ABOVE the MFMAs they hide under; hipcc sinks them to their wait otherwise)
        bf16x8_t bfr[4], af[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = wg8_cat(ob[2 * j], ob[2 * j + 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = wg8_cat(oal[2 * i], oal[2 * i + 1]);
        bf16x8_t sb0 = af[0];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = WG8_MFMA(bfr[j], af[i], acc[i][j]);
          if (DO_BIAS && i > 0) sb0 = wg8_sel(wn == i, af[i], sb0);
        }
        if (DO_BIAS) accb[0] = WG8_MFMA(ones, sb0, accb[0]);
        __builtin_amdgcn_sched_barrier(0);
        wg8_wait8(oah);
        if (ks == 0) {
          wg8_read8<32 * PITCH, 36 * PITCH>(bA, ob1);
          wg8_read8<32 * PITCH, 36 * PITCH>(aLo, oal);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = wg8_cat(oah[2 * i], oah[2 * i + 1]);
        bf16x8_t sb1 = af[0];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[4 + i][j] = WG8_MFMA(bfr[j], af[i], acc[4 + i][j]);
          if (DO_BIAS && i > 0) sb1 = wg8_sel(wn == i, af[i], sb1);
        }
        if (DO_BIAS) accb[1] = WG8_MFMA(ones, sb1, accb[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) wg8_wait16(ob1, oal);
      }
      stage ^= 1;
      const uint32_t flip = stage ? (uint32_t)STAGE : (uint32_t)-STAGE;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        adAlo[i] += flip;
        adAhi[i] += flip;
        adB[i] += flip;
      }
    }
  };
  if (do_bias) kloop(std::true_type{});
  else kloop(std::false_type{});

  // operands were multiplied swapped (x fragment first): acc[i][j][r] = dW[m0 + wm*128 + 16 i + lj][n0 + wn*64 + 16 j + 4 g4 + r]
  if (do_bias && g4 == 0) {  // wavefront wn holds the column sums of row tiles wn and 4 + wn (every register of a lane: row lj)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = m0 + wm * 128 + (4 * t + wn) * 16 + lj;
      if (n < R) P.dbias[n] += accb[t][0];
    }
  }
  const bool vec_ok = (P.ldw & 3) == 0 && (((uintptr_t)P.dW) & 15) == 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wm * 128 + 16 * i + lj;
    if (m >= R) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + 16 * j + 4 * g4;
      if (n >= Cn) continue;
      float* C = P.dW + (long)m * P.ldw + n;
      if (vec_ok && n + 4 <= Cn) {
        float4 c = *reinterpret_cast<const float4*>(C);
        c.x += acc[i][j][0]; c.y += acc[i][j][1]; c.z += acc[i][j][2]; c.w += acc[i][j][3];
        *reinterpret_cast<float4*>(C) = c;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < Cn) C[e] += acc[i][j][e];
      }
    }
  }
}

// grid: 1-D, 512 threads, 128 KB of dynamic LDS (two stages).  gridDim.x == tb.start[EA_WGRAD_MAX] (the tile count): one tile per
// workgroup.  A SMALLER grid (a multiple of 8, so that a workgroup's tiles stay on its XCD's range): every workgroup walks tiles
// blockIdx.x, + gridDim.x, ... — for launches that run on a side queue beside the compute queue and should leave it most of the CUs.
__global__ __launch_bounds__(512, 2) void wgrad_w8_kernel(const EaWgradGroup g, const W8WgradTable tb) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int total = tb.start[EA_WGRAD_MAX];
  for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
    if (vb != (int)blockIdx.x) __syncthreads();  // every wavefront is done reading the previous tile's last stage
    wg8_tile(g, tb, vb, total, dsm);
  }
}

}  // namespace

extern "C" int ea_set_wgrad_w8(int mode) {
  const int old = g_wgrad_w8;
  g_wgrad_w8 = mode;
  return old;
}

static int wgrad_w8_launch(const EaWgradGroup& g, hipStream_t stream, int* grid_out, bool forced);
// Called first by ea_wgrad_group.  Returns 1 when the group was launched here (*grid_out = workgroups), 0 to fall through, 2 when
// only the THICK problems were launched here and `rest` received the thin ones (automatic mode, rest != NULL): a problem whose dW
// has <= 64 rows (the per-head positional-projection gradients of an attention block: [64][2T'-1]) fills a quarter of a 256-row tile
// and would hold a CU for the whole reduction all the same — 24 to 56 of a layer group's 116 to 148 workgroups; the 4-wave kernel
// can do them on 64-row tiles in a second launch.
int ea_wgrad_w8_try(const EaWgradGroup& g, hipStream_t stream, int* grid_out, EaWgradGroup* rest) {
  if (!g_wgrad_w8 || g.count <= 0) return 0;
  // (A/B switch, default OFF: measured 12.41 / 12.43 vs 12.30 / 12.32 ms per step — the second launch walks the same 6 240 rows on a
  // handful of workgroups and lengthens the side queue by more than the freed CUs give back; profiles/r06_side_kernel_grids_ab.txt)
  static const bool split_thin = [] { const char* e = getenv("EA_WGRAD_W8_SPLIT_THIN"); return e && e[0] == '1'; }();
  if (g_wgrad_w8 == 1 && rest && split_thin) {
    EaWgradGroup thick;
    thick.count = 0;
    rest->count = 0;
    for (int i = 0; i < g.count; ++i) {
      if (g.p[i].N <= 64) rest->p[rest->count++] = g.p[i];
      else thick.p[thick.count++] = g.p[i];
    }
    if (thick.count == 0) return 0;  // nothing but thin problems: not this kernel's case
    if (rest->count > 0) return wgrad_w8_launch(thick, stream, grid_out, false) ? 2 : 0;
  }
  return wgrad_w8_launch(g, stream, grid_out, g_wgrad_w8 == 2);
}
static int wgrad_w8_launch(const EaWgradGroup& g, hipStream_t stream, int* grid_out, bool forced) {
  W8WgradTable tb;
  int total = 0;
  long min_rows = 1L << 40;
  for (int i = 0; i < g.count; ++i) {
    const EaWgradProblem& p = g.p[i];
    if ((p.ld_dy & 7) || (p.ld_x & 7) || ((reinterpret_cast<uintptr_t>(p.dy) | reinterpret_cast<uintptr_t>(p.x)) & 15) || p.M <= 0) return 0;
    if (((long)p.M + 64) * p.ld_dy * 2 >= (1L << 32) || ((long)p.M + 64) * p.ld_x * 2 >= (1L << 32) || p.ld_dy < 8 || p.ld_x < 8) return 0;  // 32-bit byte offsets
    tb.start[i] = total;
    tb.tiles_x[i] = (p.K + 255) / 256;
    total += ((p.N + 255) / 256) * tb.tiles_x[i];
    min_rows = p.M < min_rows ? p.M : min_rows;
  }
  // automatic: reductions long enough to amortise the 256 x 256 fp32 read-modify-write of the output, and at least 48 tiles.  Round 5
  // asked for a chip-filling grid (>= 192 tiles: the transducer joint's slabs) because ALONE the encoder layers' groups (~116 tiles
  // = 116 workgroups on 256 CUs) are slower here than on the 4-wave kernel.  But these launches never run alone: they sit on the
  // side queue beside the compute queue's data-gradient chain, and a launch that claims every CU doubles the duration of whatever
  // runs next to it (gemm_glds 18 -> 50 us).  On 116 CUs, one workgroup each, with a third of the operand traffic, the same group
  // leaves 140 CUs to the compute queue: config 3 12.89 -> 12.49 ms per step, config 2 12.91 -> 12.63 (round 6,
  // profiles/r06_side_kernel_grids_ab.txt).  Smaller row counts (config 4: ~1 500 rows) gain 1 % forced and stay with the 4-wave kernel.
  static const int min_tiles = [] { const char* e = getenv("EA_WGRAD_W8_MIN_TILES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 48; }();  // (tuning knob)
  // (second clause, round 6: the transducer recipe's encoder layers — ~116 tiles over ~1 500 rows — gain 1 % on this kernel beside the
  // compute queue; the enc-dec recipe's decoder layers — 64 tiles, similar row counts — lose 5 %: profiles/r06_wgrad_w8_configs_ab.txt)
  static const int small_rows_tiles = [] { const char* e = getenv("EA_WGRAD_W8_SMALL_ROWS_TILES"); return e ? atoi(e) : 80; }();
  const bool rule = total <= 512 && ((total >= min_tiles && min_rows >= 4096) || (small_rows_tiles > 0 && total >= small_rows_tiles && min_rows >= 1024));
  if (!forced && !rule) return 0;
  for (int i = g.count; i <= EA_WGRAD_MAX; ++i) tb.start[i] = total;
  for (int i = g.count; i < EA_WGRAD_MAX; ++i) tb.tiles_x[i] = 1;
  constexpr int lds = 2 * 2 * 64 * 512;
  static const bool attr_ok =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_w8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess;
  if (!attr_ok) return 0;
  // groups that do not fill the chip anyway (the encoder layers' ~116 tiles) run beside the compute queue's data-gradient chain, whose
  // 8-wave GEMMs are ~200 workgroups of one per CU: capped at `cap` workgroups walking two tiles each, the launch takes twice as long
  // on the side queue (which has the slack) and the chain's GEMMs fit the remaining CUs in ONE round.  0 = one workgroup per tile.
  static const int cap = [] { const char* e = getenv("EA_WGRAD_W8_GRID"); return e ? atoi(e) : 0; }();
  int grid = total;
  if (cap > 0 && total < 192 && total > cap) grid = (cap + 7) & ~7;
  hipLaunchKernelGGL(wgrad_w8_kernel, dim3(grid), dim3(512), lds, stream, g, tb);
  if (grid_out) *grid_out = grid;
  return 1;
}
