// Implicit-GEMM 3x3 convolutions of the sub-sampling front-end for gfx950 — no im2col / col2im buffers.
//
// Reference: espresso/modules/speech_convolutions.py:78-102 (Conv2d 3x3, padding 1, stride 1 or 2, then BatchNorm2d + ReLU)
// and its autograd.  Layout: channels-last [B][T][F][C] bf16, so the C channels of one (t, f) position are one contiguous
// 128 / 256-byte row — exactly one (or two) 64-deep k-steps of a GEMM whose reduction index is (tap, channel).
//
// Round 1 lowered convs 2-4 to im2col + GEMM: a 516 000 x 576 bf16 matrix (595 MB) written and read back per layer in the
// forward pass, the same again (GEMM -> col2im) in the backward pass; profiles/r02_*: the front-end cost 4.3 of the 20.2 ms
// step.  Here the A-operand tile of every k-step is GATHERED straight from the activation tensor by global_load_lds: lane ->
// (tile row, 16-byte slot); the row's source position for the step's tap is computed from the row's (b, t, f) coordinates,
// padding taps read a 128-byte zero line.  One kernel serves
//   forward        Z[b][to][fo][:]  = bias + sum_{ky,kx} X[b][to*s+ky-1][fo*s+kx-1][:] W[:, ky, kx, :]      (+ BatchNorm sums)
//   data gradient  dX[b][ti][fi][:] = sum_{taps valid for (ti, fi)} dZ[b][(ti+1-ky)/s][(fi+1-kx)/s][:] Wd[:, ky, kx, :]
// through a generic description (row grid, source-coordinate multipliers, per-tap offsets, output row mapping).  With stride 2
// the data gradient is launched once per parity class of (ti, fi): each class has its own tap list (1, 2, 2 or 4 taps — no
// multiplications by structural zeros).
//
// Tile: 128 rows (positions) x N (64 or 128 output channels = the whole channel dimension) per 256-thread workgroup, BK = 64,
// 2-stage direct-to-LDS ring (one raw barrier per k-step), MFMA 16x16x32 bf16, the LDS image / swizzle / fragment reads of
// gemm.hip.  Epilogue: fp32 LDS bounce, 16-byte stores, optional bias, optional per-channel (sum, sum of squares) of the
// bf16-rounded outputs accumulated in fp64 (BatchNorm batch statistics: replaces the separate colstats pass).
#include <hip/hip_runtime.h>

#include "gemm_common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct ConvGatherArgs {
  const bf16_t* src;   // gathered tensor [RB][ST][SF][SC]
  const bf16_t* W;     // [N][ldw] k-contiguous: k = koff[tap] + channel
  bf16_t* out;         // rows of N channels, row index from (b, rt, rf) via the o* fields
  const float* bias;   // [N] or null
  double* stats;       // [2N] (sum, sumsq) or null
  const bf16_t* zero;  // >= 128 bytes of zeros
  int RB, RT, RF;      // row grid: M = RB * RT * RF
  int ST, SF, SC;
  int at, af;          // source coordinate = r * a + b[tap]
  int ntaps;
  int bt[9], bf[9], koff[9];
  int N;
  long ldw;
  int OT, OF, ot_mul, ot_add, of_mul, of_add;  // output row = (b * OT + rt * ot_mul + ot_add) * OF + rf * of_mul + of_add
};

template <int BN_>
__global__ __launch_bounds__(256, 2) void conv_gather_kernel(const ConvGatherArgs a) {
  constexpr int BM_ = 128, NST = 2;
  constexpr int NJ = BN_ / 32;  // 2 x 2 wavefronts, each 64 x (BN_/2): NJ n-tiles of 16
  constexpr int A_BYTES = BM_ * ROW_BYTES, STAGE = A_BYTES + BN_ * ROW_BYTES;
  constexpr int NA = BM_ / 32, NB = BN_ / 32;
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wcol = (wave & 1) * (BN_ / 2);
  const uint32_t M = (uint32_t)a.RB * a.RT * a.RF;  // (< 2^31: checked by the launcher — 32-bit divisions below)
  const uint32_t m0 = blockIdx.x * BM_;
  const int cpt = a.SC / BK;  // 64-channel chunks per tap
  const int nk = a.ntaps * cpt;

  // ---- per-lane description of the NA tile rows this lane feeds (row = (wave + 4 i) * 8 + lane / 8) ----
  int rowb[NA], rowt[NA], rowf[NA];  // b * ST (or -1 for rows past M), rt * at, rf * af
  int cs[NA];                         // source 16-byte slot of this lane in that row (bank swizzle applied to the source)
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = (wave + 4 * i) * 8 + (lane >> 3);
    cs[i] = (lane & 7) ^ (r & 7) ^ ((r >> 4) & 7);
    const uint32_t p = m0 + r;
    if (p < M) {
      const uint32_t q = p / (uint32_t)a.RF, b = q / (uint32_t)a.RT;
      const int rf = (int)(p - q * a.RF), rt = (int)(q - b * a.RT);
      rowb[i] = (int)b * a.ST;
      rowt[i] = rt * a.at;
      rowf[i] = rf * a.af;
    } else {
      rowb[i] = -1;
      rowt[i] = rowf[i] = 0;
    }
  }
  const bf16_t* bp[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = (wave + 4 * i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7) ^ ((r >> 4) & 7);
    bp[i] = a.W + (long)min(r, a.N - 1) * a.ldw + c * 8;
  }
  auto issue = [&](int stage, int kt) {
    const int tap = kt / cpt, chunk = kt - tap * cpt;
    const int dt = a.bt[tap], df = a.bf[tap];
    const long ko = a.koff[tap] + chunk * BK;
    char* base = dsm + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int st = rowt[i] + dt, sf = rowf[i] + df;
      const bool ok = rowb[i] >= 0 && (unsigned)st < (unsigned)a.ST && (unsigned)sf < (unsigned)a.SF;
      const bf16_t* s = ok ? a.src + (((long)(rowb[i] + st)) * a.SF + sf) * a.SC + chunk * BK + cs[i] * 8 : a.zero + cs[i] * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(base + i * 4096), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(bp[i] + ko), (lptr_t)(base + A_BYTES + i * 4096), 16, 0, 0);
  };

  f32x4_t acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  uint32_t a_off[2][4], b_off[2][NJ];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[ks][i] = lds_off(wm * 64 + i * 16 + (lane & 15), ks * 4 + (lane >> 4));
#pragma unroll
    for (int j = 0; j < NJ; ++j) b_off[ks][j] = A_BYTES + lds_off(wcol + j * 16 + (lane & 15), ks * 4 + (lane >> 4));
  }

  if (nk > 0) issue(0, 0);
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // step kt complete in LDS; everyone is done reading the other stage
    if (kt + 1 < nk) issue(stage ^ 1, kt + 1);
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
    stage ^= 1;
  }

  // ---- epilogue: 64 rows at a time through an fp32 LDS tile; thread -> (row lane, 8 consecutive channels) ----
  constexpr int CG = BN_ / 8;        // channel groups of 8
  constexpr int RL = 256 / CG;       // row lanes (16 for N = 128, 32 for N = 64)
  float* sC = reinterpret_cast<float*>(dsm);  // [64][BN_] fp32 (<= 32 KiB; the ring is 48-64 KiB)
  const int cg = tid % CG, rlane = tid / CG;
  float bias8[8], s8[8], q8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bias8[e] = a.bias ? a.bias[cg * 8 + e] : 0.f;
    s8[e] = q8[e] = 0.f;
  }
#pragma unroll 1  // (real loops: eight unrolled copies of the row arithmetic were most of this kernel's code)
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) sC[(i * 16 + (lane >> 4) * 4 + r) * BN_ + wcol + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    __syncthreads();
#pragma unroll 1
    for (int pass = 0; pass < 64 / RL; ++pass) {
      const int rl = pass * RL + rlane;
      const uint32_t p = m0 + half * 64 + rl;
      if (p < M) {
        const uint32_t q = p / (uint32_t)a.RF, b = q / (uint32_t)a.RT;
        const int rf = (int)(p - q * a.RF), rt = (int)(q - b * a.RT);
        const long orow = ((long)b * a.OT + (long)rt * a.ot_mul + a.ot_add) * a.OF + (long)rf * a.of_mul + a.of_add;
        const float4 x0 = *reinterpret_cast<const float4*>(sC + rl * BN_ + cg * 8);
        const float4 x1 = *reinterpret_cast<const float4*>(sC + rl * BN_ + cg * 8 + 4);
        float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bias8[e];
        uint4 u;
        u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
        *reinterpret_cast<uint4*>(a.out + orow * a.N + cg * 8) = u;
        if (a.stats) {  // statistics of what BatchNorm will read: the bf16-rounded values
          const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
            s8[2 * e] += lo; q8[2 * e] += lo * lo;
            s8[2 * e + 1] += hi; q8[2 * e + 1] += hi * hi;
          }
        }
      }
    }
  }
  if (a.stats) {  // fold the row lanes that share a channel group, then one fp64 atomic pair per channel and workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(dsm);  // [RL][2][BN_]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(rlane * 2 + 0) * BN_ + cg * 8 + e] = s8[e];
      red[(rlane * 2 + 1) * BN_ + cg * 8 + e] = q8[e];
    }
    __syncthreads();
    if (tid < 2 * BN_) {
      const int which = tid / BN_, c = tid - which * BN_;
      float t = 0.f;
      for (int r = 0; r < RL; ++r) t += red[(r * 2 + which) * BN_ + c];
      atomicAdd(a.stats + which * a.N + c, (double)t);
    }
  }
}

const bf16_t* zero_line(hipStream_t stream) {
  static bf16_t* z = nullptr;
  if (!z) {
    if (hipMalloc(&z, 256) != hipSuccess) return nullptr;
    if (hipMemsetAsync(z, 0, 256, stream) != hipSuccess) return nullptr;
    hipStreamSynchronize(stream);  // once per process
  }
  return z;
}

template <int BN_>
int launch_gather(const ConvGatherArgs& a, hipStream_t stream) {
  constexpr int bytes = 2 * (128 + BN_) * ROW_BYTES;
  static bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gather_kernel<BN_>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!attr_ok) return -1;
  const long M = (long)a.RB * a.RT * a.RF;
  if (M <= 0) return 0;
  if (M >= (1L << 31) - 128) return -1;  // the kernel's row arithmetic is 32-bit
  hipLaunchKernelGGL((conv_gather_kernel<BN_>), dim3((unsigned)((M + 127) / 128)), dim3(256), bytes, stream, a);
  return EA_CHECK_LAUNCH();
}


// ---- weight gradient: dW[cout][tap][cin] += sum_p dZ[p][cout] X[src(p, tap)][cin] ----------------------------------------
// A "TN" product: the reduction index (position p) is the slow axis of BOTH operands.  gfx950's transposing LDS read
// (ds_read_b64_tr_b16: within a 16-lane group lane i receives column i of the 4 x 16 block whose rows are the 8-byte pieces
// supplied by lanes 4j .. 4j+3 — probed on hardware, tools/probes/tr_read_probe.hip) delivers MFMA fragments with k running
// down the rows of a row-major [k][64 channels] image, so both operand tiles go global -> LDS with global_load_lds exactly as
// they lie in HBM (position rows of 128 bytes; the tap's source rows gathered like in the forward kernel) — no register
// transposes, no im2col matrix.  Bank swizzle for the transposing reads: 32-byte chunk c of row r is stored at chunk
// c ^ (r & 3) ^ ((r >> 3) & 1), which makes the 8 rows x 32 bytes one 32-lane read cycle touches hit 64 distinct banks.
// Workgroup = 64 cout x 64 cin x all taps over a slice of the positions (split-K): the dZ tile of a 64-position chunk is
// loaded once and its fragments are kept in registers for the nine tap tiles.  Partial sums go to fp32 slabs, a small second
// kernel folds them into dW (fp32 atomics: 19 M of them per layer would cost more than the whole product).
struct ConvWgradArgs {
  const bf16_t* X;     // [RB][ST][SF][Cin]
  const bf16_t* dZ;    // [RB][RT][RF][Cout]
  float* slab;         // [nsplit][Cout][ntaps][Cin]
  const bf16_t* zero;
  int RB, RT, RF, ST, SF, Cin, Cout;
  int at, af, ntaps;
  int bt[9], bf[9];
  int chunks_per_wg;   // 64-position chunks per workgroup
  int ablate;          // development probe (EA_CONVW_ABLATE, results meaningless): 1 no MFMAs, 2 no global loads after the prologue,
                       // 4 no fragment reads, 8 no barrier
};

__device__ __forceinline__ int tr_sw(int r) { return (r & 3) ^ ((r >> 3) & 1); }
// Eight transposing reads + their wait in ONE asm statement: hipcc does not track the completion of loads issued from inline
// asm, so the destinations must not be visible to it (copied, spilled) before the data has landed
// (/opt/skills/guides/cdna_hip_programming.md 5.7 item 1, form (i)).
__device__ __forceinline__ void ds_read_tr16_x8(const uint32_t (&ad)[8], uint2 (&o)[8]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
      "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\tds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
      : "memory");
}

// NT = taps per workgroup: 9 (a workgroup owns all nine taps of its 64 x 64 channel tile: 144 accumulator registers, two workgroups
// per CU) or 3 (one kernel row; blockIdx.z % 3 picks it: 48 accumulators, four workgroups per CU — round 6, see the launcher)
template <int DEPTH, int NT>
__global__ __launch_bounds__(256, NT == 9 ? 2 : 4) void conv_wgrad_kernel(const ConvWgradArgs a) {
  constexpr int TILE = 64 * ROW_BYTES;  // 8 KiB: 64 position rows x 64 channels
  // Ring: two dZ tiles (a chunk's tile serves its nine taps) + four X tiles, loads issued THREE steps ahead (round 6).  Rounds 2 - 5
  // double-buffered the X tile and waited for vmcnt(0) at every step: 8 MFMAs per wavefront (~150 cycles) between a load's issue and
  // its wait against ~2 us of last-level-cache latency — 222 us per launch, 9 % of the MFMA peak, 8 KB in flight per workgroup.
  // (DEPTH = 1: the round 2 - 5 schedule, kept as the A/B reference: EA_CONV_WGRAD_DEPTH=1)
  constexpr int NBUF = DEPTH == 1 ? 2 : 4;
  static_assert(DEPTH >= 1 && DEPTH <= 3 && (NT == 9 || NT == 3), "ring of two / four X tiles");
  const int tap0 = NT == 9 ? 0 : (int)(blockIdx.z % 3) * 3;          // first tap of this workgroup
  const int zsplit = NT == 9 ? (int)blockIdx.z : (int)(blockIdx.z / 3);  // which slab / which range of chunks
  __shared__ __attribute__((aligned(16))) char lds[(2 + NBUF) * TILE];  // [A0][A1][B0][B1][B2][B3]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // wave tile: cout rows wr*32.., cin cols wc*32..
  const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64;
  const long M = (long)a.RB * a.RT * a.RF;
  const long chunk0 = (long)zsplit * a.chunks_per_wg;
  long nch = (M + 63) / 64 - chunk0;
  if (nch > a.chunks_per_wg) nch = a.chunks_per_wg;
  if (nch < 0) nch = 0;

  f32x4_t acc[NT][2][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // lane -> two rows of a tile: row = (wave + 4 i) * 8 + lane / 8, 16-byte slot lane & 7 (32-byte chunk swizzled at the source)
  int rr[2], cs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    rr[i] = (wave + 4 * i) * 8 + (lane >> 3);
    cs[i] = (lane & 7) ^ (tr_sw(rr[i]) << 1);
  }
  int rowb[2], rowt[2], rowf[2];  // this chunk's two rows: b * ST (or -1), rt * at, rf * af
  long rowp[2];
  auto decode = [&](long ch) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long p = (chunk0 + ch) * 64 + rr[i];
      rowp[i] = p;
      if (p < M) {  // (M < 2^31, checked by the launcher: 32-bit divisions — the 64-bit ones were a third of this kernel's code)
        const uint32_t q = (uint32_t)p / (uint32_t)a.RF, b = q / (uint32_t)a.RT;
        rowb[i] = (int)b * a.ST;
        rowt[i] = (int)(q - b * a.RT) * a.at;
        rowf[i] = (int)((uint32_t)p - q * a.RF) * a.af;
      } else {
        rowb[i] = -1;
        rowt[i] = rowf[i] = 0;
      }
    }
  };
  auto issue_A = [&](int abuf) {  // dZ rows of the decoded chunk
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16_t* s = rowb[i] >= 0 ? a.dZ + rowp[i] * a.Cout + co0 + cs[i] * 8 : a.zero + cs[i] * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(lds + abuf * TILE + (wave + 4 * i) * 1024), 16, 0, 0);
    }
  };
  auto issue_B = [&](int buf, int tap) {  // X rows the decoded chunk's positions read through `tap`
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int st = rowt[i] + a.bt[tap0 + tap], sf = rowf[i] + a.bf[tap0 + tap];
      const bool ok = rowb[i] >= 0 && (unsigned)st < (unsigned)a.ST && (unsigned)sf < (unsigned)a.SF;
      const bf16_t* s = ok ? a.X + (((long)(rowb[i] + st)) * a.SF + sf) * a.Cin + ci0 + cs[i] * 8 : a.zero + cs[i] * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(lds + (2 + buf) * TILE + (wave + 4 * i) * 1024), 16, 0, 0);
    }
  };
  // transposing fragment reads: tile at byte offset tile_off, the wave's 32 channels [c0, c0 + 32) (two 16-channel tiles) x 64
  // positions (two 32-deep halves) -> fr[ks][tile] = 8 bf16 per lane (k = ks*32 + 8*(lane>>4) + e, channel c0 + 16*tile + (lane&15))
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
  auto read_frags = [&](int tile_off, int c0, uint4 (&fr)[2][2]) {
    const int g = lane >> 4, i = lane & 15;
    uint32_t ad[8];
    uint2 o[8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = ks * 32 + 8 * g + (i >> 2) + 4 * h;
          const int chunk = (c0 >> 4) + t;
          ad[(ks * 2 + t) * 2 + h] = lds0 + tile_off + r * ROW_BYTES + ((chunk ^ tr_sw(r)) << 5) + (i & 3) * 8;
        }
    ds_read_tr16_x8(ad, o);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) fr[ks][t] = make_uint4(o[(ks * 2 + t) * 2].x, o[(ks * 2 + t) * 2].y, o[(ks * 2 + t) * 2 + 1].x, o[(ks * 2 + t) * 2 + 1].y);
  };

  const int nsteps = (int)nch * NT;
  // issue cursor: the loads of step `is` (chunk ich, tap itap) — the chunk's dZ tile first when the step opens a chunk, then the X
  // tile of the tap; loads return in order, so waiting for a step's X tile also covers the dZ tile issued just before it
  int is = 0, itap = 0;
  long ich = 0;
  auto issue_next = [&]() {
    if (is >= nsteps) return;
    if (!(a.ablate & 2) || is < DEPTH) {
      if (itap == 0) {
        decode(ich);
        issue_A((int)(ich & 1));
      }
      issue_B(is & (NBUF - 1), itap);
    }
    ++is;
    if (++itap == NT) { itap = 0; ++ich; }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue_next();
  uint4 fa[2][2];  // dZ fragments of the current chunk: [ksub][cout tile]
  long ch = 0;
  int tap = 0;
  for (int s = 0; s < nsteps; ++s) {
    // instructions issued after this step's X tile: the X tiles of the next two steps (2 each) + the dZ tile of a chunk one of
    // them opens (2)
    const int tap1 = tap + 1 == NT ? 0 : tap + 1, tap2 = tap1 + 1 == NT ? 0 : tap1 + 1;
    int pend = 0;
    if (DEPTH >= 2 && s + 1 < nsteps) pend += 2 + (tap1 == 0 ? 2 : 0);
    if (DEPTH >= 3 && s + 2 < nsteps) pend += 2 + (tap2 == 0 ? 2 : 0);
    if (pend >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (pend == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (pend == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (pend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(a.ablate & 8)) __builtin_amdgcn_s_barrier();  // this step's tiles are complete in every wavefront's part; everyone is done with step s - 1's X tile
    uint4 fb[2][2];
    if (!(a.ablate & 4) || s == 0) {
      if (tap == 0) read_frags((int)(ch & 1) * TILE, wr * 32, fa);
      read_frags((2 + (s & (NBUF - 1))) * TILE, wc * 32, fb);
    } else {
      fb[0][0] = fb[0][1] = fb[1][0] = fb[1][1] = fa[0][0];
    }
    __builtin_amdgcn_sched_barrier(0);
    // step s + 3 goes into the ring slot step s - 1 used (free since the barrier above); a dZ tile goes into the slot chunk ch - 1
    // used, whose fragments every wavefront copied to registers eight or more steps ago
    issue_next();
    // acc[tap] += dZ_tile^T X_tile   (static tap index: the accumulators are registers)
    if (!(a.ablate & 1)) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t == tap) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[ks][i]),
                  __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[ks][j]), acc[t][i][j], 0, 0, 0);
      }
    }
    } else {
      acc[0][0][0][0] += __uint_as_float(fb[0][0].x) + __uint_as_float(fb[1][1].y);  // (keep the fragment reads alive)
    }
    tap = tap1;
    if (tap1 == 0) ++ch;
  }

  // partial sums -> slab[z][cout][tap][cin]: lane holds acc[t][i][j][r] = (cout wr*32 + i*16 + (lane>>4)*4 + r, cin wc*32 + j*16 + (lane&15))
  float* out = a.slab + (long)zsplit * a.Cout * a.ntaps * a.Cin;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = co0 + wr * 32 + i * 16 + (lane >> 4) * 4 + r, ci = ci0 + wc * 32 + j * 16 + (lane & 15);
            out[((long)co * a.ntaps + tap0 + t) * a.Cin + ci] = acc[t][i][j][r];
          }
    }
  }
}

// cin > 0: dW is in the parameter's own layout [Cout][Cin][3][3] (torch Conv2d) instead of the kernels' [Cout][3][3][Cin] —
// the sums are scattered there, so that the gradient needs no permuting copy and no separate accumulate launch afterwards
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dW, long n, int nsplit,
                                                                int cin = 0) {
  // grid (column blocks, slab groups): a workgroup adds up its group of slabs for 1024 outputs; groups meet in dW with fp32
  // atomics (a 64 x 64 x 9 gradient split 512 ways is 75 MB of slabs over 36 column blocks: one block per column chunk walked
  // all of them serially and took longer than the MFMA kernel itself)
  const int zper = (nsplit + gridDim.y - 1) / gridDim.y;
  const int z0 = blockIdx.y * zper, z1 = min(nsplit, z0 + zper);
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n || z0 >= z1) return;
  float4 s = *reinterpret_cast<const float4*>(slab + (long)z0 * n + i);
  for (int z = z0 + 1; z < z1; ++z) {
    const float4 x = *reinterpret_cast<const float4*>(slab + (long)z * n + i);
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  if (cin > 0) {  // i = (co*9 + tap)*cin + ci, 4 consecutive ci (cin % 4 == 0)  ->  (co*cin + ci)*9 + tap
    const long ct = i / cin;
    const int ci = (int)(i - ct * cin), tap = (int)(ct % 9);
    const long co = ct / 9;
    float* d = dW + (co * cin + ci) * 9 + tap;
    const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (gridDim.y == 1) d[9 * e] += v[e];
      else atomicAdd(d + 9 * e, v[e]);
    }
    return;
  }
  if (gridDim.y == 1) {
    float4 d = *reinterpret_cast<const float4*>(dW + i);
    d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
    *reinterpret_cast<float4*>(dW + i) = d;
  } else {
    atomicAdd(dW + i, s.x); atomicAdd(dW + i + 1, s.y); atomicAdd(dW + i + 2, s.z); atomicAdd(dW + i + 3, s.w);
  }
}

static int wgrad_split(long M, int Cin, int Cout) {
  const long chunks = (M + 63) / 64;
  const int blocks = (Cin / 64) * (Cout / 64);
  static const long target = [] { const char* e = getenv("EA_CONV_WGRAD_WGS"); const long v = e ? atol(e) : 0; return v > 0 ? v : 256L; }();  // (tuning knob)
  // 256 x 3 tap groups = 3 workgroups per CU in total.  512 (round 5, before the tap groups) and 1024 measure +0.07 ms per step, 128 the
  // same as 256 (profiles/r06_side_kernel_grids_ab.txt): beside the compute queue a side-queue kernel should not claim every CU
  long nsplit = (target + blocks - 1) / blocks;
  if (nsplit > chunks / 8) nsplit = chunks / 8;      // at least 8 chunks (72 tap steps) per workgroup
  if (nsplit < 1) nsplit = 1;
  return (int)nsplit;
}

}  // namespace

// Forward: X bf16 [B][T][F][Cin] -> Z bf16 [B][To][Fo][Cout] = conv3x3(X; W, stride (sy, sx), padding 1) + bias; W bf16
// [Cout][3][3][Cin]; stats (optional, fp64 [2 Cout], accumulated): per-channel sum / sum of squares of Z.
extern "C" int ea_conv3x3_fwd(const void* X, const void* W, const float* bias, void* Z, double* stats, int B, int T, int F,
                              int Cin, int Cout, int sy, int sx, hipStream_t stream) {
  if (B <= 0 || T <= 0 || F <= 0) return 0;
  if (Cin % 64 || (Cout != 64 && Cout != 128) || sy < 1 || sx < 1) return -2;
  ConvGatherArgs a;
  a.src = (const bf16_t*)X; a.W = (const bf16_t*)W; a.out = (bf16_t*)Z; a.bias = bias; a.stats = stats;
  a.zero = zero_line(stream);
  if (!a.zero) return -1;
  const int To = (T - 1) / sy + 1, Fo = (F - 1) / sx + 1;
  a.RB = B; a.RT = To; a.RF = Fo;
  a.ST = T; a.SF = F; a.SC = Cin;
  a.at = sy; a.af = sx;
  a.ntaps = 9;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      const int t = ky * 3 + kx;
      a.bt[t] = ky - 1; a.bf[t] = kx - 1; a.koff[t] = t * Cin;
    }
  a.N = Cout; a.ldw = 9L * Cin;
  a.OT = To; a.OF = Fo; a.ot_mul = 1; a.ot_add = 0; a.of_mul = 1; a.of_add = 0;
  return Cout == 128 ? launch_gather<128>(a, stream) : launch_gather<64>(a, stream);
}

// Data gradient: dZ bf16 [B][To][Fo][Cout] -> dX bf16 [B][T][F][Cin]; Wd bf16 [Cin][3][3][Cout] (the forward weight with its
// channel axes swapped).  One launch per parity class of the input position (one class for stride 1).
extern "C" int ea_conv3x3_dgrad(const void* dZ, const void* Wd, void* dX, int B, int T, int F, int Cin, int Cout, int sy, int sx,
                                hipStream_t stream) {
  if (B <= 0 || T <= 0 || F <= 0) return 0;
  if (Cout % 64 || (Cin != 64 && Cin != 128) || sy < 1 || sx < 1 || sy > 2 || sx > 2) return -2;
  const int To = (T - 1) / sy + 1, Fo = (F - 1) / sx + 1;
  const bf16_t* zero = zero_line(stream);
  if (!zero) return -1;
  for (int pt = 0; pt < sy; ++pt)
    for (int pf = 0; pf < sx; ++pf) {
      ConvGatherArgs a;
      a.src = (const bf16_t*)dZ; a.W = (const bf16_t*)Wd; a.out = (bf16_t*)dX; a.bias = nullptr; a.stats = nullptr; a.zero = zero;
      a.RB = B; a.RT = (T - pt + sy - 1) / sy; a.RF = (F - pf + sx - 1) / sx;  // positions ti = rt*sy + pt < T
      a.ST = To; a.SF = Fo; a.SC = Cout;
      a.at = 1; a.af = 1;
      a.ntaps = 0;
      for (int ky = 0; ky < 3; ++ky) {
        if ((pt + 1 - ky) % sy) continue;  // (ti + 1 - ky) must be a multiple of the stride
        for (int kx = 0; kx < 3; ++kx) {
          if ((pf + 1 - kx) % sx) continue;
          const int t = a.ntaps++;
          // to = (rt*sy + pt + 1 - ky) / sy = rt + (pt + 1 - ky) / sy   (exact; C division of a negative multiple is exact too)
          a.bt[t] = (pt + 1 - ky) / sy; a.bf[t] = (pf + 1 - kx) / sx; a.koff[t] = (ky * 3 + kx) * Cout;
        }
      }
      a.N = Cin; a.ldw = 9L * Cout;
      a.OT = T; a.OF = F; a.ot_mul = sy; a.ot_add = pt; a.of_mul = sx; a.of_add = pf;
      if (a.RT <= 0 || a.RF <= 0) continue;
      if (a.ntaps == 0) return -2;
      const int rc = Cin == 128 ? launch_gather<128>(a, stream) : launch_gather<64>(a, stream);
      if (rc) return rc;
    }
  return 0;
}

extern "C" long ea_conv3x3_wgrad_workspace_bytes(int B, int T, int F, int Cin, int Cout, int sy, int sx) {
  const int To = (T - 1) / sy + 1, Fo = (F - 1) / sx + 1;
  return (long)wgrad_split((long)B * To * Fo, Cin, Cout) * Cout * 9 * Cin * (long)sizeof(float);
}

// Weight gradient: dW fp32 [Cout][3][3][Cin] += sum over positions of dZ (x) X-taps.  (The bias gradient of a convolution that
// feeds BatchNorm is available from BatchNorm's own sums — exactly zero in training mode — and is not computed here.)
static int conv3x3_wgrad_impl(const void* X, const void* dZ, float* dW, void* workspace, int B, int T, int F, int Cin, int Cout,
                              int sy, int sx, int param_layout, hipStream_t stream);
extern "C" int ea_conv3x3_wgrad(const void* X, const void* dZ, float* dW, void* workspace, int B, int T, int F, int Cin, int Cout,
                                int sy, int sx, hipStream_t stream) {
  return conv3x3_wgrad_impl(X, dZ, dW, workspace, B, T, F, Cin, Cout, sy, sx, 0, stream);
}
// the same with dW in the parameter's own layout [Cout][Cin][3][3] (+=): written straight into the parameter's gradient
extern "C" int ea_conv3x3_wgrad_param_layout(const void* X, const void* dZ, float* dW, void* workspace, int B, int T, int F, int Cin,
                                             int Cout, int sy, int sx, hipStream_t stream) {
  return conv3x3_wgrad_impl(X, dZ, dW, workspace, B, T, F, Cin, Cout, sy, sx, 1, stream);
}
static int conv3x3_wgrad_impl(const void* X, const void* dZ, float* dW, void* workspace, int B, int T, int F, int Cin, int Cout,
                              int sy, int sx, int param_layout, hipStream_t stream) {
  if (B <= 0 || T <= 0 || F <= 0) return 0;
  if (Cin % 64 || Cout % 64 || sy < 1 || sx < 1 || !workspace) return -2;
  ConvWgradArgs a;
  a.X = (const bf16_t*)X; a.dZ = (const bf16_t*)dZ; a.slab = (float*)workspace;
  a.zero = zero_line(stream);
  if (!a.zero) return -1;
  const int To = (T - 1) / sy + 1, Fo = (F - 1) / sx + 1;
  a.RB = B; a.RT = To; a.RF = Fo; a.ST = T; a.SF = F; a.Cin = Cin; a.Cout = Cout;
  a.at = sy; a.af = sx; a.ntaps = 9;
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) { a.bt[ky * 3 + kx] = ky - 1; a.bf[ky * 3 + kx] = kx - 1; }
  const long M = (long)B * To * Fo;
  if (M >= (1L << 31) - 128) return -2;  // the kernel decodes positions with 32-bit divisions
  const int nsplit = wgrad_split(M, Cin, Cout);
  const long chunks = (M + 63) / 64;
  a.chunks_per_wg = (int)((chunks + nsplit - 1) / nsplit);
  static const int ablate = [] { const char* e = getenv("EA_CONVW_ABLATE"); return e ? atoi(e) : 0; }();  // (development probe)
  a.ablate = ablate;
  // ring depth by shape (round 6, isolated at the recipe batch, us per call incl. the slab reduce, depth 1 / 3): conv 2 (64 -> 64
  // channels, 520 k positions) 236 / 215, conv 3 (64 -> 128) 281 / 308, conv 4 (128 -> 128, 130 k positions) 174 / 178 — the kernel is
  // bound by its LDS fragment reads and the barrier per tap, not by load latency; only the 64-output-channel shape gains
  static const int depth_env = [] { const char* e = getenv("EA_CONV_WGRAD_DEPTH"); return e ? atoi(e) : 0; }();  // (diagnostic A/B switch)
  const int depth = depth_env ? depth_env : (Cout <= 64 ? 3 : 1);
  // Taps per workgroup (round 6).  The kernel is bound by how many bytes its workgroups keep in flight against the ~2 - 3 us of a
  // last-level-cache / HBM read: TCC counters say the L2 misses are the compulsory ones (310 MB per launch, 46 % hit rate = the
  // tap re-reads), 1.35 TB/s; the ablation (profiles/r06_conv_wgrad_ablation.txt) gives 299 us -> 176 without the global loads,
  // 286 without the MFMAs.  Nine taps per workgroup = 144 accumulator registers = two workgroups per CU; three taps per
  // workgroup = four per CU with the same 64 x 64 tile, the X band re-read by the three tap-row workgroups out of L2.
  static const int taps_env = [] { const char* e = getenv("EA_CONV_WGRAD_TAPS"); return e ? atoi(e) : 0; }();  // (diagnostic A/B switch)
  // isolated, us per call incl. the slab reduce, 9 / 3 taps per workgroup: conv 2 224 / 198, conv 3 297 / 271, conv 4 178 / 149;
  // update step, same box, interleaved: 13.74 / 13.48 and 13.64 / 13.61 ms (profiles/r06_conv_wgrad_taps_ab.txt) -> 3 by default
  const int taps = taps_env == 3 || taps_env == 9 ? taps_env : 3;
  if (taps == 3) {
    hipLaunchKernelGGL((conv_wgrad_kernel<1, 3>), dim3(Cout / 64, Cin / 64, nsplit * 3), dim3(256), 0, stream, a);
  } else if (depth == 1) hipLaunchKernelGGL((conv_wgrad_kernel<1, 9>), dim3(Cout / 64, Cin / 64, nsplit), dim3(256), 0, stream, a);
  else if (depth == 2) hipLaunchKernelGGL((conv_wgrad_kernel<2, 9>), dim3(Cout / 64, Cin / 64, nsplit), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<3, 9>), dim3(Cout / 64, Cin / 64, nsplit), dim3(256), 0, stream, a);
  const long n = (long)Cout * 9 * Cin;
  const long blocks = (n / 4 + 255) / 256;
  int zg = (int)(1024 / blocks);  // ~1024 workgroups in all
  if (zg > nsplit / 4) zg = nsplit / 4;
  if (zg < 1) zg = 1;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)blocks, zg), dim3(256), 0, stream, (const float*)workspace, dW, n, nsplit,
                     param_layout ? Cin : 0);
  return EA_CHECK_LAUNCH();
}
