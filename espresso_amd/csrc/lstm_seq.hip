// Persistent LSTM recurrence for gfx950: the whole time loop of one layer (forward, or backward) in ONE launch.
//
// Reference: the step-by-step LSTMCell loop of espresso/models/speech_lstm.py:846-893 (decoder / transducer predictor / LM,
// teacher forced) and one direction of the packed nn.LSTM of the BiLSTM encoder (:470-520).  Round 1 ran one small recurrent
// GEMM + one cell kernel per step (forward) and cell kernel + split-K GEMM + reduce per step (backward): with B <= 16 rows
// those launches are pure latency (10..60 us each, ~1000 launches per update of the transducer recipe).
//
// Here the recurrent weights are stationary in REGISTERS (bf16 MFMA A fragments, loaded once), the workgroups exchange only
// the B x H hidden state (forward) / the B x 4H gate gradients (backward) through global memory and meet at a grid-wide
// barrier (agent-scope release / acquire around one atomic counter) once per step.  The cell state c (forward) and its
// gradient (backward) never leave registers.  MFMA tile = 16 weight rows x 16 batch columns x 32 of the reduction:
//   forward : rows are ordered (unit, gate) so that one lane ends up with the i, f, g, o pre-activations of ONE hidden unit
//             and one batch row in its 4 accumulator registers -> the cell update is lane-local;
//   backward: rows are hidden units, the 4H-long reduction is split over the 4 waves of a workgroup (wave w = gate w's rows),
//             partial sums meet in LDS, then thread (unit, batch row) runs the cell backward.
// All workgroups of a launch must be resident at once: at most 64 workgroups of 256 threads are launched (256 CUs).
#include "common.h"
#include "espresso_amd.h"

namespace {

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
  const float e = __expf(2.f * x);
  return 1.f - 2.f / (e + 1.f);
}

// counter[0]: arrivals; counter[1]: set when a wait gave up (a workgroup that never became resident) -> the host reports it
__device__ __forceinline__ void grid_arrive(unsigned* counter) {
  __threadfence();  // this thread's stores are visible device-wide (agent-scope release)
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void grid_wait(unsigned* counter, unsigned target) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 26)) {  // seconds: a workgroup of this grid never became resident; do not hang the device
        counter[1] = 1;
        __threadfence_system();
        // and do not carry on with stale h / dgates either (the results would be silently wrong): abort the launch, the
        // host sees hipErrorLaunchFailure at its next synchronisation
        __builtin_trap();
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale lines before reading what the other workgroups wrote
}

struct LstmSeqFwdArgs {
  const float* gx;        // [U*B][4H] input projections (+ both biases), rows t*B + b
  const bf16_t* w_hh;     // [4H][H]
  const bf16_t* h0;       // [B][H] or null
  const float* c0;        // [B][H] or null
  const uint8_t* frozen;  // [U][B] or null
  bf16_t* hs;             // [U*B][H]
  float* cs;              // [U][B][H]
  float* act;             // [U][B][4H]
  float* h_last;          // [B][H] or null
  unsigned* counter;
  int B, U, H, reverse, frozen_out_zero;
};

// KS = H / 32 reduction steps, MT = 16-row weight tiles per wave (4 hidden units each), NG = 16-column batch groups
template <int KS, int MT, int NG>
__global__ __launch_bounds__(256) void lstm_seq_fwd_kernel(const LstmSeqFwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, col = lane & 15;
  const int H = a.H, B = a.B, U = a.U;
  const int j0 = (blockIdx.x * 4 + wave) * 4 * MT;  // first hidden unit of this wave
  bf16x8_t wf[MT][KS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = lane & 15;
    const long row = (long)(m & 3) * H + j0 + mt * 4 + (m >> 2);  // tile row m = 4 * unit + gate
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[mt][ks] = *reinterpret_cast<const bf16x8_t*>(a.w_hh + row * H + ks * 32 + q * 8);
  }
  float cst[MT][NG];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int b = g * 16 + col, j = j0 + mt * 4 + q;
      cst[mt][g] = (a.c0 && b < B) ? a.c0[(long)b * H + j] : 0.f;
    }
  const unsigned nwg = gridDim.x;
  for (int n = 0; n < U; ++n) {
    const int t = a.reverse ? U - 1 - n : n;
    const int tprev = a.reverse ? t + 1 : t - 1;
    float gxv[MT][NG][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int b = g * 16 + col, j = j0 + mt * 4 + q;
        const float* gp = a.gx + ((long)t * B + min(b, B - 1)) * 4 * H + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) gxv[mt][g][r] = gp[(long)r * H];
      }
    f32x4_t acc[MT][NG];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[mt][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16_t* hp = n == 0 ? a.h0 : a.hs + (long)tprev * B * H;
    if (hp) {
      if (n > 0) grid_wait(a.counter, (unsigned)n * nwg);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const bf16_t* hr = hp + (long)min(g * 16 + col, B - 1) * H + q * 8;
        bf16x8_t hb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) hb[ks] = *reinterpret_cast<const bf16x8_t*>(hr + ks * 32);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, wf[mt][ks]),
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, hb[ks]), acc[mt][g], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int b = g * 16 + col, j = j0 + mt * 4 + q;
        if (b >= B) continue;
        const float gi = sigmoid_f(acc[mt][g][0] + gxv[mt][g][0]), gf = sigmoid_f(acc[mt][g][1] + gxv[mt][g][1]);
        const float gg = tanh_f(acc[mt][g][2] + gxv[mt][g][2]), go = sigmoid_f(acc[mt][g][3] + gxv[mt][g][3]);
        const float cp = cst[mt][g];
        float c = gf * cp + gi * gg;
        float h = go * tanh_f(c);
        const bool fz = a.frozen && a.frozen[(long)t * B + b];
        if (fz) {  // padded step of a packed sequence: the state passes through, the step emits zeros
          c = cp;
          h = 0.f;
        }
        cst[mt][g] = c;
        const long sb = (long)t * B + b;
        a.cs[sb * H + j] = c;
        a.hs[sb * H + j] = (fz && a.frozen_out_zero) ? (bf16_t)0 : f2bf(h);
        float* ap = a.act + sb * 4 * H + j;
        ap[0] = gi; ap[H] = gf; ap[2 * (long)H] = gg; ap[3 * (long)H] = go;
        if (a.h_last && n == U - 1) a.h_last[(long)b * H + j] = h;
      }
    if (n + 1 < U) grid_arrive(a.counter);
  }
}

struct LstmSeqBwdArgs {
  const bf16_t* dhs;       // [U*B][H] gradient of the emitted hidden states, or null
  const float* dh_last;    // [B][H] or null
  const float* dc_last;    // [B][H] or null
  const float* act;        // [U][B][4H]
  const float* cs;         // [U][B][H]
  const float* c0;         // [B][H] or null
  const uint8_t* frozen;   // [U][B] or null
  const bf16_t* w_hhT;     // [H][4H] (transposed recurrent weights)
  bf16_t* dG;              // [U*B][4H] gradient of the gate pre-activations (out; also the exchange buffer)
  float* dh0;              // [B][H] or null: gradient of the initial hidden state
  float* dc0;              // [B][H] or null
  unsigned* counter;
  int B, U, H, reverse;
};

template <int KS, int NG>
__global__ __launch_bounds__(256) void lstm_seq_bwd_kernel(const LstmSeqBwdArgs a) {
  __shared__ float part[2][4][NG][16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, col = lane & 15;
  const int H = a.H, B = a.B, U = a.U;
  const int j0 = blockIdx.x * 16;
  bf16x8_t wf[KS];  // W_hh^T rows j0 + (lane & 15), reduction range of this wave = gate `wave`
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    wf[ks] = *reinterpret_cast<const bf16x8_t*>(a.w_hhT + (long)(j0 + (lane & 15)) * 4 * H + (long)wave * H + ks * 32 + q * 8);
  const int tu = threadIdx.x >> 4, tb = threadIdx.x & 15;  // cell-stage ownership: unit j0 + tu, batch rows g * 16 + tb
  const int j = j0 + tu;
  float dcs[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int b = g * 16 + tb;
    dcs[g] = (a.dc_last && b < B) ? a.dc_last[(long)b * H + j] : 0.f;
  }
  const unsigned nwg = gridDim.x;
  unsigned arrivals = 0;
  // one extra pass (n = -1) turns the first step's gate gradients into the gradient of h0
  for (int n = U - 1; n >= (a.dh0 ? -1 : 0); --n) {
    const int t = n >= 0 ? (a.reverse ? U - 1 - n : n) : 0;
    const int tp = a.reverse ? t + 1 : t - 1;    // the step whose state fed this one (valid when n > 0)
    const int tnext = a.reverse ? t - 1 : t + 1;  // the step processed just before in this backward walk
    // everything the cell stage needs that does not depend on the recurrence is fetched before the barrier wait
    float gi[NG], gf[NG], gg[NG], go[NG], ct[NG], cp[NG], dh[NG];
    bool fz[NG];
    if (n >= 0) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int b = min(g * 16 + tb, B - 1);
        const long sb = (long)t * B + b;
        const float* ap = a.act + sb * 4 * H + j;
        gi[g] = ap[0]; gf[g] = ap[H]; gg[g] = ap[2 * (long)H]; go[g] = ap[3 * (long)H];
        ct[g] = a.cs[sb * H + j];
        cp[g] = n > 0 ? a.cs[((long)tp * B + b) * H + j] : (a.c0 ? a.c0[(long)b * H + j] : 0.f);
        dh[g] = a.dhs ? bf2f(a.dhs[sb * H + j]) : 0.f;
        fz[g] = a.frozen && a.frozen[sb];
      }
    }
    const int pb = n & 1;
    if (n < U - 1) {
      grid_wait(a.counter, arrivals * nwg);
      const int tsrc = n >= 0 ? tnext : (a.reverse ? U - 1 : 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const bf16_t* gr = a.dG + ((long)tsrc * B + min(g * 16 + col, B - 1)) * 4 * H + (long)wave * H + q * 8;
        bf16x8_t gb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) gb[ks] = *reinterpret_cast<const bf16x8_t*>(gr + ks * 32);
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, wf[ks]),
                                                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, gb[ks]), acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) part[pb][wave][g][q * 4 + r][col] = acc[r];
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float rec = part[pb][0][g][tu][tb] + part[pb][1][g][tu][tb] + part[pb][2][g][tu][tb] + part[pb][3][g][tu][tb];
        if (n >= 0) dh[g] += rec;
        else if (g * 16 + tb < B) a.dh0[(long)(g * 16 + tb) * H + j] = rec;
      }
    } else if (a.dh_last) {
#pragma unroll
      for (int g = 0; g < NG; ++g) dh[g] += a.dh_last[(long)min(g * 16 + tb, B - 1) * H + j];
    }
    if (n < 0) break;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int b = g * 16 + tb;
      if (b >= B) continue;
      bf16_t* d = a.dG + ((long)t * B + b) * 4 * H + j;
      if (fz[g]) {  // padded step: no dependence on the gates; the cell-state gradient passes through
        d[0] = d[H] = d[2 * (long)H] = d[3 * (long)H] = 0;
        continue;
      }
      const float tc = tanh_f(ct[g]);
      const float dc = dcs[g] + dh[g] * go[g] * (1.f - tc * tc);
      d[0] = f2bf(dc * gg[g] * gi[g] * (1.f - gi[g]));
      d[H] = f2bf(dc * cp[g] * gf[g] * (1.f - gf[g]));
      d[2 * (long)H] = f2bf(dc * gi[g] * (1.f - gg[g] * gg[g]));
      d[3 * (long)H] = f2bf(dh[g] * tc * go[g] * (1.f - go[g]));
      dcs[g] = dc * gf[g];
    }
    if (n > 0 || a.dh0) {
      grid_arrive(a.counter);
      ++arrivals;
    }
  }
  if (a.dc0) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (g * 16 + tb < B) a.dc0[(long)(g * 16 + tb) * H + j] = dcs[g];
  }
}

// The grid barrier needs every workgroup of the grid resident at the same time: the launch is refused (-> the caller's per-step
// path) unless the runtime's own occupancy answer for THIS kernel says the whole grid fits the device with room to spare (one
// workgroup per CU is all these grids ask for: nwg <= 64; the query guards against a build whose register / LDS use changed).
template <typename K>
static bool grid_fits(K kernel, int nwg) {
  int per_cu = 0, dev = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) return false;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  return nwg <= cus;  // (one slot per CU is enough; per_cu >= 1 was checked above)
}
template <int KS, int MT>
int launch_fwd(const LstmSeqFwdArgs& a, hipStream_t stream, bool probe_only = false) {
  const int nwg = a.H / (16 * MT);
  const int ng = (a.B + 15) / 16;
  static const bool ok1 = grid_fits(lstm_seq_fwd_kernel<KS, MT, 1>, 64), ok2 = grid_fits(lstm_seq_fwd_kernel<KS, MT, 2>, 64),
                    ok4 = grid_fits(lstm_seq_fwd_kernel<KS, MT, 4>, 64);
  if (!(ng == 1 ? ok1 : ng == 2 ? ok2 : ok4)) return -2;
  if (probe_only) return 0;
  if (ng == 1) hipLaunchKernelGGL((lstm_seq_fwd_kernel<KS, MT, 1>), dim3(nwg), dim3(256), 0, stream, a);
  else if (ng == 2) hipLaunchKernelGGL((lstm_seq_fwd_kernel<KS, MT, 2>), dim3(nwg), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((lstm_seq_fwd_kernel<KS, MT, 4>), dim3(nwg), dim3(256), 0, stream, a);
  return 0;
}
template <int KS>
int launch_bwd(const LstmSeqBwdArgs& a, hipStream_t stream, bool probe_only = false) {
  const int nwg = a.H / 16;
  const int ng = (a.B + 15) / 16;
  static const bool ok1 = grid_fits(lstm_seq_bwd_kernel<KS, 1>, 64), ok2 = grid_fits(lstm_seq_bwd_kernel<KS, 2>, 64),
                    ok4 = grid_fits(lstm_seq_bwd_kernel<KS, 4>, 64);
  if (!(ng == 1 ? ok1 : ng == 2 ? ok2 : ok4)) return -2;
  if (probe_only) return 0;
  if (ng == 1) hipLaunchKernelGGL((lstm_seq_bwd_kernel<KS, 1>), dim3(nwg), dim3(256), 0, stream, a);
  else if (ng == 2) hipLaunchKernelGGL((lstm_seq_bwd_kernel<KS, 2>), dim3(nwg), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((lstm_seq_bwd_kernel<KS, 4>), dim3(nwg), dim3(256), 0, stream, a);
  return 0;
}

}  // namespace

// 1 when the persistent kernels cover this shape (H a multiple of 32 among the instantiated sizes, at most 64 batch rows) AND the
// device can hold the whole grid at once: the grid barrier needs every workgroup resident (up to H / 16 = 64 workgroups of 256
// threads, 2 KB of LDS each; the CU count of the current device is the gate — small or partitioned devices fall back to the
// per-step path).  The device need NOT be idle: since round 3 the transducer's predictor LSTM runs on its own stream next to
// encoder kernels, the layer runtime's side stream and the joint's weight-gradient stream.  A workgroup of this grid then
// becomes resident as soon as any CU has 4 free wave slots; the kernels that hold them are ordinary launches that depend on
// nothing in this grid and end within a millisecond, so the whole grid is resident after at most that long, while grid_wait
// gives up (and traps, so that a broken launch cannot continue on stale state) only after ~2^26 polls of >= 64 cycles = several
// SECONDS.  What must never share the device with this kernel is another grid-barrier kernel whose grid cannot be placed
// next to it — the library has none: forward and backward recurrences of one model are ordered by autograd.
static int dispatch_fwd(const LstmSeqFwdArgs& a, hipStream_t stream, bool probe_only) {
  switch (a.H / 32) {
    case 8: return launch_fwd<8, 2>(a, stream, probe_only);
    case 10: return launch_fwd<10, 1>(a, stream, probe_only);  // 320 = 20 x 16: one tile per wave keeps every wave busy
    case 16: return launch_fwd<16, 2>(a, stream, probe_only);
    case 20: return launch_fwd<20, 1>(a, stream, probe_only);
    case 24: return launch_fwd<24, 1>(a, stream, probe_only);
    case 25: return launch_fwd<25, 1>(a, stream, probe_only);
    case 32: return launch_fwd<32, 1>(a, stream, probe_only);
    default: return -2;
  }
}
static int dispatch_bwd(const LstmSeqBwdArgs& a, hipStream_t stream, bool probe_only) {
  switch (a.H / 32) {
    case 8: return launch_bwd<8>(a, stream, probe_only);
    case 10: return launch_bwd<10>(a, stream, probe_only);
    case 16: return launch_bwd<16>(a, stream, probe_only);
    case 20: return launch_bwd<20>(a, stream, probe_only);
    case 24: return launch_bwd<24>(a, stream, probe_only);
    case 25: return launch_bwd<25>(a, stream, probe_only);
    case 32: return launch_bwd<32>(a, stream, probe_only);
    default: return -2;
  }
}

extern "C" int ea_lstm_seq_supported(int B, int H) {
  if (B < 1 || B > 64) return 0;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
  }();
  if (cus < H / 16) return 0;
  switch (H) {
    case 256: case 320: case 512: case 640: case 768: case 800: case 1024: break;
    default: return 0;
  }
  // the runtime's occupancy answer for the very kernels this shape would launch (see grid_fits)
  LstmSeqFwdArgs fa{};
  fa.B = B; fa.H = H;
  LstmSeqBwdArgs ba{};
  ba.B = B; ba.H = H;
  return dispatch_fwd(fa, nullptr, true) == 0 && dispatch_bwd(ba, nullptr, true) == 0 ? 1 : 0;
}

extern "C" int ea_lstm_seq_fwd(const float* gx, const void* w_hh, const void* h0, const float* c0, const uint8_t* frozen, void* hs,
                               float* cs, float* act, float* h_last, unsigned* counter, int B, int U, int H, int reverse,
                               int frozen_out_zero, hipStream_t stream) {
  if (U <= 0) return 0;
  if (!ea_lstm_seq_supported(B, H) || !counter) return -2;
  LstmSeqFwdArgs a{gx, (const bf16_t*)w_hh, (const bf16_t*)h0, c0, frozen, (bf16_t*)hs, cs, act, h_last, counter,
                   B, U, H, reverse, frozen_out_zero};
  if (hipMemsetAsync(counter, 0, 2 * sizeof(unsigned), stream) != hipSuccess) return -1;
  if (dispatch_fwd(a, stream, false) != 0) return -2;
  return EA_CHECK_LAUNCH();
}

extern "C" int ea_lstm_seq_bwd(const void* dhs, const float* dh_last, const float* dc_last, const float* act, const float* cs,
                               const float* c0, const uint8_t* frozen, const void* w_hhT, void* dG, float* dh0, float* dc0,
                               unsigned* counter, int B, int U, int H, int reverse, hipStream_t stream) {
  if (U <= 0) return 0;
  if (!ea_lstm_seq_supported(B, H) || !counter) return -2;
  LstmSeqBwdArgs a{(const bf16_t*)dhs, dh_last, dc_last, act, cs, c0, frozen, (const bf16_t*)w_hhT, (bf16_t*)dG, dh0, dc0, counter,
                   B, U, H, reverse};
  if (hipMemsetAsync(counter, 0, 2 * sizeof(unsigned), stream) != hipSuccess) return -1;
  if (dispatch_bwd(a, stream, false) != 0) return -2;
  return EA_CHECK_LAUNCH();
}
