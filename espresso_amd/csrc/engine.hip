// Native layer runtime: one C-ABI call runs a whole Conformer encoder layer forward (≈30 kernel launches)
// or backward (≈65 launches) back-to-back on a HIP stream, with activations in caller-provided arenas
// and parameter gradients accumulated straight into the flat fp32 gradient buffer.
//
// Why: with one Python/ctypes call per kernel the host needed ≈43 ms to enqueue a 40 ms GPU step
// (≈1400 launches); the reference has the same structure (eager PyTorch, ~20 kernels per layer forward,
// SURVEY §8a8).  The sequence below is the hand-scheduled composition of
// espresso/modules/conformer_with_relative_positional_embedding_encoder_layer.py:81-145 (ffn1 -> rel-pos MHSA ->
// conv module -> ffn2 -> final LayerNorm) and its analytic backward; it is identical, launch for launch, to
// the Python composition in espresso_amd/functional.py (which remains as the reference for tests).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <cstdio>
#include <map>
#include <mutex>
#include <vector>

#include "espresso_amd.h"

void ea_gemm_corun_hint(int on);  // gemm.hip: the launches that follow share the device with side-stream work

namespace {

struct Arena {
  char* base;
  size_t off;
  size_t peak;
  template <typename T>
  T* get(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    if (off > peak) peak = off;
    return p;
  }
};

inline int pad8(int n) { return (n + 7) / 8 * 8; }
// Row pitch of the attention backward's dBD buffer (gradient of the raw positional logits, [H*B][T][pitch], 2T-1 columns used):
// whole 128-column tiles, so that the pos_proj gradient product over it can ride in the layer's grouped weight-gradient launch
// (direct-to-LDS kernel: ea_wgrad_group needs ld_x >= K rounded up to 128); the pad columns are never read as results.
inline int pad_bd(int r) { return (r + 127) / 128 * 128; }

// Dropout mask streams of a layer call.  Every dropout site of a layer draws its keep decisions from ea_keep(site seed, element
// index) with site seed = EaLayerShape.seed + module base + site offset; the same sum is formed again in the backward pass, so
// no mask is stored.  ea_layer_dropout_seed (below, C ABI) publishes the table: the parity tests rebuild every mask from it.
constexpr uint64_t kFfn1 = 0, kAttn = 16, kConv = 32, kCross = 32, kFfn2 = 48;         // module bases
constexpr uint64_t kAct = 1, kOut = 2, kProbs = 3, kAttnOut = 4, kConvOut = 5;           // site offsets inside a module

// Deferred side work of ONE layer backward (round 2): everything that only feeds the optimizer — weight / bias gradients,
// LayerNorm / BatchNorm / depthwise-filter parameter gradients, the pos_proj chain — is collected while the data-gradient chain
// is enqueued on the main stream and launched afterwards on the side stream behind a single fork: all Linear weight and bias
// gradients of the layer as ONE grouped GEMM launch (ea_wgrad_group), all LayerNorm parameter reduces as one launch.  The
// side work of layer k then runs next to the main chain of layer k-1, whose call joins it at its end; the two calls use
// different halves of the scratch arena.  Per layer: ~36 launches and 3 stream-event operations instead of ~85 + ~28.
struct Deferred {
  EaWgradGroup grp;
  EaLnReduceGroup ln;
  std::vector<std::function<int(hipStream_t)>> pre;  // before the grouped launch (e.g. zeroing an fp32 product it accumulates into)
  std::vector<std::function<int(hipStream_t)>> ops;  // after it, in order
  Deferred() { clear(); }
  void clear() { grp.count = 0; ln.count = 0; pre.clear(); ops.clear(); }
};

struct Ctx {
  hipStream_t s;
  bool dry;  // size computation only: walk the arenas, launch nothing
  int rc;
  Arena* scratch;
  // Backward overlap: weight-gradient GEMMs and bias column-sums only feed the optimizer, so they run on a side
  // stream concurrently with the dgrad chain on `s` (fills the tails of the small launches).  While overlap is on, the
  // scratch arena is not recycled inside a layer (a side kernel may still be reading a temporary).
  hipStream_t side;
  bool overlap;
  Deferred* df = nullptr;  // non-null: deferred mode (also in the sizing pass, so that both walk the arena alike)
  const EaLayerChain* chain = nullptr;  // Conformer layer calls chained to their neighbours (ea_conformer_layer_*_chained)
  bool bits_early = false;  // the attention block's keep-bit kernel was already launched on the side stream (start of the layer call)
  bool pp_early = false;    // ... and its positional-table projection
};

#define RUN(call)                         \
  do {                                    \
    if (!c.dry && c.rc == 0) c.rc = (call); \
  } while (0)

// process-wide side stream + a small ring of events (created on first use, on the device that owns the caller's stream: one
// process drives one GPU, but the calling thread's current device is whatever the host framework left it at)
struct SideRes {
  hipStream_t stream = nullptr;
  hipEvent_t ev[32];
  int next = 0;
  bool ok = false;
  hipEvent_t done[2];                // deferred mode: side work that reads scratch half h has been enqueued up to here
  bool pending[2] = {false, false};  // ... and the main stream has not joined it yet
};
static SideRes g_side;
// The side stream must not share a HARDWARE QUEUE with the caller's stream, or nothing overlaps.  HIP deals a process's streams
// onto GPU_MAX_HW_QUEUES (4) queues by load, so whether two streams collide depends on how many streams the process created before
// (measured, round 5, profiles/r05_side_stream_queues.txt: with a one-rank RCCL group in the process the weight-gradient launches
// landed on the main stream's queue — 13.7 -> 15.3 ms per step).  Neither cure offered by the runtime is safe: more hardware queues
// (GPU_MAX_HW_QUEUES = 8) or streams in other priority classes (queues are pooled per class) put MORE than four queues to work and
// the command processor time-slices them — config 3 with a process group 29 ms, the transducer step 76 ms instead of 30.  So the
// collision is MEASURED: a candidate stream is accepted when a tiny kernel on it finishes while a 200 us spin kernel is still
// running on the caller's stream; up to 8 candidates (streams are created round-robin over the queues).
__global__ void ea_queue_probe_spin(long ticks) {  // one wavefront, ~ticks of the 100 MHz constant clock
  const long t0 = (long)wall_clock64();
  while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void ea_queue_probe_nop() {}
}  // namespace
// 1: work on `b` waits for work on `a` (same hardware queue), 0: they run side by side, < 0: error.  Blocks the host until `a`
// reaches the probe (once per stream pair, at set-up).
extern "C" int ea_streams_share_queue(hipStream_t a, hipStream_t b) {
  if (a == b) return 1;
  hipEvent_t a0, a1, b1;
  if (hipEventCreate(&a0) != hipSuccess) return -1;
  if (hipEventCreate(&a1) != hipSuccess) { (void)hipEventDestroy(a0); return -1; }
  if (hipEventCreate(&b1) != hipSuccess) { (void)hipEventDestroy(a0); (void)hipEventDestroy(a1); return -1; }
  int rc = -1;
  float spin_ms = 0.f, b_ms = 0.f;
  // b is warmed (a stream's first launch creates its queue: milliseconds) and idle-waited first, so that only the queue
  // placement, not b's own start-up or backlog, decides when its kernel runs
  hipLaunchKernelGGL(ea_queue_probe_nop, dim3(1), dim3(64), 0, b);
  if (hipStreamSynchronize(b) == hipSuccess && hipEventRecord(a0, a) == hipSuccess) {
    hipLaunchKernelGGL(ea_queue_probe_spin, dim3(1), dim3(64), 0, a, 20000L);
    if (hipEventRecord(a1, a) == hipSuccess) {
      hipLaunchKernelGGL(ea_queue_probe_nop, dim3(1), dim3(64), 0, b);
      if (hipEventRecord(b1, b) == hipSuccess && hipEventSynchronize(a1) == hipSuccess && hipEventSynchronize(b1) == hipSuccess &&
          hipEventElapsedTime(&spin_ms, a0, a1) == hipSuccess && hipEventElapsedTime(&b_ms, a0, b1) == hipSuccess)
        rc = b_ms >= 0.75f * spin_ms ? 1 : 0;  // (b_ms < 0: b finished before a even reached the probe)
      static const bool dbg = getenv("EA_SIDE_STREAM_DEBUG") != nullptr;
      if (dbg) fprintf(stderr, "[espresso_amd] queue probe %p vs %p: spin %.3f ms, other stream done after %.3f ms -> %s\n", (void*)a, (void*)b,
                       spin_ms, b_ms, rc == 1 ? "SAME queue" : rc == 0 ? "separate queues" : "error");
    }
  }
  (void)hipEventDestroy(a0);
  (void)hipEventDestroy(a1);
  (void)hipEventDestroy(b1);
  return rc;
}
namespace {
static const bool g_side_probe = [] { const char* e = getenv("EA_SIDE_STREAM_PROBE"); return !(e && e[0] == '0'); }();
static int g_side_rejected = -1;   // candidates the probe turned down before one was accepted (-1: no side stream yet)
static int g_side_unprobed = 0;    // 1: the accepted stream was NOT measured (probe switched off, or the eighth candidate)
static bool side_stream_create(hipStream_t owner) {  // (current device = the owner's)
  hipStream_t rejected[8];
  int nrej = 0;
  bool ok = false;
  for (int t = 0; t < 8; ++t) {
    hipStream_t cand = nullptr;
    if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
    const bool unprobed = !g_side_probe || t == 7;
    if (unprobed || ea_streams_share_queue(owner, cand) != 1) {
      g_side.stream = cand;
      g_side_rejected = nrej;
      g_side_unprobed = unprobed ? 1 : 0;
      ok = true;
      break;
    }
    rejected[nrej++] = cand;  // (kept alive until the choice is made: a destroyed stream's queue slot would be dealt again)
  }
  for (int i = 0; i < nrej; ++i) (void)hipStreamDestroy(rejected[i]);
  return ok;
}
static bool side_init(hipStream_t owner) {
  if (g_side.ok) return true;
  int cur = 0, dev = 0;
  if (hipGetDevice(&cur) != hipSuccess) return false;
  dev = cur;
  if (owner != nullptr) {
    hipDevice_t d;
    if (hipStreamGetDevice(owner, &d) == hipSuccess) dev = (int)d;
  }
  if (dev != cur && hipSetDevice(dev) != hipSuccess) return false;
  bool ok = side_stream_create(owner);
  for (auto& e : g_side.ev)
    ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  for (auto& e : g_side.done)
    ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  if (dev != cur) (void)hipSetDevice(cur);
  g_side.ok = ok;
  return ok;
}
// make `dst` wait for everything enqueued so far on `src`
static inline void stream_wait(Ctx& c, hipStream_t dst, hipStream_t src) {
  if (c.dry || c.rc != 0) return;
  hipEvent_t e = g_side.ev[g_side.next];
  g_side.next = (g_side.next + 1) % 32;
  if (hipEventRecord(e, src) != hipSuccess || hipStreamWaitEvent(dst, e, 0) != hipSuccess) c.rc = -1;
}
static inline void* ln_ws(Ctx& c, int M, int C) {
  return c.scratch->get<char>((size_t)ea_layernorm_bwd_workspace_bytes(M, C));
}
// stream for optimizer-only products (wgrad / bias sums)
static inline hipStream_t wstream(Ctx& c) { return c.overlap ? c.side : c.s; }
static inline void fork(Ctx& c) { if (c.overlap && !c.df) stream_wait(c, c.side, c.s); }
// main stream joins the deferred side work that used scratch half h (no-op when nothing is pending)
static inline int join_half(hipStream_t main, int h) {
  if (!g_side.ok || !g_side.pending[h]) return 0;
  g_side.pending[h] = false;
  return hipStreamWaitEvent(main, g_side.done[h], 0) == hipSuccess ? 0 : -1;
}
static inline int join_all(hipStream_t main) {
  const int a = join_half(main, 0), b = join_half(main, 1);
  return a ? a : b;
}
static inline void release(Ctx& c, size_t mark) { if (!c.overlap) c.scratch->off = mark; }

inline uint32_t drop_thr(float p) {
  if (p <= 0.f) return 0;
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
}
inline float drop_scale(float p) { return p <= 0.f ? 1.f : 1.f / (1.f - p); }

// plain GEMM helper: C = A B^T style with the flags used on the path
struct G {
  EaGemmParams p;
  G(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc) {
    memset(&p, 0, sizeof(p));
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.batch = 1; p.zdiv = 1; p.alpha = 1.f; p.out_scale = 1.f; p.splitk = 1; p.drop_scale = 1.f;
  }
  G& aks() { p.a_kstrided = 1; return *this; }
  G& bks() { p.b_kstrided = 1; return *this; }
  G& f32() { p.c_f32 = 1; return *this; }
  G& acc() { p.accumulate = 1; return *this; }
  G& bias(const float* b) { p.bias = b; return *this; }
  G& act(int a) { p.act = a; return *this; }
  G& alpha(float a) { p.alpha = a; return *this; }
  G& scale(float a) { p.out_scale = a; return *this; }
  G& resid(const void* r, long ldr) { p.resid = r; p.ldr = ldr; return *this; }
  G& c2(void* c, long ld) { p.C2 = c; p.ldc2 = ld; return *this; }
  G& aux(const void* x, long ld) { p.aux = x; p.ldaux = ld; return *this; }
  G& drop(float pr, uint64_t seed) { p.drop_thr = drop_thr(pr); p.drop_scale = drop_scale(pr); p.drop_seed = seed; return *this; }
  // columns [0, n) leave as the attention kernels' query operands: q_u = (q + pos_u) * s, q_v = (q + pos_v) * s (no launch of
  // ea_relpos_q_prep); false when the shape does not qualify (the caller then runs ea_relpos_q_prep)
  bool qsplit(void* q_u, void* q_v, const float* pos_u, const float* pos_v, int n, long ld_q, float s) {
    static const bool off = getenv("EA_NO_QSPLIT") != nullptr;  // (diagnostic A/B switch)
    if (off || n % 128 || p.N % 8) return false;
    p.q_u = q_u; p.q_v = q_v; p.pos_u = pos_u; p.pos_v = pos_v; p.qsplit_n = n; p.ld_q = ld_q; p.qscale = s;
    return true;
  }
  G& batch(int b, int zdiv, long ahi, long alo, long bhi, long blo, long chi, long clo) {
    p.batch = b; p.zdiv = zdiv; p.sA_hi = ahi; p.sA_lo = alo; p.sB_hi = bhi; p.sB_lo = blo; p.sC_hi = chi; p.sC_lo = clo;
    return *this;
  }
};

inline void gemm(Ctx& c, G& g) {
  if (!c.dry && c.rc == 0) ea_gemm_corun_hint(c.overlap || c.df != nullptr);
  RUN(ea_gemm_bf16(&g.p, c.s));
  if (!c.dry) ea_gemm_corun_hint(0);
}
inline void gemm_on(Ctx& c, G& g, hipStream_t st) { RUN(ea_gemm_bf16(&g.p, st)); }

// dW[N_out][K_in] += dy^T x (and dbias[N_out] += column sums of dy).  Deferred mode: one more problem of the layer's grouped
// launch.  Immediate mode: two-pass split-K GEMM (workspace from the scratch arena) + a column-sum launch.
inline void bias_grad(Ctx& c, const void* X, float* out, int M, int N, long ld) {
  if (c.df) {
    if (!c.dry) c.df->ops.push_back([=](hipStream_t st) { return ea_colsum_bf16(X, out, M, N, ld, st); });
    return;
  }
  RUN(ea_colsum_bf16(X, out, M, N, ld, wstream(c)));
}
inline void wgrad(Ctx& c, const void* dy, long ld_dy, const void* x, long ld_x, float* dW, int M, int N_out, int K_in,
                  float* dbias = nullptr) {
  if (c.df && c.df->grp.count < EA_WGRAD_MAX) {  // (the sizing pass counts too: both passes take the same branch)
    EaWgradProblem& q = c.df->grp.p[c.df->grp.count++];
    if (c.dry) return;
    q.dy = dy; q.x = x; q.dW = dW; q.dbias = dbias; q.M = M; q.N = N_out; q.K = K_in; q.ld_dy = ld_dy; q.ld_x = ld_x; q.ldw = K_in;
    return;
  }
  // 64x128 output tiles; aim at ~512 resident workgroups (2 per CU): deeper splits only add slab traffic for the
  // reduce pass (measured on the 2048x512 / 512x512 weight gradients at M = 6468)
  const int tiles = ((N_out + 63) / 64) * ((K_in + 127) / 128);
  int sk = 1;
  if (tiles < 384 && M >= 1024) {
    sk = (512 + tiles - 1) / tiles;
    if (sk > M / 256) sk = M / 256;
    if (sk < 1) sk = 1;
  }
  G g(dy, x, dW, N_out, K_in, M, ld_dy, ld_x, K_in);
  g.aks().bks().f32().acc();
  g.p.splitk = sk;
  if (sk > 1) g.p.workspace = c.scratch->get<float>((size_t)sk * N_out * K_in);
  if (c.df) {  // group full (never with the layers of this runtime): run it with the other deferred side work
    const EaGemmParams gp = g.p;
    if (!c.dry) c.df->ops.push_back([=](hipStream_t st) { return ea_gemm_bf16(&gp, st); });
  } else {
    gemm_on(c, g, wstream(c));
  }
  if (dbias) bias_grad(c, dy, dbias, M, N_out, ld_dy);
}
// side-stream work after the main chain of a layer backward: ONE fork, the grouped launches, the remaining small kernels; then
// the main stream joins the side work of the PREVIOUS backward call (other scratch half), which ran next to this call's chain
static bool g_defer_default = true;  // ea_set_backward_deferred
static bool g_defer_inline = false;  // A/B switch: run the deferred work on the MAIN stream at the end of the layer (no overlap)
static void run_deferred(Ctx& c, int half) {
  if (c.dry || !c.df) return;
  Deferred& d = *c.df;
  if (g_defer_inline) {
    RUN(ea_wgrad_group(&d.grp, c.s));
    RUN(ea_layernorm_param_reduce_group(&d.ln, c.s));
    for (auto& op : d.ops) RUN(op(c.s));
    d.clear();
    return;
  }
  stream_wait(c, c.side, c.s);
  for (auto& op : d.pre) RUN(op(c.side));
  RUN(ea_wgrad_group(&d.grp, c.side));
  RUN(ea_layernorm_param_reduce_group(&d.ln, c.side));
  for (auto& op : d.ops) RUN(op(c.side));
  d.clear();
  if (c.rc == 0) {
    if (hipEventRecord(g_side.done[half], c.side) != hipSuccess) c.rc = -1;
    g_side.pending[half] = true;
  }
  if (c.rc == 0) c.rc = join_half(c.s, half ^ 1);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
extern "C" {

// k-contiguous (transposed) copies of the weights whose data-gradient GEMM has a 512-wide output and a long reduction:
// layout of EaConformerLayer::wt, refreshed by every training forward (the weights only change in the optimizer step)
struct WT {
  const uint16_t *f1w1, *f2w1, *wqkv, *wo, *pw1, *pw2, *f1w2, *f2w2;
};
static WT wt_view(const EaConformerLayer* L, const EaLayerShape& sh) {
  WT w{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (!L->wt || !sh.training) return w;
  const size_t C = sh.C, F = sh.F;
  const uint16_t* p = (const uint16_t*)L->wt;
  w.f1w1 = p; p += C * F;
  w.f2w1 = p; p += C * F;
  w.wqkv = p; p += 3 * C * C;
  w.wo = p; p += C * C;
  w.pw1 = p; p += 2 * C * C;
  w.pw2 = p; p += C * C;
  w.f1w2 = p; p += C * F;
  w.f2w2 = p;
  return w;
}
static void wt_refresh(Ctx& c, const EaConformerLayer* L, const EaLayerShape& sh) {
  const WT w = wt_view(L, sh);
  if (!w.f1w1) return;
  const int C = sh.C, F = sh.F;
  const void* src[8] = {L->ffn1.w1, L->ffn2.w1, L->attn.wqkv, L->attn.wo, L->conv.pw1, L->conv.pw2, L->ffn1.w2, L->ffn2.w2};
  void* dst[8] = {(void*)w.f1w1, (void*)w.f2w1, (void*)w.wqkv, (void*)w.wo, (void*)w.pw1, (void*)w.pw2, (void*)w.f1w2, (void*)w.f2w2};
  const int rows[8] = {F, F, 3 * C, C, 2 * C, C, C, C}, cols[8] = {C, C, C, C, C, C, F, F};
  RUN(ea_transpose_bf16_batch(src, dst, rows, cols, 8, c.s));
}
// what the NEXT residual block of the backward wants as its incoming gradient: a * dropout(dx) with its own mask; produced
// by the LayerNorm backward of the current block as a second output (no separate pass over the gradient)
struct Pre {
  uint16_t* buf;   // nullptr: nothing to produce
  float a;
  uint64_t seed;
  float p;
};
// dgamma / dbeta += the partial slab of a LayerNorm backward: optimizer-only, so deferred / on the side stream
static inline void ln_param_reduce(Ctx& c, void* lnws, float* dg, float* db, int M, int C) {
  if (c.df) {
    if (c.dry) return;
    if (c.df->ln.count < EA_LNRED_MAX) {
      EaLnReduceItem& it = c.df->ln.item[c.df->ln.count++];
      it.workspace = lnws; it.dgamma = dg; it.dbeta = db; it.M = M; it.C = C;
    } else {
      c.df->ops.push_back([=](hipStream_t st) { return ea_layernorm_param_reduce(lnws, dg, db, M, C, st); });
    }
    return;
  }
  fork(c);  // the parameter-gradient reduce only feeds the optimizer
  RUN(ea_layernorm_param_reduce(lnws, dg, db, M, C, wstream(c)));
}
static inline void ln_bwd_block(Ctx& c, const void* x, const void* dxn, const float* gamma, const float* mean, const float* rstd,
                                void* dx, float* dg, float* db, int M, int C, const void* dx_add, const Pre& next) {
  void* lnws = ln_ws(c, M, C);  // outside RUN(): the dry (sizing) pass must count it too
  if (next.buf)
    RUN(ea_layernorm_bwd_dx2(x, dxn, gamma, mean, rstd, dx, dg, db, M, C, dx_add, lnws, next.buf, next.a, next.seed, drop_thr(next.p),
                             drop_scale(next.p), c.s));
  else
    RUN(ea_layernorm_bwd_dx(x, dxn, gamma, mean, rstd, dx, dg, db, M, C, nullptr, 0, 0, 1.f, dx_add, lnws, c.s));
  ln_param_reduce(c, lnws, dg, db, M, C);
}
// The last LayerNorm backward of a CHAINED layer call: this layer's ffn1 norm (x, dxn, dx_add as above) and, on its result, the
// PREVIOUS layer's final norm, in one kernel.  dx receives the gradient w.r.t. the previous layer's pre-final-norm activations.
struct ChainBwd {
  const void* x4; const float *g, *mean, *rstd; float *dg, *db;
  void* lnws;  // second partial slab (allocated by layer_bwd in every mode: the arena layout does not depend on chaining)
  Pre pre;
};
static inline void ln_bwd2_block(Ctx& c, const void* x, const void* dxn, const float* gamma, const float* mean, const float* rstd,
                                 void* dx, float* dg, float* db, int M, int C, const void* dx_add, const ChainBwd& cb) {
  void* lnws = ln_ws(c, M, C);
  RUN(ea_layernorm_bwd2_dx(x, dxn, gamma, mean, rstd, dx_add, lnws, cb.x4, cb.g, cb.mean, cb.rstd, cb.lnws, dx, M, C, cb.pre.buf, cb.pre.a,
                           cb.pre.seed, drop_thr(cb.pre.p), drop_scale(cb.pre.p), c.s));
  ln_param_reduce(c, lnws, dg, db, M, C);
  ln_param_reduce(c, cb.lnws, cb.dg, cb.db, M, C);
}

// data gradient dx[M][N] = dy[M][K] W[K][N]: k-contiguous copy Wt[N][K] when available, else W read k-strided
static inline void dgrad(Ctx& c, const void* dy, const void* W, const uint16_t* Wt, void* dx, int M, int N, int K) {
  if (Wt) {
    G g(dy, Wt, dx, M, N, K, K, K, N);
    gemm(c, g);
  } else {
    G g(dy, W, dx, M, N, K, K, N, N);
    g.bks();
    gemm(c, g);
  }
}

struct FfnSaved {
  float *mean, *rstd;
  uint16_t *xn, *z, *h;
};
static FfnSaved ffn_saved(Arena& sv, const EaLayerShape& sh) {
  const int M = sh.B * sh.T, C = sh.C, F = sh.F;
  FfnSaved f;
  f.mean = sv.get<float>(M);
  f.rstd = sv.get<float>(M);
  f.xn = sv.get<uint16_t>((size_t)M * C);
  f.z = sv.get<uint16_t>((size_t)M * F);
  f.h = sv.get<uint16_t>((size_t)M * F);
  return f;
}

static void ffn_fwd(Ctx& c, const FfnSaved& f, const EaLayerShape& sh, const EaFfnParams& w, const void* x, void* y,
                    uint64_t seed, float out_scale, int act, bool ln_done = false) {
  const int M = sh.B * sh.T, C = sh.C, F = sh.F;
  float *mean = f.mean, *rstd = f.rstd;
  uint16_t *xn = f.xn, *z = f.z, *h = f.h;
  // (ln_done: the previous layer's chained call wrote xn / mean / rstd with its own final LayerNorm)
  if (!ln_done) RUN(ea_layernorm_fwd(x, w.ln_g, w.ln_b, xn, mean, rstd, M, C, 1e-5f, nullptr, 0, 0, 1.f, c.s));
  G g1(xn, w.w1, z, M, F, C, C, C, F);
  g1.bias(w.b1).act(act).c2(h, F).drop(sh.p_act, seed + kAct);
  gemm(c, g1);
  G g2(h, w.w2, y, M, C, F, F, F, C);
  g2.bias(w.b2).drop(sh.p_drop, seed + kOut).scale(out_scale).resid(x, C);
  gemm(c, g2);
}

static void ffn_bwd(Ctx& c, const FfnSaved& f, const EaLayerShape& sh, const EaFfnParams& w, const EaFfnGrads& gw, const void* x,
                    const void* dy, void* dx, uint64_t seed, float out_scale, int act, const uint16_t* w1t, const uint16_t* w2t,
                    const uint16_t* pre, const Pre& next, const ChainBwd* chain = nullptr) {
  const int M = sh.B * sh.T, C = sh.C, F = sh.F;
  Arena& sc = *c.scratch;
  const size_t mark = sc.off;
  float *mean = f.mean, *rstd = f.rstd;
  uint16_t *xn = f.xn, *z = f.z, *h = f.h;
  const uint16_t* g2 = pre;  // out_scale * dropout(dy), already written by the previous block's LayerNorm backward
  if (!g2) {
    uint16_t* gb = sc.get<uint16_t>((size_t)M * C);
    RUN(ea_scale_dropout_bf16(dy, nullptr, gb, (long)M * C, out_scale, 0.f, seed + kOut, drop_thr(sh.p_drop), drop_scale(sh.p_drop), c.s));
    g2 = gb;
  }
  fork(c);
  wgrad(c, g2, C, h, F, gw.w2, M, C, F, gw.b2);
  uint16_t* dz = sc.get<uint16_t>((size_t)M * F);
  G gd(g2, w2t ? (const void*)w2t : w.w2, dz, M, F, C, C, w2t ? C : F, F);
  if (!w2t) gd.bks();
  gd.aux(z, F).act(act).drop(sh.p_act, seed + kAct);
  gemm(c, gd);
  fork(c);
  wgrad(c, dz, F, xn, C, gw.w1, M, F, C, gw.b1);
  uint16_t* dxn = sc.get<uint16_t>((size_t)M * C);
  dgrad(c, dz, w.w1, w1t, dxn, M, C, F);
  if (chain) ln_bwd2_block(c, x, dxn, w.ln_g, mean, rstd, dx, gw.ln_g, gw.ln_b, M, C, dy, *chain);
  else ln_bwd_block(c, x, dxn, w.ln_g, mean, rstd, dx, gw.ln_g, gw.ln_b, M, C, dy, next);
  release(c, mark);
}

// saved layout of the attention block is produced by the same get<> sequence in fwd and bwd
static bool g_flash = true;
static bool g_flash_bits_side = [] { const char* e = getenv("EA_FLASH_BITS_INLINE"); return !(e && e[0] == '1'); }();
// fused (flash) attention: head dim 64 and no additive attention mask; decided from the shape alone so that the saved
// layout of forward and backward agree
static inline bool attn_fused(const EaLayerShape& sh) {
  return g_flash && !sh.has_attn_mask && ea_flash_attention_supported(sh.C / sh.H, sh.T, sh.T, 1);
}
struct AttnSaved {
  float *mean, *rstd, *lse;
  uint16_t *xn, *qkv, *qu, *qv, *pp, *P, *Pd, *o;
  uint16_t* bits;  // attention-dropout keep bits of the fused rel-pos kernels (NULL without dropout)
};
static AttnSaved attn_saved(Arena& sv, const EaLayerShape& sh) {
  const int M = sh.B * sh.T, C = sh.C, T = sh.T, Z = sh.H * sh.B, Sp = pad8(T), R = 2 * T - 1;
  AttnSaved a;
  if (attn_fused(sh)) {
    a.mean = sv.get<float>(M);
    a.rstd = sv.get<float>(M);
    a.lse = sv.get<float>((size_t)Z * T);
    a.xn = sv.get<uint16_t>((size_t)M * C);
    a.qkv = sv.get<uint16_t>((size_t)M * 3 * C);
    a.qu = sv.get<uint16_t>((size_t)M * C);
    a.qv = sv.get<uint16_t>((size_t)M * C);
    a.pp = sv.get<uint16_t>((size_t)R * C);
    a.P = a.Pd = nullptr;
    a.o = sv.get<uint16_t>((size_t)M * C);
    a.bits = sh.p_attn > 0.f ? sv.get<uint16_t>((size_t)ea_flash_keep_bits_bytes(sh.H, sh.B, T) / 2) : nullptr;
    return a;
  }
  a.lse = nullptr;
  a.bits = nullptr;
  a.mean = sv.get<float>(M);
  a.rstd = sv.get<float>(M);
  a.xn = sv.get<uint16_t>((size_t)M * C);
  a.qkv = sv.get<uint16_t>((size_t)M * 3 * C);
  a.qu = sv.get<uint16_t>((size_t)M * C);
  a.qv = sv.get<uint16_t>((size_t)M * C);
  a.pp = sv.get<uint16_t>((size_t)R * C);
  a.P = sv.get<uint16_t>((size_t)Z * T * Sp);
  a.Pd = sh.p_attn > 0.f ? sv.get<uint16_t>((size_t)Z * T * Sp) : a.P;
  a.o = sv.get<uint16_t>((size_t)M * C);
  return a;
}

// attention-dropout keep bits of a fused attention block on the side stream (data independent: only the seed and the shape); the
// side stream first waits for the main stream — an earlier backward may still be reading the buffer.  true: launched.
static bool keep_bits_on_side(Ctx& c, const AttnSaved& a, const EaLayerShape& sh, uint64_t seed, const EaAttnParams* w = nullptr,
                              const void* pe = nullptr) {
  if (!(a.bits && attn_fused(sh) && !c.dry && c.rc == 0 && g_flash_bits_side && side_init(c.s))) return false;
  hipEvent_t e0 = g_side.ev[g_side.next];
  g_side.next = (g_side.next + 1) % 32;
  if (hipEventRecord(e0, c.s) != hipSuccess || hipStreamWaitEvent(g_side.stream, e0, 0) != hipSuccess) c.rc = -1;
  RUN(ea_flash_keep_bits(a.bits, sh.H, sh.B, sh.T, seed + kProbs, drop_thr(sh.p_attn), g_side.stream));
  if (w && pe && sh.pos_mode != 1) {  // ... and the projected positional table (weights x constant table: no activation in it either)
    const int R = 2 * sh.T - 1, C = sh.C;
    G gpp(pe, w->wpos, a.pp, R, C, C, C, C, C);
    gemm_on(c, gpp, g_side.stream);
    c.pp_early = true;
  }
  return true;
}

static void attn_fwd(Ctx& c, const AttnSaved& a, const EaLayerShape& sh, const EaAttnParams& w, const void* x, void* y,
                     const int* key_len, const float* attn_mask, const void* pe, uint64_t seed) {
  const int B = sh.B, T = sh.T, C = sh.C, H = sh.H, M = B * T, dh = C / H, Z = H * B, Sp = pad8(T), R = 2 * T - 1, Rp = pad8(R);
  const float scaling = 1.0f / sqrtf((float)dh);
  Arena& sc = *c.scratch;
  const size_t mark = sc.off;
  // attention-dropout keep bits: data independent, so the bit kernel runs on the side stream under the LayerNorm / QKV GEMM
  // of this block (the side stream first waits for the main stream: an earlier backward may still be reading the buffer)
  bool bits_on_side = c.bits_early;
  c.bits_early = false;
  if (!bits_on_side) bits_on_side = keep_bits_on_side(c, a, sh, seed);
  RUN(ea_layernorm_fwd(x, w.ln_g, w.ln_b, a.xn, a.mean, a.rstd, M, C, 1e-5f, nullptr, 0, 0, 1.f, c.s));
  G gq(a.xn, w.wqkv, a.qkv, M, 3 * C, C, C, C, 3 * C);
  gq.bias(w.bqkv);
  // positional mode 1 (learned relative table, multihead_attention.py:806-818): `pe` IS the bf16 [2T-1][C] slice of the table,
  // plain scaled queries for both terms, no pos_proj / pos_bias_u / pos_bias_v
  const bool learned = sh.pos_mode == 1;
  const uint16_t* pp = learned ? (const uint16_t*)pe : a.pp;
  const uint16_t* qvv = learned ? a.qu : a.qv;
  // q + pos_bias_u / q + pos_bias_v, scaled: written by the projection's epilogue (the q third of a.qkv then stays unwritten;
  // nothing reads it), by ea_relpos_q_prep for widths the epilogue does not take
  const bool split = learned ? gq.qsplit(a.qu, nullptr, nullptr, nullptr, C, C, scaling)
                             : gq.qsplit(a.qu, a.qv, w.pos_u, w.pos_v, C, C, scaling);
  gemm(c, gq);
  if (!split) {
    if (learned) RUN(ea_relpos_q_prep(a.qkv, 3 * C, nullptr, nullptr, a.qu, nullptr, M, C, scaling, c.s));
    else RUN(ea_relpos_q_prep(a.qkv, 3 * C, w.pos_u, w.pos_v, a.qu, a.qv, M, C, scaling, c.s));
  }
  if (attn_fused(sh)) {
    if (!learned && !c.pp_early) {
      G gpp(pe, w.wpos, a.pp, R, C, C, C, C, C);
      gemm(c, gpp);
    }
    c.pp_early = false;
    if (bits_on_side) {
      hipEvent_t e1 = g_side.ev[g_side.next];
      g_side.next = (g_side.next + 1) % 32;
      if (c.rc == 0 && (hipEventRecord(e1, g_side.stream) != hipSuccess || hipStreamWaitEvent(c.s, e1, 0) != hipSuccess)) c.rc = -1;
    }
    RUN(ea_flash_attention_fwd(a.qu, qvv, C, a.qkv + C, a.qkv + 2 * C, 3 * C, pp, C, key_len, a.o, C, a.lse, H, B, T, T, dh,
                               bits_on_side ? 4 : 0, seed + kProbs, drop_thr(sh.p_attn), drop_scale(sh.p_attn), a.bits, c.s));
    G go(a.o, w.wo, y, M, C, C, C, C, C);
    go.bias(w.bo).drop(sh.p_drop, seed + kAttnOut).resid(x, C);
    gemm(c, go);
    sc.off = mark;
    return;
  }
  float* ac = sc.get<float>((size_t)Z * T * Sp);
  float* bd = sc.get<float>((size_t)Z * T * Rp);
  G gac(a.qu, a.qkv + C, ac, T, T, dh, C, 3 * C, Sp);
  gac.f32().batch(Z, B, dh, (long)T * C, dh, (long)T * 3 * C, (long)B * T * Sp, (long)T * Sp);
  gemm(c, gac);
  if (!learned) {
    G gpp(pe, w.wpos, a.pp, R, C, C, C, C, C);
    gemm(c, gpp);
  }
  G gbd(qvv, pp, bd, T, R, dh, C, C, Rp);
  gbd.f32().batch(Z, B, dh, (long)T * C, dh, 0, (long)B * T * Rp, (long)T * Rp);
  gemm(c, gbd);
  RUN(ea_relpos_softmax_fwd(ac, bd, key_len, attn_mask, a.P, sh.p_attn > 0.f ? a.Pd : nullptr, H, B, T, T, Sp, Rp, Sp, 0,
                            seed + kProbs, drop_thr(sh.p_attn), drop_scale(sh.p_attn), c.s));
  G gpv(a.Pd, a.qkv + 2 * C, a.o, T, dh, T, Sp, 3 * C, C);
  gpv.bks().batch(Z, B, (long)B * T * Sp, (long)T * Sp, dh, (long)T * 3 * C, dh, (long)T * C);
  gemm(c, gpv);
  G go(a.o, w.wo, y, M, C, C, C, C, C);
  go.bias(w.bo).drop(sh.p_drop, seed + kAttnOut).resid(x, C);
  gemm(c, go);
  sc.off = mark;
}

// shared tail of the attention backward: pos_proj / bias / qkv weight gradients, dq = t1 + t2, dgrad to the block input, LN
static void attn_bwd_tail(Ctx& c, const AttnSaved& a, const EaLayerShape& sh, const EaAttnParams& w, const EaAttnGrads& gw,
                          const void* x, const void* dy, void* dx, const void* pe, uint16_t* dqkv, uint16_t* t1, uint16_t* t2,
                          uint16_t* dBD, const uint16_t* wqkvt, const Pre& next, float* dpe, bool have_dq = false) {
  const int B = sh.B, T = sh.T, C = sh.C, H = sh.H, M = B * T, dh = C / H, R = 2 * T - 1, Rp = pad_bd(R);
  Arena& sc = *c.scratch;
  const bool learned = sh.pos_mode == 1;
  // dpp[r][h*dh+d] = sum_{b,i} dBD[h][(b,i)][r] qv[(b,i),h,d]: tiny output (R x C), reduction over all B*T frames -> split-K
  // into fp32.  Optimizer-only: runs on the side stream.
  const int tiles = ((R + 127) / 128) * H;
  int sk = (768 + tiles - 1) / tiles;
  if (sk > (B * T) / 256) sk = (B * T) / 256;
  if (sk < 1) sk = 1;
  if (learned) {
    // learned table: the fp32 result IS the gradient of the table slice (written to the caller's buffer, [R][C]) — the caller
    // reads it right after this call returns: in deferred mode (no join before the call returns) it is the one optimizer-only
    // product that stays on the main stream
    G gpp(dBD, a.qu, dpe, R, dh, B * T, Rp, C, C);
    gpp.aks().bks().f32().batch(H, 1, (long)B * T * Rp, 0, dh, 0, dh, 0);
    gpp.p.splitk = sk;
    if (sk > 1) gpp.p.workspace = sc.get<float>((size_t)sk * H * R * dh);
    if (c.df) {
      gemm(c, gpp);
    } else {
      fork(c);
      gemm_on(c, gpp, wstream(c));
    }
  } else {
    // sinusoidal table: the product is only needed for the pos_proj weight gradient, so it is formed TRANSPOSED,
    // dppT[h*dh+d][r]: the 64-wide head dimension becomes the row tile (64x128 tiles fully used instead of half-empty
    // 128x64 ones) and the weight-gradient GEMM below reads it k-contiguous
    float* dppT32 = sc.get<float>((size_t)C * Rp);
    uint16_t* dppT = sc.get<uint16_t>((size_t)C * Rp);
    G gpp(a.qv, dBD, dppT32, dh, R, B * T, C, Rp, Rp);
    gpp.aks().bks().f32().batch(H, 1, dh, 0, (long)B * T * Rp, 0, (long)dh * Rp, 0);
    gpp.p.splitk = sk;
    if (sk > 1) gpp.p.workspace = sc.get<float>((size_t)sk * H * R * dh);
    // dWpos[n][k] += sum_r dppT[n][r] pe[r][k]
    const int wt_tiles = ((C + 63) / 64) * ((C + 127) / 128);
    int sk2 = (512 + wt_tiles - 1) / wt_tiles;
    if (sk2 > R / 256) sk2 = R / 256;
    if (sk2 < 1) sk2 = 1;
    G gw2(dppT, pe, gw.wpos, C, C, R, Rp, C, C);
    gw2.bks().f32().acc();
    gw2.p.splitk = sk2;
    if (sk2 > 1) gw2.p.workspace = sc.get<float>((size_t)sk2 * C * C);
    // Deferred mode: the product rides in the layer's grouped weight-gradient launch as one problem per head —
    // dppT32[h*dh + d][r] += sum_m qv[m][h*dh + d] * dBD[h][m][r] is a "weight gradient" with dy = the head's 64 qv columns,
    // x = the head's dBD slab (both [reduction rows][columns] as they lie in memory: global_load_lds + transposing LDS reads
    // instead of the register-transposed split-K GEMM + slab reduce: 75 -> ~25 us per layer on the side stream)
    static const bool gpp_grouped = getenv("EA_GPP_SPLITK") == nullptr;  // (diagnostic A/B switch)
    const bool in_group = c.df && gpp_grouped && dh % 64 == 0 && c.df->grp.count + H + 3 <= EA_WGRAD_MAX;
    if (in_group) {
      for (int h = 0; h < H; ++h) {
        EaWgradProblem& q = c.df->grp.p[c.df->grp.count++];
        if (c.dry) continue;
        q.dy = a.qv + (size_t)h * dh; q.x = dBD + (size_t)h * B * T * Rp; q.dW = dppT32 + (size_t)h * dh * Rp; q.dbias = nullptr;
        q.M = B * T; q.N = dh; q.K = R; q.ld_dy = C; q.ld_x = Rp; q.ldw = Rp;
      }
      if (!c.dry) {
        const EaGemmParams p2 = gw2.p;
        const long ncast = (long)C * Rp;
        c.df->pre.push_back([=](hipStream_t st) { return hipMemsetAsync(dppT32, 0, (size_t)ncast * sizeof(float), st) == hipSuccess ? 0 : -1; });
        c.df->ops.push_back([=](hipStream_t st) {
          int rc = ea_cast_f32_to_bf16(dppT32, dppT, ncast, st);  // pad columns R..Rp-1 are never read
          if (rc == 0) rc = ea_gemm_bf16(&p2, st);
          return rc;
        });
      }
    } else if (c.df) {
      if (!c.dry) {
        const EaGemmParams p1 = gpp.p, p2 = gw2.p;
        const long ncast = (long)C * Rp;
        c.df->ops.push_back([=](hipStream_t st) {
          int rc = ea_gemm_bf16(&p1, st);
          if (rc == 0) rc = ea_cast_f32_to_bf16(dppT32, dppT, ncast, st);  // pad columns R..Rp-1 are never read
          if (rc == 0) rc = ea_gemm_bf16(&p2, st);
          return rc;
        });
      }
    } else {
      fork(c);  // dBD, qv ready: the whole pos_proj gradient chain is optimizer-only
      gemm_on(c, gpp, wstream(c));
      RUN(ea_cast_f32_to_bf16(dppT32, dppT, (long)C * Rp, wstream(c)));  // pad columns R..Rp-1 are never read
      gemm_on(c, gw2, wstream(c));
    }
    bias_grad(c, t1, gw.pos_u, M, C, C);
    bias_grad(c, t2, gw.pos_v, M, C, C);
  }
  if (!have_dq) RUN(ea_add2_strided_bf16(t1, C, t2, C, dqkv, 3 * C, M, C, c.s));
  fork(c);
  wgrad(c, dqkv, 3 * C, a.xn, C, gw.wqkv, M, 3 * C, C, gw.bqkv);
  uint16_t* dxn = sc.get<uint16_t>((size_t)M * C);
  dgrad(c, dqkv, w.wqkv, wqkvt, dxn, M, C, 3 * C);
  ln_bwd_block(c, x, dxn, w.ln_g, a.mean, a.rstd, dx, gw.ln_g, gw.ln_b, M, C, dy, next);
}

static void attn_bwd(Ctx& c, const AttnSaved& a, const EaLayerShape& sh, const EaAttnParams& w, const EaAttnGrads& gw, const void* x,
                     const void* dy, void* dx, const int* key_len, const void* pe, uint64_t seed, const uint16_t* wqkvt,
                     const uint16_t* wot, const uint16_t* pre, const Pre& next, float* dpe = nullptr) {
  const int B = sh.B, T = sh.T, C = sh.C, H = sh.H, M = B * T, dh = C / H, Z = H * B, Sp = pad8(T), R = 2 * T - 1, Rp = pad_bd(R);
  const float scaling = 1.0f / sqrtf((float)dh);
  Arena& sc = *c.scratch;
  const size_t mark = sc.off;
  const void* g = dy;
  if (pre) {
    g = pre;  // dropout(dy), already written by the previous block's LayerNorm backward
  } else if (sh.p_drop > 0.f) {
    uint16_t* gg = sc.get<uint16_t>((size_t)M * C);
    RUN(ea_scale_dropout_bf16(dy, nullptr, gg, (long)M * C, 1.f, 0.f, seed + kAttnOut, drop_thr(sh.p_drop), drop_scale(sh.p_drop), c.s));
    g = gg;
  }
  fork(c);
  wgrad(c, g, C, a.o, C, gw.wo, M, C, C, gw.bo);
  uint16_t* dO = sc.get<uint16_t>((size_t)M * C);
  dgrad(c, g, w.wo, wot, dO, M, C, C);
  if (attn_fused(sh)) {
    uint16_t* dqkv = sc.get<uint16_t>((size_t)M * 3 * C);
    uint16_t* t1 = sc.get<uint16_t>((size_t)M * C);
    uint16_t* t2 = sc.get<uint16_t>((size_t)M * C);
    uint16_t* dBD = sc.get<uint16_t>((size_t)Z * T * Rp);
    float* Dd = sc.get<float>((size_t)Z * T);
    static const bool fused_dq = getenv("EA_NO_FUSED_DQ") == nullptr;  // (diagnostic A/B switch)
    RUN(ea_flash_attention_bwd(a.qu, sh.pos_mode == 1 ? a.qu : a.qv, C, a.qkv + C, a.qkv + 2 * C, 3 * C,
                               sh.pos_mode == 1 ? (const uint16_t*)pe : a.pp, C, key_len, a.o, dO, C, a.lse, Dd, t1, t2, C, dBD,
                               Rp, dqkv + C, dqkv + 2 * C, 3 * C, H, B, T, T, dh, (sh.scratch_clean && c.overlap) ? 2 : 0, scaling, seed + kProbs,
                               drop_thr(sh.p_attn),
                               drop_scale(sh.p_attn), a.bits, fused_dq ? dqkv : nullptr, 3 * C, c.s));  // dq = t1 + t2 -> q third of dqkv
    attn_bwd_tail(c, a, sh, w, gw, x, dy, dx, pe, dqkv, t1, t2, dBD, wqkvt, next, dpe, fused_dq);
    release(c, mark);
    return;
  }
  float* dPd = sc.get<float>((size_t)Z * T * Sp);
  G gdp(dO, a.qkv + 2 * C, dPd, T, T, dh, C, 3 * C, Sp);
  gdp.f32().batch(Z, B, dh, (long)T * C, dh, (long)T * 3 * C, (long)B * T * Sp, (long)T * Sp);
  gemm(c, gdp);
  uint16_t* dqkv = sc.get<uint16_t>((size_t)M * 3 * C);
  G gdv(a.Pd, dO, dqkv + 2 * C, T, dh, T, Sp, C, 3 * C);
  gdv.aks().bks().batch(Z, B, (long)B * T * Sp, (long)T * Sp, dh, (long)T * C, dh, (long)T * 3 * C);
  gemm(c, gdv);
  uint16_t* dAC = sc.get<uint16_t>((size_t)Z * T * Sp);
  uint16_t* dBD = sc.get<uint16_t>((size_t)Z * T * Rp);
  RUN(ea_relpos_softmax_bwd(a.P, dPd, dAC, dBD, H, B, T, T, Sp, Sp, Rp, seed + kProbs, drop_thr(sh.p_attn), drop_scale(sh.p_attn), c.s));
  G gdk(dAC, a.qu, dqkv + C, T, dh, T, Sp, C, 3 * C);
  gdk.aks().bks().batch(Z, B, (long)B * T * Sp, (long)T * Sp, dh, (long)T * C, dh, (long)T * 3 * C);
  gemm(c, gdk);
  uint16_t* t1 = sc.get<uint16_t>((size_t)M * C);
  G gt1(dAC, a.qkv + C, t1, T, dh, T, Sp, 3 * C, C);
  gt1.bks().alpha(scaling).batch(Z, B, (long)B * T * Sp, (long)T * Sp, dh, (long)T * 3 * C, dh, (long)T * C);
  gemm(c, gt1);
  uint16_t* t2 = sc.get<uint16_t>((size_t)M * C);
  G gt2(dBD, sh.pos_mode == 1 ? (const uint16_t*)pe : a.pp, t2, T, dh, R, Rp, C, C);
  gt2.bks().alpha(scaling).batch(Z, B, (long)B * T * Rp, (long)T * Rp, dh, 0, dh, (long)T * C);
  gemm(c, gt2);
  attn_bwd_tail(c, a, sh, w, gw, x, dy, dx, pe, dqkv, t1, t2, dBD, wqkvt, next, dpe);
  release(c, mark);
}

// BatchNorm accumulators of the convolution module without fill launches: two fp64 statistics buffers (forward) and two fp32
// sum buffers (backward) take turns; the kernel that CONSUMES one buffer clears the other for the next call (ea_bn_act_fwd_train
// / ea_bn_act_bwd_fused), so the kernel that accumulates always finds zeros.  One ring per (device, stream), allocated
// and cleared on first use (so: not inside a hipGraph capture — run one eager call first); a call that fails between the two
// launches marks the ring dirty and the next user re-zeroes it.  Channel counts above the capacity fall back to a fill per call.
struct BnRing {
  double* st[2] = {nullptr, nullptr};
  float* red[2] = {nullptr, nullptr};
  int si = 0, ri = 0, cap = 0;
  bool ok = false, failed = false, dirty = false;
};
// one ring per (device, stream): the zero-on-entry invariant only holds among launches that are ordered on ONE stream
static std::mutex g_bn_mu;
static std::map<std::pair<int, hipStream_t>, BnRing> g_bn_rings;
static BnRing* bn_ring(hipStream_t owner, int C) {
  int cur = 0, dev = 0;
  if (hipGetDevice(&cur) != hipSuccess) return nullptr;
  dev = cur;
  if (owner != nullptr) {
    hipDevice_t d;
    if (hipStreamGetDevice(owner, &d) == hipSuccess) dev = (int)d;
  }
  std::lock_guard<std::mutex> lk(g_bn_mu);
  BnRing& R = g_bn_rings[std::make_pair(dev, owner)];
  if (R.failed) return nullptr;
  if (!R.ok) {
    const int cap = 2048;
    if (dev != cur && hipSetDevice(dev) != hipSuccess) { R.failed = true; return nullptr; }
    const size_t bytes = 2 * (size_t)(2 * cap) * (sizeof(double) + sizeof(float));
    char* base = nullptr;
    const bool ok = hipMalloc(reinterpret_cast<void**>(&base), bytes) == hipSuccess && hipMemsetAsync(base, 0, bytes, owner) == hipSuccess;
    if (dev != cur) (void)hipSetDevice(cur);
    if (!ok) { R.failed = true; return nullptr; }
    R.st[0] = reinterpret_cast<double*>(base);
    R.st[1] = R.st[0] + 2 * cap;
    R.red[0] = reinterpret_cast<float*>(R.st[1] + 2 * cap);
    R.red[1] = R.red[0] + 2 * cap;
    R.cap = cap;
    R.ok = true;
  }
  if (C > R.cap) return nullptr;
  if (R.dirty) {  // a call failed between the accumulating and the consuming launch: start from zeros again
    const size_t bytes = 2 * (size_t)(2 * R.cap) * (sizeof(double) + sizeof(float));
    if (hipMemsetAsync(R.st[0], 0, bytes, owner) != hipSuccess) return nullptr;
    R.dirty = false;
  }
  return &R;
}

struct ConvSaved {
  float *mean, *rstd, *mr;
  uint16_t *xn, *Y, *U, *Z, *Hh;
};
static ConvSaved conv_saved(Arena& sv, const EaLayerShape& sh) {
  const int M = sh.B * sh.T, C = sh.C;
  ConvSaved s;
  s.mean = sv.get<float>(M);
  s.rstd = sv.get<float>(M);
  s.mr = sv.get<float>(2 * C);
  s.xn = sv.get<uint16_t>((size_t)M * C);
  s.Y = sv.get<uint16_t>((size_t)M * 2 * C);
  s.U = sv.get<uint16_t>((size_t)M * C);
  s.Z = sv.get<uint16_t>((size_t)M * C);
  s.Hh = sv.get<uint16_t>((size_t)M * C);
  return s;
}

static void conv_fwd(Ctx& c, const ConvSaved& s, const EaLayerShape& sh, const EaConvParams& w, const void* x, void* y, uint64_t seed) {
  const int B = sh.B, T = sh.T, C = sh.C, M = B * T;
  Arena& sc = *c.scratch;
  const size_t mark = sc.off;
  RUN(ea_layernorm_fwd(x, w.ln_g, w.ln_b, s.xn, s.mean, s.rstd, M, C, 1e-5f, nullptr, 0, 0, 1.f, c.s));
  G g1(s.xn, w.pw1, s.Y, M, 2 * C, C, C, C, 2 * C);
  gemm(c, g1);
  // BatchNorm batch statistics: fp64 (sum, sum of squares) accumulators; mean / rstd, the running statistics and the normalised
  // activation come out of ONE launch behind the depthwise kernel (no fill, no finalize launch)
  static const bool bn_fused = getenv("EA_BN_UNFUSED") == nullptr;  // (diagnostic A/B switch)
  const bool ring = sh.training && bn_fused && C <= 2048;  // (same decision in the sizing pass: the arena walk must not differ)
  BnRing* R = nullptr;
  if (ring && !c.dry && c.rc == 0 && !(R = bn_ring(c.s, C))) c.rc = -1;
  double* stats = nullptr;
  if (sh.training && !ring) {
    stats = sc.get<double>(2 * C);
    if (!c.dry && c.rc == 0) c.rc = hipMemsetAsync(stats, 0, 2 * C * sizeof(double), c.s) == hipSuccess ? 0 : -1;
  }
  if (R && c.rc == 0) stats = R->st[R->si];
  RUN(ea_glu_dwconv_fwd(s.Y, w.dw, s.U, s.Z, stats, B, T, C, sh.KW, c.s));
  if (ring) {
    if (R) {
      if (c.rc == 0) RUN(ea_bn_act_fwd_train(s.Z, stats, s.mr, w.bn_rm, w.bn_rv, w.bn_g, w.bn_b, s.Hh, M, C, EA_ACT_SILU, (float)M, 1e-5f, 0.1f,
                                             R->st[R->si ^ 1], 2 * R->cap, c.s));
      if (c.rc == 0) R->si ^= 1;  // (flipped only when both launches are queued)
      else R->dirty = true;
    }
  } else {
    if (sh.training) RUN(ea_bn_finalize(stats, s.mr, w.bn_rm, w.bn_rv, C, (float)M, 1e-5f, 0.1f, c.s));
    else RUN(ea_bn_from_running(w.bn_rm, w.bn_rv, s.mr, C, 1e-5f, c.s));
    RUN(ea_bn_act_fwd(s.Z, s.mr, w.bn_g, w.bn_b, s.Hh, M, C, EA_ACT_SILU, c.s));
  }
  G g2(s.Hh, w.pw2, y, M, C, C, C, C, C);
  g2.drop(sh.p_drop, seed + kConvOut).resid(x, C);
  gemm(c, g2);
  sc.off = mark;
}

static void conv_bwd(Ctx& c, const ConvSaved& s, const EaLayerShape& sh, const EaConvParams& w, const EaConvGrads& gw, const void* x,
                     const void* dy, void* dx, uint64_t seed, const uint16_t* pw1t, const uint16_t* pw2t, const uint16_t* pre,
                     const Pre& next) {
  const int B = sh.B, T = sh.T, C = sh.C, M = B * T;
  Arena& sc = *c.scratch;
  const size_t mark = sc.off;
  const void* g = dy;
  if (pre) {
    g = pre;
  } else if (sh.p_drop > 0.f) {
    uint16_t* gg = sc.get<uint16_t>((size_t)M * C);
    RUN(ea_scale_dropout_bf16(dy, nullptr, gg, (long)M * C, 1.f, 0.f, seed + kConvOut, drop_thr(sh.p_drop), drop_scale(sh.p_drop), c.s));
    g = gg;
  }
  fork(c);
  wgrad(c, g, C, s.Hh, C, gw.pw2, M, C, C);
  uint16_t* dH = sc.get<uint16_t>((size_t)M * C);
  dgrad(c, g, w.pw2, pw2t, dH, M, C, C);
  // BatchNorm backward: the two per-channel sums go to a buffer the previous call's apply kernel cleared; the apply kernel adds
  // the BatchNorm parameter gradients itself (no fill launch, no bn_param_grad launch)
  static const bool bn_fused = getenv("EA_BN_UNFUSED") == nullptr;  // (diagnostic A/B switch)
  const bool ring = bn_fused && C <= 2048;
  BnRing* R = nullptr;
  if (ring && !c.dry && c.rc == 0 && !(R = bn_ring(c.s, C))) c.rc = -1;
  float* red = nullptr;
  if (!ring) {
    red = sc.get<float>(2 * C);
    if (!c.dry && c.rc == 0) c.rc = hipMemsetAsync(red, 0, 2 * C * sizeof(float), c.s) == hipSuccess ? 0 : -1;
  }
  uint16_t* dZ = sc.get<uint16_t>((size_t)M * C);
  uint16_t* dY = sc.get<uint16_t>((size_t)M * 2 * C);
  char* wws = sc.get<char>((size_t)ea_dwconv_wgrad_workspace_bytes(B, T, C, sh.KW));
  if (ring) {
    if (R && c.rc == 0) {
      red = R->red[R->ri];
      // BatchNorm reduce, then ONE kernel: BatchNorm apply folded into the GLU / depthwise data gradient's tile staging
      RUN(ea_bn_glu_dwconv_bwd_fused(s.Z, dH, s.mr, w.bn_g, w.bn_b, red, dZ, gw.bn_g, gw.bn_b, EA_ACT_SILU, sh.training, R->red[R->ri ^ 1],
                                     2 * R->cap, s.Y, w.dw, dY, B, T, C, sh.KW, c.s));
      if (c.rc == 0) R->ri ^= 1;
      else R->dirty = true;
    }
  } else {
    RUN(ea_bn_act_bwd(s.Z, dH, s.mr, w.bn_g, w.bn_b, red, dZ, nullptr, nullptr, M, C, EA_ACT_SILU, sh.training, c.s));
    RUN(ea_glu_dwconv_bwd(dZ, s.Y, s.U, w.dw, dY, nullptr, wws, B, T, C, sh.KW, c.s));
  }
  if (c.df) {
    if (!c.dry) {
      const EaConvGrads gwv = gw;
      const uint16_t* U = s.U;
      const int KW = sh.KW;
      c.df->ops.push_back([=](hipStream_t st) {
        int rc = ring ? 0 : ea_bn_param_grad(red, gwv.bn_g, gwv.bn_b, C, st);
        if (rc == 0) rc = ea_dwconv_bwd_weight(dZ, U, gwv.dw, wws, B, T, C, KW, st);
        return rc;
      });
    }
  } else {
    fork(c);  // BatchNorm / depthwise-filter / pointwise-1 parameter gradients: optimizer-only
    if (!ring) RUN(ea_bn_param_grad(red, gw.bn_g, gw.bn_b, C, wstream(c)));
    RUN(ea_dwconv_bwd_weight(dZ, s.U, gw.dw, wws, B, T, C, sh.KW, wstream(c)));
  }
  wgrad(c, dY, 2 * C, s.xn, C, gw.pw1, M, 2 * C, C);
  uint16_t* dxn = sc.get<uint16_t>((size_t)M * C);
  dgrad(c, dY, w.pw1, pw1t, dxn, M, C, 2 * C);
  ln_bwd_block(c, x, dxn, w.ln_g, s.mean, s.rstd, dx, gw.ln_g, gw.ln_b, M, C, dy, next);
  release(c, mark);
}

// Saved-activation arena order: [x1][ffn1][x2][attn][x3][conv][x4][ffn2][final LN stats]
struct LayerSaved {
  uint16_t *x1, *x2, *x3, *x4;
  FfnSaved f1, f2;
  AttnSaved at;
  ConvSaved cv;
  float *fmean, *frstd;
};
static LayerSaved layer_saved(Arena& sv, const EaLayerShape& sh) {
  const size_t MC = (size_t)sh.B * sh.T * sh.C;
  LayerSaved L;
  L.x1 = sv.get<uint16_t>(MC);
  L.f1 = ffn_saved(sv, sh);
  L.x2 = sv.get<uint16_t>(MC);
  L.at = attn_saved(sv, sh);
  L.x3 = sv.get<uint16_t>(MC);
  L.cv = conv_saved(sv, sh);
  L.x4 = sv.get<uint16_t>(MC);
  L.f2 = ffn_saved(sv, sh);
  L.fmean = sv.get<float>((size_t)sh.B * sh.T);
  L.frstd = sv.get<float>((size_t)sh.B * sh.T);
  return L;
}

static int layer_fwd(Ctx& c, const EaConformerLayer* L, const EaLayerShape& sh, const void* x_in, void* x_out, const int* key_len,
                     const float* attn_mask, const void* pe, Arena& sv) {
  const int M = sh.B * sh.T, C = sh.C;
  LayerSaved S = layer_saved(sv, sh);
  const uint64_t seed = sh.seed;
  const EaLayerChain* ch = c.chain;
  // the attention block's keep bits need nothing but the seed: launched now, they run under the first feed-forward block instead of
  // beside the QKV projection that the attention kernel also waits for (EA_KEEP_BITS_EARLY=0: in the attention block, as before)
  static const bool bits_early = [] { const char* e = getenv("EA_KEEP_BITS_EARLY"); return !(e && e[0] == '0'); }();
  static const bool pp_early = [] { const char* e = getenv("EA_POS_PROJ_EARLY"); return !(e && e[0] == '0'); }();
  if (bits_early) c.bits_early = keep_bits_on_side(c, S.at, sh, seed + kAttn, pp_early ? &L->attn : nullptr, pe);
  ffn_fwd(c, S.f1, sh, L->ffn1, x_in, S.x1, seed + kFfn1, 0.5f, EA_ACT_SILU, ch && ch->ln1_done);
  attn_fwd(c, S.at, sh, L->attn, S.x1, S.x2, key_len, attn_mask, pe, seed + kAttn);
  conv_fwd(c, S.cv, sh, L->conv, S.x2, S.x3, seed + kConv);
  ffn_fwd(c, S.f2, sh, L->ffn2, S.x3, S.x4, seed + kFfn2, 0.5f, EA_ACT_SILU);
  if (ch && ch->next) {  // ... and the next layer's opening LayerNorm, written into ITS saved arena
    Arena nsv{(char*)ch->next_saved, 0, 0};
    const LayerSaved N = layer_saved(nsv, sh);
    RUN(ea_layernorm_fwd2(S.x4, L->final_ln_g, L->final_ln_b, x_out, S.fmean, S.frstd, ch->next->ffn1.ln_g, ch->next->ffn1.ln_b, N.f1.xn,
                          N.f1.mean, N.f1.rstd, M, C, 1e-5f, c.s));
  } else {
    RUN(ea_layernorm_fwd(S.x4, L->final_ln_g, L->final_ln_b, x_out, S.fmean, S.frstd, M, C, 1e-5f, nullptr, 0, 0, 1.f, c.s));
  }
  if (!sh.wt_fresh) wt_refresh(c, L, sh);  // the backward of this step reads the k-contiguous copies (wt_fresh: the caller refreshed them already, off this stream)
  return c.rc;
}

static bool g_fuse_predrop = true;  // A/B switch (ea_set_fused_predrop)
static int layer_bwd(Ctx& c, const EaConformerLayer* L, const EaLayerShape& sh, const void* x_in, const void* dy, void* dx,
                     const int* key_len, const void* pe, Arena& sv) {
  const int M = sh.B * sh.T, C = sh.C;
  LayerSaved S = layer_saved(sv, sh);
  Arena& sc = *c.scratch;
  const uint64_t seed = sh.seed;
  // one gradient buffer per stage (no ping-pong): with dropout off a block's side-stream weight-gradient GEMM reads its
  // incoming gradient in place, so that buffer must stay untouched until the join at the end of the layer
  uint16_t* dA = sc.get<uint16_t>((size_t)M * C);
  uint16_t* dB = sc.get<uint16_t>((size_t)M * C);
  uint16_t* dC = sc.get<uint16_t>((size_t)M * C);
  uint16_t* dD = sc.get<uint16_t>((size_t)M * C);
  // each block's incoming gradient after its residual dropout (and the 0.5 FFN scale) is written by the LayerNorm backward
  // that produces the gradient itself (second output) — no separate pass
  const bool dp = sh.p_drop > 0.f, fz = g_fuse_predrop;
  uint16_t* pf2 = fz ? sc.get<uint16_t>((size_t)M * C) : nullptr;
  uint16_t* pcv = fz && dp ? sc.get<uint16_t>((size_t)M * C) : nullptr;
  uint16_t* pat = fz && dp ? sc.get<uint16_t>((size_t)M * C) : nullptr;
  uint16_t* pf1 = fz ? sc.get<uint16_t>((size_t)M * C) : nullptr;
  const Pre to_ffn2{pf2, 0.5f, seed + kFfn2 + kOut, sh.p_drop}, to_conv{pcv, 1.f, seed + kConv + kConvOut, sh.p_drop};
  const Pre to_attn{pat, 1.f, seed + kAttn + kAttnOut, sh.p_drop}, to_ffn1{pf1, 0.5f, seed + kFfn1 + kOut, sh.p_drop}, none{nullptr, 1.f, 0, 0.f};
  const EaLayerChain* ch = c.chain;
  void* lnws2 = ln_ws(c, M, C);  // (second partial slab of a chained call's closing kernel; taken in every mode)
  if (ch && ch->final_ln_done) {  // the next layer's chained call ran this layer's final-LayerNorm backward
    dA = (uint16_t*)dy;
    pf2 = (uint16_t*)ch->pre_in;
  } else {
    ln_bwd_block(c, S.x4, dy, L->final_ln_g, S.fmean, S.frstd, dA, L->grads.final_ln_g, L->grads.final_ln_b, M, C, nullptr, to_ffn2);
  }
  ChainBwd cb;
  const bool chained = ch && ch->prev;
  if (chained) {
    Arena psv{(char*)ch->prev_saved, 0, 0};
    const LayerSaved P = layer_saved(psv, sh);
    cb = ChainBwd{P.x4, ch->prev->final_ln_g, P.fmean, P.frstd, ch->prev->grads.final_ln_g, ch->prev->grads.final_ln_b, lnws2,
                  Pre{(uint16_t*)ch->prev_pre, 0.5f, ch->prev_seed + kFfn2 + kOut, sh.p_drop}};
  }
  const WT wt = wt_view(L, sh);
  ffn_bwd(c, S.f2, sh, L->ffn2, L->grads.ffn2, S.x3, dA, dB, seed + kFfn2, 0.5f, EA_ACT_SILU, wt.f2w1, wt.f2w2, pf2, to_conv);
  conv_bwd(c, S.cv, sh, L->conv, L->grads.conv, S.x2, dB, dC, seed + kConv, wt.pw1, wt.pw2, pcv, to_attn);
  attn_bwd(c, S.at, sh, L->attn, L->grads.attn, S.x1, dC, dD, key_len, pe, seed + kAttn, wt.wqkv, wt.wo, pat, to_ffn1);
  ffn_bwd(c, S.f1, sh, L->ffn1, L->grads.ffn1, x_in, dD, dx, seed + kFfn1, 0.5f, EA_ACT_SILU, wt.f1w1, wt.f1w2, pf1, none, chained ? &cb : nullptr);
  if (c.df) run_deferred(c, sh.defer - 1);
  else if (c.overlap) stream_wait(c, c.s, c.side);  // join: gradients complete (and scratch reusable) once `s` passes this point
  return c.rc;
}

// ---------------------------------------------------------------------------------------------------------
// Transformer encoder layer (pre-LN) — fairseq/modules/transformer_layer.py:135-214 as used by
// espresso/models/transformer/speech_transformer_encoder.py (layer_type "transformer"):
//   x = x + dropout(self_attn(LN(x)));  x = x + dropout(fc2(dropout_act(act(fc1(LN(x))))))
// = the attention block and one FFN block of the Conformer runtime (out_scale 1, activation from the shape).  Uses the
// `attn` and `ffn1` members of EaConformerLayer.  Saved order: [x1][attn][ffn].
struct TLayerSaved {
  uint16_t* x1;
  AttnSaved at;
  FfnSaved f;
};
static TLayerSaved tlayer_saved(Arena& sv, const EaLayerShape& sh) {
  TLayerSaved L;
  L.x1 = sv.get<uint16_t>((size_t)sh.B * sh.T * sh.C);
  L.at = attn_saved(sv, sh);
  L.f = ffn_saved(sv, sh);
  return L;
}
// k-contiguous weight copies of the transformer layer: [fc1^T (C x F)][wqkv^T (C x 3C)][wo^T (C x C)][fc2^T (F x C)]
static WT twt_view(const EaConformerLayer* L, const EaLayerShape& sh) {
  WT w{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (!L->wt || !sh.training) return w;
  const size_t C = sh.C, F = sh.F;
  const uint16_t* p = (const uint16_t*)L->wt;
  w.f1w1 = p; p += C * F;
  w.wqkv = p; p += 3 * C * C;
  w.wo = p; p += C * C;
  w.f1w2 = p;
  return w;
}
static int tlayer_fwd(Ctx& c, const EaConformerLayer* L, const EaLayerShape& sh, const void* x_in, void* x_out, const int* key_len,
                      const float* attn_mask, const void* pe, Arena& sv) {
  TLayerSaved S = tlayer_saved(sv, sh);
  const uint64_t seed = sh.seed;
  attn_fwd(c, S.at, sh, L->attn, x_in, S.x1, key_len, attn_mask, pe, seed + kAttn);
  ffn_fwd(c, S.f, sh, L->ffn1, S.x1, x_out, seed + kFfn1, 1.f, sh.act);
  const WT w = twt_view(L, sh);
  if (w.f1w1) {
    const int C = sh.C, F = sh.F;
    const void* src[4] = {L->ffn1.w1, L->attn.wqkv, L->attn.wo, L->ffn1.w2};
    void* dst[4] = {(void*)w.f1w1, (void*)w.wqkv, (void*)w.wo, (void*)w.f1w2};
    const int rows[4] = {F, 3 * C, C, C}, cols[4] = {C, C, C, F};
    RUN(ea_transpose_bf16_batch(src, dst, rows, cols, 4, c.s));
  }
  return c.rc;
}
static int tlayer_bwd(Ctx& c, const EaConformerLayer* L, const EaLayerShape& sh, const void* x_in, const void* dy, void* dx,
                      const int* key_len, const void* pe, float* dpe, Arena& sv) {
  const int M = sh.B * sh.T, C = sh.C;
  TLayerSaved S = tlayer_saved(sv, sh);
  Arena& sc = *c.scratch;
  const uint64_t seed = sh.seed;
  uint16_t* dA = sc.get<uint16_t>((size_t)M * C);
  uint16_t* pat = sh.p_drop > 0.f ? sc.get<uint16_t>((size_t)M * C) : nullptr;
  const Pre to_attn{pat, 1.f, seed + kAttn + kAttnOut, sh.p_drop}, none{nullptr, 1.f, 0, 0.f};
  const WT wt = twt_view(L, sh);
  ffn_bwd(c, S.f, sh, L->ffn1, L->grads.ffn1, S.x1, dy, dA, seed + kFfn1, 1.f, sh.act, wt.f1w1, wt.f1w2, nullptr, to_attn);
  attn_bwd(c, S.at, sh, L->attn, L->grads.attn, x_in, dA, dx, key_len, pe, seed + kAttn, wt.wqkv, wt.wo, pat, none, dpe);
  if (c.df) run_deferred(c, sh.defer - 1);
  else if (c.overlap) stream_wait(c, c.s, c.side);
  return c.rc;
}

// ---------------------------------------------------------------------------------------------------------
// Transformer decoder layer (teacher-forced training): causal self-attention, encoder-decoder attention, FFN; pre-LN.
// Both attention blocks run on the fused flash kernels without relative positions (head dim 64).
//   saved: [x1][self: mean rstd lse xn qkv qs o][x2][cross: mean rstd lse xn q qs kv o][ffn]
struct DSelfSaved { float *mean, *rstd, *lse; uint16_t *xn, *qkv, *qs, *o; };
struct DCrossSaved { float *mean, *rstd, *lse; uint16_t *xn, *q, *qs, *kv, *o; };
struct DLayerSaved { uint16_t *x1, *x2; DSelfSaved sa; DCrossSaved ca; FfnSaved f; };
static DLayerSaved dlayer_saved(Arena& sv, const EaLayerShape& sh) {
  const size_t M = (size_t)sh.B * sh.T, Ms = (size_t)sh.B * sh.S, C = sh.C, Z = (size_t)sh.H * sh.B;
  DLayerSaved L;
  L.x1 = sv.get<uint16_t>(M * C);
  L.sa.mean = sv.get<float>(M); L.sa.rstd = sv.get<float>(M); L.sa.lse = sv.get<float>(Z * sh.T);
  L.sa.xn = sv.get<uint16_t>(M * C); L.sa.qkv = sv.get<uint16_t>(M * 3 * C); L.sa.qs = sv.get<uint16_t>(M * C);
  L.sa.o = sv.get<uint16_t>(M * C);
  L.x2 = sv.get<uint16_t>(M * C);
  L.ca.mean = sv.get<float>(M); L.ca.rstd = sv.get<float>(M); L.ca.lse = sv.get<float>(Z * sh.T);
  L.ca.xn = sv.get<uint16_t>(M * C); L.ca.q = sv.get<uint16_t>(M * C); L.ca.qs = sv.get<uint16_t>(M * C);
  L.ca.kv = sv.get<uint16_t>(Ms * 2 * C); L.ca.o = sv.get<uint16_t>(M * C);
  L.f = ffn_saved(sv, sh);
  return L;
}
// k-contiguous weight copies: [fc1^T C x F][fc2^T F x C][wqkv^T C x 3C][wo^T C x C][xq^T C x C][xkv^T C x 2C][xo^T C x C]
struct DWT { const uint16_t *fc1, *fc2, *wqkv, *wo, *xq, *xkv, *xo; };
static DWT dwt_view(const EaDecoderLayer* L, const EaLayerShape& sh) {
  DWT w{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (!L->wt || !sh.training) return w;
  const size_t C = sh.C, F = sh.F;
  const uint16_t* p = (const uint16_t*)L->wt;
  w.fc1 = p; p += C * F;
  w.fc2 = p; p += C * F;
  w.wqkv = p; p += 3 * C * C;
  w.wo = p; p += C * C;
  w.xq = p; p += C * C;
  w.xkv = p; p += 2 * C * C;
  w.xo = p;
  return w;
}
static inline bool dshape_ok(const EaLayerShape& sh) {
  return sh.B > 0 && sh.T > 0 && sh.S > 0 && sh.C % 8 == 0 && sh.H > 0 && sh.C / sh.H == 64 && sh.C % sh.H == 0 && sh.F % 8 == 0 &&
         !sh.has_attn_mask;
}
static int dlayer_fwd(Ctx& c, const EaDecoderLayer* L, const EaLayerShape& sh, const void* x_in, const void* enc, void* x_out,
                      const int* enc_len, Arena& sv) {
  const int B = sh.B, T = sh.T, S = sh.S, C = sh.C, H = sh.H, M = B * T, Ms = B * S, dh = C / H;
  const float scaling = 1.0f / sqrtf((float)dh);
  DLayerSaved D = dlayer_saved(sv, sh);
  const uint64_t seed = sh.seed;
  {  // causal self-attention block: x1 = x + dropout(out_proj(attn(LN(x))))
    const EaAttnParams& w = L->self_attn;
    RUN(ea_layernorm_fwd(x_in, w.ln_g, w.ln_b, D.sa.xn, D.sa.mean, D.sa.rstd, M, C, 1e-5f, nullptr, 0, 0, 1.f, c.s));
    G gq(D.sa.xn, w.wqkv, D.sa.qkv, M, 3 * C, C, C, C, 3 * C);
    gq.bias(w.bqkv);
    const bool split = gq.qsplit(D.sa.qs, nullptr, nullptr, nullptr, C, C, scaling);  // scaled queries from the projection's epilogue
    gemm(c, gq);
    if (!split) RUN(ea_relpos_q_prep(D.sa.qkv, 3 * C, nullptr, nullptr, D.sa.qs, nullptr, M, C, scaling, c.s));
    RUN(ea_flash_attention_fwd(D.sa.qs, nullptr, C, D.sa.qkv + C, D.sa.qkv + 2 * C, 3 * C, nullptr, 0, nullptr, D.sa.o, C, D.sa.lse, H,
                               B, T, T, dh, 1, seed + kAttn + kProbs, drop_thr(sh.p_attn), drop_scale(sh.p_attn), nullptr, c.s));
    G go(D.sa.o, w.wo, D.x1, M, C, C, C, C, C);
    go.bias(w.bo).drop(sh.p_drop, seed + kAttn + kAttnOut).resid(x_in, C);
    gemm(c, go);
  }
  {  // encoder-decoder attention block: x2 = x1 + dropout(out_proj(attn(q = LN(x1), k = v = enc)))
    const EaXAttnParams& w = L->cross;
    RUN(ea_layernorm_fwd(D.x1, w.ln_g, w.ln_b, D.ca.xn, D.ca.mean, D.ca.rstd, M, C, 1e-5f, nullptr, 0, 0, 1.f, c.s));
    G gq(D.ca.xn, w.wq, D.ca.q, M, C, C, C, C, C);
    gq.bias(w.bq);
    const bool split = gq.qsplit(D.ca.qs, nullptr, nullptr, nullptr, C, C, scaling);
    gemm(c, gq);
    if (!split) RUN(ea_relpos_q_prep(D.ca.q, C, nullptr, nullptr, D.ca.qs, nullptr, M, C, scaling, c.s));
    G gkv(enc, w.wkv, D.ca.kv, Ms, 2 * C, C, C, C, 2 * C);
    gkv.bias(w.bkv);
    gemm(c, gkv);
    RUN(ea_flash_attention_fwd(D.ca.qs, nullptr, C, D.ca.kv, D.ca.kv + C, 2 * C, nullptr, 0, enc_len, D.ca.o, C, D.ca.lse, H, B, T, S,
                               dh, 0, seed + kCross + kProbs, drop_thr(sh.p_attn), drop_scale(sh.p_attn), nullptr, c.s));
    G go(D.ca.o, w.wo, D.x2, M, C, C, C, C, C);
    go.bias(w.bo).drop(sh.p_drop, seed + kCross + kAttnOut).resid(D.x1, C);
    gemm(c, go);
  }
  ffn_fwd(c, D.f, sh, L->ffn, D.x2, x_out, seed + kFfn1, 1.f, sh.act);
  const DWT w = dwt_view(L, sh);
  if (w.fc1) {
    const int F = sh.F;
    const void* src[7] = {L->ffn.w1, L->ffn.w2, L->self_attn.wqkv, L->self_attn.wo, L->cross.wq, L->cross.wkv, L->cross.wo};
    void* dst[7] = {(void*)w.fc1, (void*)w.fc2, (void*)w.wqkv, (void*)w.wo, (void*)w.xq, (void*)w.xkv, (void*)w.xo};
    const int rows[7] = {F, C, 3 * C, C, C, 2 * C, C}, cols[7] = {C, F, C, C, C, C, C};
    RUN(ea_transpose_bf16_batch(src, dst, rows, cols, 7, c.s));
  }
  return c.rc;
}
static int dlayer_bwd(Ctx& c, const EaDecoderLayer* L, const EaLayerShape& sh, const void* x_in, const void* enc, const void* dy,
                      void* dx, void* denc, const int* enc_len, Arena& sv) {
  const int B = sh.B, T = sh.T, S = sh.S, C = sh.C, H = sh.H, M = B * T, Ms = B * S, dh = C / H, Z = H * B;
  const float scaling = 1.0f / sqrtf((float)dh);
  DLayerSaved D = dlayer_saved(sv, sh);
  Arena& sc = *c.scratch;
  const uint64_t seed = sh.seed;
  const DWT wt = dwt_view(L, sh);
  const bool dp = sh.p_drop > 0.f;
  // The layer's optimizer-only products (seven weight + bias gradients, three LayerNorm parameter reduces) are collected while the
  // data-gradient chain is enqueued and launched behind ONE fork as one grouped weight-gradient launch + one grouped reduce —
  // instead of 7 split-K GEMMs, 7 slab reduces, 7 column sums and 3 reduces with a fork each; the call still joins them before
  // it returns (the scratch arena is the caller's to reuse).
  static Deferred local;  // one process drives one GPU from one thread (as in bwd_deferred)
  const bool grouped = c.overlap && !c.dry && !c.df && g_defer_default;
  if (grouped) {
    local.clear();
    c.df = &local;
  }
  // what has been collected so far goes to the side stream (behind everything the main stream has been given up to here)
  auto flush_group = [&]() {
    if (!grouped) return;
    stream_wait(c, c.side, c.s);
    RUN(ea_wgrad_group(&local.grp, c.side));
    RUN(ea_layernorm_param_reduce_group(&local.ln, c.side));
    for (auto& op : local.ops) RUN(op(c.side));
    local.clear();
  };
  uint16_t* dA = sc.get<uint16_t>((size_t)M * C);  // gradient at x2
  uint16_t* dB = sc.get<uint16_t>((size_t)M * C);  // gradient at x1
  uint16_t* pca = dp ? sc.get<uint16_t>((size_t)M * C) : nullptr;
  uint16_t* psa = dp ? sc.get<uint16_t>((size_t)M * C) : nullptr;
  const Pre to_cross{pca, 1.f, seed + kCross + kAttnOut, sh.p_drop}, to_self{psa, 1.f, seed + kAttn + kAttnOut, sh.p_drop}, none{nullptr, 1.f, 0, 0.f};
  ffn_bwd(c, D.f, sh, L->ffn, L->g_ffn, D.x2, dy, dA, seed + kFfn1, 1.f, sh.act, wt.fc1, wt.fc2, nullptr, to_cross);
  {  // encoder-decoder attention block
    const EaXAttnParams& w = L->cross;
    const EaXAttnGrads& gw = L->g_cross;
    const void* g = dp ? (const void*)pca : (const void*)dA;
    fork(c);
    wgrad(c, g, C, D.ca.o, C, gw.wo, M, C, C, gw.bo);
    uint16_t* dO = sc.get<uint16_t>((size_t)M * C);
    dgrad(c, g, w.wo, wt.xo, dO, M, C, C);
    uint16_t* dq = sc.get<uint16_t>((size_t)M * C);
    uint16_t* dkv = sc.get<uint16_t>((size_t)Ms * 2 * C);
    float* Dd = sc.get<float>((size_t)Z * T);
    RUN(ea_flash_attention_bwd(D.ca.qs, nullptr, C, D.ca.kv, D.ca.kv + C, 2 * C, nullptr, 0, enc_len, D.ca.o, dO, C, D.ca.lse, Dd, dq,
                               nullptr, C, nullptr, 0, dkv, dkv + C, 2 * C, H, B, T, S, dh, 0, scaling, seed + kCross + kProbs, drop_thr(sh.p_attn),
                               drop_scale(sh.p_attn), nullptr, nullptr, 0, c.s));
    fork(c);
    wgrad(c, dq, C, D.ca.xn, C, gw.wq, M, C, C, gw.bq);
    wgrad(c, dkv, 2 * C, enc, C, gw.wkv, Ms, 2 * C, C, gw.bkv);
    dgrad(c, dkv, w.wkv, wt.xkv, denc, Ms, C, 2 * C);
    uint16_t* dxn = sc.get<uint16_t>((size_t)M * C);
    dgrad(c, dq, w.wq, wt.xq, dxn, M, C, C);
    ln_bwd_block(c, D.x1, dxn, w.ln_g, D.ca.mean, D.ca.rstd, dB, gw.ln_g, gw.ln_b, M, C, dA, to_self);
  }
  flush_group();  // FFN + encoder-decoder attention products (five of the seven weight gradients) run next to the self-attention chain
  {  // causal self-attention block
    const EaAttnParams& w = L->self_attn;
    const EaAttnGrads& gw = L->g_self;
    const void* g = dp ? (const void*)psa : (const void*)dB;
    fork(c);
    wgrad(c, g, C, D.sa.o, C, gw.wo, M, C, C, gw.bo);
    uint16_t* dO = sc.get<uint16_t>((size_t)M * C);
    dgrad(c, g, w.wo, wt.wo, dO, M, C, C);
    uint16_t* dqkv = sc.get<uint16_t>((size_t)M * 3 * C);
    float* Dd = sc.get<float>((size_t)Z * T);
    // t1 (gradient of the scaled queries, already multiplied by the scale) goes straight into the q third of dqkv
    RUN(ea_flash_attention_bwd(D.sa.qs, nullptr, C, D.sa.qkv + C, D.sa.qkv + 2 * C, 3 * C, nullptr, 0, nullptr, D.sa.o, dO, C, D.sa.lse,
                               Dd, dqkv, nullptr, 3 * C, nullptr, 0, dqkv + C, dqkv + 2 * C, 3 * C, H, B, T, T, dh, 1, scaling,
                               seed + kAttn + kProbs, drop_thr(sh.p_attn), drop_scale(sh.p_attn), nullptr, nullptr, 0, c.s));
    fork(c);
    wgrad(c, dqkv, 3 * C, D.sa.xn, C, gw.wqkv, M, 3 * C, C, gw.bqkv);
    uint16_t* dxn = sc.get<uint16_t>((size_t)M * C);
    dgrad(c, dqkv, w.wqkv, wt.wqkv, dxn, M, C, 3 * C);
    ln_bwd_block(c, x_in, dxn, w.ln_g, D.sa.mean, D.sa.rstd, dx, gw.ln_g, gw.ln_b, M, C, dB, none);
  }
  flush_group();
  if (grouped) c.df = nullptr;
  if (c.overlap) stream_wait(c, c.s, c.side);
  return c.rc;
}
static bool darenas_fit(const EaLayerShape& sh, bool backward, bool overlap, long saved_bytes, long scratch_bytes) {
  EaDecoderLayer L;
  memset(&L, 0, sizeof(L));
  Arena sv{nullptr, 0, 0}, sc{nullptr, 0, 0};
  Ctx c{nullptr, true, 0, &sc, nullptr, overlap};
  if (backward) dlayer_bwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sv);
  else dlayer_fwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, sv);
  return (long)sv.peak <= saved_bytes && (long)sc.peak <= scratch_bytes;
}

static bool g_overlap_default = true;
int ea_set_backward_overlap(int on) {
  const int old = g_overlap_default;
  g_overlap_default = on != 0;
  return old;
}
int ea_set_backward_deferred(int on) {
  const int old = g_defer_default;
  g_defer_default = on != 0;
  return old;
}
int ea_backward_flush(hipStream_t stream) { return join_all(stream); }
int ea_set_backward_deferred_inline(int on) {
  const int old = g_defer_inline;
  g_defer_inline = on != 0;
  return old;
}

// deferred mode of one backward call: requested by the caller (shape.defer = 1 + scratch half), side stream available, not
// disabled (the learned-table attention's `dpe`, which the caller reads right after the call, is computed on the main stream)
static inline bool want_deferred(const EaLayerShape& sh, bool overlap) {
  return overlap && g_defer_default && (sh.defer == 1 || sh.defer == 2);
}
// scratch sizing shared by the conformer / transformer workspace queries: the arena must hold either one immediate-mode
// backward or two deferred-mode halves
static long scratch_need(const std::function<void(Ctx&)>& walk) {
  Arena sc{nullptr, 0, 0};
  Ctx c{nullptr, true, 0, &sc, nullptr, g_overlap_default};
  walk(c);
  long need = (long)sc.peak + 256;
  if (g_overlap_default && g_defer_default) {
    Arena sd{nullptr, 0, 0};
    Deferred dummy;
    Ctx d{nullptr, true, 0, &sd, nullptr, true};
    d.df = &dummy;
    walk(d);
    const long two = 2 * (((long)sd.peak + 511) & ~255L);
    if (two > need) need = two;
  }
  return need;
}

static bool shape_ok(const EaLayerShape& sh) {
  return sh.B > 0 && sh.T > 0 && sh.C % 8 == 0 && sh.H > 0 && sh.C % sh.H == 0 && sh.F % 8 == 0 && sh.T <= 1024;
}

int ea_conformer_layer_workspace(const EaLayerShape* shape, long* saved_bytes, long* scratch_bytes) {
  if (!shape_ok(*shape)) return -2;
  EaConformerLayer L;
  memset(&L, 0, sizeof(L));
  Arena sv{nullptr, 0, 0};
  EaLayerShape sh = *shape;
  sh.defer = 1;
  *scratch_bytes = scratch_need([&](Ctx& c) {
    Arena a{nullptr, 0, 0}, b{nullptr, 0, 0};
    layer_fwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, nullptr, a);
    layer_bwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, nullptr, b);
    if (a.peak > sv.peak) sv.peak = a.peak;
  });
  *saved_bytes = (long)sv.peak + 256;
  return 0;
}

// arena capacities are checked with a dry (sizing) pass before anything is launched
static bool arenas_fit(const EaLayerShape& sh, bool backward, bool overlap, long saved_bytes, long scratch_bytes, bool transformer = false,
                       bool deferred = false) {
  EaConformerLayer L;
  memset(&L, 0, sizeof(L));
  Arena sv{nullptr, 0, 0}, sc{nullptr, 0, 0};
  Ctx c{nullptr, true, 0, &sc, nullptr, overlap};
  Deferred dummy;
  if (deferred) c.df = &dummy;
  if (transformer) {
    if (backward) tlayer_bwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sv);
    else tlayer_fwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, nullptr, sv);
  } else if (backward) layer_bwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, nullptr, sv);
  else layer_fwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, nullptr, sv);
  return (long)sv.peak <= saved_bytes && (long)sc.peak <= scratch_bytes;
}

// One backward call in deferred mode: the main chain runs out of scratch half (defer - 1); its side work is launched behind one
// fork at the end and joined by the NEXT backward call (or ea_backward_flush / any forward call).
static int bwd_deferred(const EaConformerLayer* layer, const EaLayerShape& sh, const void* x_in, const void* dy, void* dx,
                        const int* key_len, const void* pe, float* dpe, void* saved, long saved_bytes, void* scratch,
                        long scratch_bytes, hipStream_t stream, bool transformer, const EaLayerChain* chain = nullptr) {
  static Deferred df;  // one process drives one GPU from one thread (the autograd engine's device thread)
  const int half = sh.defer - 1;
  const long half_bytes = (scratch_bytes / 2) & ~255L;
  if (!arenas_fit(sh, true, true, saved_bytes, half_bytes, transformer, true)) return -5;
  if (join_half(stream, half) != 0) return -1;  // only if the caller did not alternate the halves
  Arena sv{(char*)saved, 0, 0}, sc{(char*)scratch + (size_t)half * half_bytes, 0, 0};
  Ctx c{stream, false, 0, &sc, g_side.stream, true};
  df.clear();
  c.df = &df;
  c.chain = chain;
  return transformer ? tlayer_bwd(c, layer, sh, x_in, dy, dx, key_len, pe, dpe, sv) : layer_bwd(c, layer, sh, x_in, dy, dx, key_len, pe, sv);
}

// what a chained call needs beyond the plain one: narrow rows (the paired LayerNorm kernels) and neighbour arenas of full size
static int chain_ok(const EaLayerShape& sh, const EaLayerChain* ch, bool backward) {
  if (!ch) return 0;
  if ((ch->next || ch->prev || ch->ln1_done || ch->final_ln_done) && sh.C > 512) return -2;
  if (!backward && ch->next && (!ch->next_saved || !arenas_fit(sh, false, false, ch->next_saved_bytes, 1L << 60))) return -5;
  if (backward && ch->prev && (!ch->prev_saved || !ch->prev_pre || !arenas_fit(sh, false, false, ch->prev_saved_bytes, 1L << 60))) return -5;
  if (backward && ch->final_ln_done && !ch->pre_in) return -2;
  return 0;
}

int ea_conformer_layer_fwd_chained(const EaConformerLayer* layer, const EaLayerShape* shape, const EaLayerChain* chain, const void* x_in,
                                   void* x_out, const int* key_len, const float* attn_mask, const void* pe, void* saved,
                                   long saved_bytes, void* scratch, long scratch_bytes, hipStream_t stream) {
  if (!shape_ok(*shape)) return -2;
  if (!arenas_fit(*shape, false, false, saved_bytes, scratch_bytes)) return -5;
  if (const int rc = chain_ok(*shape, chain, false)) return rc;
  Arena sv{(char*)saved, 0, 0}, sc{(char*)scratch, 0, 0};
  Ctx c{stream, false, 0, &sc, nullptr, false};
  c.chain = chain;
  if ((attn_mask != nullptr) != (shape->has_attn_mask != 0)) return -2;
  if (join_all(stream) != 0) return -1;  // a forward reuses the scratch arena: no deferred side work may still be reading it
  return layer_fwd(c, layer, *shape, x_in, x_out, key_len, attn_mask, pe, sv);
}
int ea_conformer_layer_fwd(const EaConformerLayer* layer, const EaLayerShape* shape, const void* x_in, void* x_out,
                           const int* key_len, const float* attn_mask, const void* pe, void* saved, long saved_bytes,
                           void* scratch, long scratch_bytes, hipStream_t stream) {
  return ea_conformer_layer_fwd_chained(layer, shape, nullptr, x_in, x_out, key_len, attn_mask, pe, saved, saved_bytes, scratch, scratch_bytes,
                                        stream);
}

int ea_conformer_layer_refresh_wt(const EaConformerLayer* layer, const EaLayerShape* shape, hipStream_t stream) {
  if (!shape_ok(*shape)) return -2;
  Arena sc{nullptr, 0, 0};
  Ctx c{stream, false, 0, &sc, nullptr, false};
  EaLayerShape sh = *shape;
  sh.training = 1;
  wt_refresh(c, layer, sh);
  return c.rc;
}

int ea_set_fused_predrop(int on) {
  const int old = g_fuse_predrop;
  g_fuse_predrop = on != 0;
  return old;
}

int ea_set_flash_attention(int on) {
  const int old = g_flash;
  g_flash = on != 0;
  return old;
}

int ea_conformer_layer_bwd_chained(const EaConformerLayer* layer, const EaLayerShape* shape, const EaLayerChain* chain, const void* x_in,
                                   const void* dy, void* dx, const int* key_len, const void* pe, void* saved, long saved_bytes,
                                   void* scratch, long scratch_bytes, hipStream_t stream) {
  if (!shape_ok(*shape)) return -2;
  if (const int rc = chain_ok(*shape, chain, true)) return rc;
  const bool ov = g_overlap_default && side_init(stream);
  if (want_deferred(*shape, ov))
    return bwd_deferred(layer, *shape, x_in, dy, dx, key_len, pe, nullptr, saved, saved_bytes, scratch, scratch_bytes, stream, false, chain);
  if (!arenas_fit(*shape, true, ov, saved_bytes, scratch_bytes)) return -5;
  if (join_all(stream) != 0) return -1;
  Arena sv{(char*)saved, 0, 0}, sc{(char*)scratch, 0, 0};
  Ctx c{stream, false, 0, &sc, ov ? g_side.stream : nullptr, ov};
  c.chain = chain;
  if (ov) stream_wait(c, c.side, c.s);
  return layer_bwd(c, layer, *shape, x_in, dy, dx, key_len, pe, sv);
}
int ea_conformer_layer_bwd(const EaConformerLayer* layer, const EaLayerShape* shape, const void* x_in, const void* dy, void* dx,
                           const int* key_len, const void* pe, void* saved, long saved_bytes, void* scratch, long scratch_bytes,
                           hipStream_t stream) {
  return ea_conformer_layer_bwd_chained(layer, shape, nullptr, x_in, dy, dx, key_len, pe, saved, saved_bytes, scratch, scratch_bytes, stream);
}

// A stack of layers per call: the same layer calls, issued back to back from here instead of from the caller's Python loop
// (include/espresso_amd.h: what it saves is host time).
int ea_conformer_stack_fwd(const EaStackLayer* L, int n, void* x_out, const int* key_len, const float* attn_mask, const void* pe,
                           void* scratch, long scratch_bytes, int chain, hipStream_t stream) {
  if (n <= 0 || !L) return -2;
  for (int k = 0; k < n; ++k) {
    EaLayerChain ch;
    memset(&ch, 0, sizeof(ch));
    if (chain) {
      ch.ln1_done = k > 0;
      if (k + 1 < n) { ch.next = L[k + 1].layer; ch.next_saved = L[k + 1].saved; ch.next_saved_bytes = L[k + 1].saved_bytes; }
    }
    void* out = k + 1 < n ? L[k + 1].x_in : x_out;
    const int rc = ea_conformer_layer_fwd_chained(L[k].layer, &L[k].shape, &ch, L[k].x_in, out, key_len, attn_mask, pe, L[k].saved,
                                                  L[k].saved_bytes, scratch, scratch_bytes, stream);
    if (rc != 0) return rc;
  }
  return 0;
}

int ea_conformer_stack_bwd(const EaStackLayer* L, int n, const void* dy, void* dx, void* const* dbuf, void* const* pre,
                           const int* key_len, const void* pe, void* scratch, long scratch_bytes, int chain, hipStream_t stream) {
  if (n <= 0 || !L || (n > 1 && !dbuf) || (chain && n > 1 && !pre)) return -2;
  for (int k = n - 1; k >= 0; --k) {
    EaLayerChain ch;
    memset(&ch, 0, sizeof(ch));
    if (chain) {
      if (k + 1 < n) { ch.final_ln_done = 1; ch.pre_in = pre[(k + 1) % 3]; }
      if (k > 0) {
        ch.prev = L[k - 1].layer; ch.prev_saved = L[k - 1].saved; ch.prev_saved_bytes = L[k - 1].saved_bytes;
        ch.prev_seed = L[k - 1].shape.seed; ch.prev_pre = pre[k % 3];
      }
    }
    const void* gin = k + 1 < n ? dbuf[(k + 1) & 1] : dy;
    void* gout = k > 0 ? dbuf[k & 1] : dx;
    const int rc = ea_conformer_layer_bwd_chained(L[k].layer, &L[k].shape, &ch, L[k].x_in, gin, gout, key_len, pe, L[k].saved,
                                                  L[k].saved_bytes, scratch, scratch_bytes, stream);
    if (rc != 0) return rc;
  }
  return 0;
}

int ea_transformer_layer_workspace(const EaLayerShape* shape, long* saved_bytes, long* scratch_bytes) {
  if (!shape_ok(*shape)) return -2;
  EaConformerLayer L;
  memset(&L, 0, sizeof(L));
  Arena sv{nullptr, 0, 0};
  EaLayerShape sh = *shape;
  sh.defer = 1;
  *scratch_bytes = scratch_need([&](Ctx& c) {
    Arena a{nullptr, 0, 0}, b{nullptr, 0, 0};
    tlayer_fwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, nullptr, a);
    tlayer_bwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, b);
    if (a.peak > sv.peak) sv.peak = a.peak;
  });
  if (sv.peak == 0) {  // (walk order above: the first, immediate-mode walk always sets it)
    Arena a{nullptr, 0, 0}, sc{nullptr, 0, 0};
    Ctx c{nullptr, true, 0, &sc, nullptr, false};
    tlayer_fwd(c, &L, sh, nullptr, nullptr, nullptr, nullptr, nullptr, a);
    sv.peak = a.peak;
  }
  *saved_bytes = (long)sv.peak + 256;
  return 0;
}

int ea_transformer_layer_fwd(const EaConformerLayer* layer, const EaLayerShape* shape, const void* x_in, void* x_out,
                             const int* key_len, const float* attn_mask, const void* pe, void* saved, long saved_bytes,
                             void* scratch, long scratch_bytes, hipStream_t stream) {
  if (!shape_ok(*shape)) return -2;
  if ((attn_mask != nullptr) != (shape->has_attn_mask != 0)) return -2;
  if (!arenas_fit(*shape, false, false, saved_bytes, scratch_bytes, true)) return -5;
  if (join_all(stream) != 0) return -1;
  Arena sv{(char*)saved, 0, 0}, sc{(char*)scratch, 0, 0};
  Ctx c{stream, false, 0, &sc, nullptr, false};
  return tlayer_fwd(c, layer, *shape, x_in, x_out, key_len, attn_mask, pe, sv);
}

int ea_transformer_layer_bwd(const EaConformerLayer* layer, const EaLayerShape* shape, const void* x_in, const void* dy, void* dx,
                             const int* key_len, const void* pe, float* dpe, void* saved, long saved_bytes, void* scratch,
                             long scratch_bytes, hipStream_t stream) {
  if (!shape_ok(*shape)) return -2;
  if ((shape->pos_mode == 1) != (dpe != nullptr)) return -2;
  const bool ov = g_overlap_default && side_init(stream);
  if (want_deferred(*shape, ov)) return bwd_deferred(layer, *shape, x_in, dy, dx, key_len, pe, dpe, saved, saved_bytes, scratch, scratch_bytes, stream, true);
  if (!arenas_fit(*shape, true, ov, saved_bytes, scratch_bytes, true)) return -5;
  if (join_all(stream) != 0) return -1;
  Arena sv{(char*)saved, 0, 0}, sc{(char*)scratch, 0, 0};
  Ctx c{stream, false, 0, &sc, ov ? g_side.stream : nullptr, ov};
  if (ov) stream_wait(c, c.side, c.s);
  return tlayer_bwd(c, layer, *shape, x_in, dy, dx, key_len, pe, dpe, sv);
}

int ea_decoder_layer_workspace(const EaLayerShape* shape, long* saved_bytes, long* scratch_bytes) {
  if (!dshape_ok(*shape)) return -2;
  EaDecoderLayer L;
  memset(&L, 0, sizeof(L));
  Arena sv{nullptr, 0, 0}, sc{nullptr, 0, 0};
  Ctx c{nullptr, true, 0, &sc, nullptr, g_overlap_default};
  dlayer_fwd(c, &L, *shape, nullptr, nullptr, nullptr, nullptr, sv);
  Arena sv2{nullptr, 0, 0};
  dlayer_bwd(c, &L, *shape, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sv2);
  *saved_bytes = (long)sv.peak + 256;
  *scratch_bytes = (long)sc.peak + 256;
  return 0;
}

int ea_decoder_layer_fwd(const EaDecoderLayer* layer, const EaLayerShape* shape, const void* x_in, const void* enc, void* x_out,
                         const int* enc_len, void* saved, long saved_bytes, void* scratch, long scratch_bytes, hipStream_t stream) {
  if (!dshape_ok(*shape)) return -2;
  if (!darenas_fit(*shape, false, false, saved_bytes, scratch_bytes)) return -5;
  if (join_all(stream) != 0) return -1;
  Arena sv{(char*)saved, 0, 0}, sc{(char*)scratch, 0, 0};
  Ctx c{stream, false, 0, &sc, nullptr, false};
  return dlayer_fwd(c, layer, *shape, x_in, enc, x_out, enc_len, sv);
}

int ea_decoder_layer_bwd(const EaDecoderLayer* layer, const EaLayerShape* shape, const void* x_in, const void* enc, const void* dy,
                         void* dx, void* denc, const int* enc_len, void* saved, long saved_bytes, void* scratch, long scratch_bytes,
                         hipStream_t stream) {
  if (!dshape_ok(*shape)) return -2;
  const bool ov = g_overlap_default && side_init(stream);
  if (!darenas_fit(*shape, true, ov, saved_bytes, scratch_bytes)) return -5;
  if (join_all(stream) != 0) return -1;
  Arena sv{(char*)saved, 0, 0}, sc{(char*)scratch, 0, 0};
  Ctx c{stream, false, 0, &sc, ov ? g_side.stream : nullptr, ov};
  if (ov) stream_wait(c, c.side, c.s);
  return dlayer_bwd(c, layer, *shape, x_in, enc, dy, dx, denc, enc_len, sv);
}

}  // extern "C"

// include/espresso_amd.h: what the hardware-queue probe decided for the layer runtime's side stream
extern "C" int ea_side_stream_report(int* rejected, int* unprobed) {
  if (rejected) *rejected = g_side_rejected;
  if (unprobed) *unprobed = g_side_unprobed;
  return g_side.ok ? 1 : 0;
}

// include/espresso_amd.h: seed of one dropout site of a layer call (EA_SITE_*), given EaLayerShape.seed
extern "C" uint64_t ea_layer_dropout_seed(uint64_t layer_seed, int site) {
  switch (site) {
    case EA_SITE_FFN1_ACT: return layer_seed + kFfn1 + kAct;
    case EA_SITE_FFN1_OUT: return layer_seed + kFfn1 + kOut;
    case EA_SITE_ATTN_PROBS: return layer_seed + kAttn + kProbs;
    case EA_SITE_ATTN_OUT: return layer_seed + kAttn + kAttnOut;
    case EA_SITE_CONV_OUT: return layer_seed + kConv + kConvOut;
    case EA_SITE_FFN2_ACT: return layer_seed + kFfn2 + kAct;
    case EA_SITE_FFN2_OUT: return layer_seed + kFfn2 + kOut;
    case EA_SITE_CROSS_PROBS: return layer_seed + kCross + kProbs;
    case EA_SITE_CROSS_OUT: return layer_seed + kCross + kAttnOut;
    default: return 0;
  }
}
