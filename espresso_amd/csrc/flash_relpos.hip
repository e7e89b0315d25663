// Fused relative-position self-attention for gfx950, encoder hot path (T == S, head dim 64, no causal mask):
// forward, backward-Q and backward-KV kernels of the Conformer / Transformer-XL attention
//   S[i][j] = (q_i + u) . k_j  +  (q_i + v) . p_{T-1-i+j}        fairseq/modules/multihead_attention.py:788-831
//   P = softmax(S + key padding), Pd = dropout(P), O = Pd V        :835-907
// Same mathematics and operand layout as flash_attention.hip (which keeps serving causal / cross / absolute-position
// attention); what is different is how the data moves, because those kernels were VALU-bound (750-1300 vector
// instructions per 64-key tile against 34-64 MFMAs; tools/isa_mix.py):
//   * operand tiles go global -> LDS with global_load_lds (no staging registers, no ds_write pass, no per-row predication:
//     out-of-range rows are clamped, their products only reach masked entries); K / V (or Qu / dO) tiles are double
//     buffered, the positional table is a ring of three 64-row blocks because consecutive tiles share half their window;
//   * ONE row-major [64][64] bf16 image per operand serves both orientations: k-contiguous fragments by ds_read_b128,
//     transposed fragments (V^T, K^T, PP^T, Qu^T, dO^T) by ds_read_b64_tr_b16 — the second, transposed LDS images and
//     their register transposes are gone.  One 16-byte-slot swizzle is conflict free for both (tools/lds_layout_check.py);
//   * P.V, dS.K, dS^T.Qu ... run as 16x16x32 MFMAs: the C layout of two neighbouring 16-key score tiles IS a B operand
//     if the 32 keys are taken in the order (tile a rows 4g..4g+3, tile b rows 4g..4g+3); the transposing reads deliver
//     the A operand in the same order;
//   * attention dropout: the keep decisions are evaluated ONCE per element by a separate bit kernel (same counter hash
//     and element index as everywhere else: ea_keep(seed, (z*T+i)*S + j)), 16 bits per lane and tile; the three
//     attention kernels read two bytes instead of hashing 16 elements each (3 quarter-rate multiplies per element);
//   * softmax in the exp2 domain (one fma + v_exp_f32 per element), key-padding mask only on the tile that contains the
//     sentence end, partial row sums reduced across lanes once at the end;
//   * backward KV: a wave owns 16 query rows x 64 keys while it recomputes P and dS (identical code and lane layout to the
//     Q kernel: 10 instead of 16 positional MFMAs, one skew round trip), the four waves exchange dS / Pd through LDS and
//     then each owns 16 keys for dK^T = Qu^T dS, dV^T = dO^T Pd.  K and V stay in registers as A fragments.
#include "flash_internal.h"

namespace {

typedef short bf16x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_b4_t;

constexpr int DH = 64, TQ = 64, TK = 64;
constexpr int IMG = 64 * 128;            // one [64 rows][64 bf16] image
constexpr int BDP = 18;                  // pitch (floats) of the per-wave skew buffer [80][BDP]
constexpr int BNC = 80 * BDP * 4;        // bytes of one wave's skew buffer (also holds the un-skew / exchange tiles)
constexpr int OBP = 88;                  // pitch (bf16) of the un-skew buffer [16 rows][80 positions]
constexpr int PP_OFF = 4 * IMG, BNC_OFF = 7 * IMG;
constexpr int LDS_BYTES = 7 * IMG + 4 * BNC;  // 2 stages x 2 images + 3 positional blocks + 4 skew buffers = 80 384
constexpr int AUX_OFF = LDS_BYTES;            // KV kernel: 2 stages x {lse[64], D[64]} fp32 behind the skew buffers
constexpr int LDS_BYTES_KV = LDS_BYTES + 1024;  // 81 408 <= 81 920 = half of a CU's LDS
constexpr float LOG2E = 1.4426950408889634f;

// 16-byte slot s of image row r lives at slot s ^ swz(r): conflict free for ds_read_b128 operand fragments AND for both
// ds_read_b64_tr_b16 patterns (tools/lds_layout_check.py)
__device__ __forceinline__ int swz(int r) { return ((r >> 2) & 1) | (((r >> 1) & 1) << 1) | ((((r >> 2) ^ (r >> 3)) & 1) << 2); }

__device__ __forceinline__ int xcd_remap(int id, int total) {
  const int xcd = id & 7, q = total >> 3, r = total & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}
__device__ __forceinline__ f32x4_t mfma32(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a),
                                                 __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma16(bf16x4_t a, bf16x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8_t cat(bf16x4_t lo, bf16x4_t hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x4_t pack4(float a, float b, float c, float d) {
  uint2 pk;
  pk.x = pack_bf2(a, b);
  pk.y = pack_bf2(c, d);
  return __builtin_bit_cast(bf16x4_t, pk);
}
// transposing read (ds_read_b64_tr_b16): lane (g = lane>>4, i = lane&15) passes the address of element
// [k0(g) + (i>>2)][m0 + 4*(i&3)] of a row-major image and receives the four elements [k0(g) + e][m0 + i], e = 0..3.
// Issued from inline asm in batches with their own wait: through the builtin, hipcc puts `s_waitcnt vmcnt(0)` in front of
// the first transposing read of every tile (it cannot tell them from the global_load_lds writes in flight), which drains
// the operand prefetch in the middle of the tile it should overlap with.  The compiler does not track loads issued from
// asm, so the results must not be visible to it before the wait inside the same statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }
// 4 addresses x 4 immediate offsets -> o[a*4 + q] = image at ad[a] + OFFq
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void trr16(const uint32_t (&ad)[4], uint2 (&o)[16]) {
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
// 2 addresses x 4 immediate offsets
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void trr8(const uint32_t (&ad)[2], uint2 (&o)[8]) {
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
// 20 addresses, no offsets (the five 16-row blocks of a wave's positional window x four 16-column tiles)
__device__ __forceinline__ void trr20(const uint32_t (&ad)[20], uint2 (&o)[20]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %20\n\tds_read_b64_tr_b16 %1, %21\n\tds_read_b64_tr_b16 %2, %22\n\tds_read_b64_tr_b16 %3, %23\n\t"
      "ds_read_b64_tr_b16 %4, %24\n\tds_read_b64_tr_b16 %5, %25\n\tds_read_b64_tr_b16 %6, %26\n\tds_read_b64_tr_b16 %7, %27\n\t"
      "ds_read_b64_tr_b16 %8, %28\n\tds_read_b64_tr_b16 %9, %29\n\tds_read_b64_tr_b16 %10, %30\n\tds_read_b64_tr_b16 %11, %31\n\t"
      "ds_read_b64_tr_b16 %12, %32\n\tds_read_b64_tr_b16 %13, %33\n\tds_read_b64_tr_b16 %14, %34\n\tds_read_b64_tr_b16 %15, %35\n\t"
      "ds_read_b64_tr_b16 %16, %36\n\tds_read_b64_tr_b16 %17, %37\n\tds_read_b64_tr_b16 %18, %38\n\tds_read_b64_tr_b16 %19, %39\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]),
        "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(o[16]),
        "=&v"(o[17]), "=&v"(o[18]), "=&v"(o[19])
      : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]), "v"(ad[8]), "v"(ad[9]),
        "v"(ad[10]), "v"(ad[11]), "v"(ad[12]), "v"(ad[13]), "v"(ad[14]), "v"(ad[15]), "v"(ad[16]), "v"(ad[17]), "v"(ad[18]),
        "v"(ad[19])
      : "memory");
}
// 8 bytes of the un-skew buffer as two SCALAR float loads (merged into one ds_read_b64 / ds_read2 by the compiler): hipcc
// puts `s_waitcnt vmcnt(0)` in front of vector-typed LDS loads that follow a global_load_lds (type-based alias analysis
// cannot separate them from the in-flight LDS writes), which would drain the next tile's prefetch in mid-tile
__device__ __forceinline__ uint2 lds_u2(const char* p) {
  const float* f = reinterpret_cast<const float*>(p);
  return make_uint2(__float_as_uint(f[0]), __float_as_uint(f[1]));
}
__device__ __forceinline__ bf16x8_t cat2(uint2 lo, uint2 hi) {
  return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
}
// the [64 d] x [64 keys / rows] transposed operand of an image as eight 16x32 fragments: f[dt*2 + kb], keys kb*32 .. +31 in the
// permuted order (rows 4g..4g+3 of the block's first 16, then of its second 16)
__device__ __forceinline__ void tr_image(const char* img, const uint32_t (&tro)[4], bf16x8_t (&f)[8]) {
  const uint32_t b = lds_addr(img);
  const uint32_t ad[4] = {b + tro[0], b + tro[1], b + tro[2], b + tro[3]};
  uint2 o[16];
  trr16<0, 2048, 4096, 6144>(ad, o);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    f[dt * 2 + 0] = cat2(o[dt * 4 + 0], o[dt * 4 + 1]);
    f[dt * 2 + 1] = cat2(o[dt * 4 + 2], o[dt * 4 + 3]);
  }
}
__device__ __forceinline__ bf16x8_t ldf(const char* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

// -DRP_PROF: s_memtime stamps at the phase boundaries of the backward kernels, per-wave sums written to a.prof and averaged on
// the host by ea_rp_bwd (EA_RP_PROF=1).  Diagnostic only: the stamps wait for lgkmcnt and so add sync points of their own.
#ifdef RP_PROF
#define PROF_DECL unsigned long long prof_t = __builtin_amdgcn_s_memtime(); unsigned prof_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_MARK(k) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof_acc[k] += (unsigned)(n_ - prof_t); prof_t = n_; }
#define PROF_DUMP(kern) if (a.prof && (threadIdx.x & 63) == 0) { unsigned long long* o_ = a.prof + ((long)(kern) * 8192 + (long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 10; for (int k_ = 0; k_ < 10; ++k_) o_[k_] = prof_acc[k_]; }
#else
#define PROF_DECL
#define PROF_MARK(k)
#define PROF_DUMP(kern)
#endif

struct LaneK {
  int lane, w, li, g4;
  uint32_t offk[2];  // k-contiguous fragment: row li (+16 rows = +2048 bytes), slot ks*4 + g4
  uint32_t tro[4];   // transposing read inside a 16-row block: row 4*g4 + (li>>2), columns dt*16 + 4*(li&3)
  uint32_t bd_w;     // skew buffer (floats): write [(ct*16 + g4*4 + r)][li]
  uint32_t bd_r;     //                       read  [(15 - li + jt*16 + g4*4 + r)][li]
};
__device__ __forceinline__ LaneK lane_consts() {
  LaneK L;
  const int tid = threadIdx.x;
  L.lane = tid & 63;
  L.w = __builtin_amdgcn_readfirstlane(tid >> 6);
  L.li = L.lane & 15;
  L.g4 = L.lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) L.offk[ks] = L.li * 128 + (((ks * 4 + L.g4) ^ swz(L.li)) << 4);
  const int rowl = 4 * L.g4 + (L.li >> 2);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
    L.tro[dt] = rowl * 128 + (((dt * 2 + ((L.li & 3) >> 1)) ^ swz(rowl)) << 4) + (L.li & 1) * 8;
  L.bd_w = (L.g4 * 4) * BDP + L.li;
  L.bd_r = (15 - L.li + L.g4 * 4) * BDP + L.li;
  return L;
}

// one [64][64] bf16 image: rows row0 .. row0+63 of a matrix (clamped to [0, rmax]) -> LDS at dst.  A wave instruction moves
// 8 rows (lane -> row lane>>3, slot lane&7); wave w carries row groups w and w+4; the swizzle is applied to the source chunk.
// The per-lane source pointers for row0 = 0 are formed once; a tile whose rows are all in range adds one wave-uniform offset.
struct ImgSrc {
  const bf16_t* p[2];
  long ld;
  int rmax;
};
__device__ __forceinline__ ImgSrc img_src(const bf16_t* base, long ld, int rmax, int w, int lane) {
  ImgSrc s;
  s.ld = ld;
  s.rmax = rmax;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int r = (w + 4 * n) * 8 + (lane >> 3);
    s.p[n] = base + (long)r * ld + (((lane & 7) ^ swz(r)) << 3);
  }
  return s;
}
__device__ __forceinline__ void issue_img(char* dst, const ImgSrc& s, int row0, int w, int lane) {
  if (row0 >= 0 && row0 + 63 <= s.rmax) {  // wave-uniform
    const long off = (long)row0 * s.ld;
#pragma unroll
    for (int n = 0; n < 2; ++n)
      __builtin_amdgcn_global_load_lds((gptr_t)(s.p[n] + off), (lptr_t)(dst + (w + 4 * n) * 1024), 16, 0, 0);
  } else {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int r = (w + 4 * n) * 8 + (lane >> 3);
      const int g = min(max(row0 + r, 0), s.rmax);
      __builtin_amdgcn_global_load_lds((gptr_t)(s.p[n] + (long)(g - r) * s.ld), (lptr_t)(dst + (w + 4 * n) * 1024), 16, 0, 0);
    }
  }
}

// positional logits of one wave (16 query rows x 64 keys touch 79 consecutive relative positions = window rows
// c0w .. c0w+78 of the 128-row window [lo block | hi block], c0w = 48 - 16 w): BD^T[c'][i] on 5 MFMA tiles, then the skew
// S^T[j][i] += BD^T[15 - i_w + j][i_w] through the wave's LDS buffer.
__device__ __forceinline__ void add_band(f32x4_t (&acc_s)[4], const bf16x8_t (&qv)[2], const char* blk_lo, const char* blk_hi,
                                         float* bd, const LaneK& L) {
  // software pipeline over the five band tiles: the table fragments of tile ct+1 are read while tile ct is on the MFMA pipe,
  // and the skew reads of key tile jt = ct-1 (they need band tiles ct-1 and ct) follow the stores of tile ct directly.  LDS
  // serves a wavefront's instructions in order, so the cross-lane store -> load hand-off needs no wait; the compiler keeps
  // the order because stores and loads go to the same array through index expressions it cannot tell apart.
  bf16x8_t pf[2][2];
  auto frag = [&](int ct, bf16x8_t (&f)[2]) {
    const int q = ct - L.w + 3;  // wave-uniform, 0..7
    const char* p = ((q >> 2) ? blk_hi : blk_lo) + (q & 3) * 2048;
    f[0] = ldf(p + L.offk[0]);
    f[1] = ldf(p + L.offk[1]);
  };
  frag(0, pf[0]);
#pragma unroll
  for (int ct = 0; ct < 5; ++ct) {
    if (ct + 1 < 5) frag(ct + 1, pf[(ct + 1) & 1]);
    f32x4_t t = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    t = mfma32(pf[ct & 1][0], qv[0], t);
    t = mfma32(pf[ct & 1][1], qv[1], t);
#pragma unroll
    for (int r = 0; r < 4; ++r) bd[L.bd_w + (ct * 16 + r) * BDP] = t[r];
    if (ct >= 1) {
      const int jt = ct - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_s[jt][r] += bd[L.bd_r + (jt * 16 + r) * BDP];
    }
  }
}

// keep decision of element (jt, r) from the lane's 16-bit piece: all-ones / zero mask
__device__ __forceinline__ uint32_t keep_mask(uint32_t piece, int k) { return (uint32_t)__builtin_amdgcn_sbfe((int)piece, k, 1); }
__device__ __forceinline__ float fand(float x, uint32_t m) { return __uint_as_float(__float_as_uint(x) & m); }

// ---- keep bits --------------------------------------------------------------------------------------------------------
// bits[((z*nkt + kt)*Tpad + i)*4 + g] : bit (jt*4 + r) = keep decision of key j = kt*64 + jt*16 + g*4 + r for query row i of
// (head, sentence) z — exactly the 16 elements lane (i & 15, g) of the attention kernels holds for key tile kt.
// Grid-stride over a CAPPED grid (2 workgroups per CU): the kernel runs on the side stream next to the QKV projection of the
// same layer; one workgroup per 256 pieces (5 488 workgroups at the recipe's batch) took every wave slot of the device and the
// projection's workgroups queued behind them (53 us in the step against 27 alone).
__global__ __launch_bounds__(256) void keep_bits_kernel(uint16_t* bits, uint64_t seed, uint32_t thr, int Z, int T, int nkt) {
  const int Tpad = nkt * 64;
  const long n = (long)Z * nkt * Tpad * 4;
  const long stride = (long)gridDim.x * 256;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < n; id += stride) {
    const int g = (int)(id & 3);
    long t = id >> 2;
    const int i = (int)(t % Tpad);
    t /= Tpad;
    const int kt = (int)(t % nkt);
    const int z = (int)(t / nkt);
    uint32_t piece = 0;
    if (i < T) {
      const uint64_t rowbase = ((uint64_t)z * T + (uint64_t)i) * (uint64_t)T;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = kt * 64 + jt * 16 + g * 4 + r;
          if (j < T && ea_hash(seed, rowbase + (uint64_t)j) >= thr) piece |= 1u << (jt * 4 + r);
        }
    }
    bits[id] = (uint16_t)piece;
  }
}

// ======================================================================================================================
// forward
template <bool DROP>
__global__ __launch_bounds__(256, 2) void rp_fwd_kernel(const FlashFwdArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];  // the ONLY LDS object (a second one de-pipelines glds loops)
  const LaneK L = lane_consts();
  const int lane = L.lane, w = L.w, li = L.li, g4 = L.g4;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int z = vid / a.nq, qt = vid % a.nq;
  const int h = z / a.B, b = z % a.B;
  const int T = a.T;
  const int i0 = qt * TQ;
  const int i = i0 + 16 * w + li;  // this lane's query row
  const int kl = a.klen ? min(a.klen[b], T) : T;
  const int nt = (kl + TK - 1) / TK;
  const int nkt = (T + TK - 1) / TK;

  bf16x8_t qu[2], qv[2];
  {
    const int ic = min(i, T - 1);  // rows past the end: a valid row's data, never stored
    const long o = ((long)b * T + ic) * a.ldq + h * DH + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qu[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a.qu + o + ks * 32));
      qv[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a.qv + o + ks * 32));
    }
  }
  const bf16_t* Kb = a.k + (long)b * T * a.ldkv + h * DH;
  const bf16_t* Vb = a.v + (long)b * T * a.ldkv + h * DH;
  const bf16_t* PPb = a.pp + h * DH;
  const int R = 2 * T - 1;
  const ImgSrc srcK = img_src(Kb, a.ldkv, T - 1, w, lane), srcV = img_src(Vb, a.ldkv, T - 1, w, lane);
  const ImgSrc srcP = img_src(PPb, a.ldpp, R - 1, w, lane);
  const int pbase = (T - 1) - (i0 + TQ - 1);  // first table row of positional block 0
  float* bd = reinterpret_cast<float*>(lds + BNC_OFF + w * BNC);
  // the lane's 16-bit piece is read inside its aligned 32-bit word (pieces g4 and g4^1): a 16-bit load is widened with an AND
  // that hipcc places right behind the prefetch, together with a wait for everything in flight
  const uint32_t* bitp = DROP ? reinterpret_cast<const uint32_t*>(a.bits) + (((long)z * nkt) * (nkt * 64) + i) * 2 + (g4 >> 1) : nullptr;
  const long bit_step = (long)nkt * 64 * 2;
  const int ksh = (g4 & 1) * 16;

  f32x4_t acc_o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc_o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;  // l_run: this lane's share of the row sum (4 lanes per row)
  uint32_t piece = 0, piece_next = 0;

  if (nt > 0) {
    issue_img(lds, srcK, 0, w, lane);
    issue_img(lds + IMG, srcV, 0, w, lane);
    issue_img(lds + PP_OFF, srcP, pbase, w, lane);
    issue_img(lds + PP_OFF + IMG, srcP, pbase + 64, w, lane);
    if (DROP) piece_next = bitp[0];
  }
  int slot_lo = 0;  // ring slot of positional block t
  for (int t = 0; t < nt; ++t) {
    const int j0 = t * TK;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile t is in LDS (every wave's part); everyone is done with tile t-1
    const int slot_hi = slot_lo == 2 ? 0 : slot_lo + 1;
    const int slot_nx = slot_hi == 2 ? 0 : slot_hi + 1;
    if (DROP) {
      piece = piece_next;
      asm volatile("" : "+v"(piece));  // pin the copy (and the compiler's wait for the load) HERE, ahead of the next tile's prefetch
    }
    if (t + 1 < nt) {
      // register-destination loads first: a later wait for one of them is then `vmcnt(6)`, not a wait for the prefetch
      if (DROP) piece_next = bitp[(long)(t + 1) * bit_step];
      char* st = lds + ((t + 1) & 1) * 2 * IMG;
      issue_img(st, srcK, j0 + TK, w, lane);
      issue_img(st + IMG, srcV, j0 + TK, w, lane);
      issue_img(lds + PP_OFF + slot_nx * IMG, srcP, pbase + 64 * (t + 2), w, lane);
    }
    const char* sK = lds + (t & 1) * 2 * IMG;
    const char* sV = sK + IMG;

    // content logits (transposed): acc_s[jt][r] = S[i][j0 + jt*16 + g4*4 + r]
    f32x4_t acc_s[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc_s[jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    {  // fragments of key tile jt+1 are read while tile jt is on the MFMA pipe
      bf16x8_t fr[2][2];
      fr[0][0] = ldf(sK + L.offk[0]);
      fr[0][1] = ldf(sK + L.offk[1]);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        if (jt + 1 < 4) {
          fr[(jt + 1) & 1][0] = ldf(sK + (jt + 1) * 2048 + L.offk[0]);
          fr[(jt + 1) & 1][1] = ldf(sK + (jt + 1) * 2048 + L.offk[1]);
        }
        acc_s[jt] = mfma32(fr[jt & 1][0], qu[0], acc_s[jt]);
        acc_s[jt] = mfma32(fr[jt & 1][1], qu[1], acc_s[jt]);
      }
    }
    add_band(acc_s, qv, lds + PP_OFF + slot_lo * IMG, lds + PP_OFF + slot_hi * IMG, bd, L);

    if (j0 + TK > kl) {  // the tile that contains the sentence end
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + jt * 16 + g4 * 4 + r >= kl) acc_s[jt][r] = -INFINITY;
    }
    // online softmax, exp2 domain (column i lives in the 4 lanes {li, li+16, li+32, li+48})
    float tmax = fmaxf(fmaxf(acc_s[0][0], acc_s[0][1]), fmaxf(acc_s[0][2], acc_s[0][3]));
#pragma unroll
    for (int jt = 1; jt < 4; ++jt) tmax = fmaxf(tmax, fmaxf(fmaxf(acc_s[jt][0], acc_s[jt][1]), fmaxf(acc_s[jt][2], acc_s[jt][3])));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float msc = (m_new == -INFINITY) ? 0.f : m_new * LOG2E;
    const float alpha = fexp2(m_run * LOG2E - msc);  // first tile: exp2(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fexp2(fmaf(acc_s[jt][r], LOG2E, -msc));
        psum += p;
        acc_s[jt][r] = DROP ? fand(p, keep_mask(piece, ksh + jt * 4 + r)) : p;  // 1/(1-p) is applied once at the end
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_o[dt][r] *= alpha;

    // O^T[d][i] += sum_j V^T[d][j] Pd^T[j][i], 32 keys per MFMA
    {
      bf16x8_t vf[8];
      tr_image(sV, L.tro, vf);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const bf16x8_t pb = cat(pack4(acc_s[2 * kb][0], acc_s[2 * kb][1], acc_s[2 * kb][2], acc_s[2 * kb][3]),
                                pack4(acc_s[2 * kb + 1][0], acc_s[2 * kb + 1][1], acc_s[2 * kb + 1][2], acc_s[2 * kb + 1][3]));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc_o[dt] = mfma32(vf[dt * 2 + kb], pb, acc_o[dt]);
      }
    }
    slot_lo = slot_hi;
  }

  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (i < T) {
    const float inv = (DROP ? a.inv_keep : 1.f) / l_run;  // all-masked rows: 0/0 = NaN, exactly like the reference softmax
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

// ======================================================================================================================
// backward, shared core: for one wave's 16 query rows x the 64 keys of a tile
//   P = exp(S - lse), dS = P * (dPd * keep/(1-p) - D), Pd = P * keep/(1-p)        (transposed C layout: lane column = row i)
// Inputs: acc_s = content logits (S^T), acc_dp = dO . V^T (dPd^T), both already on MFMA; the positional band is added here.
template <bool DROP, bool WANT_PD>
__device__ __forceinline__ void softmax_bwd_tile(f32x4_t (&acc_s)[4], const f32x4_t (&acc_dp)[4], bf16x4_t (&dsb)[4], bf16x4_t (&pdb)[4],
                                                 float lse2, float Di, float inv_keep, uint32_t piece, int ksh, int jrel_end, int g4) {
  // jrel_end = kl - j0: keys at or past it are padding (>= 64: nothing to mask)
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    float ds[4], pd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p = fexp2(fmaf(acc_s[jt][r], LOG2E, -lse2));
      if (jrel_end < 64 && jt * 16 + g4 * 4 + r >= jrel_end) p = 0.f;
      float dp = acc_dp[jt][r];
      if (DROP) {
        const uint32_t m = keep_mask(piece, ksh + jt * 4 + r);
        dp = fand(dp * inv_keep, m);
        if (WANT_PD) pd[r] = fand(p * inv_keep, m);
      } else if (WANT_PD) {
        pd[r] = p;
      }
      ds[r] = p * (dp - Di);
    }
    dsb[jt] = pack4(ds[0], ds[1], ds[2], ds[3]);
    if (WANT_PD) pdb[jt] = pack4(pd[0], pd[1], pd[2], pd[3]);
  }
}

// ---- Q kernel: workgroup = (z, 64 query rows), loop over key tiles -> t1, t2 (gradients of q+u, q+v), dBD, D
template <bool DROP>
__global__ __launch_bounds__(256, 2) void rp_bwd_q_kernel(const FlashBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  const LaneK L = lane_consts();
  const int lane = L.lane, w = L.w, li = L.li, g4 = L.g4;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int z = vid / a.nq, qt = vid % a.nq;
  const int h = z / a.B, b = z % a.B;
  const int T = a.T;
  const int i0 = qt * TQ;
  const int i = i0 + 16 * w + li;
  const int kl = a.klen ? min(a.klen[b], T) : T;
  const int nt = (kl + TK - 1) / TK;
  const int nkt = (T + TK - 1) / TK;

  bf16x8_t qu[2], qv[2], dO[2];
  float Di = 0.f;
  {
    const int ic = min(i, T - 1);
    const long oq = ((long)b * T + ic) * a.ldq + h * DH + g4 * 8;
    const long oo = ((long)b * T + ic) * a.ldo + h * DH + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qu[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a.qu + oq + ks * 32));
      qv[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a.qv + oq + ks * 32));
      const uint4 g = *reinterpret_cast<const uint4*>(a.dout + oo + ks * 32);
      const uint4 o = *reinterpret_cast<const uint4*>(a.out + oo + ks * 32);
      dO[ks] = __builtin_bit_cast(bf16x8_t, g);
      const uint32_t wg[4] = {g.x, g.y, g.z, g.w}, wo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        Di += __uint_as_float(wg[e] << 16) * __uint_as_float(wo[e] << 16) +
              __uint_as_float(wg[e] & 0xffff0000u) * __uint_as_float(wo[e] & 0xffff0000u);
    }
  }
  Di += __shfl_xor(Di, 16, 64);
  Di += __shfl_xor(Di, 32, 64);
  // rows past the end: lse = +inf makes every probability (and dS) exactly zero
  const float lse2 = (i < T) ? a.lse[(long)z * T + i] * LOG2E : INFINITY;
  if (i < T && g4 == 0 && a.D) a.D[(long)z * T + i] = Di;

  f32x4_t acc_t1[4], acc_t2[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    acc_t1[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    acc_t2[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const bf16_t* Kb = a.k + (long)b * T * a.ldkv + h * DH;
  const bf16_t* Vb = a.v + (long)b * T * a.ldkv + h * DH;
  const bf16_t* PPb = a.pp + h * DH;
  const int R = 2 * T - 1;
  const ImgSrc srcK = img_src(Kb, a.ldkv, T - 1, w, lane), srcV = img_src(Vb, a.ldkv, T - 1, w, lane);
  const ImgSrc srcP = img_src(PPb, a.ldpp, R - 1, w, lane);
  const int pbase = (T - 1) - (i0 + TQ - 1);
  float* bd = reinterpret_cast<float*>(lds + BNC_OFF + w * BNC);
  bf16_t* ob = reinterpret_cast<bf16_t*>(bd);  // un-skew buffer [16 rows][OBP], aliases the skew buffer (dead by then)
  // the lane's 16-bit piece is read inside its aligned 32-bit word (pieces g4 and g4^1): a 16-bit load is widened with an AND
  // that hipcc places right behind the prefetch, together with a wait for everything in flight
  const uint32_t* bitp = DROP ? reinterpret_cast<const uint32_t*>(a.bits) + (((long)z * nkt) * (nkt * 64) + i) * 2 + (g4 >> 1) : nullptr;
  const long bit_step = (long)nkt * 64 * 2;
  const int ksh = (g4 & 1) * 16;
  // un-skewed band of row li covers positions c' in [15-li, 78-li] of the wave's 80; what else the buffer holds is stale
  uint2 mlo, mhi;  // AND masks for the B-operand pieces c' = 4*g4 + e and c' = 64 + 4*g4 + e
  {
    uint32_t m[4], n[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      m[e] = (4 * g4 + e >= 15 - li) ? 0xffffu : 0u;
      n[e] = (64 + 4 * g4 + e <= 78 - li) ? 0xffffu : 0u;
    }
    mlo = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
    mhi = make_uint2(n[0] | (n[1] << 16), n[2] | (n[3] << 16));
  }
  const int ob_w = li * (OBP - 1) + 15 + g4 * 4;  // element index of [li][15 - li + g4*4] (+ jt*16 + r)
  const char* ob_rd = reinterpret_cast<const char*>(ob) + li * (OBP * 2) + g4 * 8;  // bytes: [li][4*g4] (+ 32 c' per 64 bytes)

  uint32_t piece = 0, piece_next = 0;
  int jcov = 0;
  uint32_t dbd_pk[8];
  int dbd_j0 = -1;  // key tile whose dBD rows wait in dbd_pk
  auto store_dbd = [&]() {
    const int j = dbd_j0 + lane;
    if (dbd_j0 >= 0 && j < T) {
      const int row_w = i0 + 16 * w, nrow = T - row_w;  // wavefront-uniform
      bf16_t* dp = a.dBD + ((long)z * T + row_w) * a.ld_bd + (T - 1 - row_w) + j;
      if (nrow >= 16) {
#pragma unroll
        for (int iw = 0; iw < 16; ++iw) dp[(long)iw * (a.ld_bd - 1)] = (bf16_t)(dbd_pk[iw >> 1] >> (16 * (iw & 1)));
      } else {
#pragma unroll
        for (int iw = 0; iw < 16; ++iw)
          if (iw < nrow) dp[(long)iw * (a.ld_bd - 1)] = (bf16_t)(dbd_pk[iw >> 1] >> (16 * (iw & 1)));
      }
    }
  };
  if (nt > 0) {
    issue_img(lds, srcK, 0, w, lane);
    issue_img(lds + IMG, srcV, 0, w, lane);
    issue_img(lds + PP_OFF, srcP, pbase, w, lane);
    issue_img(lds + PP_OFF + IMG, srcP, pbase + 64, w, lane);
    if (DROP) piece_next = bitp[0];
  }
  int slot_lo = 0;
  PROF_DECL
  for (int t = 0; t < nt; ++t) {
    const int j0 = t * TK;
    jcov = min(T, j0 + TK);
    PROF_MARK(9)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PROF_MARK(0)
    store_dbd();  // previous tile's rows
    const int slot_hi = slot_lo == 2 ? 0 : slot_lo + 1;
    const int slot_nx = slot_hi == 2 ? 0 : slot_hi + 1;
    if (DROP) {
      piece = piece_next;
      asm volatile("" : "+v"(piece));  // pin the copy (and the compiler's wait for the load) HERE, ahead of the next tile's prefetch
    }
    if (t + 1 < nt) {
      // register-destination loads first: a later wait for one of them is then `vmcnt(6)`, not a wait for the prefetch
      if (DROP) piece_next = bitp[(long)(t + 1) * bit_step];
      char* st = lds + ((t + 1) & 1) * 2 * IMG;
      issue_img(st, srcK, j0 + TK, w, lane);
      issue_img(st + IMG, srcV, j0 + TK, w, lane);
      issue_img(lds + PP_OFF + slot_nx * IMG, srcP, pbase + 64 * (t + 2), w, lane);
    }
    PROF_MARK(1)
    const char* sK = lds + (t & 1) * 2 * IMG;
    const char* sV = sK + IMG;
    const char* blk_lo = lds + PP_OFF + slot_lo * IMG;
    const char* blk_hi = lds + PP_OFF + slot_hi * IMG;

    f32x4_t acc_s[4], acc_dp[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      acc_s[jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      acc_dp[jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    {  // fragments of key tile jt+1 are read while tile jt is on the MFMA pipe
      bf16x8_t fr[2][4];
      auto frag = [&](int jt, bf16x8_t (&f)[4]) {
        f[0] = ldf(sK + jt * 2048 + L.offk[0]);
        f[1] = ldf(sK + jt * 2048 + L.offk[1]);
        f[2] = ldf(sV + jt * 2048 + L.offk[0]);
        f[3] = ldf(sV + jt * 2048 + L.offk[1]);
      };
      frag(0, fr[0]);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        if (jt + 1 < 4) frag(jt + 1, fr[(jt + 1) & 1]);
        acc_s[jt] = mfma32(fr[jt & 1][0], qu[0], acc_s[jt]);
        acc_dp[jt] = mfma32(fr[jt & 1][2], dO[0], acc_dp[jt]);
        acc_s[jt] = mfma32(fr[jt & 1][1], qu[1], acc_s[jt]);
        acc_dp[jt] = mfma32(fr[jt & 1][3], dO[1], acc_dp[jt]);
      }
    }
    PROF_MARK(2)
    add_band(acc_s, qv, blk_lo, blk_hi, bd, L);
    PROF_MARK(3)
    bf16x4_t dsb[4], pdb[4];
    softmax_bwd_tile<DROP, false>(acc_s, acc_dp, dsb, pdb, lse2, Di, a.inv_keep, piece, ksh, kl - j0, g4);
    PROF_MARK(4)

    // t1^T[d][i] += sum_j K^T[d][j] dS^T[j][i]
    if (!(a.dbg & 4)) {
      bf16x8_t kf[8];
      tr_image(sK, L.tro, kf);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const bf16x8_t db = cat(dsb[2 * kb], dsb[2 * kb + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc_t1[dt] = mfma32(kf[dt * 2 + kb], db, acc_t1[dt]);
      }
    }
    PROF_MARK(5)
    // un-skew: dBD^T[15 - i_w + j][i_w] = dS^T[j][i_w], kept as bf16 [row i_w][position c']
    wave_lds_sync();  // the skew reads above are done before the same bytes are rewritten
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const uint2 pk = __builtin_bit_cast(uint2, dsb[jt]);
      ob[ob_w + jt * 16 + 0] = (bf16_t)(pk.x & 0xffffu);
      ob[ob_w + jt * 16 + 1] = (bf16_t)(pk.x >> 16);
      ob[ob_w + jt * 16 + 2] = (bf16_t)(pk.y & 0xffffu);
      ob[ob_w + jt * 16 + 3] = (bf16_t)(pk.y >> 16);
    }
    wave_lds_sync();
    // t2^T[d][i] += sum_c' PP^T[d][c0w + c'] dBD^T[c'][i] over the wave's 80 positions: 2 x 32 (permuted) + 16
    if (!(a.dbg & 2)) {
      uint32_t ad[20];
      uint2 pf[20];
#pragma unroll
      for (int q16 = 0; q16 < 5; ++q16) {
        const int qq = q16 - w + 3;  // 16-row block of the window
        const uint32_t pbk = lds_addr(((qq >> 2) ? blk_hi : blk_lo) + (qq & 3) * 2048);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ad[q16 * 4 + dt] = pbk + L.tro[dt];
      }
      trr20(ad, pf);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        uint2 blo = lds_u2(ob_rd + cb * 64);
        const uint2 bhi = lds_u2(ob_rd + cb * 64 + 32);
        if (cb == 0) {
          blo.x &= mlo.x;
          blo.y &= mlo.y;
        }
        const bf16x8_t db = cat2(blo, bhi);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc_t2[dt] = mfma32(cat2(pf[(2 * cb) * 4 + dt], pf[(2 * cb + 1) * 4 + dt]), db, acc_t2[dt]);
      }
      uint2 bl = lds_u2(ob_rd + 128);
      bl.x &= mhi.x;
      bl.y &= mhi.y;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        acc_t2[dt] = mfma16(__builtin_bit_cast(bf16x4_t, pf[16 + dt]), __builtin_bit_cast(bf16x4_t, bl), acc_t2[dt]);
    }
    PROF_MARK(6)
    // dBD[z][row][T-1-row + j] = dS[row][j]: lane = key, one 128-byte row segment per store instruction.  The values are only
    // READ here (two rows per register); the stores go out at the top of the next tile, ahead of its prefetch: vmcnt counts
    // loads and stores in one queue, so stores issued behind the prefetch would make the next `s_waitcnt vmcnt(0)` wait for
    // their write acknowledgements (11 % of the kernel in the s_memtime profile)
    if (!(a.dbg & 1)) {
      const bf16_t* bs = ob + 15 + lane;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        dbd_pk[k] = (uint32_t)bs[(2 * k) * (OBP - 1)] | ((uint32_t)bs[(2 * k + 1) * (OBP - 1)] << 16);
      dbd_j0 = j0;
    }
    wave_lds_sync();  // un-skew buffer reads done before the next tile's skew writes
    slot_lo = slot_hi;
    PROF_MARK(7)
  }
  store_dbd();  // last tile

  if (i < T) {
    bf16_t* o1 = a.t1 + ((long)b * T + i) * a.ldt + h * DH;
    bf16_t* o2 = a.t2 + ((long)b * T + i) * a.ldt + h * DH;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 pk;
      pk.x = pack_bf2(acc_t1[dt][0] * a.scaling, acc_t1[dt][1] * a.scaling);
      pk.y = pack_bf2(acc_t1[dt][2] * a.scaling, acc_t1[dt][3] * a.scaling);
      *reinterpret_cast<uint2*>(o1 + dt * 16 + g4 * 4) = pk;
      pk.x = pack_bf2(acc_t2[dt][0] * a.scaling, acc_t2[dt][1] * a.scaling);
      pk.y = pack_bf2(acc_t2[dt][2] * a.scaling, acc_t2[dt][3] * a.scaling);
      *reinterpret_cast<uint2*>(o2 + dt * 16 + g4 * 4) = pk;
    }
    if (a.dq) {  // the query projection's gradient, summed before rounding (saves the separate t1 + t2 pass)
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
  PROF_MARK(9)
  if (!a.dbd_prezeroed) {
    // columns of dBD no (row, key) pair of this workgroup wrote: r < T-1-row or r >= T-1-row + jcov
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
  PROF_MARK(8)
  PROF_DUMP(0)
}

// ---- KV kernel: workgroup = (z, 64 keys), loop over query tiles -> dK, dV
template <bool DROP>
__global__ __launch_bounds__(256, 2) void rp_bwd_kv_kernel(const FlashBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES_KV];
  const LaneK L = lane_consts();
  const int lane = L.lane, w = L.w, li = L.li, g4 = L.g4;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int z = vid / a.nk, kt = vid % a.nk;
  const int h = z / a.B, b = z % a.B;
  const int T = a.T;
  const int j0 = kt * TK;
  const int kl = a.klen ? min(a.klen[b], T) : T;
  const int nq = (j0 < kl) ? a.nq : 0;  // padded keys: zero gradient
  const int nkt = a.nk;

  // K, V as A fragments for all four 16-key tiles (held for the whole kernel)
  bf16x8_t kf[4][2], vf[4][2];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    const int j = min(j0 + jt * 16 + li, T - 1);
    const long o = ((long)b * T + j) * a.ldkv + h * DH + g4 * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[jt][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a.k + o + ks * 32));
      vf[jt][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a.v + o + ks * 32));
    }
  }
  f32x4_t acc_dk[4], acc_dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    acc_dk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    acc_dv[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  const bf16_t* Qub = a.qu + (long)b * T * a.ldq + h * DH;
  const bf16_t* Qvb = a.qv + (long)b * T * a.ldq + h * DH;
  const bf16_t* dOb = a.dout + (long)b * T * a.ldo + h * DH;
  const bf16_t* PPb = a.pp + h * DH;
  const int R = 2 * T - 1;
  const ImgSrc srcQ = img_src(Qub, a.ldq, T - 1, w, lane), srcG = img_src(dOb, a.ldo, T - 1, w, lane);
  const ImgSrc srcP = img_src(PPb, a.ldpp, R - 1, w, lane);
  const int pb0 = (T - 1) - (TQ - 1) + j0;  // first table row of positional block 0 (query tile 0); block u starts 64 u lower
  float* bd = reinterpret_cast<float*>(lds + BNC_OFF + w * BNC);
  char* exw = lds + BNC_OFF + w * BNC;  // this wave's rows of the exchange tiles: dS at +0, Pd at +2048 ([16][64] bf16 each)
  // exchange write: row li, 8-byte piece jt*4 + g4 ; exchange read (transposing, k = 32 rows): rows 32 ks2 + 8 g4 + 4 hh + (li>>2)
  const uint32_t exo = li * 128 + ((((g4 >> 1) ^ swz(li))) << 4) + (g4 & 1) * 8;
  uint32_t exr[2], trk[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int rl = 8 * (g4 & 1) + 4 * hh + (li >> 2);  // row inside a wave's 16
    exr[hh] = (g4 >> 1) * BNC + rl * 128 + (((2 * w + ((li & 3) >> 1)) ^ swz(rl)) << 4) + (li & 1) * 8;
    const int rr = 8 * g4 + 4 * hh + (li >> 2);        // row inside a 32-row half of an image
    trk[hh] = rr * 128 + ((((li & 3) >> 1) ^ swz(rr)) << 4) + (li & 1) * 8;
  }
  const char* exbase = lds + BNC_OFF;
  const uint32_t* bitp = DROP ? reinterpret_cast<const uint32_t*>(a.bits) + (((long)z * nkt + kt) * (nkt * 64) + 16 * w + li) * 2 + (g4 >> 1) : nullptr;
  const int ksh = (g4 & 1) * 16;

  bf16x8_t qv[2], qv_next[2];
  uint32_t piece_next = 0;
  // per-lane operands of query tile i0 (row i = i0 + 16 w + li): Qv fragments and the keep piece into registers; lse and D of
  // the tile's 64 rows into LDS (4-byte global_load_lds by waves 0 / 1): as register loads their destinations got paired
  // with live values in packed-math operands and the compiler waited for everything in flight in mid-tile
  auto row_loads = [&](int i0, int stage) {
    const int i = i0 + 16 * w + li;
    const int ic = min(i, T - 1);
    const long o = (long)ic * a.ldq + g4 * 8;
    qv_next[0] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Qvb + o));
    qv_next[1] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Qvb + o + 32));
    if (DROP) piece_next = bitp[(long)i0 * 2];
    if (w < 2) {
      const float* src = (w == 0 ? a.lse : a.D) + (long)z * T + min(i0 + lane, T - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + AUX_OFF + stage * 512 + w * 256), 4, 0, 0);
    }
  };
  if (nq > 0) {
    issue_img(lds, srcQ, 0, w, lane);
    issue_img(lds + IMG, srcG, 0, w, lane);
    issue_img(lds + PP_OFF, srcP, pb0, w, lane);               // block 0 -> slot 0
    issue_img(lds + PP_OFF + 2 * IMG, srcP, pb0 + 64, w, lane);  // block -1 -> slot 2
    row_loads(0, 0);
  }
  int slot_lo = 0;  // ring slot of block `it` (window rows 0..63); block it-1 (rows 64..127) sits one slot below
  PROF_DECL
  for (int it = 0; it < nq; ++it) {
    const int i0 = it * TQ;
    PROF_MARK(9)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile `it` landed; everyone is done with the exchange tiles of tile it-1
    PROF_MARK(0)
    const int slot_hi = slot_lo == 0 ? 2 : slot_lo - 1;
    const int slot_nx = slot_lo == 2 ? 0 : slot_lo + 1;
    qv[0] = qv_next[0];
    qv[1] = qv_next[1];
    const float* aux = reinterpret_cast<const float*>(lds + AUX_OFF + (it & 1) * 512) + 16 * w + li;
    const float lse_raw = aux[0], Di = aux[64];
    uint32_t piece = piece_next;
    {  // pin the copies (and the compiler's waits for these loads) HERE, ahead of the next tile's prefetch
      uint4 q0 = __builtin_bit_cast(uint4, qv[0]), q1 = __builtin_bit_cast(uint4, qv[1]);
      asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w), "+v"(piece));
      qv[0] = __builtin_bit_cast(bf16x8_t, q0);
      qv[1] = __builtin_bit_cast(bf16x8_t, q1);
    }
    const float lse2 = (i0 + 16 * w + li < T) ? lse_raw * LOG2E : INFINITY;
    if (it + 1 < nq) {
      row_loads(i0 + TQ, (it + 1) & 1);  // register-destination loads first: a later wait for one of them is `vmcnt(6)`, not a wait for the prefetch
      char* st = lds + ((it + 1) & 1) * 2 * IMG;
      issue_img(st, srcQ, i0 + TQ, w, lane);
      issue_img(st + IMG, srcG, i0 + TQ, w, lane);
      issue_img(lds + PP_OFF + slot_nx * IMG, srcP, pb0 - 64 * (it + 1), w, lane);
    }
    PROF_MARK(1)
    const char* sQ = lds + (it & 1) * 2 * IMG;
    const char* sG = sQ + IMG;
    // phase 1: this wave's 16 rows x 64 keys
    bf16x8_t qu[2], dO[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qu[ks] = ldf(sQ + w * 2048 + L.offk[ks]);
      dO[ks] = ldf(sG + w * 2048 + L.offk[ks]);
    }
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
        acc_s[jt] = mfma32(kf[jt][ks], qu[ks], acc_s[jt]);
        acc_dp[jt] = mfma32(vf[jt][ks], dO[ks], acc_dp[jt]);
      }
    PROF_MARK(2)
    add_band(acc_s, qv, lds + PP_OFF + slot_lo * IMG, lds + PP_OFF + slot_hi * IMG, bd, L);
    PROF_MARK(3)
    bf16x4_t dsb[4], pdb[4];
    softmax_bwd_tile<DROP, true>(acc_s, acc_dp, dsb, pdb, lse2, Di, a.inv_keep, piece, ksh, kl - j0, g4);
    PROF_MARK(4)
    wave_lds_sync();  // skew reads done before the exchange tiles overwrite the buffer
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      *reinterpret_cast<bf16x4_t*>(exw + (exo ^ (jt << 5))) = dsb[jt];
      *reinterpret_cast<bf16x4_t*>(exw + 2048 + (exo ^ (jt << 5))) = pdb[jt];
    }
    PROF_MARK(5)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // not __syncthreads(): that also waits for the prefetch (vmcnt)
    __builtin_amdgcn_s_barrier();
    PROF_MARK(6)
    // phase 2: this wave's 16 keys, all 64 rows:  dK^T[d][j] += sum_i Qu^T[d][i] dS[i][j],  dV^T[d][j] += sum_i dO^T[d][i] Pd[i][j]
    {
      const uint32_t eb = lds_addr(exbase);
      const uint32_t ead[2] = {eb + exr[0], eb + exr[1]};
      uint2 bx[8];  // bx[hh*4 + q]: q = 0 dS rows 0..31, 1 Pd rows 0..31, 2 dS rows 32..63, 3 Pd rows 32..63
      trr8<0, 2048, 2 * BNC, 2 * BNC + 2048>(ead, bx);
      const uint32_t qb = lds_addr(sQ);
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // two 16-column tiles of d per batch of reads
        const int d0 = half * 2;
        const uint32_t aad[4] = {qb + (trk[0] ^ (uint32_t)(d0 << 5)), qb + (trk[1] ^ (uint32_t)(d0 << 5)),
                                 qb + (trk[0] ^ (uint32_t)((d0 + 1) << 5)), qb + (trk[1] ^ (uint32_t)((d0 + 1) << 5))};
        uint2 ax[16];  // ax[(dd*2 + hh)*4 + q]: q = 0 Qu rows 0..31, 1 Qu rows 32..63, 2 dO rows 0..31, 3 dO rows 32..63
        trr16<0, 4096, IMG, IMG + 4096>(aad, ax);
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
          for (int ks2 = 0; ks2 < 2; ++ks2) {
            const bf16x8_t aq = cat2(ax[(dd * 2 + 0) * 4 + ks2], ax[(dd * 2 + 1) * 4 + ks2]);
            const bf16x8_t ag = cat2(ax[(dd * 2 + 0) * 4 + 2 + ks2], ax[(dd * 2 + 1) * 4 + 2 + ks2]);
            acc_dk[d0 + dd] = mfma32(aq, cat2(bx[0 * 4 + 2 * ks2], bx[1 * 4 + 2 * ks2]), acc_dk[d0 + dd]);
            acc_dv[d0 + dd] = mfma32(ag, cat2(bx[0 * 4 + 2 * ks2 + 1], bx[1 * 4 + 2 * ks2 + 1]), acc_dv[d0 + dd]);
          }
      }
    }
    slot_lo = slot_nx;
    PROF_MARK(7)
  }
  const int j = j0 + 16 * w + li;
  if (j < T) {
    bf16_t* ok = a.dk + ((long)b * T + j) * a.lddkv + h * DH;
    bf16_t* ov = a.dv + ((long)b * T + j) * a.lddkv + h * DH;
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
  PROF_MARK(8)
  PROF_DUMP(1)
}

}  // namespace

// EA_FLASH_V1=1 (or ea_set_flash_relpos(0)): the general kernels of flash_attention.hip serve the encoder too (A/B runs, tests)
static int g_rp_on = [] { const char* e = getenv("EA_FLASH_V1"); return e && e[0] == '1' ? 0 : 1; }();
extern "C" int ea_set_flash_relpos(int on) {
  const int prev = g_rp_on;
  g_rp_on = on ? 1 : 0;
  return prev;
}
bool ea_rp_eligible(bool relpos, int T, int S, int causal, uint32_t thr, const void* bits) {
  return g_rp_on && relpos && T == S && !(causal & 1) && (thr == 0 || bits != nullptr);
}

int ea_rp_keep_bits(uint16_t* bits, int H, int B, int T, uint64_t seed, uint32_t thr, hipStream_t stream) {
  const int nkt = (T + TK - 1) / TK;
  const long n = (long)H * B * nkt * (nkt * 64) * 4;
  static const long cap = [] { const char* e = getenv("EA_KEEP_BITS_WGS"); return e ? atol(e) : 256L; }();  // (diagnostic override)
  long blocks = (n + 255) / 256;
  if (cap > 0 && blocks > cap) blocks = cap;
  hipLaunchKernelGGL(keep_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, bits, seed, thr, H * B, T, nkt);
  return EA_CHECK_LAUNCH();
}

int ea_rp_fwd(const FlashFwdArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.nq * a.H * a.B));
  if (a.thr) hipLaunchKernelGGL(rp_fwd_kernel<true>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(rp_fwd_kernel<false>, grid, dim3(256), 0, stream, a);
  return EA_CHECK_LAUNCH();
}

int ea_rp_bwd(const FlashBwdArgs& a_in, hipStream_t stream) {
  static const int dbg = [] { const char* e = getenv("EA_RP_DBG"); return e ? atoi(e) : 0; }();
  FlashBwdArgs a = a_in;
  a.dbg = dbg;
  a.prof = nullptr;
#ifdef RP_PROF
  static unsigned long long* prof_buf = nullptr;
  static const int prof_on = [] { const char* e = getenv("EA_RP_PROF"); return e && e[0] == '1' ? 1 : 0; }();
  if (prof_on) {
    if (!prof_buf) hipMalloc(&prof_buf, 2 * 8192 * 10 * sizeof(unsigned long long));
    hipMemsetAsync(prof_buf, 0, 2 * 8192 * 10 * sizeof(unsigned long long), stream);
    a.prof = prof_buf;
  }
#endif
  const dim3 gq((unsigned)(a.nq * a.H * a.B)), gk((unsigned)(a.nk * a.H * a.B));
  if (a.thr) {
    hipLaunchKernelGGL(rp_bwd_q_kernel<true>, gq, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(rp_bwd_kv_kernel<true>, gk, dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL(rp_bwd_q_kernel<false>, gq, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(rp_bwd_kv_kernel<false>, gk, dim3(256), 0, stream, a);
  }
#ifdef RP_PROF
  if (a.prof) {
    static int calls = 0;
    if (++calls == 20) {  // one report, from a warm call
      hipStreamSynchronize(stream);
      static unsigned long long h[2 * 8192 * 10];
      hipMemcpy(h, a.prof, sizeof(h), hipMemcpyDeviceToHost);
      const char* names[2] = {"Q ", "KV"};
      const int nw[2] = {(int)gq.x * 4, (int)gk.x * 4};
      for (int k = 0; k < 2; ++k) {
        double sum[10] = {0};
        int cnt = 0;
        for (int wv = 0; wv < nw[k] && wv < 8192; ++wv) {
          const unsigned long long* o = h + ((long)k * 8192 + wv) * 10;
          unsigned long long tot = 0;
          for (int p = 0; p < 10; ++p) tot += o[p];
          if (!tot) continue;
          ++cnt;
          for (int p = 0; p < 10; ++p) sum[p] += (double)o[p];
        }
        fprintf(stderr, "[rp prof] %s waves %d, mean s_memtime ticks per wave by phase:", names[k], cnt);
        for (int p = 0; p < 10; ++p) fprintf(stderr, " p%d=%.0f", p, cnt ? sum[p] / cnt : 0.0);
        fprintf(stderr, "\n");
      }
    }
  }
#endif
  return EA_CHECK_LAUNCH();
}
