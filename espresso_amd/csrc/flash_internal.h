// Shared argument blocks of the fused attention kernels (flash_attention.hip: general kernels; flash_relpos.hip: the
// encoder self-attention hot path, relative positions, T == S, no causal mask).
#pragma once
#include "common.h"

struct FlashFwdArgs {
  const bf16_t* qu; const bf16_t* qv; long ldq;
  const bf16_t* k; const bf16_t* v; long ldkv;
  const bf16_t* pp; long ldpp;
  const int* klen;
  bf16_t* out; long ldo;
  float* lse;
  int H, B, T, S, causal, nq;
  uint64_t seed; uint32_t thr; float inv_keep;
  const uint16_t* bits;  // keep-bit pieces (ea_flash_keep_bits layout) or NULL
};

struct FlashBwdArgs {
  const bf16_t* qu; const bf16_t* qv; long ldq;
  const bf16_t* k; const bf16_t* v; long ldkv;
  const bf16_t* pp; long ldpp;
  const int* klen;
  const bf16_t* out; const bf16_t* dout; long ldo;
  const float* lse;
  float* D;                       // [H*B][T]   (written by the Q kernel, read by the KV kernel)
  bf16_t* t1; bf16_t* t2; long ldt;  // [B*T][ldt] gradients of (q+u), (q+v) (unscaled q space)
  bf16_t* dq; long lddq;             // optional [B*T][lddq]: t1 + t2 summed in fp32 (rel-pos only)
  bf16_t* dBD; int ld_bd;         // [H*B][T][ld_bd]
  bf16_t* dk; bf16_t* dv; long lddkv;
  int H, B, T, S, causal, nq, nk, dbd_prezeroed;
  float scaling;
  uint64_t seed; uint32_t thr; float inv_keep;
  const uint16_t* bits;
  unsigned long long* prof;  // RP_PROF builds: per-wave phase cycle sums
  int dbg;  // diagnostic builds of the rel-pos kernels: EA_RP_DBG bit mask of phases to skip (results are then wrong)
};

// flash_relpos.hip
bool ea_rp_eligible(bool relpos, int T, int S, int causal, uint32_t thr, const void* bits);
int ea_rp_keep_bits(uint16_t* bits, int H, int B, int T, uint64_t seed, uint32_t thr, hipStream_t stream);
int ea_rp_fwd(const FlashFwdArgs& a, hipStream_t stream);
int ea_rp_bwd(const FlashBwdArgs& a, hipStream_t stream);
