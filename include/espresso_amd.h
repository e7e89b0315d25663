/* espresso_amd — C ABI of the MI355X (gfx950) hot path of freewym/espresso.
 *
 * The reference has no FFI on this path: its boundary is fairseq's Python plugin registry
 * (SURVEY.md §8b), and all compute below it is ATen/cuDNN/torchaudio calls.  This header is the
 * boundary this framework introduces UNDER that registry: one entry point per reference
 * function on the hot path, plain device pointers + sizes + a HIP stream, no torch types.  The
 * host-side mirror of the reference interface (espresso_amd/{data,modules,models,criterions,
 * tasks,tools}) binds these through ctypes (espresso_amd/_lib.py); INTEGRATION.md shows the stub
 * a reference maintainer would add.  Each declaration cites the reference code it replaces
 * (paths relative to the reference root).
 *
 * Conventions: every function returns 0 on success, a negative value on a launch/argument
 * error; all pointers are DEVICE pointers unless the name ends in _host; `stream` is a
 * hipStream_t (0 = default stream); kernels are asynchronous with respect to the host.
 * bf16 tensors are raw uint16 bit patterns.  "TBC" = (time, batch, channel) row-major.
 */
#ifndef ESPRESSO_AMD_H
#define ESPRESSO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ea_stream_t;

int ea_version(void);

/* ------------------------------------------------------------------------------------------
 * Dense products (MFMA).  Replaces torch.nn.functional.linear / torch.bmm / conv1d(k=1) /
 * im2col-conv2d calls at: fairseq/modules/conformer_layer.py:79-101,134-146;
 * fairseq/modules/multihead_attention.py:650-688,788-831,884-907;
 * espresso/models/transformer/speech_transformer_encoder.py:341-343;
 * espresso/models/transformer/speech_transformer_encoder_model.py:207-208;
 * espresso/modules/speech_convolutions.py:78-102.
 *
 *   C[z][m][n] = epi( alpha * sum_k A(z;m,k) * B(z;n,k) ),  z in [0,batch)
 *   A(z;m,k) = A[(z/zdiv)*sA_hi + (z%zdiv)*sA_lo + (a_kstrided ? k*lda + m : m*lda + k)]  (bf16)
 *   B likewise with n.                                                                    (bf16)
 *   epi(v):  v += bias[n]
 *            if aux:  v *= keep(drop) * act'(aux[m][n])                 (backward through act)
 *            elif C2: C = bf16(v);  C2 = bf16(keep(drop) * act(v));     (two outputs) done
 *            else:    v = keep(drop) * act(v)
 *            v = v * out_scale + resid[m][n]
 *            C = c_f32 ? (accumulate ? C + v : v) : bf16(v)
 *   splitk > 1: two-pass split-K (fp32 slabs + reduce) for wgrad: long reduction, few output tiles.
 */
enum { EA_ACT_NONE = 0, EA_ACT_RELU = 1, EA_ACT_SILU = 2 };

typedef struct EaGemmParams {
  const void* A;
  const void* B;
  void* C;
  void* C2;           /* optional bf16 second output, leading dim ldc2, same batch offsets as C */
  const float* bias;  /* [N] fp32 or NULL */
  const void* resid;  /* bf16 (or fp32 if resid_f32) [M][ldr] or NULL */
  const void* aux;    /* bf16 pre-activation [M][ldaux] or NULL */
  int M, N, K, batch, zdiv;
  int a_kstrided, b_kstrided, c_f32, accumulate, resid_f32, act;
  /* Row pitches in elements.  CONTRACT for ldc > N (a padded output pitch): the columns N .. ldc - 1 of a row are PADDING that
   * the library may overwrite with unspecified finite-or-not values (the 8-wave kernels store whole 32-column groups when
   * N % 4 == 0 and ldc >= roundup32(N)); a caller that later reduces over the pitch instead of over N must zero the pad itself
   * (as ea_rnnt_grad does for the joint's gradient). */
  long lda, ldb, ldc, ldc2, ldr, ldaux;
  long sA_hi, sA_lo, sB_hi, sB_lo, sC_hi, sC_lo, sR_hi, sR_lo, sX_hi, sX_lo;
  float alpha, out_scale;
  /* dropout: element dropped when hash(seed, z*M*N + m*N + n) < drop_thr; kept * drop_scale */
  uint64_t drop_seed;
  uint32_t drop_thr;
  float drop_scale;
  /* split-K: the k range is cut into `splitk` chunks reduced by different workgroups; the fp32 partial
   * slabs go to `workspace` (ea_gemm_splitk_workspace_bytes) and a second kernel sums them into C
   * (honouring `accumulate`).  Requires c_f32 and no other epilogue.  kchunk is filled in by the library. */
  int splitk, kchunk;
  void* workspace;
  /* Relative-position query preparation fused into the QKV projection (fairseq/modules/multihead_attention.py:679-688:
   * q_with_bias_u = q + pos_bias_u, q_with_bias_v = q + pos_bias_v, later scaled by head_dim ** -0.5).  When q_u != NULL,
   * columns n < qsplit_n of the product are NOT written to C; the rounded bf16 value q = bf16(alpha * acc + bias[n]) is formed
   * as usual and  q_u[m][n] = bf16((q + pos_u[n]) * qscale),  q_v[m][n] = bf16((q + pos_v[n]) * qscale)  go to two bf16
   * [M][ld_q] buffers instead (pos_u / pos_v NULL: zeros; q_v NULL: q_u only) — exactly ea_relpos_q_prep applied to those
   * columns.  Requires qsplit_n % 128 == 0, N % 8 == 0, batch == 1, no split-K; the other columns take the normal epilogue. */
  void* q_u;
  void* q_v;
  const float* pos_u;
  const float* pos_v;
  long ld_q;
  int qsplit_n;
  float qscale;
} EaGemmParams;

int ea_gemm_bf16(const EaGemmParams* p, ea_stream_t stream);
long ea_gemm_splitk_workspace_bytes(int M, int N, int batch, int splitk);
/* tuning hook: 0 = automatic tile height (64-row tiles when 128x128 tiles under-fill the chip), 1 = always 128,
 * 2 = always 64; returns the previous value. */
int ea_set_gemm_variant(int v);
/* live profiling of ea_gemm_bf16 for roofline reports: enable(1) clears and starts recording one HIP-event
 * pair per launch on the launch stream; read() synchronises and returns launches, summed ms and flops. */
/* tuning hook: XCD-aware workgroup -> tile mapping; mask bit 0 = direct-to-LDS kernel, bit 1 = register-staged kernel (both
 * additionally gated on the grid shape); returns the previous mask */
int ea_set_gemm_xcd_swizzle(int mask);
/* tuning hook: direct-to-LDS ring kernel for launches whose operands are both k-contiguous (0 = off, 1 = automatic ring depth, 2..4 = forced number of stages); returns the previous value */
int ea_set_gemm_glds(int stages);
/* bf16 GEMM outputs of at least `bytes` are written with non-temporal stores (default 0 = never: measured neutral on the training step); returns the old value */
long ea_set_gemm_nt_store_min_bytes(long bytes);
/* tuning hook: 8-wavefront large-tile kernels (csrc/gemm_w8.hip) for launches with both operands k-contiguous and the lean bf16
 * epilogue: 0 = never, 1 = automatic (default: a 256 x 256 or 128 x 128 tile per CU when that grid fills the chip), 2..6 = forced
 * tile configuration (diagnostic); returns the previous value */
int ea_set_gemm_w8(int mode);
int ea_gemm_profile_enable(int on);
long ea_gemm_profile_read(double* total_ms, double* total_flops);
/* algorithmic HBM bytes of the recorded launches (operands read once, outputs written once) */
int ea_gemm_profile_bytes(double* total_bytes);
/* writes one line per recorded launch ("M N K batch a_ks b_ks splitk bm64 epilogue-bits ms"); returns the count or -1 */
long ea_gemm_profile_dump(const char* path);

/* Grouped weight-gradient GEMM: ONE launch computes, for every problem i,
 *   dW_i[n][k] += sum_m dy_i[m][n] * x_i[m][k]      (fp32, leading dim ldw)
 *   db_i[n]    += sum_m dy_i[m][n]                  (when dbias != NULL)
 * i.e. torch.nn.Linear's weight / bias gradients (torch autograd of F.linear as called by
 * fairseq/modules/conformer_layer.py:134-146, multihead_attention.py:650-688, the pointwise convolutions :79-101) for all the
 * Linear layers of one encoder layer at once: no split-K slabs, no reduce pass, no separate column-sum launches. */
#define EA_WGRAD_MAX 16
typedef struct EaWgradProblem {
  const void* dy; /* bf16 [M][ld_dy], columns 0..N-1 */
  const void* x;  /* bf16 [M][ld_x], columns 0..K-1 */
  float* dW;      /* fp32 [N][ldw] */
  float* dbias;   /* fp32 [N] or NULL */
  int M, N, K;
  long ld_dy, ld_x, ldw;
} EaWgradProblem;
typedef struct EaWgradGroup {
  int count;
  EaWgradProblem p[EA_WGRAD_MAX];
} EaWgradGroup;
int ea_wgrad_group(const EaWgradGroup* group, ea_stream_t stream);
/* tuning hook: 1 (default) = groups whose operand rows are 16-byte aligned and cover whole tiles (ld_dy >= N rounded up to the
 * tile height, ld_x >= K rounded up to 128) take the direct-to-LDS kernel with transposing fragment reads, 0 = always the
 * register-staged kernel; returns the previous value */
int ea_set_wgrad_transposing_reads(int on);
/* tuning hook for the 8-wavefront 256 x 256 kernel (csrc/wgrad_w8.hip: groups of long reductions whose 256 x 256 tile grid fills
 * the chip — the transducer joint's output layer cut into row slabs): 0 = never, 1 (default) = automatic, 2 = every group whose
 * operands are 16-byte aligned; returns the previous value.  Results are bit-identical to the 4-wave kernels. */
int ea_set_wgrad_w8(int mode);

/* ------------------------------------------------------------------------------------------
 * LayerNorm (eps, affine) — fairseq/modules/layer_norm.py:28-33; with optional fused
 * "dropout then zero padded rows" of espresso/models/transformer/speech_transformer_encoder.py:348-357.
 * x,y,dy,dx: bf16 [M][C]; gamma,beta,mean,rstd,dgamma,dbeta: fp32.  row_zero: uint8 [M] or NULL.
 * bwd accumulates (+=) into dgamma/dbeta; dx_add (bf16 [M][C] or NULL) is added to dx. */
int ea_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                     int M, int C, float eps, const uint8_t* row_zero, uint64_t drop_seed, uint32_t drop_thr,
                     float drop_scale, ea_stream_t stream);
/* fp32 islands of the reference's autocast run (fairseq/tasks/fairseq_task.py:516: LayerNorm outputs are fp32): the same
 * LayerNorm with an fp32 OUTPUT (x stays bf16, a Linear's output), and its backward with an fp32 incoming gradient.  Used by
 * the transducer joint network, which adds and rectifies the two LayerNorm outputs in fp32
 * (espresso/models/transformer/speech_transformer_transducer_base.py:292-294). */
int ea_layernorm_fwd_f32out(const void* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            int M, int C, float eps, const uint8_t* row_zero, uint64_t drop_seed, uint32_t drop_thr,
                            float drop_scale, ea_stream_t stream);
int ea_layernorm_bwd_f32dy(const void* x, const float* dy, const float* gamma, const float* mean, const float* rstd,
                           void* dx, float* dgamma, float* dbeta, int M, int C, const uint8_t* row_zero,
                           uint64_t drop_seed, uint32_t drop_thr, float drop_scale, const void* dx_add,
                           void* workspace, ea_stream_t stream);
/* workspace (ea_layernorm_bwd_workspace_bytes, or NULL): per-block dgamma/dbeta partial rows folded by a second small
 * kernel instead of 2*C global atomics per block. */
long ea_layernorm_bwd_workspace_bytes(int M, int C);
int ea_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd,
                     void* dx, float* dgamma, float* dbeta, int M, int C, const uint8_t* row_zero,
                     uint64_t drop_seed, uint32_t drop_thr, float drop_scale, const void* dx_add,
                     void* workspace, ea_stream_t stream);
/* the two passes of ea_layernorm_bwd separately (workspace required): the parameter-gradient reduce only feeds the optimizer,
 * so a runtime can put it on another stream */
int ea_layernorm_bwd_dx(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                        float* dgamma, float* dbeta, int M, int C, const uint8_t* row_zero, uint64_t drop_seed,
                        uint32_t drop_thr, float drop_scale, const void* dx_add, void* workspace, ea_stream_t stream);
int ea_layernorm_param_reduce(const void* workspace, float* dgamma, float* dbeta, int M, int C, ea_stream_t stream);
/* the same for up to EA_LNRED_MAX LayerNorms in one launch (all LayerNorms of one encoder layer's backward) */
#define EA_LNRED_MAX 8
typedef struct EaLnReduceItem { const void* workspace; float* dgamma; float* dbeta; int M, C; } EaLnReduceItem;
typedef struct EaLnReduceGroup { int count; EaLnReduceItem item[EA_LNRED_MAX]; } EaLnReduceGroup;
int ea_layernorm_param_reduce_group(const EaLnReduceGroup* group, ea_stream_t stream);
/* ea_layernorm_bwd_dx plus a second output out2[i] = a2 * dropout(dx[i]) with its own (seed2, thr2, scale2) mask — exactly
 * ea_scale_dropout_bf16(dx, NULL, out2, M*C, a2, 0, seed2, thr2, scale2) without the extra pass (the next residual block of
 * the backward starts with that product: FairseqDropout backward + the 0.5 FFN scale) */
int ea_layernorm_bwd_dx2(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                         float* dgamma, float* dbeta, int M, int C, const void* dx_add, void* workspace, void* out2, float a2,
                         uint64_t seed2, uint32_t thr2, float scale2, ea_stream_t stream);
/* Two LayerNorms back to back over the same rows, C <= 512: y1 = LN(x; g1, b1) (rounded to bf16: a Conformer layer's output,
 * espresso/modules/conformer_with_relative_positional_embedding_encoder_layer.py:143) and y2 = LN(y1; g2, b2) (the NEXT layer's
 * first operation, fairseq/modules/conformer_layer.py:141) in one launch — same results as two ea_layernorm_fwd calls. */
int ea_layernorm_fwd2(const void* x, const float* g1, const float* b1, void* y1, float* mean1, float* rstd1, const float* g2,
                      const float* b2, void* y2, float* mean2, float* rstd2, int M, int C, float eps, ea_stream_t stream);
/* Backward of that pair in one pass: d = bf16(LNbwd(x1, dy1; gamma1, mean1, rstd1) + dx_add), dx = bf16(LNbwd(x2, d; gamma2, mean2,
 * rstd2)), out2 = a2 * dropout(dx) (NULL: skipped) — the same values as ea_layernorm_bwd_dx followed by ea_layernorm_bwd_dx2, without
 * writing d.  ws1 / ws2 (ea_layernorm_bwd_workspace_bytes each) receive the dgamma / dbeta partials of norm 1 (the later one in the
 * forward) and norm 2: fold them with ea_layernorm_param_reduce. */
int ea_layernorm_bwd2_dx(const void* x1, const void* dy1, const float* gamma1, const float* mean1, const float* rstd1,
                         const void* dx_add, void* ws1, const void* x2, const float* gamma2, const float* mean2, const float* rstd2,
                         void* ws2, void* dx, int M, int C, void* out2, float a2, uint64_t seed2, uint32_t thr2, float scale2,
                         ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Streaming helpers (AMP weight cast fairseq/tasks/fairseq_task.py:516; FairseqDropout backward
 * fairseq/modules/fairseq_dropout.py:23-25; nn.Linear bias gradient). */
int ea_cast_f32_to_bf16(const float* src, void* dst, long n, ea_stream_t stream);
/* n <= 8 bf16 matrices in one launch: dst[i][c][r] = src[i][r][c] (rows[i] x cols[i], dense).  Used for the k-contiguous
 * copies of the Linear weights that the backward data-gradient GEMMs read (nn.Linear backward, x_grad = y_grad @ W). */
int ea_transpose_bf16_batch(const void* const* src, void* const* dst, const int* rows, const int* cols, int n, ea_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * Waveform ingestion (host code: no device work, no stream) for the `wave` entries of the data json —
 * fairseq/data/audio/audio_utils.py:74-118 get_waveform (soundfile: PCM WAV and FLAC, mono = channel 0 as
 * espresso/tools/utils.py:438-440, normalization=False -> int16 scale) called per utterance by
 * espresso/data/feat_text_dataset.py:128-155 from `dataset.num_workers` DataLoader worker processes.  Files are decoded to int16
 * straight into the caller's (pinned) staging buffer; the batch call decodes many files on `num_threads` host threads without
 * touching Python objects.  Error codes: -1 unreadable, -3 not a WAV / FLAC file, -4 unsupported sample format, -5 capacity too
 * small, -6 malformed FLAC stream, -7 FLAC frame checksum mismatch. */
int ea_audio_probe(const char* path, long* num_samples, int* sample_rate, int* channels, int* bits);
long ea_audio_read_i16(const char* path, int16_t* dst, long capacity, int* sample_rate);
/* 1: the MD5 of the decoded audio equals the signature in the FLAC STREAMINFO block (or none is recorded; WAV: always 1), 0: differs */
int ea_audio_verify(const char* path);
/* file i -> dst[offsets[i] .. offsets[i+1]); lengths[i] = samples decoded (or the file's negative error); returns the failures */
int ea_audio_read_batch_i16(const char* const* paths, int n, int16_t* dst, const long* offsets, int num_threads, long* lengths,
                            int* sample_rates);

/* fp32 [M][ld_src] -> bf16 [M][ld_dst], N valid columns per row, columns N .. ld_dst-1 zero-filled (M <= 65535): the fp32
 * gradient of the vocabulary logits re-pitched for the bf16 weight / data gradient GEMMs of the output layer
 * (espresso/models/transformer/speech_transformer_encoder_model.py:207-208 fc_out, torch autograd of F.linear). */
int ea_cast_f32_to_bf16_rows(const float* src, long ld_src, void* dst, long ld_dst, long M, int N, ea_stream_t stream);
int ea_cast_bf16_to_f32(const void* src, float* dst, long n, ea_stream_t stream);
/* out = a*x*keep(idx) + b*y  (bf16; y may be NULL) */
int ea_scale_dropout_bf16(const void* x, const void* y, void* out, long n, float a, float b, uint64_t seed,
                          uint32_t thr, float inv_keep, ea_stream_t stream);
/* out[n] += sum_m X[m*ld+n]  (X bf16, out fp32) */
int ea_colsum_bf16(const void* X, float* out, int M, int N, long ld, ea_stream_t stream);
/* Token + positional embedding of the decoder — fairseq/models/transformer/transformer_decoder.py:296-330
 * (embed_scale * embed_tokens(prev_output_tokens) + embed_positions); bwd accumulates into dW (pad row skipped). */
int ea_embedding_fwd(const int* tokens, const int* positions, const float* W, const float* pos_table, void* out, int M,
                     int C, float scale, ea_stream_t stream);
int ea_embedding_bwd(const int* tokens, const void* dy, float* dW, int M, int C, float scale, int pad_idx,
                     ea_stream_t stream);
int ea_zero_rows_bf16(void* x, const uint8_t* row_zero, int M, int C, ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused attention (flash_attention.hip): scores + rel-pos skew + key-padding / causal mask + fp32 online softmax +
 * attention dropout + P.V in one kernel; replaces the bmm / as_strided / softmax / dropout / bmm sequence of
 * fairseq/modules/multihead_attention.py:788-907 without materialising the (B*H, T, S) tensors.
 *   qu, qv : bf16 [B*T][ldq], (q + pos_bias_u) * scaling and (q + pos_bias_v) * scaling (ea_relpos_q_prep); qv = NULL
 *            selects plain attention (absolute positions, decoder self-attention, cross-attention).
 *   k, v   : bf16, row (b*S + j) * ldkv, head h at column h*dh of the given base pointers
 *   pp     : bf16 [2T-1][ldpp] projected sinusoidal table (row r <-> relative position, used column r = T-1-i+j)
 *   key_len: int [B] valid keys per sentence or NULL; causal: mask j > i + (S - T)
 *   out    : bf16 [B*T][ldo] (head h at column h*dh); lse: fp32 [H*B][T] row logsumexp (NULL to skip)
 * dropout index of element (z = h*B + b, i, j) is (z*T + i)*S + j, identical to ea_relpos_softmax_fwd.
 * ea_flash_attention_supported: dh == 64 (and T == S for rel-pos); other shapes use the unfused kernels. */
int ea_flash_attention_supported(int dh, int T, int S, int relpos);
int ea_flash_attention_fwd(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldkv,
                           const void* pp, long ldpp, const int* key_len, void* out, long ldo, float* lse, int H, int B,
                           int T, int S, int dh, int causal, uint64_t drop_seed, uint32_t drop_thr, float drop_scale,
                           void* keep_bits, ea_stream_t stream);
/* Attention-dropout keep decisions as bits (flash_relpos.hip): the rel-pos encoder kernels (qv != NULL, T == S, no causal
 * mask) read 16 keep bits per lane and key tile instead of hashing every element in each of the three kernels.  The
 * decisions are the same ea_keep(seed, (z*T + i)*S + j) as everywhere else.  keep_bits: ea_flash_keep_bits_bytes(H, B, T)
 * bytes of device memory; ea_flash_attention_fwd fills it (or expects it filled by ea_flash_keep_bits when bit 2 of
 * `causal` is set) and ea_flash_attention_bwd reads it.  keep_bits == NULL with dropout on selects the general kernels. */
long ea_flash_keep_bits_bytes(int H, int B, int T);
/* 0: the general kernels also serve the rel-pos encoder attention (A/B measurements, tests); returns the previous value */
int ea_set_flash_relpos(int on);
int ea_flash_keep_bits(void* keep_bits, int H, int B, int T, uint64_t drop_seed, uint32_t drop_thr, ea_stream_t stream);
/* Backward of ea_flash_attention_fwd (probabilities recomputed from lse; same dropout stream).
 *   out, dout : bf16 [B*T][ldo] forward output and its gradient;  D: fp32 [H*B][T] scratch (row dot dO.O)
 *   t1, t2    : bf16 [B*T][ldt], scaling * dL/dqu and scaling * dL/dqv, i.e. the gradients w.r.t. (q + pos_bias_u) and
 *               (q + pos_bias_v) before scaling (t2 / dBD / pp unused when qv == NULL)
 *   dBD       : bf16 [H*B][T][ld_bd] gradient of the raw positional logits (un-skewed, zero outside the band), the
 *               operand of the pos_proj weight gradient dpp[r] = sum_{b,i} dBD[.,i,r] qv[b,i]
 *   dk, dv    : bf16, row (b*S + j) * lddkv, head h at column h*dh (e.g. the k / v thirds of a packed dqkv buffer)  * ea_flash_attention_bwd `causal`: bit 0 = causal mask; bit 1 = the caller guarantees that the part of every dBD row outside
 * the band [T-1-i, T-1-i+S) is already zero (e.g. the same buffer was filled by a previous call with the same H, B, T, S and not
 * touched since), so the kernel only writes the band.
 *   dq        : optional (NULL, or bf16 [B*T][lddq], head h at column h*dh; qv != NULL only): t1 + t2 with the sum formed in
 *               fp32 — the gradient of the query projection's output, e.g. the q third of a packed dqkv buffer (without the
 *               positional term t1 itself is that gradient) */
int ea_flash_attention_bwd(const void* qu, const void* qv, long ldq, const void* k, const void* v, long ldkv,
                           const void* pp, long ldpp, const int* key_len, const void* out, const void* dout, long ldo,
                           const float* lse, float* D, void* t1, void* t2, long ldt, void* dBD, int ld_bd, void* dk,
                           void* dv, long lddkv, int H, int B, int T, int S, int dh, int causal, float scaling,
                           uint64_t drop_seed, uint32_t drop_thr, float drop_scale, const void* keep_bits,
                           void* dq, long lddq, ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Relative-position attention glue — fairseq/modules/multihead_attention.py:679-688 (q+u, q+v,
 * scaling), :788-831 (skew), :835-874 (masks, fp32 softmax, dropout).  Score tensors are
 * [H][B][T][ld] fp32 (ac: ld_ac, bd: ld_bd >= T+S-1); probabilities bf16 [H][B][T][ld_p].
 * bd == NULL: plain attention (decoder self/cross attention); causal != 0 adds the future mask
 * (fairseq/models/transformer/transformer_decoder.py:386-398). */
int ea_relpos_q_prep(const void* qkv, long ldq, const float* u, const float* v, void* qu, void* qv, int M,
                     int C, float scaling, ea_stream_t stream);
int ea_relpos_softmax_fwd(const float* ac, const float* bd, const int* key_len, const float* attn_mask, void* P,
                          void* Pd, int H, int B, int T, int S, int ld_ac, int ld_bd, int ld_p, int causal,
                          uint64_t drop_seed, uint32_t drop_thr, float drop_scale, ea_stream_t stream);
int ea_relpos_softmax_bwd(const void* P, const float* dPd, void* dAC, void* dBD, int H, int B, int T, int S,
                          int ld_p, int ld_dp, int ld_bd, uint64_t drop_seed, uint32_t drop_thr,
                          float drop_scale, ea_stream_t stream);
int ea_add2_strided_bf16(const void* a, long lda, const void* b, long ldb, void* out, long ldo, int M, int C,
                         ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Conformer convolution module middle — fairseq/modules/conformer_layer.py:79-101:
 * GLU -> depthwise Conv1d (k in {3,7,15,31}, pad (k-1)/2, no bias) -> BatchNorm1d -> SiLU.
 * Y: bf16 [B*T][2C] (pointwise_conv1 output); U,Z,H: bf16 [B*T][C]; w: fp32 [C][KW];
 * stats: fp64 [2][C] (sum, sum of squares) zeroed by the caller — double accumulators make the atomics' order invisible
 * in the fp32 statistics (reproducible forward); red: fp32 [2][C] zeroed by the caller; mean_rstd: fp32 [2][C].
 * ea_bn_act_* are also used for BatchNorm2d+ReLU of the sub-sampler (act = EA_ACT_RELU). */
int ea_glu_dwconv_fwd(const void* Y, const float* w, void* U, void* Z, double* stats, int B, int T, int C, int KW,
                      ea_stream_t stream);
int ea_bn_finalize(const double* stats, float* mean_rstd, float* running_mean, float* running_var, int C, float n,
                   float eps, float momentum, ea_stream_t stream);
int ea_bn_from_running(const float* running_mean, const float* running_var, float* mean_rstd, int C, float eps,
                       ea_stream_t stream);
int ea_bn_act_fwd(const void* Z, const float* mean_rstd, const float* gamma, const float* beta, void* H, long M,
                  int C, int act, ea_stream_t stream);
int ea_bn_act_bwd(const void* Z, const void* dH, const float* mean_rstd, const float* gamma, const float* beta,
                  float* red, void* dZ, float* dgamma, float* dbeta, long M, int C, int act, int training,
                  ea_stream_t stream);
/* dw += ... ; wgrad_ws: ea_dwconv_wgrad_workspace_bytes(B,T,C,KW) bytes of scratch (per-block partial slabs) */
long ea_dwconv_wgrad_workspace_bytes(int B, int T, int C, int KW);
int ea_glu_dwconv_bwd(const void* dZ, const void* Y, const void* U, const float* w, void* dY, float* dw,
                      void* wgrad_ws, int B, int T, int C, int KW, ea_stream_t stream);
/* ea_glu_dwconv_bwd with dw == NULL computes dY only; the depthwise weight gradient (optimizer-only) on its own: */
int ea_dwconv_bwd_weight(const void* dZ, const void* U, float* dw, void* wgrad_ws, int B, int T, int C, int KW, ea_stream_t stream);
/* Training-mode BatchNorm + activation forward in one launch (ea_bn_finalize + ea_bn_act_fwd): mean / rstd come straight from
 * the fp64 batch sums `stats` ([2][C]: sum, sum of squares over n rows), are recorded in mean_rstd for the backward pass, the
 * running statistics are updated (torch.nn.BatchNorm1d: momentum, unbiased variance), and zero_next (fp64 [zero_n] or NULL) is
 * cleared — the statistics buffer of the next call, so that no fill launch precedes the kernel that accumulates into it. */
int ea_bn_act_fwd_train(const void* Z, const double* stats, float* mean_rstd, float* running_mean, float* running_var,
                        const float* gamma, const float* beta, void* H, long M, int C, int act, float n, float eps,
                        float momentum, double* zero_next, int zero_n, ea_stream_t stream);
/* ea_bn_act_bwd with the parameter gradients added by the apply kernel itself (no third launch) and zero_next (fp32 [zero_n]
 * or NULL) cleared for the next call's `red`. */
int ea_bn_act_bwd_fused(const void* Z, const void* dH, const float* mean_rstd, const float* gamma, const float* beta,
                        float* red, void* dZ, float* dgamma, float* dbeta, long M, int C, int act, int training,
                        float* zero_next, int zero_n, ea_stream_t stream);
/* ... and the same followed by ea_glu_dwconv_bwd's data gradient (dY [B*T][2C] from dZ, Y and the depthwise filter w), with the
 * BatchNorm backward's second pass folded into the convolution kernel's tile staging: the BatchNorm reduce + ONE launch instead of
 * reduce + apply + convolution (the Conformer convolution module's backward, fairseq/modules/conformer_layer.py:79-101).  dZ is
 * written as before (ea_dwconv_bwd_weight reads it). */
int ea_bn_glu_dwconv_bwd_fused(const void* Z, const void* dH, const float* mean_rstd, const float* gamma, const float* beta, float* red,
                               void* dZ, float* dgamma, float* dbeta, int act, int training, float* zero_next, int zero_n,
                               const void* Y, const float* w, void* dY, int B, int T, int C, int KW, ea_stream_t stream);
/* ea_bn_act_bwd with dgamma == dbeta == NULL skips the parameter gradients; they are then taken from `red` by: */
int ea_bn_param_grad(const float* red, float* dgamma, float* dbeta, int C, ea_stream_t stream);
/* First sub-sampler layer (espresso/modules/speech_convolutions.py:78-102, layer 0): BatchNorm (+ activation) backward and the
 * 3x3 / 1-input-channel convolution's weight + bias gradient in one pass over Z and dH — dZ is formed in registers, never stored
 * (the first layer needs no data gradient).  X fp32 [B][T][F] (the feature matrix), Z / dH bf16 [B][To][Fo][CO], CO % 64 == 0;
 * red fp32 [2 CO] zeroed by the caller (receives BatchNorm's sums); dW fp32 [CO][3][3] and dbias fp32 [CO] (may be NULL)
 * accumulated; dgamma / dbeta accumulated when non-NULL. */
int ea_conv1_bn_bwd(const float* X, const void* Z, const void* dH, const float* mean_rstd, const float* gamma, const float* beta,
                    float* red, float* dgamma, float* dbeta, float* dW, float* dbias, int B, int T, int F, int CO, int sy, int sx,
                    int act, int training, ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Conv2d sub-sampler — espresso/modules/speech_convolutions.py:78-102.  Channels-last bf16
 * activations [B][T][F][C]; first conv (C_in=1) direct from fp32 features [B][T][F]; later convs
 * via im2col (k = (ky*3+kx)*C + c) + ea_gemm_bf16.  Output sizes: To=(T-1)/sy+1, Fo=(F-1)/sx+1. */
int ea_conv1_fwd(const float* X, const float* W, const float* bias, void* Z, double* stats, int B, int T, int F,
                 int CO, int sy, int sx, ea_stream_t stream);
int ea_conv1_wgrad(const float* X, const void* dZ, float* dW, float* dbias, int B, int T, int F, int CO, int sy,
                   int sx, ea_stream_t stream);
/* Implicit-GEMM 3x3 convolutions (csrc/conv_igemm.hip; espresso/modules/speech_convolutions.py:78-102 and its autograd), channels-last
 * bf16 [B][T][F][C], padding 1, strides 1 or 2, Cin a multiple of 64, Cout 64 or 128 — no im2col / col2im buffers.
 *   fwd:   Z[B][To][Fo][Cout] = conv(X; W[Cout][3][3][Cin]) + bias ; stats (optional, fp64 [2*Cout], accumulated) receives the
 *          per-channel sum / sum of squares of the bf16 outputs (BatchNorm batch statistics)
 *   dgrad: dX[B][T][F][Cin] from dZ[B][To][Fo][Cout] and Wd[Cin][3][3][Cout] (the forward weight with its channel axes swapped)
 *   wgrad: dW[Cout][3][3][Cin] (fp32, +=) and dbias[Cout] (fp32, += ; may be NULL) from X and dZ; workspace from
 *          ea_conv3x3_wgrad_workspace_bytes */
int ea_conv3x3_fwd(const void* X, const void* W, const float* bias, void* Z, double* stats, int B, int T, int F, int Cin,
                   int Cout, int sy, int sx, ea_stream_t stream);
int ea_conv3x3_dgrad(const void* dZ, const void* Wd, void* dX, int B, int T, int F, int Cin, int Cout, int sy, int sx,
                     ea_stream_t stream);
long ea_conv3x3_wgrad_workspace_bytes(int B, int T, int F, int Cin, int Cout, int sy, int sx);
int ea_conv3x3_wgrad(const void* X, const void* dZ, float* dW, void* workspace, int B, int T, int F, int Cin, int Cout, int sy,
                     int sx, ea_stream_t stream);
/* ea_conv3x3_wgrad with dW in the parameter's own layout [Cout][Cin][3][3] (torch.nn.Conv2d, +=): written straight into the
 * parameter's gradient, no permuting copy and no accumulate launch afterwards */
int ea_conv3x3_wgrad_param_layout(const void* X, const void* dZ, float* dW, void* workspace, int B, int T, int F, int Cin, int Cout,
                                  int sy, int sx, ea_stream_t stream);
int ea_im2col3x3(const void* A, void* col, int B, int T, int F, int C, int sy, int sx, ea_stream_t stream);
int ea_col2im3x3(const void* dcol, void* dA, int B, int T, int F, int C, int sy, int sx, ea_stream_t stream);
int ea_colstats_bf16(const void* X, double* stats, long M, int C, ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Losses.
 * log-softmax: espresso/models/transformer/speech_transformer_encoder_model.py:141-150.
 * CTC: espresso/criterions/ctc_loss.py:59-100 (F.ctc_loss, blank=<s>, sum, zero_infinity).
 *   lprobs fp32 [B][T][V]; targets int32 [B][Lmax]; nll fp32 [B] (per-utterance -log p);
 *   ea_ctc_grad: dlogits = grad_scale * d(sum nll)/d(logits), bf16 or fp32 [B][T][ld_out], columns
 *   [V, ld_out) zeroed; workspace: ea_ctc_workspace_bytes(B,T,Lmax) bytes, kept between the two calls.
 * Label-smoothed CE: espresso/criterions/label_smoothed_cross_entropy_v2.py:94-119 (uniform).
 *   out_loss[0] += sum loss, out_loss[1] += sum nll over non-pad rows. */
int ea_log_softmax_f32(const float* in, long ld_in, float* out, long M, int V, ea_stream_t stream);
int ea_log_softmax_bf16(const void* in, long ld_in, float* out, long M, int V, ea_stream_t stream);
long ea_ctc_workspace_bytes(int B, int T, int Lmax);
int ea_ctc_loss(const float* lprobs, const int* targets, const int* in_len, const int* tgt_len, float* nll,
                void* workspace, int B, int T, int V, int Lmax, int blank, ea_stream_t stream);
/* backward of ea_ctc_loss: reads the lattice left in `workspace`; grad_scale_dev (device fp32 scalar,
 * may be NULL) multiplies grad_scale — the autograd grad_output, without a host sync. */
int ea_ctc_grad(const float* lprobs, const void* workspace, const float* nll, const int* targets,
                const int* in_len, const int* tgt_len, void* dlogits, long ld_out, int dlogits_bf16, int B, int T,
                int V, int Lmax, int blank, float grad_scale, const float* grad_scale_dev, int zero_infinity,
                ea_stream_t stream);
/* CTC greedy decoding — espresso/tools/ctc_decoder.py:172-188 (max over V, unique_consecutive, drop blank, score =
 * sum of per-frame maxima, alignments = first frame of each emitted token).  x: fp32 or bf16 [B][T][ld] log-probs. */
int ea_ctc_greedy_decode(const void* x, long ld, int x_bf16, const int* in_len, int* best, float* bestv, int* tokens,
                         int* align, int* out_len, float* score, int B, int T, int V, int blank, int pad,
                         ea_stream_t stream);
/* Label-smoothed CE — espresso/criterions/label_smoothed_cross_entropy_v2.py:49-119.  smoothing 0 = uniform, 1 = unigram
 * (prior fp32 [V], sums to one), 2 = temporal (neighbouring targets of the same sentence, weights 2:5:5:2; rows are
 * b*tgt_len + u).  out_loss[0] += sum loss, out_loss[1] += sum nll (pad rows skipped). */
int ea_label_smoothed_ce(const void* logits, long ld, int logits_bf16, const int* target, float* out_loss,
                         void* dlogits, long ld_out, int dlogits_bf16, long M, int V, int pad_idx, float eps,
                         float grad_scale, int smoothing, const float* prior, int tgt_len, ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Front-end: Kaldi fbank + global CMVN + SpecAugment + padding — espresso/tools/utils.py:426-454
 * (torchaudio.compliance.kaldi.fbank defaults), fairseq/data/audio/feature_transforms/global_cmvn.py:26-29,
 * espresso/data/feature_transforms/adaptive_specaugment.py:77-136, espresso/tools/utils.py:97-113.
 * wav: concatenated fp32 samples (int16 scale), offsets int64 [B+1]; feat fp32 [B][Tmax][nmel]
 * (rows >= n_frames(b) are written as 0); utt_sum fp32 [B] zeroed by caller (sum of features, for
 * the mean mask value); out_len int32 [B] = 1+(N-frame_len)/frame_shift.  window/twiddle/mel_* are
 * host-built tables (espresso_amd/data/fbank_tables.py).  Mask lists: int32 [B][n][2] = (start,width). */
int ea_fbank_batch(const float* wav, const long* offsets, int B, const float* window, const float* twiddle,
                   const int* mel_start, const int* mel_len, const int* mel_woff, const float* mel_w,
                   const float* cmvn_mean, const float* cmvn_std, float* feat, float* utt_sum, int* out_len,
                   int Tmax, int nmel, int frame_len, int frame_shift, float preemph, float log_floor,
                   ea_stream_t stream);
/* the same on int16 samples (what ea_audio_read_batch_i16 delivers): the conversion to fp32 is exact, half the bytes move */
int ea_fbank_batch_i16(const int16_t* wav, const long* offsets, int B, const float* window, const float* twiddle,
                   const int* mel_start, const int* mel_len, const int* mel_woff, const float* mel_w,
                   const float* cmvn_mean, const float* cmvn_std, float* feat, float* utt_sum, int* out_len,
                   int Tmax, int nmel, int frame_len, int frame_shift, float preemph, float log_floor,
                   ea_stream_t stream);
int ea_specaugment(float* feat, const int* lengths, const float* utt_sum, const int* fmask, const int* tmask,
                   int nf, int nt, int B, int Tmax, int nmel, int use_mean, float mask_value, ea_stream_t stream);

/* Global CMVN statistics on the device — espresso/tools/compute_global_cmvn_stats.py:55-120 (per-utterance numpy
 * sum / variance merged on the host).  acc fp64 [2*nmel+1] = (sum[f], sum of squares[f], frame count), accumulated
 * across calls over the valid frames (t < lengths[b]) of feat fp32 [B][Tmax][nmel]; zeroed by the caller once. */
int ea_feature_stats(const float* feat, const int* lengths, double* acc, int B, int Tmax, int nmel, ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer on flat fp32 buffers — fairseq/utils.py:347-397 (clip_grad_norm_),
 * fairseq/optim/adam.py:215-240.  coef[0] multiplies every gradient (pre_scale * clip), coef[1]
 * receives the (pre-scaled) gradient norm; both stay on the device. */
int ea_grad_sumsq(const float* g, long n, float* out, ea_stream_t stream);
/* scale = pre_scale / denom_dev[0] (denom_dev: device fp32 scalar such as the all-reduced sample_size; may be NULL) */
int ea_clip_coef(const float* sumsq, float pre_scale, const float* denom_dev, float max_norm, float* coef,
                 ea_stream_t stream);
int ea_adam_step(float* p, float* g, float* m, float* v, void* p_bf16, long n, const float* coef, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int step, int zero_grad,
                 ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Native layer runtime (csrc/engine.hip): a whole Conformer encoder layer per call —
 * espresso/modules/conformer_with_relative_positional_embedding_encoder_layer.py:81-145 forward and its
 * backward — as one back-to-back launch sequence on `stream`.  Weights: bf16 matrices (w1, w2, wqkv =
 * [q;k;v] rows, wo, wpos, pw1, pw2) + fp32 vectors; gradients are ACCUMULATED (+=) into the fp32
 * buffers in `grads` (the flat gradient buffer).  x: bf16 [B*T][C] batch-major rows.  `pe`: bf16
 * [2T-1][C] sinusoidal table.  `saved` carries activations from fwd to bwd, `scratch` is reusable;
 * sizes from ea_conformer_layer_workspace; fwd / bwd get the capacities of the two arenas and return -5 before
 * launching anything when an arena is too small.  Dropout masks are re-derived from `seed`. */
typedef struct EaFfnParams { const float *ln_g, *ln_b; const void* w1; const float* b1; const void* w2; const float* b2; } EaFfnParams;
typedef struct EaFfnGrads { float *ln_g, *ln_b, *w1, *b1, *w2, *b2; } EaFfnGrads;
typedef struct EaAttnParams {
  const float *ln_g, *ln_b; const void* wqkv; const float* bqkv; const void* wo; const float* bo;
  const float *pos_u, *pos_v; const void* wpos;
} EaAttnParams;
typedef struct EaAttnGrads { float *ln_g, *ln_b, *wqkv, *bqkv, *wo, *bo, *pos_u, *pos_v, *wpos; } EaAttnGrads;
typedef struct EaConvParams {
  const float *ln_g, *ln_b; const void* pw1; const float* dw; const float *bn_g, *bn_b; float *bn_rm, *bn_rv; const void* pw2;
} EaConvParams;
typedef struct EaConvGrads { float *ln_g, *ln_b, *pw1, *dw, *bn_g, *bn_b, *pw2; } EaConvGrads;
typedef struct EaLayerGrads { EaFfnGrads ffn1; EaAttnGrads attn; EaConvGrads conv; EaFfnGrads ffn2; float *final_ln_g, *final_ln_b; } EaLayerGrads;
typedef struct EaConformerLayer {
  EaFfnParams ffn1; EaAttnParams attn; EaConvParams conv; EaFfnParams ffn2; const float *final_ln_g, *final_ln_b;
  EaLayerGrads grads;
  /* optional (may be NULL): bf16 scratch of 4*C*F + 7*C*C elements owned by the caller.  A training forward refreshes it with
   * the transposes of ffn1.w1, ffn2.w1, wqkv, wo, pw1, pw2, ffn1.w2, ffn2.w2 (in this order); the backward's data-gradient
   * GEMMs then read k-contiguous weights (direct-to-LDS kernel where it pays) instead of transposing them in registers. */
  void* wt;
} EaConformerLayer;
typedef struct EaLayerShape {
  int B, T, C, H, F, KW, training; float p_drop, p_act, p_attn; uint64_t seed; int has_attn_mask;
  /* backward only: the scratch arena is untouched since the previous ea_conformer_layer_bwd call with the same shape and
   * settings (consecutive layers of one backward pass) - lets the attention backward skip re-zeroing its band buffer */
  int scratch_clean;
  /* relative-position mode of the attention block: 0 = sinusoidal table projected by pos_proj with pos_bias_u / pos_bias_v (the
   * Conformer recipes), 1 = learned table (`pe` is the bf16 [2T-1][C] slice of the table, used as is; its fp32 gradient is
   * returned through `dpe`).  `act`: FFN activation of the Transformer layer (EA_ACT_RELU / EA_ACT_SILU); the Conformer
   * layer always uses SiLU. */
  int pos_mode, act;
  int S; /* decoder layer only: padded encoder length (keys of the encoder-decoder attention); T is then the target length */
  /* backward only (conformer / transformer encoder layers): 0 = the call returns with every gradient ordered on `stream`
   * (side work forked and joined inside the call); 1 or 2 = DEFERRED mode using scratch half (defer - 1): the optimizer-only
   * work (all weight / bias / LayerNorm / BatchNorm parameter gradients) is launched on the side stream at the end of the call
   * and joined by the next backward call, which must use the other half — so it overlaps that call's data-gradient chain.
   * The gradients of a deferred call are ordered on `stream` once the NEXT backward call, ea_backward_flush or any layer
   * forward call has been issued; `saved` must stay alive until then. */
  int defer;
  /* forward only (Conformer layer, training): 1 = the k-contiguous weight copies in EaConformerLayer.wt are already fresh for this
   * update (the caller ran ea_conformer_layer_refresh_wt after the optimizer step, typically on another stream under the
   * sub-sampler's forward pass) - the forward call then skips its own transposes.  0 = the call refreshes them itself. */
  int wt_fresh;
} EaLayerShape;

/* Dropout sites of a layer call (FairseqDropout calls of the reference, cited per site) and the mask stream each one uses.
 * Every site keeps element `idx` iff ea_dropout_hash(site seed, idx) >= thr, thr = min(floor(p * 2^32), 2^32 - 1), and scales
 * kept elements by 1 / (1 - p); site seed = ea_layer_dropout_seed(EaLayerShape.seed, site).  Element index per site, with
 * activation rows m = b*T + t (batch-major):
 *   *_ACT    m*F + f        fairseq/modules/conformer_layer.py:144 / transformer_layer.py:212 (activation_dropout, after the activation)
 *   *_OUT    m*C + c        conformer_layer.py:146 / transformer_layer.py:216 (dropout on the FFN output, before 0.5*y + x)
 *   *_PROBS  ((h*B + b)*T + i)*S + j   fairseq/modules/multihead_attention.py:874 (attention_dropout on the softmax output)
 *   ATTN_OUT / CROSS_OUT  m*C + c      espresso/modules/conformer_with_relative_positional_embedding_encoder_layer.py:125,
 *                                      transformer_layer.py:196, 456, 481 (dropout on out_proj's output, before the residual)
 *   CONV_OUT m*C + c        conformer_layer.py:100 (dropout at the end of the convolution module)
 * The Transformer encoder / decoder layers use FFN1_* for their one FFN; CROSS_* exist in the decoder layer only. */
enum {
  EA_SITE_FFN1_ACT = 0, EA_SITE_FFN1_OUT = 1, EA_SITE_ATTN_PROBS = 2, EA_SITE_ATTN_OUT = 3, EA_SITE_CONV_OUT = 4,
  EA_SITE_FFN2_ACT = 5, EA_SITE_FFN2_OUT = 6, EA_SITE_CROSS_PROBS = 7, EA_SITE_CROSS_OUT = 8
};
uint64_t ea_layer_dropout_seed(uint64_t layer_seed, int site);
/* Host evaluation of the device's counter-based dropout hash (csrc/common.h ea_hash) for idx0 .. idx0 + n - 1: the pin for
 * the tests' numpy restatement (oracle/dropout_ref.py); no GPU needed. */
int ea_dropout_hash_host(uint64_t seed, uint64_t idx0, long n, uint32_t* out);

/* tuning hook: run weight-gradient GEMMs / bias sums of the layer backward on a side stream (default on); returns the
 * previous value.  Sizes from ea_conformer_layer_workspace depend on this setting: query them after changing it. */
int ea_set_backward_overlap(int on);
/* tuning hook: honour EaLayerShape.defer (default on); returns the previous value.  Workspace sizes depend on it. */
int ea_set_backward_deferred(int on);
/* tuning hook (A/B): run the deferred work on the caller's stream at the end of each layer backward instead of the side stream */
int ea_set_backward_deferred_inline(int on);
/* make `stream` wait for all deferred side work issued so far (call after the last layer backward of a pass) */
int ea_backward_flush(ea_stream_t stream);
/* Do two streams share a HARDWARE queue?  1 = work on `b` waits for work on `a`, 0 = they run side by side, < 0 = error.
 * HIP deals a process's streams onto 4 hardware queues by load; two streams on one queue serialise silently.  The layer
 * runtime picks its side stream with this probe (a tiny kernel on `b` must finish while a 200 us spin kernel runs on `a`); callers
 * that create their own side streams can do the same.  Blocks the host until `a` reaches the probe; EA_SIDE_STREAM_PROBE=0
 * disables the runtime's own use of it. */
int ea_streams_share_queue(ea_stream_t a, ea_stream_t b);
/* What the probe decided for the layer runtime's own side stream (created by the first backward call): returns 1 when the stream
 * exists; *rejected = candidate streams that shared the caller's hardware queue and were turned down (-1: none created yet),
 * *unprobed = 1 when the accepted stream was not measured (EA_SIDE_STREAM_PROBE=0, or the eighth candidate).  Reported per rank in
 * the bench line: on a multi-rank job RCCL's streams change how HIP deals queues, so every rank's verdict should be visible. */
int ea_side_stream_report(int* rejected, int* unprobed);
int ea_conformer_layer_workspace(const EaLayerShape* shape, long* saved_bytes, long* scratch_bytes);
/* The k-contiguous (transposed) bf16 copies of a Conformer layer's eight weight matrices, which its BACKWARD data-gradient GEMMs
 * read (EaConformerLayer.wt): one batched transpose on `stream`.  The weights only change in the optimizer step, and nothing in the
 * forward pass reads the copies: a caller can refresh all layers on a side stream right after the optimizer step and pass
 * EaLayerShape.wt_fresh = 1 to the forward calls (12 launches off the compute stream per update step). */
int ea_conformer_layer_refresh_wt(const EaConformerLayer* layer, const EaLayerShape* shape, ea_stream_t stream);
int ea_conformer_layer_fwd(const EaConformerLayer* layer, const EaLayerShape* shape, const void* x_in, void* x_out,
                           const int* key_len, const float* attn_mask, const void* pe, void* saved, long saved_bytes,
                           void* scratch, long scratch_bytes, ea_stream_t stream);
int ea_conformer_layer_bwd(const EaConformerLayer* layer, const EaLayerShape* shape, const void* x_in, const void* dy,
                           void* dx, const int* key_len, const void* pe, void* saved, long saved_bytes, void* scratch,
                           long scratch_bytes, ea_stream_t stream);
/* CHAINED calls over a stack of Conformer layers of one shape (C <= 512): layer k's closing LayerNorm and layer k+1's opening
 * LayerNorm (of ffn1) touch the same rows back to back, in the forward and — mirrored — in the backward pass.  A chained call does
 * both with one kernel (ea_layernorm_fwd2 / ea_layernorm_bwd2_dx): 2 launches and two passes over the activations fewer per layer
 * boundary and update, results bit-identical to the plain calls.  All members may be NULL / 0 (= the plain call).
 *   forward of layer k:    `next` / `next_saved`: the following layer and ITS saved arena; this call also writes that layer's ffn1
 *                          LayerNorm output and statistics there.
 *   forward of layer k+1:  `ln1_done` = 1: `saved` already holds them (the previous call was chained to this layer).
 *   backward of layer k+1: `prev` / `prev_saved` / `prev_seed` (EaLayerShape.seed of layer k's forward) / `prev_pre` (bf16 [B*T][C]
 *                          buffer): the call ends with layer k's final-LayerNorm backward; `dx` receives the gradient w.r.t. layer
 *                          k's PRE-final-norm activations, `prev_pre` the 0.5 * dropout(.) copy layer k's ffn2 block starts from, and
 *                          layer k's final_ln gradients are accumulated with this call's side work.
 *   backward of layer k:   `final_ln_done` = 1: `dy` is that gradient and `pre_in` that copy; the call skips its final-norm backward.
 * The caller must hand layer k+1's `dx` to layer k unchanged (layer k's output has no other consumer), and `prev_pre` must stay
 * untouched until layer k's own side work has been joined (deferred mode: by the backward call AFTER layer k's, or by
 * ea_backward_flush) — its weight-gradient launch reads the buffer. */
typedef struct EaLayerChain {
  const EaConformerLayer* next; void* next_saved; long next_saved_bytes; int ln1_done;
  const EaConformerLayer* prev; void* prev_saved; long prev_saved_bytes; uint64_t prev_seed; void* prev_pre;
  int final_ln_done; const void* pre_in;
} EaLayerChain;
int ea_conformer_layer_fwd_chained(const EaConformerLayer* layer, const EaLayerShape* shape, const EaLayerChain* chain, const void* x_in,
                                   void* x_out, const int* key_len, const float* attn_mask, const void* pe, void* saved,
                                   long saved_bytes, void* scratch, long scratch_bytes, ea_stream_t stream);
int ea_conformer_layer_bwd_chained(const EaConformerLayer* layer, const EaLayerShape* shape, const EaLayerChain* chain, const void* x_in,
                                   const void* dy, void* dx, const int* key_len, const void* pe, void* saved, long saved_bytes,
                                   void* scratch, long scratch_bytes, ea_stream_t stream);
/* A STACK of n Conformer layers of one shape in one call per direction — the layer loop of
 * espresso/models/transformer/speech_transformer_encoder.py:374-390 and its backward.  Same launches as n chained layer calls
 * (`chain` != 0: EaLayerChain between consecutive members; 0: plain calls); what it saves is host time: at small batches (the
 * transducer recipe: ~1 500 rows per micro-batch) a layer call issued from the Python loop costs 142 us (forward) / 181 us (backward)
 * of host time against 64 us for the same call issued back to back from C (cold caches after the interpreter ran;
 * profiles/r06_transducer_host_split.txt).
 * L[k].x_in: bf16 [B*T][C] input of layer k — for k > 0 ALSO the output buffer of layer k-1; x_out: output of the last layer.
 * L[k].shape: the layer's own seed / wt_fresh; backward: its own `defer` (alternating halves as for consecutive plain calls) and
 * `scratch_clean`.  Backward: dy / dx of the stack; dbuf: two bf16 [B*T][C] buffers for the gradients between layers; pre: three
 * bf16 [B*T][C] buffers used in rotation for EaLayerChain.prev_pre (must stay untouched until ea_backward_flush / the next
 * backward call; ignored when chain == 0).  Gradient ordering on `stream`: as after the same sequence of layer calls (the last
 * member's deferred side work is joined by the next backward call or ea_backward_flush). */
typedef struct EaStackLayer { const EaConformerLayer* layer; EaLayerShape shape; void* saved; long saved_bytes; void* x_in; } EaStackLayer;
int ea_conformer_stack_fwd(const EaStackLayer* L, int n, void* x_out, const int* key_len, const float* attn_mask, const void* pe,
                           void* scratch, long scratch_bytes, int chain, ea_stream_t stream);
int ea_conformer_stack_bwd(const EaStackLayer* L, int n, const void* dy, void* dx, void* const* dbuf, void* const* pre,
                           const int* key_len, const void* pe, void* scratch, long scratch_bytes, int chain, ea_stream_t stream);
/* Transformer encoder layer (pre-LN; fairseq/modules/transformer_layer.py:135-214 with the rel-pos MHA of
 * multihead_attention.py:650-907) in the same runtime: uses the `attn` and `ffn1` members (+ their grads) of EaConformerLayer;
 * `wt` (optional) holds 2*C*F + 4*C*C bf16 elements.  dpe: fp32 [2T-1][C], required iff shape->pos_mode == 1. */
int ea_transformer_layer_workspace(const EaLayerShape* shape, long* saved_bytes, long* scratch_bytes);
int ea_transformer_layer_fwd(const EaConformerLayer* layer, const EaLayerShape* shape, const void* x_in, void* x_out,
                             const int* key_len, const float* attn_mask, const void* pe, void* saved, long saved_bytes,
                             void* scratch, long scratch_bytes, ea_stream_t stream);
int ea_transformer_layer_bwd(const EaConformerLayer* layer, const EaLayerShape* shape, const void* x_in, const void* dy,
                             void* dx, const int* key_len, const void* pe, float* dpe, void* saved, long saved_bytes,
                             void* scratch, long scratch_bytes, ea_stream_t stream);
/* Transformer DECODER layer (pre-LN; fairseq/modules/transformer_layer.py:241-529 as built by
 * espresso/modules/transformer_with_relative_positional_embedding_layer.py:66-116) for teacher-forced training: causal
 * self-attention, encoder-decoder attention over `enc` (bf16 [B*S][C], valid lengths enc_len), FFN.  Head dim 64 (fused
 * attention kernels).  x: bf16 [B*T][C].  denc: bf16 [B*S][C], this layer's gradient w.r.t. `enc` (written, not accumulated).
 * `wt` (optional): 2*C*F + 8*C*C bf16 elements for the k-contiguous weight copies. */
typedef struct EaXAttnParams { const float *ln_g, *ln_b; const void* wq; const float* bq; const void* wkv; const float* bkv; const void* wo; const float* bo; } EaXAttnParams;
typedef struct EaXAttnGrads { float *ln_g, *ln_b, *wq, *bq, *wkv, *bkv, *wo, *bo; } EaXAttnGrads;
typedef struct EaDecoderLayer {
  EaAttnParams self_attn; EaXAttnParams cross; EaFfnParams ffn;
  EaAttnGrads g_self; EaXAttnGrads g_cross; EaFfnGrads g_ffn;
  void* wt;
} EaDecoderLayer;
int ea_decoder_layer_workspace(const EaLayerShape* shape, long* saved_bytes, long* scratch_bytes);
int ea_decoder_layer_fwd(const EaDecoderLayer* layer, const EaLayerShape* shape, const void* x_in, const void* enc, void* x_out,
                         const int* enc_len, void* saved, long saved_bytes, void* scratch, long scratch_bytes, ea_stream_t stream);
int ea_decoder_layer_bwd(const EaDecoderLayer* layer, const EaLayerShape* shape, const void* x_in, const void* enc, const void* dy,
                         void* dx, void* denc, const int* enc_len, void* saved, long saved_bytes, void* scratch, long scratch_bytes,
                         ea_stream_t stream);
/* tuning / test hook: use the fused attention kernels inside the layer runtime when the shape allows (default on);
 * returns the previous value.  Workspace sizes depend on it. */
int ea_set_flash_attention(int on);
/* tuning / test hook: LayerNorm backward writes the next block's dropout-scaled gradient as a second output (default on) */
int ea_set_fused_predrop(int on);

/* ------------------------------------------------------------------------------------------
 * Look-ahead word LM fusion (csrc/lookahead.hip) — espresso/models/tensorized_lookahead_language_model.py:83-269 over the
 * tensorized prefix tree of espresso/tools/tensorized_prefix_tree.py:14-108 (int32 device copies: children [NN][D],
 * prev_subword [NN], word_idx [NN], word_set [NN][2]; node 0 = none, node 1 = root).
 * ea_softmax_cumsum: cumsum[n][v] = sum_{w<=v} softmax(logits[n])[w] (fp32, rows ld apart), lp_tok[n] = log softmax[n][tok];
 *   rows with row_mask[n] == 0 keep their previous values (row_mask NULL = all rows).
 * ea_lookahead_advance: nodes[n] <- root if prev_tok[n] == space else the child reached by prev_tok[n] (none if absent).
 * ea_lookahead_logprobs: out fp32 [N][Vs] = sub-word log-probabilities of Eqn. 15 (floor log(1e-10)); lp_word_eos[n] is used
 *   as the <eos> score of hypotheses whose last sub-word is <space>. */
int ea_softmax_cumsum(const float* logits, long ld, const uint8_t* row_mask, float* cumsum, float* lp_tok, int N, int V, int tok,
                      ea_stream_t stream);
int ea_lookahead_advance(int* nodes, const int* prev_tok, const int* children, const int* prev_subword, int N, int D,
                         int space_idx, int root_id, ea_stream_t stream);
int ea_lookahead_logprobs(const int* nodes, const int* prev_tok, const float* cumsum, const float* lp_word_eos, const int* children,
                          const int* prev_subword, const int* word_idx, const int* word_set, float* out, int N, int Vw, int Vs,
                          int D, float oov_penalty, int open_vocab, int word_unk, int sub_space, int sub_eos, int sub_pad,
                          ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LSTM cell element-wise stages (csrc/lstm.hip) — torch.nn.LSTMCell as driven by espresso/models/speech_lstm.py:846-893
 * (transducer predictor, LSTM LM, attention decoder).  The packed pre-activations gates_pre = x W_ih^T + b_ih + h W_hh^T
 * + b_hh ([B][ldg] fp32, gate order i,f,g,o) come from ea_gemm_bf16 (fp32 output, fp32 residual).
 *   fwd: c_out/h fp32 [B][H]; h_out_bf16 [B][ldh] (GEMM operand of the next step / layer); gates_act fp32 [B][4H] saved for
 *        bwd (NULL at inference); keep_row uint8 [B] (NULL = none): rows whose state is frozen (h_prev_f32 required).
 *   bwd: dh = dh_bf16 (from the layer above, [B][ld_dh], may be NULL) + dh_f32 (recurrent, may be NULL); dc_in may be NULL;
 *        dgates bf16 [B][lddg]; dc_prev fp32 [B][H].
 * ea_gather_rows: out[n] = in[parent[n]] over rows of W elements of 2 or 4 bytes (beam reorder of cached LSTM states,
 * speech_lstm.py:981-999). */
int ea_lstm_cell_fwd(const float* gates_pre, long ldg, const float* c_prev, float* c_out, float* h_out_f32, void* h_out_bf16,
                     long ldh, float* gates_act, const uint8_t* keep_row, const float* h_prev_f32, int frozen_out_zero, int B,
                     int H, ea_stream_t stream);
int ea_lstm_cell_bwd(const void* dh_bf16, long ld_dh, const float* dh_f32, const float* dc_in, const float* gates_act,
                     const float* c_prev, const float* c, void* dgates, long lddg, float* dc_prev, const uint8_t* frozen, int B,
                     int H, ea_stream_t stream);
/* Whole time loop of one LSTM layer in one persistent launch (csrc/lstm_seq.hip) — the LSTMCell loop of
 * espresso/models/speech_lstm.py:846-893 and one direction of the packed nn.LSTM of :470-520.  Recurrent weights stay in
 * registers, workgroups exchange h_t (fwd) / dgates_t (bwd) through global memory behind a grid barrier per step.
 *   gx fp32 [U*B][4H] = x W_ih^T + b_ih + b_hh (rows t*B + b); w_hh bf16 [4H][H]; w_hhT bf16 [H][4H]; h0 bf16 [B][H] / c0 fp32
 *   [B][H] (NULL = zero state); frozen uint8 [U][B] (NULL = none); reverse: walk t = U-1 .. 0.
 *   fwd out: hs bf16 [U*B][H], cs fp32 [U][B][H], act fp32 [U][B][4H] (activated gates i,f,g,o), h_last fp32 [B][H] (may be NULL).
 *   bwd in : dhs bf16 [U*B][H] (may be NULL), dh_last / dc_last fp32 [B][H] (may be NULL);
 *   bwd out: dG bf16 [U*B][4H] (gradient of the gate pre-activations), dh0 / dc0 fp32 [B][H] (may be NULL).
 *   counter: 2 x uint32 of device scratch (zeroed by the call); counter[1] != 0 afterwards = a barrier wait timed out.
 * ea_lstm_seq_supported: 1 <= B <= 64 and H in {256, 320, 512, 640, 768, 800, 1024}; other shapes return -2 (callers use
 * the per-step ea_gemm_bf16 + ea_lstm_cell_* path). */
int ea_lstm_seq_supported(int B, int H);
int ea_lstm_seq_fwd(const float* gx, const void* w_hh, const void* h0, const float* c0, const uint8_t* frozen, void* hs, float* cs,
                    float* act, float* h_last, unsigned* counter, int B, int U, int H, int reverse, int frozen_out_zero,
                    ea_stream_t stream);
int ea_lstm_seq_bwd(const void* dhs, const float* dh_last, const float* dc_last, const float* act, const float* cs, const float* c0,
                    const uint8_t* frozen, const void* w_hhT, void* dG, float* dh0, float* dc0, unsigned* counter, int B, int U,
                    int H, int reverse, ea_stream_t stream);
/* frozen_out_zero / frozen: packed-sequence semantics of the BiLSTM encoder (speech_lstm.py:470-520, pack_padded_sequence /
 * pad_packed_sequence with padding_value 0): rows past their length keep their state, emit zeros and get no gradient.
 *
 * Bahdanau attention of one decoder step — espresso/modules/speech_attention.py:38-87 (normalize=True):
 *   score[t][b] = sum_a nv[a] tanh(qp[b][a] + key[t][b][a] + bias[a]), nv = g v/||v|| (caller), softmax over t < len[b],
 *   ctx[b] = sum_t p[t][b] value[t][b].  qp bf16 [B][A]; key bf16 [T][B][A]; value bf16 [T][B][Cv]; p fp32 [T][B]; ctx bf16 [B][ldc].
 * bwd (one step): dqp bf16 [B][A]; dkey_acc fp32 [T][B][A] and dvalue_acc fp32 [T][B][Cv] are accumulated in place over the
 * decoder steps; dnv_acc / dbias_acc fp32 [A] by atomics. */
int ea_bahdanau_fwd(const void* qp, const void* key, const void* value, const float* nv, const float* bias, const int* len,
                    float* p_out, void* ctx, long ldc, int T, int B, int A, int Cv, const int* kv_col, int Bkv,
                    ea_stream_t stream);
/* kv_col (int [B], or NULL): beam search — row b attends over column kv_col[b] of key / value / len, which then have Bkv
 * columns (one per sentence) instead of B. */
int ea_bahdanau_bwd(const void* dctx, long ldd, const void* qp, const void* key, const void* value, const float* nv,
                    const float* bias, const int* len, const float* p, void* dqp, float* dkey_acc, float* dvalue_acc,
                    float* dnv_acc, float* dbias_acc, int T, int B, int A, int Cv, ea_stream_t stream);
int ea_gather_rows(const void* in, void* out, const int* parent, int N, int W, int elem_bytes, ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batched beam-search decoding (csrc/decode.hip) — fairseq/sequence_generator.py:355-609, fairseq/search.py:103-144,
 * fairseq/modules/multihead_attention.py:716-760,878-897,964-989.
 * ea_decode_attention: out[n] = softmax(q[n] . K[r]^T) V[r], r = kv_row ? kv_row[n] : n, over len ? len[r] : max_len keys;
 *   q/out bf16 [N][ldq] (q pre-scaled), K/V bf16 rows of `row_stride` elements, key j at j*ldkv + koff/voff + h*dh.
 * ea_kv_append_reorder: new[n][0:L] = old[parent[n]][0:L], new[n][L] = kv_new[n]  (rows of W bf16, capacity Lmax).
 * ea_beam_mask_rows: sequence_generator.py:395-424 on fp32 lprobs [N][V] in place.
 * ea_beam_topk: per sentence the k (<=128) best of lprobs[(s*beam+b)][v] + prev_scores[s*beam+b], b < nbeam_used. */
int ea_decode_attention(const void* q, const void* K, const void* V, const int* kv_row, const int* len, void* out, int N, int H,
                        int dh, long ldq, long row_stride, long ldkv, int koff, int voff, int max_len, ea_stream_t stream);
int ea_kv_append_reorder(const void* old_cache, void* new_cache, const void* kv_new, const int* parent, int N, int L, int Lmax,
                         int W, ea_stream_t stream);
int ea_beam_mask_rows(float* lprobs, int N, int V, int pad, int unk, int eos, float unk_penalty, int only_eos, int forbid_eos,
                      float eos_factor, int use_eos_factor, ea_stream_t stream);
int ea_beam_topk(const float* lprobs, const float* prev_scores, int bsz, int beam, int nbeam_used, int V, int k,
                 float* cand_score, int* cand_tok, int* cand_beam, ea_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * RNN-T loss — torchaudio.functional.rnnt_loss as called at espresso/criterions/transducer_loss.py:130-140
 * (blank = "<s>", clamp = -1, fused log-softmax).  logits fp32 or bf16 (logits_bf16) [B][T][U1][V] (U1 = Umax + 1; bf16 is
 * what the reference's fc_out produces under bf16 autocast), targets int32 [B][Umax], loss fp32 [B] = -log p(y|x);
 * grad = grad_scale * d(sum loss)/d(logits), fp32 or bf16, same layout as logits.
 * ld: row pitch of logits AND grad in elements (>= V).  The joint of this library writes rows padded to a multiple of 64
 * (V = 5004 -> 5056): 16-byte accesses here, aligned fast paths / direct-to-LDS GEMMs for the three joint products; the grad
 * kernel writes the pad columns as zeros so that they can be part of a GEMM reduction.
 * workspace: ea_rnnt_workspace_bytes(B,T,U1) bytes, kept between the two calls. */
long ea_rnnt_workspace_bytes(int B, int T, int U1);
int ea_rnnt_loss(const void* logits, int logits_bf16, const int* targets, const int* logit_lengths, const int* target_lengths,
                 float* loss, void* workspace, int B, int T, int U1, int V, long ld, int Umax, int blank, ea_stream_t stream);
int ea_rnnt_grad(const void* logits, int logits_bf16, const int* targets, const int* logit_lengths, const int* target_lengths,
                 const float* loss, const void* workspace, void* grad, int grad_bf16, int B, int T, int U1, int V, long ld,
                 int Umax, int blank, float grad_scale, const float* grad_scale_dev, ea_stream_t stream);
/* The alpha / beta sweep of ea_rnnt_loss alone: lpb / lpy fp32 [B][T][U1] = log p(blank | t,u), log p(y_{u+1} | t,u). */
int ea_rnnt_scan(const float* lpb, const float* lpy, const int* logit_lengths, const int* target_lengths, float* alpha,
                 float* beta, float* loss, int B, int T, int U1, ea_stream_t stream);
/* Joint output layer FUSED with the RNN-T loss (csrc/joint_rnnt.hip, round 6): replaces fc_out of
 * espresso/models/transformer/speech_transformer_transducer_base.py:276-299 followed by torchaudio.functional.rnnt_loss
 * (espresso/criterions/transducer_loss.py:130-140) without ever writing the (B, T, U1, V) logits.
 *   Z bf16 [B*T*U1][J] = relu(E + D) (ea_joint_add_relu_f32), W bf16 [V][J], bias fp32 [V] or NULL; J % 64 == 0, 16-byte aligned.
 * ea_joint_rnnt_loss: loss[b] as ea_rnnt_loss, computed from the fp32 accumulators of the product (the unfused path rounds the
 *   logits to bf16 first); the workspace keeps lse / log-probs / alpha / beta for the gradient call.
 * ea_joint_rnnt_grad: dl bf16 [B*T*U1][ld] = d loss / d logits * grad_scale (* grad_scale_dev[0]); ld % 32 == 0, ld >= V, columns
 *   V .. ld - 1 are written as zeros (dl is the operand of the joint's data- and weight-gradient GEMMs, which reduce over ld).
 * Both return -2 for shapes they do not take (the caller then materialises the logits and uses ea_rnnt_loss / ea_rnnt_grad). */
long ea_joint_rnnt_workspace_bytes(int B, int T, int U1, int V);
int ea_joint_rnnt_loss(const void* Z, const void* W, const float* bias, const int* targets, const int* logit_lengths,
                       const int* target_lengths, float* loss, void* workspace, int B, int T, int U1, int V, int J, int Umax,
                       int blank, ea_stream_t stream);
int ea_joint_rnnt_grad(const void* Z, const void* W, const float* bias, const int* targets, const int* logit_lengths,
                       const int* target_lengths, const float* loss, void* workspace, void* dl, long ld, int B, int T, int U1,
                       int V, int J, int Umax, int blank, float grad_scale, const float* grad_scale_dev, ea_stream_t stream);
/* Joint network element-wise stages (espresso/models/transformer/speech_transformer_transducer_base.py:276-299):
 * Z[b][t][u] = relu(E[b][t] + D[b][u]) (bf16, E [B*T][J], D [B*U1][J], Z [B*T*U1][J], J % 8 == 0) and its backward
 * reductions dE[b][t] = sum_u dZ[b][t][u], dD[b][u] = sum_t dZ[b][t][u] (either output may be NULL). */
int ea_joint_add_relu(const void* E, const void* D, void* Z, int B, int T, int U1, int J, ea_stream_t stream);
int ea_joint_reduce(const void* dZ, void* dE, void* dD, int B, int T, int U1, int J, ea_stream_t stream);
/* The training path's form (round 6): E and D are the fp32 LayerNorm outputs (ea_layernorm_fwd_f32out), the sum and the ReLU are
 * evaluated in fp32 as in the reference's autocast run (:292-294) and only Z — fc_out's bf16 GEMM operand — is rounded; dE / dD
 * leave as fp32 sums (inputs of ea_layernorm_bwd_f32dy).  relu'(E + D) is recovered from Z > 0, which is exact for the fp32 sum. */
int ea_joint_add_relu_f32(const float* E, const float* D, void* Z, int B, int T, int U1, int J, ea_stream_t stream);
int ea_joint_reduce_f32(const void* dZ, float* dE, float* dD, int B, int T, int U1, int J, ea_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ESPRESSO_AMD_H */
