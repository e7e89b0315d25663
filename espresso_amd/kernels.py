"""Thin, typed Python wrappers over the C ABI (include/espresso_amd.h).

Every function takes CUDA(=HIP) torch tensors, checks dtype/contiguity, and launches the HIP kernel
on torch's current stream.  torch is used here ONLY for device memory and streams.  Nothing in this
module computes on the host and nothing falls back to eager PyTorch: a missing library or a CPU
tensor raises.
"""
import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import EaGemmParams, check

ACT_NONE, ACT_RELU, ACT_SILU = 0, 1, 2
_ACT = {None: 0, "none": 0, "relu": 1, "silu": 2, "swish": 2}


def _stream():
    """Raw handle of torch's current HIP stream on the current device (the C accessors: `torch.cuda.current_stream()` builds a
    Python Stream object through several device-index lookups, ~9 us per call — per kernel launch that was 12 % of the
    transducer beam search's host time)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("espresso_amd kernels need device tensors (no CPU path)")
    return ctypes.c_void_p(t.data_ptr())


def drop_params(p: float):
    """(threshold, keep-scale) for dropout probability p as the kernels expect them."""
    if p <= 0.0:
        return 0, 1.0
    thr = min(int(p * 4294967296.0), 4294967295)
    return thr, 1.0 / (1.0 - p)


def gemm(
    A: torch.Tensor,
    B: torch.Tensor,
    C: torch.Tensor,
    M: int,
    N: int,
    K: int,
    *,
    lda: int,
    ldb: int,
    ldc: int,
    a_kstrided: bool = False,
    b_kstrided: bool = False,
    batch: int = 1,
    zdiv: int = 1,
    sA=(0, 0),
    sB=(0, 0),
    sC=(0, 0),
    a_off: int = 0,
    b_off: int = 0,
    c_off: int = 0,
    bias: Optional[torch.Tensor] = None,
    act=None,
    alpha: float = 1.0,
    out_scale: float = 1.0,
    resid: Optional[torch.Tensor] = None,
    ldr: int = 0,
    sR=(0, 0),
    r_off: int = 0,
    C2: Optional[torch.Tensor] = None,
    ldc2: int = 0,
    aux: Optional[torch.Tensor] = None,
    ldaux: int = 0,
    sX=(0, 0),
    accumulate: bool = False,
    drop_p: float = 0.0,
    drop_seed: int = 0,
    splitk: int = 1,
    qsplit=None,
):
    """C[z][m][n] = epi(alpha * sum_k A(z;m,k) B(z;n,k)).  Offsets (a_off, ...) are in elements.
    qsplit = (q_u, q_v | None, pos_u | None, pos_v | None, n, ld_q, scale): columns [0, n) go to q_u / q_v as
    (q + pos) * scale instead of C (EaGemmParams.q_u in the header)."""
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16
    assert C.dtype in (torch.bfloat16, torch.float32)
    p = EaGemmParams()
    p.A = A.data_ptr() + 2 * a_off
    p.B = B.data_ptr() + 2 * b_off
    p.C = C.data_ptr() + C.element_size() * c_off
    p.C2 = (C2.data_ptr() + 2 * c_off) if C2 is not None else None
    if bias is not None:
        assert bias.dtype == torch.float32
        p.bias = bias.data_ptr()
    if resid is not None:
        assert resid.dtype in (torch.bfloat16, torch.float32)
        p.resid = resid.data_ptr() + resid.element_size() * r_off
        p.resid_f32 = 1 if resid.dtype == torch.float32 else 0
    if aux is not None:
        assert aux.dtype == torch.bfloat16
        p.aux = aux.data_ptr()
    p.M, p.N, p.K, p.batch, p.zdiv = M, N, K, batch, zdiv
    p.a_kstrided, p.b_kstrided = int(a_kstrided), int(b_kstrided)
    p.c_f32 = 1 if C.dtype == torch.float32 else 0
    p.accumulate = int(accumulate)
    p.act = _ACT[act] if not isinstance(act, int) else act
    p.lda, p.ldb, p.ldc, p.ldc2, p.ldr, p.ldaux = lda, ldb, ldc, ldc2, ldr, ldaux
    p.sA_hi, p.sA_lo = sA
    p.sB_hi, p.sB_lo = sB
    p.sC_hi, p.sC_lo = sC
    p.sR_hi, p.sR_lo = sR
    p.sX_hi, p.sX_lo = sX
    p.alpha, p.out_scale = alpha, out_scale
    thr, scale = drop_params(drop_p)
    p.drop_seed, p.drop_thr, p.drop_scale = drop_seed, thr, scale
    p.splitk = max(1, int(splitk))
    ws = None
    if p.splitk > 1:
        ws = torch.empty(p.splitk * batch * M * N, dtype=torch.float32, device=C.device)
        p.workspace = ws.data_ptr()
    if qsplit is not None:
        q_u, q_v, pos_u, pos_v, qn, ld_q, qscale = qsplit
        assert q_u.dtype == torch.bfloat16 and (q_v is None or q_v.dtype == torch.bfloat16)
        assert (pos_u is None or pos_u.dtype == torch.float32) and (pos_v is None or pos_v.dtype == torch.float32)
        p.q_u, p.q_v, p.pos_u, p.pos_v = _p(q_u), _p(q_v), _p(pos_u), _p(pos_v)
        p.qsplit_n, p.ld_q, p.qscale = int(qn), int(ld_q), float(qscale)
    check(_lib.lib().ea_gemm_bf16(ctypes.byref(p), _stream()), "ea_gemm_bf16")
    return C


def layernorm_fwd(x, gamma, beta, eps=1e-5, row_zero=None, drop_p=0.0, drop_seed=0, save_stats=True, out_f32=False):
    """out_f32: the output stays fp32 (an fp32 island of the reference's autocast run: the joint network's LayerNorms)."""
    M, C = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    y = torch.empty(M, C, dtype=torch.float32, device=x.device) if out_f32 else torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    thr, scale = drop_params(drop_p)
    fn = _lib.lib().ea_layernorm_fwd_f32out if out_f32 else _lib.lib().ea_layernorm_fwd
    check(fn(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, C, eps, _p(row_zero), drop_seed, thr, scale, _stream()),
          "ea_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(x, dy, gamma, mean, rstd, dgamma, dbeta, row_zero=None, drop_p=0.0, drop_seed=0, dx_add=None):
    M, C = x.shape
    assert dy.is_contiguous() and dy.dtype in (torch.bfloat16, torch.float32)
    dx = torch.empty_like(x)
    thr, scale = drop_params(drop_p)
    ws = torch.empty(_lib.lib().ea_layernorm_bwd_workspace_bytes(M, C) // 4, dtype=torch.float32, device=x.device)
    fn = _lib.lib().ea_layernorm_bwd_f32dy if dy.dtype == torch.float32 else _lib.lib().ea_layernorm_bwd
    check(fn(_p(x), _p(dy), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma), _p(dbeta), M, C, _p(row_zero), drop_seed, thr, scale,
             _p(dx_add), _p(ws), _stream()), "ea_layernorm_bwd")
    return dx


def cast_f32_to_bf16(src, dst=None):
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    check(_lib.lib().ea_cast_f32_to_bf16(_p(src), _p(dst), src.numel(), _stream()), "ea_cast_f32_to_bf16")
    return dst


def cast_f32_to_bf16_rows(src, ld_dst):
    """src fp32 [M][N] (row stride src.stride(0)) -> bf16 [M][ld_dst] buffer (columns N.. zero); returns the [M][N] view."""
    M, N = src.shape
    assert src.dtype == torch.float32 and src.stride(1) == 1 and ld_dst >= N
    dst = torch.empty(M, ld_dst, dtype=torch.bfloat16, device=src.device)
    check(_lib.lib().ea_cast_f32_to_bf16_rows(_p(src), src.stride(0), _p(dst), ld_dst, M, N, _stream()), "ea_cast_f32_to_bf16_rows")
    return dst if ld_dst == N else dst[:, :N]


def cast_bf16_to_f32(src, dst=None):
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    check(_lib.lib().ea_cast_bf16_to_f32(_p(src), _p(dst), src.numel(), _stream()), "ea_cast_bf16_to_f32")
    return dst


def scale_dropout(x, a=1.0, y=None, b=0.0, drop_p=0.0, drop_seed=0, out=None):
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    thr, scale = drop_params(drop_p)
    check(_lib.lib().ea_scale_dropout_bf16(_p(x), _p(y), _p(out), x.numel(), a, b, drop_seed, thr, scale, _stream()),
          "ea_scale_dropout_bf16")
    return out


def colsum(X, out, M, N, ld):
    """out[n] += sum_m X[m*ld + n]  (fp32 accumulate into `out`)."""
    assert X.dtype == torch.bfloat16 and out.dtype == torch.float32
    check(_lib.lib().ea_colsum_bf16(_p(X), _p(out), M, N, ld, _stream()), "ea_colsum_bf16")
    return out


def colsum_ptr(X_ptr: int, out, M, N, ld):
    check(_lib.lib().ea_colsum_bf16(ctypes.c_void_p(X_ptr), _p(out), M, N, ld, _stream()), "ea_colsum_bf16")
    return out


def zero_rows(x, row_zero):
    M, C = x.shape
    check(_lib.lib().ea_zero_rows_bf16(_p(x), _p(row_zero), M, C, _stream()), "ea_zero_rows_bf16")
    return x


def relpos_q_prep(qkv, ldq, u, v, M, C, scaling, want_qv=True):
    qu = torch.empty(M, C, dtype=torch.bfloat16, device=qkv.device)
    qv = torch.empty(M, C, dtype=torch.bfloat16, device=qkv.device) if want_qv else None
    check(_lib.lib().ea_relpos_q_prep(_p(qkv), ldq, _p(u), _p(v), _p(qu), _p(qv), M, C, scaling, _stream()),
          "ea_relpos_q_prep")
    return qu, qv


def relpos_softmax_fwd(ac, bd, key_len, attn_mask, H, B, T, S, ld_ac, ld_bd, ld_p, causal=False, drop_p=0.0,
                       drop_seed=0):
    P = torch.empty(H * B * T, ld_p, dtype=torch.bfloat16, device=ac.device)
    Pd = torch.empty_like(P) if drop_p > 0 else None
    thr, scale = drop_params(drop_p)
    check(
        _lib.lib().ea_relpos_softmax_fwd(_p(ac), _p(bd), _p(key_len), _p(attn_mask), _p(P), _p(Pd), H, B, T, S, ld_ac,
                                         ld_bd, ld_p, int(causal), drop_seed, thr, scale, _stream()),
        "ea_relpos_softmax_fwd",
    )
    return P, (Pd if Pd is not None else P)


def flash_attention_supported(dh, T, S, relpos):
    return bool(_lib.lib().ea_flash_attention_supported(dh, T, S, int(relpos)))


def flash_keep_bits_buffer(H, B, T, device):
    """Device buffer for the attention-dropout keep bits of the rel-pos encoder kernels (csrc/flash_relpos.hip)."""
    return torch.empty(int(_lib.lib().ea_flash_keep_bits_bytes(H, B, T)) // 2, dtype=torch.int16, device=device)


def flash_attention_fwd(qu, qv, k, v, pp, key_len, H, B, T, S, ldq, ldkv, ldpp=0, causal=False, drop_p=0.0, drop_seed=0,
                        want_lse=True, want_bits=False):
    """Fused attention forward.  qu/qv: [B*T][ldq] bf16; k, v: tensors (views allowed) whose data_ptr is head 0 of row 0,
    rows ldkv apart.  Returns (out [B*T][H*64] bf16, lse [H*B][T] fp32 or None); with want_bits also the keep-bit buffer
    the rel-pos encoder kernels filled (None when dropout is off or the general kernels ran): hand it to flash_attention_bwd."""
    dh = 64
    out = torch.empty(B * T, H * dh, dtype=torch.bfloat16, device=qu.device)
    lse = torch.empty(H * B, T, dtype=torch.float32, device=qu.device) if want_lse else None
    thr, scale = drop_params(drop_p)
    bits = None
    if want_bits and thr and qv is not None and T == S and not causal:
        bits = flash_keep_bits_buffer(H, B, T, qu.device)
    check(
        _lib.lib().ea_flash_attention_fwd(_p(qu), _p(qv), ldq, _p(k), _p(v), ldkv, _p(pp), ldpp, _p(key_len), _p(out),
                                          H * dh, _p(lse), H, B, T, S, dh, int(causal), drop_seed, thr, scale, _p(bits), _stream()),
        "ea_flash_attention_fwd",
    )
    return (out, lse, bits) if want_bits else (out, lse)


def flash_attention_bwd(qu, qv, k, v, pp, key_len, out, dout, lse, dk, dv, H, B, T, S, ldq, ldkv, lddkv, ldpp=0, causal=False,
                        scaling=1.0, drop_p=0.0, drop_seed=0, keep_bits=None, dq=None, lddq=0):
    """Fused attention backward.  dk / dv: destination views (row stride lddkv).  keep_bits: the buffer flash_attention_fwd
    returned with want_bits (None: the general kernels re-evaluate the dropout hash).  dq (rel-pos only): destination view for
    t1 + t2 (row stride lddq).  Returns (t1, t2, dBD)."""
    dh = 64
    C = H * dh
    relpos = qv is not None
    t1 = torch.empty(B * T, C, dtype=torch.bfloat16, device=qu.device)
    t2 = torch.empty_like(t1) if relpos else None
    Rp = (2 * T - 1 + 7) // 8 * 8
    dBD = torch.empty(H * B * T, Rp, dtype=torch.bfloat16, device=qu.device) if relpos else None
    D = torch.empty(H * B, T, dtype=torch.float32, device=qu.device)
    thr, scale = drop_params(drop_p)
    check(
        _lib.lib().ea_flash_attention_bwd(_p(qu), _p(qv), ldq, _p(k), _p(v), ldkv, _p(pp), ldpp, _p(key_len), _p(out), _p(dout),
                                          C, _p(lse), _p(D), _p(t1), _p(t2), C, _p(dBD), Rp, _p(dk), _p(dv), lddkv, H, B, T, S,
                                          dh, int(causal), scaling, drop_seed, thr, scale, _p(keep_bits), _p(dq), lddq,
                                          _stream()),
        "ea_flash_attention_bwd",
    )
    return t1, t2, dBD


def relpos_softmax_bwd(P, dPd, H, B, T, S, ld_p, ld_dp, ld_bd, want_bd=True, drop_p=0.0, drop_seed=0):
    dAC = torch.empty(H * B * T, ld_p, dtype=torch.bfloat16, device=P.device)
    dBD = torch.empty(H * B * T, ld_bd, dtype=torch.bfloat16, device=P.device) if want_bd else None
    thr, scale = drop_params(drop_p)
    check(
        _lib.lib().ea_relpos_softmax_bwd(_p(P), _p(dPd), _p(dAC), _p(dBD), H, B, T, S, ld_p, ld_dp, ld_bd, drop_seed,
                                         thr, scale, _stream()),
        "ea_relpos_softmax_bwd",
    )
    return dAC, dBD


def add2_strided(a, lda, b, ldb, out, ldo, M, C, out_off=0):
    check(
        _lib.lib().ea_add2_strided_bf16(_p(a), lda, _p(b), ldb, ctypes.c_void_p(out.data_ptr() + 2 * out_off), ldo, M,
                                        C, _stream()),
        "ea_add2_strided_bf16",
    )
    return out


def glu_dwconv_fwd(Y, w, B, T, C, KW, stats):
    U = torch.empty(B * T, C, dtype=torch.bfloat16, device=Y.device)
    Z = torch.empty(B * T, C, dtype=torch.bfloat16, device=Y.device)
    check(_lib.lib().ea_glu_dwconv_fwd(_p(Y), _p(w), _p(U), _p(Z), _p(stats), B, T, C, KW, _stream()),
          "ea_glu_dwconv_fwd")
    return U, Z


def bn_finalize(stats, C, n, eps, momentum, running_mean=None, running_var=None):
    mean_rstd = torch.empty(2, C, dtype=torch.float32, device=stats.device)
    check(_lib.lib().ea_bn_finalize(_p(stats), _p(mean_rstd), _p(running_mean), _p(running_var), C, float(n), eps,
                                    momentum, _stream()), "ea_bn_finalize")
    return mean_rstd


def bn_from_running(running_mean, running_var, eps):
    C = running_mean.numel()
    mean_rstd = torch.empty(2, C, dtype=torch.float32, device=running_mean.device)
    check(_lib.lib().ea_bn_from_running(_p(running_mean), _p(running_var), _p(mean_rstd), C, eps, _stream()),
          "ea_bn_from_running")
    return mean_rstd


def bn_act_fwd(Z, mean_rstd, gamma, beta, act):
    M, C = Z.shape
    H = torch.empty_like(Z)
    check(_lib.lib().ea_bn_act_fwd(_p(Z), _p(mean_rstd), _p(gamma), _p(beta), _p(H), M, C, _ACT[act], _stream()),
          "ea_bn_act_fwd")
    return H


def conv1_bn_bwd(X, Z, dH, mean_rstd, gamma, beta, dgamma, dbeta, dW, dbias, B, T, F, CO, sy, sx, act, training=True):
    """BatchNorm(+act) backward of the first sub-sampler layer fused with conv1's weight / bias gradient (no dZ tensor)."""
    from . import functional as _F

    red = _F._pool_zeros((2, CO), torch.float32, Z.device)
    check(_lib.lib().ea_conv1_bn_bwd(_p(X), _p(Z), _p(dH), _p(mean_rstd), _p(gamma), _p(beta), _p(red), _p(dgamma), _p(dbeta), _p(dW),
                                     _p(dbias), B, T, F, CO, sy, sx, _ACT[act], int(training), _stream()), "ea_conv1_bn_bwd")


def bn_act_bwd(Z, dH, mean_rstd, gamma, beta, dgamma, dbeta, act, training=True):
    M, C = Z.shape
    from . import functional as _F  # (zero pool: one fill per update step instead of one per call)

    red = _F._pool_zeros((2, C), torch.float32, Z.device)
    dZ = torch.empty_like(Z)
    check(
        _lib.lib().ea_bn_act_bwd(_p(Z), _p(dH), _p(mean_rstd), _p(gamma), _p(beta), _p(red), _p(dZ), _p(dgamma),
                                 _p(dbeta), M, C, _ACT[act], int(training), _stream()),
        "ea_bn_act_bwd",
    )
    return dZ


def glu_dwconv_bwd(dZ, Y, U, w, dw, B, T, C, KW):
    dY = torch.empty(B * T, 2 * C, dtype=torch.bfloat16, device=Y.device)
    ws = torch.empty(int(_lib.lib().ea_dwconv_wgrad_workspace_bytes(B, T, C, KW)), dtype=torch.uint8, device=Y.device)
    check(_lib.lib().ea_glu_dwconv_bwd(_p(dZ), _p(Y), _p(U), _p(w), _p(dY), _p(dw), _p(ws), B, T, C, KW, _stream()),
          "ea_glu_dwconv_bwd")
    return dY


def conv1_fwd(X, W, bias, B, T, F, CO, sy, sx, stats):
    To, Fo = (T - 1) // sy + 1, (F - 1) // sx + 1
    Z = torch.empty(B * To * Fo, CO, dtype=torch.bfloat16, device=X.device)
    check(_lib.lib().ea_conv1_fwd(_p(X), _p(W), _p(bias), _p(Z), _p(stats), B, T, F, CO, sy, sx, _stream()),
          "ea_conv1_fwd")
    return Z


def conv1_wgrad(X, dZ, dW, dbias, B, T, F, CO, sy, sx):
    check(_lib.lib().ea_conv1_wgrad(_p(X), _p(dZ), _p(dW), _p(dbias), B, T, F, CO, sy, sx, _stream()), "ea_conv1_wgrad")


def conv3x3_fwd(X, W16, bias, B, T, F, Cin, Cout, sy, sx, stats=None):
    """Implicit-GEMM conv: X bf16 [B*T*F][Cin] (channels-last) -> Z bf16 [B*To*Fo][Cout]; W16 bf16 [Cout][9*Cin] (tap-major)."""
    To, Fo = (T - 1) // sy + 1, (F - 1) // sx + 1
    Z = torch.empty((B * To * Fo, Cout), dtype=torch.bfloat16, device=X.device)
    check(_lib.lib().ea_conv3x3_fwd(_p(X), _p(W16), _p(bias), _p(Z), _p(stats), B, T, F, Cin, Cout, sy, sx, _stream()), "ea_conv3x3_fwd")
    return Z


def conv3x3_dgrad(dZ, Wd16, B, T, F, Cin, Cout, sy, sx):
    """dX bf16 [B*T*F][Cin] from dZ bf16 [B*To*Fo][Cout]; Wd16 bf16 [Cin][9*Cout]."""
    dX = torch.empty((B * T * F, Cin), dtype=torch.bfloat16, device=dZ.device)
    check(_lib.lib().ea_conv3x3_dgrad(_p(dZ), _p(Wd16), _p(dX), B, T, F, Cin, Cout, sy, sx, _stream()), "ea_conv3x3_dgrad")
    return dX


def conv3x3_wgrad(X, dZ, dW, B, T, F, Cin, Cout, sy, sx, param_layout=False):
    """dW fp32 [Cout][9*Cin] += weight gradient of the 3x3 conv from X bf16 [B*T*F][Cin] and dZ bf16 [B*To*Fo][Cout];
    param_layout: dW is the parameter's gradient itself, [Cout][Cin][3][3]."""
    lib = _lib.lib()
    nb = lib.ea_conv3x3_wgrad_workspace_bytes(B, T, F, Cin, Cout, sy, sx)
    ws = torch.empty(nb, dtype=torch.uint8, device=X.device)
    fn = lib.ea_conv3x3_wgrad_param_layout if param_layout else lib.ea_conv3x3_wgrad
    check(fn(_p(X), _p(dZ), _p(dW), _p(ws), B, T, F, Cin, Cout, sy, sx, _stream()), "ea_conv3x3_wgrad")
    return dW


def im2col3x3(A, B, T, F, C, sy, sx):
    To, Fo = (T - 1) // sy + 1, (F - 1) // sx + 1
    col = torch.empty(B * To * Fo, 9 * C, dtype=torch.bfloat16, device=A.device)
    check(_lib.lib().ea_im2col3x3(_p(A), _p(col), B, T, F, C, sy, sx, _stream()), "ea_im2col3x3")
    return col


def col2im3x3(dcol, B, T, F, C, sy, sx):
    dA = torch.empty(B * T * F, C, dtype=torch.bfloat16, device=dcol.device)
    check(_lib.lib().ea_col2im3x3(_p(dcol), _p(dA), B, T, F, C, sy, sx, _stream()), "ea_col2im3x3")
    return dA


def colstats(X, stats):
    M, C = X.shape
    check(_lib.lib().ea_colstats_bf16(_p(X), _p(stats), M, C, _stream()), "ea_colstats_bf16")
    return stats


def log_softmax(x, M, V, ld):
    out = torch.empty(M, V, dtype=torch.float32, device=x.device)
    if x.dtype == torch.float32:
        check(_lib.lib().ea_log_softmax_f32(_p(x), ld, _p(out), M, V, _stream()), "ea_log_softmax_f32")
    else:
        assert x.dtype == torch.bfloat16
        check(_lib.lib().ea_log_softmax_bf16(_p(x), ld, _p(out), M, V, _stream()), "ea_log_softmax_bf16")
    return out


def ctc_loss_fwd(lprobs, targets, in_len, tgt_len, B, T, V, Lmax, blank):
    """lprobs fp32 [B][T][V] -> (nll[B], workspace holding the gathered lattice + alpha + beta)."""
    assert lprobs.dtype == torch.float32 and lprobs.is_contiguous()
    assert targets.dtype == torch.int32 and in_len.dtype == torch.int32 and tgt_len.dtype == torch.int32
    dev = lprobs.device
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    ws = torch.empty(int(_lib.lib().ea_ctc_workspace_bytes(B, T, Lmax)), dtype=torch.uint8, device=dev)
    check(
        _lib.lib().ea_ctc_loss(_p(lprobs), _p(targets), _p(in_len), _p(tgt_len), _p(nll), _p(ws), B, T, V, Lmax, blank,
                               _stream()),
        "ea_ctc_loss",
    )
    return nll, ws


def ctc_loss_grad(lprobs, ws, nll, targets, in_len, tgt_len, B, T, V, Lmax, blank, ld_out=None, grad_bf16=True,
                  grad_scale=1.0, grad_scale_dev=None, zero_infinity=True):
    ld_out = V if ld_out is None else ld_out
    dl = torch.empty(B * T, ld_out, dtype=torch.bfloat16 if grad_bf16 else torch.float32, device=lprobs.device)
    check(
        _lib.lib().ea_ctc_grad(_p(lprobs), _p(ws), _p(nll), _p(targets), _p(in_len), _p(tgt_len), _p(dl), ld_out,
                               int(grad_bf16), B, T, V, Lmax, blank, grad_scale, _p(grad_scale_dev),
                               int(zero_infinity), _stream()),
        "ea_ctc_grad",
    )
    return dl


SMOOTHING = {"uniform": 0, "unigram": 1, "temporal": 2}


def label_smoothed_ce(logits, ld, target, M, V, pad_idx, eps, want_grad=True, grad_bf16=True, grad_scale=1.0,
                      grad_ld=None, smoothing="uniform", prior=None, tgt_len=0):
    assert target.dtype == torch.int32
    dev = logits.device
    out = torch.zeros(2, dtype=torch.float32, device=dev)
    dl = None
    grad_ld = V if grad_ld is None else grad_ld
    if want_grad:
        dl = torch.zeros(M, grad_ld, dtype=torch.bfloat16 if grad_bf16 else torch.float32, device=dev)
    check(
        _lib.lib().ea_label_smoothed_ce(_p(logits), ld, int(logits.dtype == torch.bfloat16), _p(target), _p(out),
                                        _p(dl), grad_ld, int(grad_bf16), M, V, pad_idx, eps, grad_scale, SMOOTHING[smoothing],
                                        _p(prior), tgt_len, _stream()),
        "ea_label_smoothed_ce",
    )
    return out, dl


def fbank_batch(wav, offsets, B, tables, cmvn_mean, cmvn_std, Tmax, nmel=80, frame_len=400, frame_shift=160,
                preemph=0.97, log_floor=1.1920928955078125e-07, want_sum=True):
    dev = wav.device
    feat = torch.empty(B, Tmax, nmel, dtype=torch.float32, device=dev)
    utt_sum = torch.zeros(B, dtype=torch.float32, device=dev) if want_sum else None
    out_len = torch.empty(B, dtype=torch.int32, device=dev)
    assert wav.dtype in (torch.float32, torch.int16)
    fn = _lib.lib().ea_fbank_batch if wav.dtype == torch.float32 else _lib.lib().ea_fbank_batch_i16
    check(
        fn(_p(wav), _p(offsets), B, _p(tables["window"]), _p(tables["twiddle"]), _p(tables["mel_start"]), _p(tables["mel_len"]),
           _p(tables["mel_woff"]), _p(tables["mel_w"]), _p(cmvn_mean), _p(cmvn_std), _p(feat), _p(utt_sum), _p(out_len), Tmax, nmel,
           frame_len, frame_shift, preemph, log_floor, _stream()),
        "ea_fbank_batch",
    )
    return feat, out_len, utt_sum


def feature_stats(feat, lengths, acc):
    """acc fp64 [2*nmel+1] += (sum, sum of squares, frames) over the valid frames of feat fp32 [B][Tmax][nmel]."""
    B, Tmax, nmel = feat.shape
    assert acc.dtype == torch.float64 and acc.numel() == 2 * nmel + 1 and feat.dtype == torch.float32 and feat.is_contiguous()
    check(_lib.lib().ea_feature_stats(_p(feat), _p(lengths), _p(acc), B, Tmax, nmel, _stream()), "ea_feature_stats")
    return acc


def specaugment(feat, lengths, utt_sum, fmask, tmask, use_mean=True, mask_value=0.0):
    B, Tmax, nmel = feat.shape
    nf = fmask.shape[1] if fmask is not None else 0
    nt = tmask.shape[1] if tmask is not None else 0
    check(
        _lib.lib().ea_specaugment(_p(feat), _p(lengths), _p(utt_sum), _p(fmask), _p(tmask), nf, nt, B, Tmax, nmel,
                                  int(use_mean), mask_value, _stream()),
        "ea_specaugment",
    )
    return feat


def grad_sumsq(g, out):
    check(_lib.lib().ea_grad_sumsq(_p(g), g.numel(), _p(out), _stream()), "ea_grad_sumsq")
    return out


def clip_coef(sumsq, pre_scale, max_norm, coef, denom_dev=None):
    check(_lib.lib().ea_clip_coef(_p(sumsq), pre_scale, _p(denom_dev), max_norm, _p(coef), _stream()), "ea_clip_coef")
    return coef


def adam_step(p, g, m, v, p_bf16, coef, lr, beta1, beta2, eps, weight_decay, step, zero_grad=True):
    check(
        _lib.lib().ea_adam_step(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), _p(coef), lr, beta1, beta2, eps,
                                weight_decay, step, int(zero_grad), _stream()),
        "ea_adam_step",
    )


def ctc_greedy_decode(x, in_len, B, T, V, blank, pad, ld=None, want_align=True):
    """x: [B*T][V] fp32/bf16 log-probs (batch-major).  Returns tokens [B][T] (pad filled), lengths [B], scores [B], align [B][T]."""
    dev = x.device
    ld = x.stride(0) if ld is None else ld
    best = torch.empty(B * T, dtype=torch.int32, device=dev)
    bestv = torch.empty(B * T, dtype=torch.float32, device=dev)
    tokens = torch.empty(B, T, dtype=torch.int32, device=dev)
    align = torch.zeros(B, T, dtype=torch.int32, device=dev) if want_align else None
    out_len = torch.empty(B, dtype=torch.int32, device=dev)
    score = torch.empty(B, dtype=torch.float32, device=dev)
    check(
        _lib.lib().ea_ctc_greedy_decode(_p(x), ld, int(x.dtype == torch.bfloat16), _p(in_len), _p(best), _p(bestv), _p(tokens),
                                        _p(align), _p(out_len), _p(score), B, T, V, blank, pad, _stream()),
        "ea_ctc_greedy_decode",
    )
    return tokens, out_len, score, align


def embedding_fwd(tokens, positions, W, pos_table, scale):
    M, C = tokens.numel(), W.shape[1]
    out = torch.empty(M, C, dtype=torch.bfloat16, device=W.device)
    check(_lib.lib().ea_embedding_fwd(_p(tokens), _p(positions), _p(W), _p(pos_table), _p(out), M, C, scale, _stream()),
          "ea_embedding_fwd")
    return out


def embedding_bwd(tokens, dy, dW, scale, pad_idx):
    M, C = dy.shape
    check(_lib.lib().ea_embedding_bwd(_p(tokens), _p(dy), _p(dW), M, C, scale, pad_idx, _stream()), "ea_embedding_bwd")
    return dW


def decode_attention(q, Kc, Vc_off, kv_row, lens, N, H, dh, row_stride, ldkv, koff, voff, max_len):
    """One query per hypothesis.  q bf16 [N][C]; Kc: bf16 cache tensor holding K and V (offsets koff / voff)."""
    out = torch.empty_like(q)
    check(
        _lib.lib().ea_decode_attention(_p(q), _p(Kc), _p(Kc), _p(kv_row), _p(lens), _p(out), N, H, dh, q.stride(0), row_stride,
                                       ldkv, koff, voff, max_len, _stream()),
        "ea_decode_attention",
    )
    return out


def kv_append_reorder(old_cache, new_cache, kv_new, parent, N, L, Lmax, W):
    check(_lib.lib().ea_kv_append_reorder(_p(old_cache), _p(new_cache), _p(kv_new), _p(parent), N, L, Lmax, W, _stream()),
          "ea_kv_append_reorder")
    return new_cache


def beam_mask_rows(lprobs, pad, unk, eos, unk_penalty=0.0, only_eos=False, forbid_eos=False, eos_factor=None):
    N, V = lprobs.shape
    assert lprobs.dtype == torch.float32 and lprobs.is_contiguous()
    check(
        _lib.lib().ea_beam_mask_rows(_p(lprobs), N, V, pad, unk, eos, unk_penalty, int(only_eos), int(forbid_eos),
                                     0.0 if eos_factor is None else float(eos_factor), int(eos_factor is not None), _stream()),
        "ea_beam_mask_rows",
    )
    return lprobs


def beam_topk(lprobs, prev_scores, bsz, beam, nbeam_used, k):
    V = lprobs.shape[1]
    dev = lprobs.device
    cs = torch.empty(bsz, k, dtype=torch.float32, device=dev)
    ct = torch.empty(bsz, k, dtype=torch.int32, device=dev)
    cb = torch.empty(bsz, k, dtype=torch.int32, device=dev)
    check(_lib.lib().ea_beam_topk(_p(lprobs), _p(prev_scores), bsz, beam, nbeam_used, V, k, _p(cs), _p(ct), _p(cb), _stream()),
          "ea_beam_topk")
    return cs, ct, cb


def _rnnt_pitch(logits):
    """(B, T, U1, V, ld) of a [B][T][U1][V] logits tensor that is contiguous or a column-slice view of rows with pitch ld."""
    B, T, U1, V = logits.shape
    assert logits.dtype in (torch.float32, torch.bfloat16) and logits.stride(3) == 1
    ld = logits.stride(2)
    assert ld >= V and logits.stride(1) == U1 * ld and logits.stride(0) == T * U1 * ld, "logits rows must be equally spaced"
    return B, T, U1, V, ld


def rnnt_loss_fwd(logits, targets, logit_lengths, target_lengths, blank):
    """logits fp32 or bf16 [B][T][U1][V] (rows may be padded: a [..., :V] view of a wider buffer); returns (loss [B], workspace)."""
    B, T, U1, V, ld = _rnnt_pitch(logits)
    Umax = targets.shape[1]
    assert U1 == Umax + 1
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    ws = torch.empty(int(_lib.lib().ea_rnnt_workspace_bytes(B, T, U1)), dtype=torch.uint8, device=logits.device)
    check(_lib.lib().ea_rnnt_loss(_p(logits), int(logits.dtype == torch.bfloat16), _p(targets), _p(logit_lengths), _p(target_lengths), _p(loss), _p(ws), B, T, U1, V,
                                  ld, Umax, blank, _stream()), "ea_rnnt_loss")
    return loss, ws


def rnnt_loss_grad(logits, targets, logit_lengths, target_lengths, loss, ws, blank, grad_scale_dev=None, grad_bf16=False):
    """Gradient with the logits' own row pitch (pad columns zero); returned as the [..., :V] view when the rows are padded."""
    B, T, U1, V, ld = _rnnt_pitch(logits)
    grad = torch.empty((B, T, U1, ld), dtype=torch.bfloat16 if grad_bf16 else torch.float32, device=logits.device)
    check(_lib.lib().ea_rnnt_grad(_p(logits), int(logits.dtype == torch.bfloat16), _p(targets), _p(logit_lengths), _p(target_lengths), _p(loss), _p(ws), _p(grad),
                                  int(grad_bf16), B, T, U1, V, ld, targets.shape[1], blank, 1.0, _p(grad_scale_dev), _stream()),
          "ea_rnnt_grad")
    return grad if ld == V else grad[..., :V]


def joint_rnnt_supported(Z, w16) -> bool:
    """Shapes the fused joint + RNN-T kernels take (csrc/joint_rnnt.hip): joint dim a multiple of 64, 16-byte aligned operands."""
    J = Z.shape[1]
    return (Z.is_cuda and Z.dtype == torch.bfloat16 and w16.dtype == torch.bfloat16 and J % 64 == 0 and Z.is_contiguous()
            and w16.is_contiguous() and Z.data_ptr() % 16 == 0 and w16.data_ptr() % 16 == 0 and Z.shape[0] * J < 2 ** 31)


def joint_rnnt_loss_fwd(Z, w16, bias, targets, logit_lengths, target_lengths, B, T, U1, blank):
    """Per-utterance RNN-T loss of logits = Z w16^T + bias WITHOUT materialising them (fp32 accumulators -> log-sum-exp per vocabulary
    tile -> alpha / beta).  Z bf16 [B*T*U1][J].  Returns (loss [B], workspace kept for joint_rnnt_loss_grad)."""
    V, J = w16.shape
    Umax = targets.shape[1]
    assert U1 == Umax + 1 and Z.shape[0] == B * T * U1
    loss = torch.empty(B, dtype=torch.float32, device=Z.device)
    ws = torch.empty(int(_lib.lib().ea_joint_rnnt_workspace_bytes(B, T, U1, V)), dtype=torch.uint8, device=Z.device)
    check(_lib.lib().ea_joint_rnnt_loss(_p(Z), _p(w16), _p(bias), _p(targets), _p(logit_lengths), _p(target_lengths), _p(loss), _p(ws),
                                        B, T, U1, V, J, Umax, blank, _stream()), "ea_joint_rnnt_loss")
    return loss, ws


def joint_rnnt_loss_grad(Z, w16, bias, targets, logit_lengths, target_lengths, loss, ws, B, T, U1, blank, ld, grad_scale_dev=None):
    """d loss / d logits as bf16 [B*T*U1][ld] (pad columns V .. ld - 1 zero), the logits recomputed tile by tile."""
    V, J = w16.shape
    dl = torch.empty(B * T * U1, ld, dtype=torch.bfloat16, device=Z.device)
    check(_lib.lib().ea_joint_rnnt_grad(_p(Z), _p(w16), _p(bias), _p(targets), _p(logit_lengths), _p(target_lengths), _p(loss), _p(ws),
                                        _p(dl), ld, B, T, U1, V, J, targets.shape[1], blank, 1.0, _p(grad_scale_dev), _stream()),
          "ea_joint_rnnt_grad")
    return dl


# ------------------------------------------------------------------------------------------------ LSTM
def lstm_cell_fwd(gates_pre, c_prev, c_out, h_f32, h_bf16, ldh, gates_act, B, H, keep_row=None, h_prev_f32=None, ldg=None,
                  frozen_out_zero=False):
    check(_lib.lib().ea_lstm_cell_fwd(_p(gates_pre), ldg if ldg is not None else 4 * H, _p(c_prev), _p(c_out), _p(h_f32), _p(h_bf16),
                                      ldh, _p(gates_act), _p(keep_row), _p(h_prev_f32), int(frozen_out_zero), B, H, _stream()),
          "ea_lstm_cell_fwd")


def lstm_cell_bwd(dh_bf16, ld_dh, dh_f32, dc_in, gates_act, c_prev, c, dgates, lddg, dc_prev, B, H, frozen=None):
    check(_lib.lib().ea_lstm_cell_bwd(_p(dh_bf16), ld_dh, _p(dh_f32), _p(dc_in), _p(gates_act), _p(c_prev), _p(c), _p(dgates), lddg,
                                      _p(dc_prev), _p(frozen), B, H, _stream()), "ea_lstm_cell_bwd")


def wgrad_group(problems):
    """ea_wgrad_group: for every problem (dy bf16 [M][ld_dy], x bf16 [M][ld_x], dW fp32 [N][ldw] accumulated in place, dbias fp32 [N]
    or None accumulated in place, M, N, K, ld_dy, ld_x, ldw) one grid computes dW += dy^T x and dbias += colsum(dy)."""
    import ctypes

    assert 0 < len(problems) <= 16
    grp = _lib.EaWgradGroup()
    grp.count = len(problems)
    for i, (dy, x, dW, db, M, N, Kk, ld_dy, ld_x, ldw) in enumerate(problems):
        q = grp.p[i]
        q.dy, q.x, q.dW, q.dbias = dy.data_ptr(), x.data_ptr(), dW.data_ptr(), (db.data_ptr() if db is not None else None)
        q.M, q.N, q.K, q.ld_dy, q.ld_x, q.ldw = M, N, Kk, ld_dy, ld_x, ldw
    check(_lib.lib().ea_wgrad_group(ctypes.byref(grp), _stream()), "ea_wgrad_group")


def lstm_seq_supported(B, H):
    return bool(_lib.lib().ea_lstm_seq_supported(int(B), int(H)))


def lstm_seq_fwd(gx, w_hh16, h0_16, c0, frozen, hs, cs, act, h_last, counter, B, U, H, reverse=False, frozen_out_zero=False):
    check(_lib.lib().ea_lstm_seq_fwd(_p(gx), _p(w_hh16), _p(h0_16), _p(c0), _p(frozen), _p(hs), _p(cs), _p(act), _p(h_last),
                                     _p(counter), B, U, H, int(reverse), int(frozen_out_zero), _stream()), "ea_lstm_seq_fwd")


def lstm_seq_bwd(dhs, dh_last, dc_last, act, cs, c0, frozen, w_hhT16, dG, dh0, dc0, counter, B, U, H, reverse=False):
    check(_lib.lib().ea_lstm_seq_bwd(_p(dhs), _p(dh_last), _p(dc_last), _p(act), _p(cs), _p(c0), _p(frozen), _p(w_hhT16), _p(dG),
                                     _p(dh0), _p(dc0), _p(counter), B, U, H, int(reverse), _stream()), "ea_lstm_seq_bwd")


def bahdanau_fwd(qp, key, value, nv, bias, lens, T, B, ctx=None, ldc=None, kv_col=None, Bkv=0):
    A, Cv = qp.shape[1], value.shape[-1]
    p = torch.empty(T, B, dtype=torch.float32, device=qp.device)
    if ctx is None:
        ctx = torch.empty(B, Cv, dtype=torch.bfloat16, device=qp.device)
        ldc = Cv
    check(_lib.lib().ea_bahdanau_fwd(_p(qp), _p(key), _p(value), _p(nv), _p(bias), _p(lens), _p(p), _p(ctx), ldc, T, B, A, Cv,
                                     _p(kv_col), Bkv, _stream()), "ea_bahdanau_fwd")
    return p, ctx


def bahdanau_bwd(dctx, qp, key, value, nv, bias, lens, p, dkey_acc, dvalue_acc, dnv_acc, dbias_acc, T, B):
    A, Cv = qp.shape[1], value.shape[-1]
    dqp = torch.empty(B, A, dtype=torch.bfloat16, device=qp.device)
    check(_lib.lib().ea_bahdanau_bwd(_p(dctx), dctx.stride(0), _p(qp), _p(key), _p(value), _p(nv), _p(bias), _p(lens), _p(p), _p(dqp),
                                     _p(dkey_acc), _p(dvalue_acc), _p(dnv_acc), _p(dbias_acc), T, B, A, Cv, _stream()), "ea_bahdanau_bwd")
    return dqp


def gather_rows(src, parent, out=None):
    """out[n] = src[parent[n]] for a contiguous [N][...] fp32 / bf16 tensor; parent int32 on device."""
    assert src.is_contiguous() and src.element_size() in (2, 4)
    N = parent.numel()
    W = src.numel() // src.shape[0]
    if out is None:
        out = torch.empty((N,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    check(_lib.lib().ea_gather_rows(_p(src), _p(out), _p(parent), N, W, src.element_size(), _stream()), "ea_gather_rows")
    return out


def joint_add_relu(E, D, B, T, U1):
    """Z [B*T*U1][J] bf16 = relu(E[b,t] + D[b,u]).  E, D fp32 (the model's path: sum and ReLU in fp32 as in the reference's
    autocast run, only Z is rounded) or both bf16."""
    J = E.shape[1]
    assert E.dtype == D.dtype and E.is_contiguous() and D.is_contiguous()
    Z = torch.empty(B * T * U1, J, dtype=torch.bfloat16, device=E.device)
    fn = _lib.lib().ea_joint_add_relu_f32 if E.dtype == torch.float32 else _lib.lib().ea_joint_add_relu
    check(fn(_p(E), _p(D), _p(Z), B, T, U1, J, _stream()), "ea_joint_add_relu")
    return Z


def joint_reduce(dZ, B, T, U1, out_f32=False):
    J = dZ.shape[1]
    dt = torch.float32 if out_f32 else torch.bfloat16
    dE = torch.empty(B * T, J, dtype=dt, device=dZ.device)
    dD = torch.empty(B * U1, J, dtype=dt, device=dZ.device)
    fn = _lib.lib().ea_joint_reduce_f32 if out_f32 else _lib.lib().ea_joint_reduce
    check(fn(_p(dZ), _p(dE), _p(dD), B, T, U1, J, _stream()), "ea_joint_reduce")
    return dE, dD
