"""Parity checks of the HIP path against the oracle (oracle/) and the golden fixtures produced by
the reference's own code (tests/golden/).  Shared by the `-m gpu` tests and __graft_entry__.smoke().
Every HIP call goes through the C ABI (espresso_amd/_lib.py -> libespresso_amd.so)."""
import math
import os

import numpy as np
import ctypes
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# ------------------------------------------------------------------ GEMM
def check_gemm(M, N, K, a_ks, b_ks, batch=1, bias=False, act=None, resid=False, c_f32=False, seed=0, splitk=1, variant=0):
    from espresso_amd import _lib
    from espresso_amd import kernels as Kk

    _lib.lib().ea_set_gemm_variant(variant)

    g = torch.Generator(device="cpu").manual_seed(seed)
    A = bf(torch.randn(batch, K, M, generator=g) if a_ks else torch.randn(batch, M, K, generator=g)).to(DEV)
    B = bf(torch.randn(batch, K, N, generator=g) if b_ks else torch.randn(batch, N, K, generator=g)).to(DEV)
    Af = A.float().transpose(1, 2) if a_ks else A.float()
    Bf = B.float().transpose(1, 2) if b_ks else B.float()
    ref = torch.bmm(Af, Bf.transpose(1, 2))
    bvec = torch.randn(N, generator=g).to(DEV) if bias else None
    if bias:
        ref = ref + bvec
    if act == "relu":
        ref = torch.relu(ref)
    elif act == "silu":
        ref = torch.nn.functional.silu(ref)
    R = None
    if resid:
        R = bf(torch.randn(batch, M, N, generator=g)).to(DEV)
        ref = ref * 0.5 + R.float()
    C = torch.full((batch, M, N), 0.0 if splitk > 1 else float("nan"), dtype=torch.float32 if c_f32 else torch.bfloat16, device=DEV)
    Kk.gemm(A, B, C, M, N, K, lda=M if a_ks else K, ldb=N if b_ks else K, ldc=N, a_kstrided=a_ks, b_kstrided=b_ks,
            batch=batch, zdiv=1, sA=(M * K, 0), sB=(N * K, 0), sC=(M * N, 0), bias=bvec, act=act,
            out_scale=0.5 if resid else 1.0, resid=R, ldr=N, sR=(M * N, 0), splitk=splitk)
    torch.cuda.synchronize()
    _lib.lib().ea_set_gemm_variant(0)
    return rel_err(C, ref)


def _bf_ulp_ok(got, want_f32, ulps=1.0, atol=0.0):
    """got bf16 tensor vs the fp32 value it should be the rounding of: within `ulps` bf16 steps of that value's magnitude
    (+ `atol` where the value is a difference of larger terms: fp32 association / fma contraction differ there)."""
    got, want = got.float().cpu(), want_f32.float().cpu()
    step = torch.exp2(torch.floor(torch.log2(want.abs().clamp_min(2.0 ** -60))) - 7)
    return float((((got - want).abs() - atol).clamp_min(0) / step).max()) <= ulps + 1e-3


def check_gemm_dropout(M=777, N=520, K=192, batch=2, p=0.1, seed=0, variant=0):
    """The dropout of the GEMM epilogue (csrc/gemm.hip `has_drop`: FairseqDropout of conformer_layer.py:144,146 and of the
    attention / conv-module output projections) against the restated mask stream (oracle/dropout_ref.py): keep decisions bit
    for bit, keep rate, the 1/(1-p) scale, and the position of the dropout relative to bias / activation / scale / residual in
    the four epilogue forms the layers use."""
    from espresso_amd import _lib
    from espresso_amd import kernels as Kk
    from oracle import dropout_ref as D

    _lib.lib().ea_set_gemm_variant(variant)
    g = torch.Generator(device="cpu").manual_seed(seed)
    # positive operands: every accumulator is > 0, so "output == 0" <=> "dropped"
    A = bf(torch.rand(batch, M, K, generator=g) + 0.1).to(DEV)
    Bm = bf(torch.rand(batch, N, K, generator=g) + 0.1).to(DEV)
    bvec = (torch.rand(N, generator=g) + 0.5).to(DEV)
    R = bf(torch.randn(batch, M, N, generator=g)).to(DEV)
    dseed = (0x5EED << 32) | (1234 + seed)
    kw = dict(lda=K, ldb=K, ldc=N, batch=batch, zdiv=1, sA=(M * K, 0), sB=(N * K, 0), sC=(M * N, 0))
    mask = D.scale_mask(dseed, (batch, M, N), p)  # index (z*M + m)*N + n, value 0 or 1/(1-p)
    keep = mask != 0
    res = {"keep_rate": float(keep.float().mean())}

    def run(drop, **extra):
        C = torch.full((batch, M, N), float("nan"), dtype=extra.pop("cdtype", torch.float32), device=DEV)
        Kk.gemm(A, Bm, C, M, N, K, drop_p=p if drop else 0.0, drop_seed=dseed if drop else 0, **kw, **extra)
        torch.cuda.synchronize()
        return C

    # (1) plain fp32 output: y = acc * keep / (1-p)
    y0, y = run(False).cpu(), run(True).cpu()
    res["mask_bits_equal"] = bool(((y != 0) == keep).all())
    res["scale_err"] = float(((y - y0 * mask).abs() / y0.abs()).max())
    # (2) FFN first GEMM: C = acc + bias (pre-activation, not dropped), C2 = drop(act(acc + bias)) in bf16
    C2 = torch.full((batch, M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    z = run(True, bias=bvec, act="silu", C2=C2, ldc2=N, cdtype=torch.bfloat16)
    pre = y0 + bvec.cpu()
    res["c2_pre_ok"] = _bf_ulp_ok(z, pre)
    res["c2_drop_ok"] = _bf_ulp_ok(C2, torch.nn.functional.silu(pre) * mask) and bool(((C2.cpu().float() != 0) == keep).all())
    # (3) FFN second GEMM: out = 0.5 * drop(acc + bias) + residual
    o = run(True, bias=bvec, out_scale=0.5, resid=R, ldr=N, sR=(M * N, 0), cdtype=torch.bfloat16)
    res["resid_ok"] = _bf_ulp_ok(o, 0.5 * (pre * mask) + R.float().cpu(), atol=1e-4)
    # (4) backward through activation + activation dropout: dz = drop(acc) * act'(aux)
    aux = bf(torch.randn(batch, M, N, generator=g)).to(DEV)
    d = run(True, aux=aux, ldaux=N, sX=(M * N, 0), act="silu", cdtype=torch.bfloat16)
    xs = aux.float().cpu()
    sg = torch.sigmoid(xs)
    res["dact_ok"] = _bf_ulp_ok(d, y0 * mask * (sg * (1 + xs * (1 - sg))), ulps=2.0)
    _lib.lib().ea_set_gemm_variant(0)
    return res


def check_elementwise_dropout(M=1501, C=512, p=0.1, seed=0):
    """`ea_scale_dropout_bf16` (FairseqDropout backward + the 0.5 FFN scale), the LayerNorm kernel's fused output dropout
    (speech_transformer_encoder.py:348-350), and the LayerNorm backward's second output (`ea_layernorm_bwd_dx2`) against the
    restated mask stream."""
    import ctypes

    from espresso_amd import _lib
    from espresso_amd import kernels as Kk
    from oracle import dropout_ref as D

    g = torch.Generator(device="cpu").manual_seed(seed)
    x = bf(torch.randn(M, C, generator=g) + 3.0).to(DEV)  # away from zero: "== 0" <=> dropped
    yb = bf(torch.randn(M, C, generator=g)).to(DEV)
    s1, s2 = (0x5EED << 32) | 77, (0x5EED << 32) | 78
    m1, m2 = D.scale_mask(s1, (M, C), p), D.scale_mask(s2, (M, C), p)
    res = {}
    o = Kk.scale_dropout(x, a=0.5, y=yb, b=1.0, drop_p=p, drop_seed=s1)
    torch.cuda.synchronize()
    res["scale_dropout_ok"] = _bf_ulp_ok(o, 0.5 * x.float().cpu() * m1 + yb.float().cpu(), atol=2e-6)
    o = Kk.scale_dropout(x, a=1.0, drop_p=p, drop_seed=s1)
    res["scale_dropout_bits"] = bool(((o.float().cpu() != 0) == (m1 != 0)).all())
    # LayerNorm forward with output dropout and zeroed rows
    gam, bet = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.rand(C, generator=g) + 4.0).to(DEV)
    rz = (torch.rand(M, generator=g) < 0.1).to(torch.uint8).to(DEV)
    yl, mean, rstd = Kk.layernorm_fwd(x, gam, bet, 1e-5, rz, p, s2)
    ref = torch.nn.functional.layer_norm(x.float().cpu(), (C,), gam.cpu(), bet.cpu(), 1e-5) * m2 * (1 - rz.cpu().float()).unsqueeze(1)
    res["ln_drop_ok"] = _bf_ulp_ok(yl, ref)
    res["ln_drop_bits"] = bool(((yl.float().cpu() != 0) == ((m2 != 0) & (rz.cpu() == 0).unsqueeze(1))).all())
    # LayerNorm backward through the same dropout: dx of y = drop(LN(x)) equals LN-backward of (dy * mask)
    dy = bf(torch.randn(M, C, generator=g)).to(DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = Kk.layernorm_bwd(x, dy, gam, mean, rstd, dg, db, rz, p, s2)
    xr = x.float().cpu().requires_grad_(True)
    gr, br = gam.cpu().clone().requires_grad_(True), bet.cpu().clone().requires_grad_(True)
    (torch.nn.functional.layer_norm(xr, (C,), gr, br, 1e-5) * m2 * (1 - rz.cpu().float()).unsqueeze(1) * dy.float().cpu()).sum().backward()
    res["ln_bwd_dx"] = rel_err(dx, xr.grad)
    res["ln_bwd_dgamma"] = rel_err(dg, gr.grad)
    res["ln_bwd_dbeta"] = rel_err(db, br.grad)
    # second output of the LayerNorm backward: out2 = a2 * dropout(dx; seed2)
    lib = _lib.lib()
    ws = torch.empty(lib.ea_layernorm_bwd_workspace_bytes(M, C), dtype=torch.uint8, device=DEV)
    dx2, out2 = torch.empty_like(x), torch.empty_like(x)
    dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    thr, sc = Kk.drop_params(p)
    _lib.check(lib.ea_layernorm_bwd_dx2(x.data_ptr(), dy.data_ptr(), gam.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx2.data_ptr(),
                                        dg2.data_ptr(), db2.data_ptr(), M, C, None, ws.data_ptr(), out2.data_ptr(), 0.5, s1, thr, sc,
                                        Kk._stream()), "ln_bwd_dx2")
    torch.cuda.synchronize()
    res["ln_bwd_out2_ok"] = _bf_ulp_ok(out2, 0.5 * dx2.float().cpu() * m1, ulps=1.0)
    return res


def check_gemm_query_split(M=333, C=256, Kin=192, with_v=True, glds=1, seed=0):
    """The QKV projection's query-split epilogue (EaGemmParams.q_u: columns [0, C) leave as (q + pos_u) * s and (q + pos_v) * s)
    against the two-step path it replaces — plain projection, then ea_relpos_q_prep: bit-identical q_u / q_v, and the k / v
    columns of the packed output unchanged."""
    from espresso_amd import _lib
    from espresso_amd import kernels as Kk

    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = bf(torch.randn(M, Kin, generator=g)).to(DEV)
    W = bf(torch.randn(3 * C, Kin, generator=g) * Kin ** -0.5).to(DEV)
    bias = torch.randn(3 * C, generator=g).to(DEV)
    u = torch.randn(C, generator=g).to(DEV)
    v = torch.randn(C, generator=g).to(DEV) if with_v else None
    s = 0.125
    old = lib.ea_set_gemm_glds(glds)
    try:
        ref = torch.zeros(M, 3 * C, dtype=torch.bfloat16, device=DEV)
        Kk.gemm(A, W, ref, M, 3 * C, Kin, lda=Kin, ldb=Kin, ldc=3 * C, bias=bias)
        qu_ref, qv_ref = Kk.relpos_q_prep(ref, 3 * C, u, v, M, C, s, want_qv=with_v)
        out = torch.full((M, 3 * C), 7.0, dtype=torch.bfloat16, device=DEV)
        qu = torch.zeros(M, C, dtype=torch.bfloat16, device=DEV)
        qv = torch.zeros(M, C, dtype=torch.bfloat16, device=DEV) if with_v else None
        Kk.gemm(A, W, out, M, 3 * C, Kin, lda=Kin, ldb=Kin, ldc=3 * C, bias=bias, qsplit=(qu, qv, u, v, C, C, s))
        torch.cuda.synchronize()
    finally:
        lib.ea_set_gemm_glds(old)
    return {"qu_equal": bool((qu == qu_ref).all()), "qv_equal": bool((qv == qv_ref).all()) if with_v else True,
            "kv_equal": bool((out[:, C:] == ref[:, C:]).all()), "q_third_untouched": bool((out[:, :C] == 7.0).all())}


def check_gemm_w8(cfg, kind, M=777, N=640, K=192, seed=0):
    """The 8-wavefront large-tile kernels (csrc/gemm_w8.hip, forced tile configuration `cfg`) against the 4-wave kernels on the
    same launch: every output bit for bit (same products, same rounding points, same accumulation order along k; the epilogue runs
    from registers on transposed accumulators).  kind: plain | bias_relu | qsplit | act2 (bias + SiLU + dropout, two outputs) |
    aux (dropout * SiLU'(aux)) | resid (bias + dropout, 0.5 y + residual).  Ragged M; N a multiple of 128, not always of 256."""
    from espresso_amd import _lib
    from espresso_amd import kernels as Kk

    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    extra = bf(torch.randn(M, N, generator=g)).to(DEV)

    def run():
        C = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        outs = [C]
        kw = dict(lda=K, ldb=K, ldc=N)
        if kind == "bias_relu":
            kw.update(bias=bias, act="relu")
        elif kind == "qsplit":
            qn = 256 if N >= 384 else 128
            qu = torch.full((M, qn), 7.0, dtype=torch.bfloat16, device=DEV)
            qv = torch.full((M, qn), 7.0, dtype=torch.bfloat16, device=DEV)
            kw.update(bias=bias, qsplit=(qu, qv, bias[:qn].contiguous(), (bias[:qn] * 0.5).contiguous(), qn, qn, 0.125))
            outs += [qu, qv]
        elif kind == "act2":
            C2 = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
            kw.update(bias=bias, act="silu", C2=C2, ldc2=N, drop_p=0.1, drop_seed=1234 + seed)
            outs.append(C2)
        elif kind == "aux":
            kw.update(aux=extra, ldaux=N, act="silu", drop_p=0.1, drop_seed=77 + seed)
        elif kind == "resid":
            kw.update(bias=bias, resid=extra, ldr=N, out_scale=0.5, drop_p=0.1, drop_seed=9 + seed)
        Kk.gemm(A, W, C, M, N, K, **kw)
        torch.cuda.synchronize()
        return outs

    old = lib.ea_set_gemm_w8(0)
    try:
        ref = run()
        lib.ea_set_gemm_w8(cfg)
        got = run()
    finally:
        lib.ea_set_gemm_w8(old)
    return {"equal": all(bool(torch.equal(a, b)) for a, b in zip(got, ref)), "finite": all(bool(torch.isfinite(a.float()).all()) for a in got),
            "written": any(bool((a != 7.0).any()) for a in got)}


def check_conv3x3(Cin=64, Cout=128, sy=2, sx=2, B=2, T=37, F=21, seed=0):
    """Implicit-GEMM 3x3 convolution (forward + BatchNorm sums, data gradient) vs torch conv2d / its autograd on the same bf16
    operands (fp32 CPU): odd T / F (ragged parity classes, padding taps on every border), strides 1 and 2."""
    from espresso_amd import kernels as Kk

    g = torch.Generator().manual_seed(seed)
    X = bf(torch.randn(B, T, F, Cin, generator=g))
    W = bf(torch.randn(Cout, 3, 3, Cin, generator=g) * (9 * Cin) ** -0.5)
    bias = torch.randn(Cout, generator=g)
    xr = X.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = W.float().permute(0, 3, 1, 2)
    zr = torch.nn.functional.conv2d(xr, wr, bias, stride=(sy, sx), padding=1)
    To, Fo = zr.shape[2], zr.shape[3]
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=DEV)
    Z = Kk.conv3x3_fwd(X.to(DEV).reshape(-1, Cin), W.to(DEV).reshape(Cout, 9 * Cin), bias.to(DEV), B, T, F, Cin, Cout, sy, sx, stats=stats)
    torch.cuda.synchronize()
    zh = Z.float().cpu().view(B, To, Fo, Cout)
    zref = zr.detach().permute(0, 2, 3, 1)
    res = {"fwd_rel": float((zh - zref).abs().max() / zref.abs().max())}
    s_ref = torch.cat([zh.double().sum((0, 1, 2)), (zh.double() ** 2).sum((0, 1, 2))])
    res["stats_rel"] = float((stats.cpu() - s_ref).abs().max() / s_ref.abs().max())
    dZ = bf(torch.randn(B, To, Fo, Cout, generator=g))
    (dxr,) = torch.autograd.grad(zr, xr, dZ.float().permute(0, 3, 1, 2))
    Wd = W.permute(3, 1, 2, 0).contiguous()  # [Cin][3][3][Cout]
    dX = Kk.conv3x3_dgrad(dZ.to(DEV).reshape(-1, Cout), Wd.to(DEV).reshape(Cin, 9 * Cout), B, T, F, Cin, Cout, sy, sx)
    torch.cuda.synchronize()
    dref = dxr.permute(0, 2, 3, 1)
    res["dgrad_rel"] = float((dX.float().cpu().view(B, T, F, Cin) - dref).abs().max() / dref.abs().max())
    # weight gradient (transposing-LDS-read kernel), accumulated into a non-zero buffer
    wleaf = wr.clone().requires_grad_(True)
    zr2 = torch.nn.functional.conv2d(X.float().permute(0, 3, 1, 2), wleaf, None, stride=(sy, sx), padding=1)
    (dwr,) = torch.autograd.grad(zr2, wleaf, dZ.float().permute(0, 3, 1, 2))
    dW0 = torch.randn(Cout, 9 * Cin, generator=g)
    dW = Kk.conv3x3_wgrad(X.to(DEV).reshape(-1, Cin), dZ.to(DEV).reshape(-1, Cout), dW0.clone().to(DEV), B, T, F, Cin, Cout, sy, sx)
    torch.cuda.synchronize()
    wref = dW0 + dwr.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    res["wgrad_rel"] = float((dW.cpu() - wref).abs().max() / wref.abs().max())
    return res


def check_wgrad_w8(seed=0):
    """The 8-wavefront 256 x 256 weight-gradient kernel (csrc/wgrad_w8.hip, forced) against the 4-wave transposing kernel on the
    same group: dW bit for bit, db within fp32 reassociation (1e-6 relative), and both against fp64 torch.  Ragged M (rows past M
    from the zero page: M = 64 k + 1, M < 64), N / K that are not multiples of 256, row pitches shorter than the last tile
    (the joint's 5004 columns at pitch 5056), accumulation into non-zero dW / db, problems with and without bias."""
    import ctypes

    from espresso_amd import _lib
    from espresso_amd import kernels as Kk

    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(seed)
    # (M, N, K, pad_dy, pad_x, bias)
    probs = [(4161, 1256, 512, 24, 0, True), (777, 512, 384, 0, 8, True), (63, 256, 256, 0, 0, True), (1000, 296, 128, 24, 0, False),
             (129, 640, 768, 0, 0, True), (520, 5004, 512, 52, 0, True)]
    data = []
    for (M, N, K, pdy, px, bias) in probs:
        dy = bf(torch.randn(M, N + pdy, generator=g)).to(DEV)
        x = bf(torch.randn(M, K + px, generator=g)).to(DEV)
        dW0 = torch.randn(N, K, generator=g).to(DEV)
        db0 = torch.randn(N, generator=g).to(DEV) if bias else None
        data.append((dy, x, dW0, db0))

    def run(mode):
        old = lib.ea_set_wgrad_w8(mode)
        grp = _lib.EaWgradGroup()
        grp.count = len(probs)
        outs = []
        for i, ((M, N, K, pdy, px, bias), (dy, x, dW0, db0)) in enumerate(zip(probs, data)):
            dW, db = dW0.clone(), (db0.clone() if bias else None)
            q = grp.p[i]
            q.dy, q.x, q.dW, q.dbias = dy.data_ptr(), x.data_ptr(), dW.data_ptr(), (db.data_ptr() if bias else None)
            q.M, q.N, q.K, q.ld_dy, q.ld_x, q.ldw = M, N, K, N + pdy, K + px, K
            outs.append((dW, db))
        try:
            _lib.check(lib.ea_wgrad_group(ctypes.byref(grp), Kk._stream()), "ea_wgrad_group")
            torch.cuda.synchronize()
        finally:
            lib.ea_set_wgrad_w8(old)
        return outs

    ref = run(0)
    got = run(2)
    out = {"dW_bits_differ": 0, "db_rel_vs_4wave": 0.0, "dW_rel": 0.0, "db_rel": 0.0}
    for (M, N, K, pdy, px, bias), (dy, x, dW0, db0), (rW, rb), (gW, gb) in zip(probs, data, ref, got):
        out["dW_bits_differ"] += int((rW.view(torch.int32) != gW.view(torch.int32)).sum())
        r64 = dW0.double() + dy[:, :N].double().t() @ x[:, :K].double()
        out["dW_rel"] = max(out["dW_rel"], float((gW.double() - r64).abs().max() / r64.abs().max()))
        if bias:
            out["db_rel_vs_4wave"] = max(out["db_rel_vs_4wave"], float((gb - rb).abs().max() / rb.abs().max()))
            b64 = db0.double() + dy[:, :N].double().sum(0)
            out["db_rel"] = max(out["db_rel"], float((gb.double() - b64).abs().max() / b64.abs().max()))
    return out


def check_wgrad_group(seed=0, variant=0, aligned=False, tr=1):
    """ea_wgrad_group (all weight / bias gradients of a layer in one launch) vs fp32 torch on the same bf16 operands:
    ragged N / K / M, padded leading dimensions, accumulation into non-zero dW / db, problems with and without bias.
    aligned: whole-tile problems only (N % 128, K % 128, row pitches % 8) — the group then takes the direct-to-LDS kernel with
    transposing fragment reads (tr = 1) or, for comparison, the register-staged kernel (tr = 0); M stays ragged (rows past M
    come from the zero page), including M < 64 and M = 64 k + 1."""
    import ctypes

    from espresso_amd import _lib
    from espresso_amd import kernels as Kk

    lib = _lib.lib()
    lib.ea_set_gemm_variant(variant)
    g = torch.Generator(device="cpu").manual_seed(seed)
    # (M, N, K, pad_dy, pad_x, bias)
    probs = [(1000, 512, 128, 0, 0, True), (1000, 128, 512, 0, 8, True), (777, 200, 72, 8, 16, True), (333, 64, 136, 0, 0, False),
             (64, 136, 40, 8, 0, True), (4100, 256, 256, 0, 0, True), (1000, 96, 128, 32, 0, False)]
    if aligned:
        probs = [(1000, 512, 128, 0, 0, True), (777, 128, 512, 0, 8, True), (6128, 2048, 512, 0, 0, True), (333, 1536, 512, 16, 0, False),
                 (64, 256, 128, 8, 0, True), (63, 128, 128, 0, 0, True), (129, 128, 256, 0, 0, True), (4100, 256, 256, 0, 0, True)]
    old_tr = lib.ea_set_wgrad_transposing_reads(tr)
    grp = _lib.EaWgradGroup()
    grp.count = len(probs)
    keep, refs = [], []
    for i, (M, N, K, pdy, px, bias) in enumerate(probs):
        dy = bf(torch.randn(M, N + pdy, generator=g)).to(DEV)
        x = bf(torch.randn(M, K + px, generator=g)).to(DEV)
        dW = torch.randn(N, K, generator=g).to(DEV)
        db = torch.randn(N, generator=g).to(DEV) if bias else None
        refW = dW.double() + dy[:, :N].double().t() @ x[:, :K].double()
        refb = db.double() + dy[:, :N].double().sum(0) if bias else None
        q = grp.p[i]
        q.dy, q.x, q.dW, q.dbias = dy.data_ptr(), x.data_ptr(), dW.data_ptr(), (db.data_ptr() if bias else None)
        q.M, q.N, q.K, q.ld_dy, q.ld_x, q.ldw = M, N, K, N + pdy, K + px, K
        keep.append((dy, x, dW, db))
        refs.append((refW, refb))
    _lib.check(lib.ea_wgrad_group(ctypes.byref(grp), Kk._stream()), "ea_wgrad_group")
    torch.cuda.synchronize()
    lib.ea_set_gemm_variant(0)
    lib.ea_set_wgrad_transposing_reads(old_tr)
    worst_w = worst_b = 0.0
    for (dy, x, dW, db), (refW, refb) in zip(keep, refs):
        worst_w = max(worst_w, float((dW.double() - refW).abs().max() / refW.abs().max()))
        if db is not None:
            worst_b = max(worst_b, float((db.double() - refb).abs().max() / refb.abs().max()))
    return {"dW_rel": worst_w, "db_rel": worst_b}


def check_wgrad_w8_layer_group(M=4200, C=512, Fd=2048, H=8, R=519, seed=1):
    """An encoder layer's grouped weight gradient at the size where the AUTOMATIC rule takes the 8-wave kernel (>= 48 tiles of 256 x 256,
    >= 4096 rows; round 6): the layer's eight weight matrices plus the per-head positional-projection problems (dW of 64 rows), with
    biases, accumulated into non-zero gradients — 8-wave (mode 1 = automatic) against the 4-wave kernel (mode 0) bit for bit, and
    that the automatic mode really took the 8-wave kernel (the group's launch grid)."""
    import ctypes

    from espresso_amd import _lib
    from espresso_amd import kernels as Kk

    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(seed)
    Rp = (R + 7) // 8 * 8
    # (N = dW rows = dy columns, K = dW columns = x columns, bias)
    shapes = [(Fd, C, True), (C, Fd, True), (3 * C, C, True), (C, C, True), (2 * C, C, False), (C, C, False), (Fd, C, True), (C, Fd, True)]
    probs = []
    for (N, K, bias) in shapes:
        dy = bf(torch.randn(M, N, generator=g) * 0.1).to(DEV)
        x = bf(torch.randn(M, K, generator=g)).to(DEV)
        probs.append((dy, N, x, K, K, torch.randn(N, K, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV) if bias else None, N, K))
    qv = bf(torch.randn(M, C, generator=g) * 0.1).to(DEV)
    dh = C // H
    for h in range(H):  # thin problems: dy = 64 columns of a wider matrix, x = this head's [M][Rp] slab
        xh = bf(torch.randn(M, Rp, generator=g)).to(DEV)
        probs.append((qv[:, h * dh:], dh, xh, R, Rp, torch.randn(dh, Rp, generator=g).to(DEV), None, C, Rp))

    def run(mode):
        old = lib.ea_set_wgrad_w8(mode)
        grp = _lib.EaWgradGroup()
        grp.count = len(probs)
        outs = []
        for i, (dy, N, x, K, ldw, dW0, db0, ld_dy, ld_x) in enumerate(probs):
            dW, db = dW0.clone(), (db0.clone() if db0 is not None else None)
            q = grp.p[i]
            q.dy, q.x, q.dW, q.dbias = dy.data_ptr(), x.data_ptr(), dW.data_ptr(), (db.data_ptr() if db is not None else None)
            q.M, q.N, q.K, q.ld_dy, q.ld_x, q.ldw = M, N, K, ld_dy, ld_x, ldw
            outs.append((dW, db))
        try:
            _lib.check(lib.ea_wgrad_group(ctypes.byref(grp), Kk._stream()), "ea_wgrad_group")
            torch.cuda.synchronize()
        finally:
            lib.ea_set_wgrad_w8(old)
        return outs

    ref, got = run(0), run(1)
    res = {"dW_bits_differ": 0, "db_rel": 0.0, "dW_rel_fp64": 0.0}
    for (dy, N, x, K, ldw, dW0, db0, ld_dy, ld_x), (rW, rb), (gW, gb) in zip(probs, ref, got):
        res["dW_bits_differ"] += int((rW[:, :K].contiguous().view(torch.int32) != gW[:, :K].contiguous().view(torch.int32)).sum())
        if rb is not None:
            res["db_rel"] = max(res["db_rel"], float((rb - gb).abs().max() / rb.abs().max()))
    dy, N, x, K, ldw, dW0, db0, ld_dy, ld_x = probs[3]
    want = dW0.double() + dy[:, :N].double().t() @ x[:, :K].double()
    res["dW_rel_fp64"] = float((got[3][0].double() - want).abs().max() / want.abs().max())
    tiles = sum(((N + 255) // 256) * ((K + 255) // 256) for (_, N, _, K, *_rest) in probs)
    res["tiles_256"] = tiles
    return res


def check_joint_wgrad(n=20000, V=5004, J=512, seed=0):
    """functional._joint_wgrad (the transducer output layer's weight / bias gradient at recipe width: padded row pitch 5056, row
    slabs through one grouped launch, slab outputs summed) against fp64 on the same bf16 operands."""
    from espresso_amd import functional as F

    g = torch.Generator(device="cpu").manual_seed(seed)
    Vp = (V + 63) // 64 * 64
    dl = torch.zeros(n, Vp, dtype=torch.bfloat16)
    dl[:, :V] = bf(torch.randn(n, V, generator=g) * 0.1)
    Z = bf(torch.relu(torch.randn(n, J, generator=g)))
    dl, Z = dl.to(DEV), Z.to(DEV)
    dw, db = F._joint_wgrad(dl, Z, n, V, J, Vp)
    torch.cuda.synchronize()
    ref_w = dl[:, :V].double().t() @ Z.double()
    ref_b = dl[:, :V].double().sum(0)
    return {"dW_rel": float((dw.double() - ref_w).abs().max() / ref_w.abs().max()),
            "db_rel": float((db.double() - ref_b).abs().max() / ref_b.abs().max()), "shape_ok": tuple(dw.shape) == (V, J) and tuple(db.shape) == (V,)}


def check_deferred_backward_matches_immediate(layer_type="conformer", p_drop=0.0):
    """The same update step with the layer backward's side work deferred (grouped weight-gradient launch, joined by the next
    layer's call / the end-of-backward flush) and immediate (split-K launches joined inside every call): same loss, same
    gradients up to the fp32 summation order."""
    from espresso_amd import functional as F

    g, sd, _, _ = load_fixture(f"ref_{layer_type}_ctc_tiny")
    feats = torch.from_numpy(g["feats"]).to(DEV)
    lengths = torch.from_numpy(g["lengths"]).to(DEV)
    out = []
    for deferred in (True, False):
        F.set_backward_deferred(deferred)
        try:
            model = build_tiny_model(layer_type).to(DEV)
            load_ref_state(model, sd)
            model.train()
            for rep in range(2):  # second pass: halves / scratch tags in their steady state
                for p in model.parameters():
                    p.grad = None
                o = model(feats, lengths)
                lo = o["encoder_out"][0].float()
                with F.accumulating_backward():  # (the scope a trainer opens; outside it the sink is never used)
                    (lo * torch.linspace(-1, 1, lo.shape[-1], device=DEV)).sum().backward()
            torch.cuda.synchronize()
            out.append({n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None})
        finally:
            F.set_backward_deferred(True)
    worst = ("", 0.0)
    for n in out[0]:
        if (".pre_encoder.convolutions." in n and n.endswith(".bias")) or n.endswith("self_attn.k_proj.bias"):
            continue  # true gradient exactly zero: both runs hold round-off noise there
        a, b = out[0][n], out[1][n]
        e = float((a - b).abs().max() / (b.abs().max() + 1e-6))
        if e > worst[1]:
            worst = (n, e)
    return {"worst_grad": worst, "n": len(out[0])}


def check_layernorm_pair_kernels(M=6240, C=512, p=0.1, seed=5):
    """ea_layernorm_fwd2 / ea_layernorm_bwd2_dx (two LayerNorms over the same rows in one launch) against the two separate launches
    they replace — ea_layernorm_fwd twice; ea_layernorm_bwd_dx then ea_layernorm_bwd_dx2 — on the same inputs: every output compared
    BITWISE (both norms' statistics and outputs; the gradient, its dropped copy, both partial slabs)."""
    from espresso_amd import _lib
    from espresso_amd import kernels as Kk

    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(seed)
    st = Kk._stream()
    dev = DEV
    x = bf(torch.randn(M, C, generator=g) * 1.7 + 0.3).to(dev)
    g1, b1 = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    g2, b2 = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    e16 = lambda: torch.empty(M, C, dtype=torch.bfloat16, device=dev)
    f32 = lambda: torch.empty(M, dtype=torch.float32, device=dev)
    # forward: separate
    y1a, y2a, m1a, r1a, m2a, r2a = e16(), e16(), f32(), f32(), f32(), f32()
    _lib.check(lib.ea_layernorm_fwd(P(x), P(g1), P(b1), P(y1a), P(m1a), P(r1a), M, C, 1e-5, None, 0, 0, 1.0, st), "fwd a")
    _lib.check(lib.ea_layernorm_fwd(P(y1a), P(g2), P(b2), P(y2a), P(m2a), P(r2a), M, C, 1e-5, None, 0, 0, 1.0, st), "fwd b")
    y1b, y2b, m1b, r1b, m2b, r2b = e16(), e16(), f32(), f32(), f32(), f32()
    _lib.check(lib.ea_layernorm_fwd2(P(x), P(g1), P(b1), P(y1b), P(m1b), P(r1b), P(g2), P(b2), P(y2b), P(m2b), P(r2b), M, C, 1e-5, st), "fwd2")
    eq = lambda a, b: bool(torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a.view(torch.int32),
                                       b.view(torch.int16) if b.dtype == torch.bfloat16 else b.view(torch.int32)))
    res = {"fwd_y1": eq(y1a, y1b), "fwd_y2": eq(y2a, y2b), "fwd_stats": eq(m1a, m1b) and eq(r1a, r1b) and eq(m2a, m2b) and eq(r2a, r2b)}
    # backward: norm 2 first (incoming gradient dy2, residual-path gradient `add`), then norm 1 on its result
    dy2 = bf(torch.randn(M, C, generator=g) * 0.05).to(dev)
    add = bf(torch.randn(M, C, generator=g) * 0.05).to(dev)
    nws = lib.ea_layernorm_bwd_workspace_bytes(M, C) // 4
    wsa1, wsa2, wsb1, wsb2 = (torch.zeros(nws, dtype=torch.float32, device=dev) for _ in range(4))
    dga, dba = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    d_mid, dxa, o2a = e16(), e16(), e16()
    thr, scale = Kk.drop_params(p)
    _lib.check(lib.ea_layernorm_bwd_dx(P(y1a), P(dy2), P(g2), P(m2a), P(r2a), P(d_mid), P(dga), P(dba), M, C, None, 0, 0, 1.0, P(add), P(wsa1), st), "bwd a")
    _lib.check(lib.ea_layernorm_bwd_dx2(P(x), P(d_mid), P(g1), P(m1a), P(r1a), P(dxa), P(dga), P(dba), M, C, None, P(wsa2), P(o2a), 0.5, 1234 * 64 + 7,
                                        thr, scale, st), "bwd b")
    dxb, o2b = e16(), e16()
    _lib.check(lib.ea_layernorm_bwd2_dx(P(y1a), P(dy2), P(g2), P(m2a), P(r2a), P(add), P(wsb1), P(x), P(g1), P(m1a), P(r1a), P(wsb2), P(dxb), M, C,
                                        P(o2b), 0.5, 1234 * 64 + 7, thr, scale, st), "bwd2")
    torch.cuda.synchronize()
    res.update({"bwd_dx": eq(dxa, dxb), "bwd_out2": eq(o2a, o2b), "bwd_partials_later_norm": eq(wsa1, wsb1), "bwd_partials_earlier_norm": eq(wsa2, wsb2),
                "dx_max_diff": float((dxa.float() - dxb.float()).abs().max()), "dx_nonzero": float(dxa.float().abs().mean()) > 0,
                "ws_max_rel": float((wsa2 - wsb2).abs().max() / wsa2.abs().max())})
    return res


def check_layer_chain_matches_plain(B=8, T=1100, p_drop=0.1, embed_dim=64, heads=4, ffn=128, seed=11):
    """Chained Conformer layer calls (EaLayerChain: layer k's final LayerNorm and layer k+1's first in one kernel, forward and backward)
    against the plain calls, same dropout masks: identical forward output when both use the rows-at-once kernels (B*T' >= 2048 rows),
    gradients equal up to the fp32 atomics of the parameter reductions.  Also reports how many calls were chained."""
    from espresso_amd import functional as F

    gen = torch.Generator(device="cpu").manual_seed(seed)
    feats = torch.randn(B, T, 80, generator=gen).to(DEV)
    lengths = torch.tensor([T - 37 * i for i in range(B)], dtype=torch.long).clamp(min=T // 3).to(DEV)
    model = build_tiny_model("conformer", embed_dim=embed_dim, heads=heads, ffn=ffn, dropout=p_drop).to(DEV)
    model.train()
    runs = []
    for chain in (True, False, False):  # (the second plain run measures the run-to-run noise of the plain path itself)
        F.set_layer_chain(chain)
        try:
            for rep in range(2):
                for p in model.parameters():
                    p.grad = None
                for m in model.modules():  # same BatchNorm running statistics going in
                    if hasattr(m, "running_mean") and m.running_mean is not None:
                        m.running_mean.zero_(); m.running_var.fill_(1.0)
                F.set_dropout_seed(seed)
                o = model(feats, lengths)
                lo = o["encoder_out"][0].float()
                with F.accumulating_backward():
                    (lo * torch.linspace(-1, 1, lo.shape[-1], device=DEV)).sum().backward()
            torch.cuda.synchronize()
            runs.append((lo.detach().cpu().clone(),
                         {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}))
        finally:
            F.set_layer_chain(True)
    rows = int(lo.shape[0] * lo.shape[1])

    def worst_diff(ga, gb):
        worst = ("", 0.0)
        for n in ga:
            if (".pre_encoder.convolutions." in n and n.endswith(".bias")) or n.endswith("self_attn.k_proj.bias"):
                continue  # true gradient exactly zero: both runs hold round-off noise there
            e = float((ga[n] - gb[n]).abs().max() / (gb[n].abs().max() + 1e-6))
            if e > worst[1]:
                worst = (n, e)
        return worst

    return {"rows": rows, "out_equal": bool(torch.equal(runs[0][0], runs[1][0])),
            "out_max_diff": float((runs[0][0] - runs[1][0]).abs().max()), "worst_grad": worst_diff(runs[0][1], runs[1][1]),
            "plain_vs_plain": worst_diff(runs[2][1], runs[1][1]), "n": len(runs[0][1])}


def check_layer_stack_matches_loop(B=8, T=1100, p_drop=0.1, seed=17):
    """One C call per direction over the run of Conformer layers (ea_conformer_stack_fwd / _bwd) against the Python loop of layer
    nodes: same dropout seeds in the same order, same launches — training output and eval output identical, gradients inside the
    loop path's own run-to-run noise (BatchNorm backward sums by atomics)."""
    from espresso_amd import functional as F

    gen = torch.Generator(device="cpu").manual_seed(seed)
    feats = torch.randn(B, T, 80, generator=gen).to(DEV)
    lengths = torch.tensor([T - 37 * i for i in range(B)], dtype=torch.long).clamp(min=T // 3).to(DEV)
    model = build_tiny_model("conformer", dropout=p_drop).to(DEV)
    runs = []
    for stack in (True, False, False):
        F.set_layer_stack(stack)
        try:
            for m in model.modules():
                if hasattr(m, "running_mean") and m.running_mean is not None:
                    m.running_mean.zero_(); m.running_var.fill_(1.0)
            model.eval()
            with torch.no_grad():
                ev = model(feats, lengths)["encoder_out"][0].float().cpu()
            model.train()
            for rep in range(2):
                for p in model.parameters():
                    p.grad = None
                for m in model.modules():
                    if hasattr(m, "running_mean") and m.running_mean is not None:
                        m.running_mean.zero_(); m.running_var.fill_(1.0)
                F.set_dropout_seed(seed)
                lo = model(feats, lengths)["encoder_out"][0].float()
                with F.accumulating_backward():
                    (lo * torch.linspace(-1, 1, lo.shape[-1], device=DEV)).sum().backward()
            torch.cuda.synchronize()
            nbt = [int(l.conv_module.batch_norm.num_batches_tracked) for l in model.encoder.layers]
            runs.append((lo.detach().cpu().clone(), ev,
                         {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}, nbt))
        finally:
            F.set_layer_stack(True)

    def worst_diff(ga, gb):
        worst = ("", 0.0)
        for n in ga:
            if (".pre_encoder.convolutions." in n and n.endswith(".bias")) or n.endswith("self_attn.k_proj.bias"):
                continue
            e = float((ga[n] - gb[n]).abs().max() / (gb[n].abs().max() + 1e-6))
            if e > worst[1]:
                worst = (n, e)
        return worst

    return {"out_equal": bool(torch.equal(runs[0][0], runs[1][0])), "eval_equal": bool(torch.equal(runs[0][1], runs[1][1])),
            "worst_grad": worst_diff(runs[0][2], runs[1][2]), "loop_vs_loop": worst_diff(runs[2][2], runs[1][2]), "n": len(runs[0][2]),
            "counters": (runs[0][3], runs[1][3])}


def check_layer_chain_fallbacks(B=8, T=1100, seed=3):
    """The situations in which a chained call must NOT be used, or must not leak state, each against the switch-off run:
    (a) eval under no_grad (forward chaining only) — identical logits; (b) collected hidden states (the layer outputs have a second
    consumer: the encoder does not chain) — states and logits identical; (c) two forward passes before one backward (the second
    finds every binding's arena busy and takes private arenas, unchained; the first pass's chained contexts are consumed later) —
    gradients inside the plain path's own run-to-run noise."""
    from espresso_amd import functional as F

    gen = torch.Generator(device="cpu").manual_seed(seed)
    feats = torch.randn(B, T, 80, generator=gen).to(DEV)
    feats2 = torch.randn(B, T, 80, generator=gen).to(DEV)
    lengths = torch.tensor([T - 41 * i for i in range(B)], dtype=torch.long).clamp(min=T // 3).to(DEV)
    model = build_tiny_model("conformer", dropout=0.0).to(DEV)
    res = {}
    outs = {}
    for chain in (True, False, False):
        F.set_layer_chain(chain)
        try:
            model.eval()
            for m in model.modules():  # (the training passes below move the running statistics)
                if hasattr(m, "running_mean") and m.running_mean is not None:
                    m.running_mean.zero_(); m.running_var.fill_(1.0)
            with torch.no_grad():
                ev = model(feats, lengths)["encoder_out"][0].float().cpu()
                enc = model.encoder(feats, lengths, return_all_hiddens=True)
                st = [s.float().cpu() for s in enc["encoder_states"]] + [enc["encoder_out"][0].float().cpu()]
            model.train()
            for m in model.modules():
                if hasattr(m, "running_mean") and m.running_mean is not None:
                    m.running_mean.zero_(); m.running_var.fill_(1.0)
            for p in model.parameters():
                p.grad = None
            o1 = model(feats, lengths)["encoder_out"][0].float()
            o2 = model(feats2, lengths)["encoder_out"][0].float()
            w = torch.linspace(-1, 1, o1.shape[-1], device=DEV)
            with F.accumulating_backward():
                ((o1 * w).sum() + (o2 * w).sum() * 0.5).backward()
            torch.cuda.synchronize()
            grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
            outs.setdefault("runs", []).append((ev, st, grads, o1.detach().cpu(), o2.detach().cpu()))
        finally:
            F.set_layer_chain(True)
    a, b, c = outs["runs"]

    def worst(ga, gb):
        w_ = 0.0
        for n in ga:
            if (".pre_encoder.convolutions." in n and n.endswith(".bias")) or n.endswith("self_attn.k_proj.bias"):
                continue
            w_ = max(w_, float((ga[n] - gb[n]).abs().max() / (gb[n].abs().max() + 1e-6)))
        return w_

    res["eval_equal"] = bool(torch.equal(a[0], b[0]))
    res["states_equal"] = len(a[1]) == len(b[1]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    res["n_states"] = len(a[1])
    res["train_out_equal"] = bool(torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]))
    res["grad_diff"], res["plain_noise"], res["n_grads"] = worst(a[2], b[2]), worst(c[2], b[2]), len(a[2])
    return res


def check_direct_param_grads(fixture="ref_conformer_ctc_dh64"):
    """Sub-sampler / fc_out / embedding-LayerNorm parameter gradients accumulated straight into the flat gradient buffer by the
    kernels (functional._grad_sink: no pooled temporary, no AccumulateGrad launch per parameter, the conv weight gradient
    scattered into the parameter's own layout) vs the autograd route: two micro-batches accumulated each way."""
    from espresso_amd import functional as F
    from espresso_amd.optim.flat import FlatParams

    g, sd, _, _ = load_fixture(fixture)
    d, H, ffn = _fixture_shape(fixture)
    feats, lengths = torch.from_numpy(g["feats"]).to(DEV), torch.from_numpy(g["lengths"]).to(DEV)
    out, reported = [], []
    for direct in (True, False):
        old = F.set_direct_param_grads(direct)
        seen = []
        try:
            model = build_tiny_model("conformer", embed_dim=d, heads=H, ffn=ffn).to(DEV)
            load_ref_state(model, sd)
            flat = FlatParams(model, DEV)
            flat.zero_grad()
            for p_ in flat.params:  # what the data-parallel wrapper registers: must fire once per backward either way
                p_.register_post_accumulate_grad_hook(lambda q, seen=seen: seen.append(q))
            model.train()
            for rep in range(2):
                o = model(feats, lengths)
                lo = o["encoder_out"][0].float()
                with F.accumulating_backward():  # (the scope a trainer opens; outside it the sink is never used)
                    (lo * torch.linspace(-1, 1, lo.shape[-1], device=DEV)).sum().backward()
            torch.cuda.synchronize()
            out.append({n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None})
            ids = {id(p): n for n, p in model.named_parameters()}
            import collections

            cnt = collections.Counter(ids[id(p)] for p in seen if id(p) in ids and "layers." not in ids[id(p)])
            reported.append(dict(cnt))
        finally:
            F.set_direct_param_grads(old)
    worst = ("", 0.0)
    for n in out[1]:
        if (".pre_encoder.convolutions." in n and n.endswith(".bias")) or n.endswith("self_attn.k_proj.bias"):
            continue
        a, b = out[0][n], out[1][n]
        e = float((a - b).abs().max() / (b.abs().max() + 1e-6))
        if e > worst[1]:
            worst = (n, e)
    return {"worst_grad": worst, "n": len(out[1]), "hook_counts_direct": reported[0], "hook_counts_autograd_route": reported[1]}


# ------------------------------------------------------------------ model-level parity vs the reference fixture
def load_fixture(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    grads = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad::")}
    bn_after = {k[10:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("bn_after::")}
    return g, sd, grads, bn_after


class _Task:
    feat_dim, feat_in_channels = 80, 1
    blank_symbol = "<s>"

    def __init__(self, V):
        from espresso_amd.data.asr_dictionary import AsrDictionary

        self.target_dictionary = AsrDictionary.from_symbols([f"t{i}" for i in range(V - 5)], enable_bos=True)
        assert len(self.target_dictionary) == V


def build_tiny_model(layer_type, V=40, embed_dim=64, heads=4, ffn=128, learned_pos=False, legacy=None, dropout=0.0):
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerConfig
    from espresso_amd.models.transformer.speech_transformer_encoder_model import SpeechTransformerEncoderModel

    cfg = SpeechTransformerConfig()
    e = cfg.encoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = embed_dim, ffn, 2, heads
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, layer_type
    e.learned_pos = learned_pos
    e.conv_channels = "[64, 64, 16, 16]"
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = dropout
    cfg.layernorm_embedding = True
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    if legacy is not None:  # the options of a fixture written with `legacy=...` (oracle/gen_golden.py)
        e.relative_positional_embeddings = False
        e.learned_pos = bool(legacy.get("learned_pos", False))
        cfg.layernorm_embedding = bool(legacy.get("layernorm_embedding", False))
        e.normalize_before = bool(legacy.get("normalize_before", True))
        e.chunk_size = int(legacy.get("chunk_size", 0))
        e.chunk_left_window = int(legacy.get("chunk_left_window", 0))
        e.chunk_right_window = int(legacy.get("chunk_right_window", 0))
    return SpeechTransformerEncoderModel.build_model(cfg, _Task(V))


def load_ref_state(model, sd):
    sd = {"encoder." + k: v for k, v in sd.items()}
    sd = model.upgrade_state_dict_named(sd, "")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing


def _fixture_shape(fixture):
    """(embed_dim, heads, ffn) of an encoder fixture: `*_dh64` = head dim 64 (the recipes' 512 / 8 shape class)."""
    return (128, 2, 256) if fixture.endswith("_dh64") or fixture == "ref_transformer_ctc_legacy" else (64, 4, 128)


def check_encoder_vs_reference(layer_type="conformer", fixture=None):
    """Load the reference's weights into the HIP model; compare eval logits, train logits (BN batch stats), CTC loss and every
    parameter gradient
      (a) with what the reference's own modules produced in fp32 (golden fixture) — north_star's bf16 tolerance 1e-2 on
          losses / log-probs, greedy ids identical — and
      (b) with the oracle run under bf16 emulation (oracle/torch_ref.py rounds where the HIP path stores): what remains is
          accumulation order and 1-ulp rounding flips, so the bound is a factor 5-10 tighter and a real 10 % arithmetic bug in
          any gradient cannot hide behind the expected bf16-vs-fp32 gap."""
    from espresso_amd import functional as F
    from oracle import torch_ref

    name = fixture or f"ref_{layer_type}_ctc_tiny"
    g, sd, grads, bn_after = load_fixture(name)
    learned = "learnedpos" in name
    layer_type = layer_type.split("_")[0]
    d, H, ffn = _fixture_shape(name)
    legacy = None
    if "meta" in g.files:
        import json

        legacy = json.loads(str(g["meta"]))
    lkw = lambda training: torch_ref.legacy_encoder_kwargs(legacy, torch.from_numpy(g["lengths"]), training) if legacy is not None else {}
    model = build_tiny_model(layer_type, embed_dim=d, heads=H, ffn=ffn, learned_pos=learned, legacy=legacy).to(DEV)
    load_ref_state(model, sd)
    feats = torch.from_numpy(g["feats"]).to(DEV)
    lengths = torch.from_numpy(g["lengths"]).to(DEV)
    res = {}
    model.eval()
    with torch.no_grad():
        out = model(feats, lengths)
    lo = out["encoder_out"][0].float().cpu()
    ref = torch.from_numpy(g["out::eval_logits"])
    res["eval_logits_abs"] = float((lo - ref).abs().max())
    res["eval_lengths_equal"] = bool((out["src_lengths"][0].cpu().numpy() == g["out::out_lengths"]).all())
    # ---- bf16-emulating oracle, eval mode ----
    flash = d // H == 64
    with torch.no_grad(), torch_ref.bf16_emulation(True, flash=flash):
        emu_eval, _ = torch_ref.encoder(torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), sd, H=H, layer_type=layer_type,
                                        training=False, **lkw(False))
    res["eval_logits_vs_emulation"] = float((lo - emu_eval).abs().max())
    # the same in units of the bf16 spacing at each logit's magnitude (a bf16 logit of magnitude 2..4 cannot be closer than 0.0156)
    ulp = torch.exp2(torch.floor(torch.log2(torch.maximum(lo.abs(), emu_eval.abs()).clamp_min(2.0 ** -20))) - 7)
    nul = ((lo - emu_eval).abs() / ulp)
    res["eval_logits_vs_emulation_ulps"] = float(nul.max())
    res["eval_logits_identical_frac"] = float((nul == 0).float().mean())
    # greedy ids over valid frames: identical wherever the reference's own top-2 margin exceeds the bf16 noise floor
    # (a frame whose two best logits are closer than the rounding of bf16 logits has no defined argmax at this precision)
    ol = g["out::out_lengths"]
    agree, agree_clear, n_clear = [], 0, 0
    for b in range(lo.shape[1]):
        a_hip, a_ref = lo[: ol[b], b].argmax(-1), ref[: ol[b], b].argmax(-1)
        agree.append(float((a_hip == a_ref).float().mean()))
        top2 = ref[: ol[b], b].topk(2, -1).values
        clear = (top2[:, 0] - top2[:, 1]) > 0.05
        agree_clear += int(((a_hip == a_ref) & clear).sum())
        n_clear += int(clear.sum())
    res["eval_greedy_agree"] = min(agree)
    res["eval_greedy_agree_clear_margin"] = agree_clear / max(1, n_clear)
    res["clear_margin_frames"] = n_clear
    # train mode
    model.train()
    out = model(feats, lengths)
    lo = out["encoder_out"][0].float().cpu()
    res["train_logits_abs"] = float((lo.detach() - torch.from_numpy(g["out::train_logits"])).abs().max())
    tgt = torch.from_numpy(g["targets"]).to(DEV)
    tl = (tgt != 1).sum(-1)
    B, Tp = out["encoder_padding_mask"][0].shape
    nll, _ = F.ctc_loss(out["_logits_bt"][0], tgt.to(torch.int32).contiguous(), out["src_lengths"][0].to(torch.int32),
                        tl.to(torch.int32), B, Tp, blank=0)
    loss = nll.sum()
    loss.backward()
    torch.cuda.synchronize()
    res["train_loss"] = float(loss.detach())
    res["ref_loss"] = float(g["out::train_loss"])
    # ---- bf16-emulating oracle, train mode: loss and every gradient ----
    sde = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "version"
               and not k.endswith("_float_tensor") else v.clone()) for k, v in sd.items()}
    with torch_ref.bf16_emulation(True, flash=flash):
        lt, ole = torch_ref.encoder(torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), sde, H=H, layer_type=layer_type,
                                    training=True, **lkw(True))
        tg = torch.from_numpy(g["targets"])
        eloss = torch_ref.ctc_loss_sum(lt, tg, ole, (tg != 1).sum(-1))
        eloss.backward()
    res["emu_loss"] = float(eloss.detach())
    res["train_logits_vs_emulation"] = float((lo.detach() - lt.detach()).abs().max())
    # per-parameter gradient error relative to that gradient's own scale.  Excluded: parameters whose true gradient is exactly
    # zero, so that the reference's value is fp32 round-off — a conv bias in front of BatchNorm (the batch mean removes it) and
    # the key bias (softmax is invariant to a constant added to every key).
    errs, errs_emu = [], []
    for n, p in model.encoder.named_parameters():
        if n.startswith("pre_encoder.convolutions.") and n.endswith(".bias"):
            continue
        if n.endswith("self_attn.k_proj.bias"):
            continue
        r = grads[n]
        gr = p.grad.float().cpu()
        errs.append((float((gr - r).abs().max() / (float(r.abs().max()) + 1e-12)), n))
        ge = sde[n].grad
        errs_emu.append((float((gr - ge).abs().max() / (float(ge.abs().max()) + 1e-12)), n))
    errs.sort(reverse=True)
    errs_emu.sort(reverse=True)
    res["worst_grad"] = (errs[0][1], errs[0][0])
    res["worst5"] = [(n, round(e, 4)) for e, n in errs[:5]]
    res["worst_grad_vs_emulation"] = (errs_emu[0][1], errs_emu[0][0])
    res["median_grad_vs_emulation"] = errs_emu[len(errs_emu) // 2][0]
    bn = 0.0
    msd = model.encoder.state_dict()
    for k, v in bn_after.items():
        bn = max(bn, float((msd[k].float().cpu() - v).abs().max()))
    res["bn_running_abs"] = bn
    return res


def _grad_errors(named_params, oracle_sd, skip=lambda n: False):
    """[(max |hip - oracle| / max |oracle|, name)] sorted worst first."""
    errs = []
    for n, p in named_params:
        if skip(n) or p.grad is None:
            continue
        ge = oracle_sd[n].grad
        errs.append((float((p.grad.float().cpu() - ge).abs().max() / (float(ge.abs().max()) + 1e-12)), n))
    errs.sort(reverse=True)
    return errs


def _skip_zero_grad_params(n):
    # true gradient exactly zero (a conv bias in front of BatchNorm, the key bias under softmax): the oracle's value is round-off
    return (n.startswith("pre_encoder.convolutions.") and n.endswith(".bias")) or n.endswith("self_attn.k_proj.bias")


def check_encoder_dropout_vs_oracle(layer_type="conformer", fixture=None, p=0.1, native=True, seed=4321):
    """TRAINING-MODE parity (dropout = attention_dropout = activation_dropout = p, the recipes' 0.1): the HIP model's loss, logits
    and every parameter gradient against the oracle running the reference's FairseqDropout sites with the HIP path's own keep
    decisions (oracle/dropout_ref.py rebuilds every mask from the (site, seed) list the HIP forward reports; the hash itself is
    pinned on the CPU and in check_gemm_dropout / check_elementwise_dropout).  Same bounds as the dropout-off comparison.
    Control: the oracle with masks from different seeds must disagree far beyond the bound (the test can see a wrong mask)."""
    from espresso_amd import _lib
    from espresso_amd import functional as F
    from oracle import dropout_ref as D
    from oracle import torch_ref

    name = fixture or f"ref_{layer_type}_ctc_tiny"
    g, sd, _, _ = load_fixture(name)
    learned = "learnedpos" in name
    layer_type = layer_type.split("_")[0]
    d, H, ffn = _fixture_shape(name)
    model = build_tiny_model(layer_type, embed_dim=d, heads=H, ffn=ffn, learned_pos=learned, dropout=p).to(DEV)
    load_ref_state(model, sd)
    for l in model.encoder.layers:
        l.use_native_runtime = native
    feats, lengths = torch.from_numpy(g["feats"]).to(DEV), torch.from_numpy(g["lengths"]).to(DEV)
    tgt = torch.from_numpy(g["targets"]).to(DEV)
    tl = (tgt != 1).sum(-1)
    model.train()
    F.set_dropout_seed(seed)
    with F.trace_dropout_seeds() as tr:
        out = model(feats, lengths)
    lo = out["encoder_out"][0].float().cpu().detach()
    B, Tp = out["encoder_padding_mask"][0].shape
    nll, _ = F.ctc_loss(out["_logits_bt"][0], tgt.to(torch.int32).contiguous(), out["src_lengths"][0].to(torch.int32),
                        tl.to(torch.int32), B, Tp, blank=0)
    loss = nll.sum()
    loss.backward()
    torch.cuda.synchronize()
    res = {"train_loss": float(loss.detach()), "sites": [e[0] for e in tr.entries]}
    lib = _lib.lib()
    flash = d // H == 64

    def oracle(trace, emulate):
        sde = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "version"
                   and not k.endswith("_float_tensor") else v.clone()) for k, v in sd.items()}
        plan = D.MaskPlan(trace, lib.ea_layer_dropout_seed)
        with torch_ref.bf16_emulation(emulate, flash=flash), torch_ref.dropout_masks(plan):
            lt, ole = torch_ref.encoder(torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), sde, H=H, layer_type=layer_type,
                                        training=True)
            tg = torch.from_numpy(g["targets"])
            el = torch_ref.ctc_loss_sum(lt, tg, ole, (tg != 1).sum(-1))
            el.backward()
        plan.done()
        return float(el.detach()), lt.detach(), sde, plan

    emu_loss, emu_logits, sde, plan = oracle(tr.entries, True)
    res["n_site_masks"] = len(plan.queue)
    res["emu_loss"] = emu_loss
    res["train_logits_vs_emulation"] = float((lo - emu_logits).abs().max())
    errs = _grad_errors(model.encoder.named_parameters(), sde, _skip_zero_grad_params)
    res["worst_grad_vs_emulation"] = (errs[0][1], errs[0][0])
    res["median_grad_vs_emulation"] = errs[len(errs) // 2][0]
    res["fp32_loss"], _, sdf, _ = oracle(tr.entries, False)
    errs32 = _grad_errors(model.encoder.named_parameters(), sdf, _skip_zero_grad_params)
    res["worst_grad_vs_fp32"] = (errs32[0][1], errs32[0][0])
    # how far the two ORACLE runs (fp32 / bf16-emulating: same arithmetic, same masks) are from each other, per tensor: a tensor
    # on which they disagree by more than the 8 % bound is rounding-chaotic at this precision (ReLU / dropout kinks: a derivative
    # flips when a pre-activation within one bf16 step of zero rounds to the other side), and no implementation can be held
    # closer to either of them than they are to each other
    gap = {n: float((sde[n].grad - sdf[n].grad).abs().max() / (float(sdf[n].grad.abs().max()) + 1e-12)) for _, n in errs}
    res["worst_excess_over_bound"] = max(e / max(0.08, gap[n]) for e, n in errs)
    res["oracle_gap_of_worst"] = gap[errs[0][1]]
    gaps = sorted(gap.values())
    res["median_oracle_gap"] = gaps[len(gaps) // 2]
    # control: same sites, other seeds
    wrong = [[s, sd_ + 977 * 64, pp] for s, sd_, pp in tr.entries]
    res["wrong_mask_loss"], wl, sdw, _ = oracle(wrong, True)
    errsw = _grad_errors(model.encoder.named_parameters(), sdw, _skip_zero_grad_params)
    res["wrong_mask_median_grad"] = errsw[len(errsw) // 2][0]
    res["wrong_mask_logits"] = float((lo - wl).abs().max())
    return res


# ------------------------------------------------------------------ CTC alone (fp32 path, 1e-3 bar)
def check_ctc(B=5, T=60, V=57, Lmax=9, seed=0):
    from espresso_amd import functional as F

    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(B, T, V, generator=g) * 2).float()
    in_len = torch.tensor([T, T - 7, T // 2, 11, T][:B])
    tgt_len = torch.tensor([Lmax, 3, 5, 1, 0][:B])
    tgt = torch.ones(B, Lmax, dtype=torch.long)
    for b in range(B):
        tgt[b, : tgt_len[b]] = torch.randint(1, V, (int(tgt_len[b]),), generator=g)
    tgt[0, 1] = tgt[0, 0]  # repeated label
    lg = logits.clone().requires_grad_(True)
    lp = torch.log_softmax(lg, -1).transpose(0, 1)
    flat = torch.cat([tgt[b, : tgt_len[b]] for b in range(B)])
    ref = torch.nn.functional.ctc_loss(lp, flat, in_len, tgt_len, blank=0, reduction="none", zero_infinity=True)
    ref.sum().backward()
    x = logits.to(DEV).reshape(B * T, V).clone().requires_grad_(True)
    nll, lprobs = F.ctc_loss(x, tgt.to(torch.int32).to(DEV), in_len.to(torch.int32).to(DEV), tgt_len.to(torch.int32).to(DEV), B, T, 0)
    nll2 = torch.where(torch.isinf(nll), torch.zeros_like(nll), nll)
    nll2.sum().backward()
    torch.cuda.synchronize()
    return {
        "nll_abs": float((nll2.cpu() - ref.detach()).abs().max()),
        "grad_abs": float((x.grad.cpu().view(B, T, V) - lg.grad).abs().max()),
        "lprobs_abs": float((lprobs.cpu().view(B, T, V) - torch.log_softmax(logits, -1)).abs().max()),
    }


# ------------------------------------------------------------------ fbank + CMVN + SpecAugment vs numpy oracle
def check_frontend(seed=0):
    from espresso_amd.data.feature_transforms import AdaptiveSpecAugmentTransform, GlobalCMVN, numpy_seed_value
    from espresso_amd.data.gpu_frontend import GpuFbankFrontend
    from oracle import fbank_ref

    rng = np.random.default_rng(seed)
    ns = [16000 * 3 + 123, 399, 400, 16000, 16000 * 5 + 7]
    wavs = [(rng.standard_normal(n) * 3000).astype(np.float32) for n in ns]
    mean = rng.standard_normal(80) * 2 + 8
    std = rng.random(80) + 2.0
    sa = AdaptiveSpecAugmentTransform.from_config_dict({"freq_mask_N": 2, "freq_mask_F": 27, "time_mask_pm": 0.04, "time_mask_ps": 0.04})
    fe = GpuFbankFrontend(DEV, cmvn=GlobalCMVN(mean=mean, std=std), specaug=sa, seed=1)
    off = np.zeros(len(ns) + 1, dtype=np.int64)
    off[1:] = np.cumsum(ns)
    wav = torch.from_numpy(np.concatenate(wavs)).to(DEV)
    offs = torch.from_numpy(off).to(DEV)
    res = {}
    feat, lens, frames = fe(wav, offs, ns, train=False)
    torch.cuda.synchronize()
    feat = feat.cpu().numpy()
    worst = 0.0
    for b, w in enumerate(wavs):
        ref = fbank_ref.global_cmvn(fbank_ref.fbank(w), mean, std)
        assert ref.shape[0] == frames[b] == int(lens[b])
        if ref.shape[0]:
            worst = max(worst, float(np.abs(feat[b, : ref.shape[0]] - ref).max()))
        assert np.all(feat[b, ref.shape[0]:] == 0)
    res["fbank_cmvn_abs"] = worst
    # SpecAugment with the reference's RNG stream
    idx = [5, 17, 2, 9, 11]
    feat2, lens, frames = fe(wav, offs, ns, train=True, epoch=2, indices=idx)
    torch.cuda.synchronize()
    feat2 = feat2.cpu().numpy()
    worst = 0.0
    for b, w in enumerate(wavs):
        ref = fbank_ref.global_cmvn(fbank_ref.fbank(w), mean, std)
        if ref.shape[0] == 0:
            continue
        state = np.random.get_state()
        np.random.seed(numpy_seed_value(1, 2, idx[b]))
        fm, tm = sa.draw_masks(ref.shape[0], 80)
        np.random.set_state(state)
        ref2 = fbank_ref.specaugment_apply(ref.astype(np.float64), fm, tm, None)
        worst = max(worst, float(np.abs(feat2[b, : ref.shape[0]] - ref2).max()))
    res["specaug_abs"] = worst
    return res


# ------------------------------------------------------------------ Adam / clip
def check_adam(n=100003, steps=3):
    from espresso_amd import kernels as Kk

    g = torch.Generator().manual_seed(0)
    p = torch.randn(n, generator=g)
    # reference arithmetic: fairseq/utils.py:347-397 clip + fairseq/optim/adam.py:215-240 (denom = sqrt(v) + eps,
    # step_size = lr * sqrt(1 - b2^t) / (1 - b1^t)) restated with CPU torch ops
    ref_p, rm, rv = p.clone(), torch.zeros(n), torch.zeros(n)
    dp, m, v = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    p16 = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    sumsq = torch.zeros(1, device=DEV)
    coef = torch.zeros(2, device=DEV)
    for s in range(1, steps + 1):
        grad = torch.randn(n, generator=g) * 3
        gg = grad / 4.0
        gg = gg * min(1.0, 2.0 / (float(gg.norm()) + 1e-6))
        rm = rm * 0.9 + 0.1 * gg
        rv = rv * 0.98 + 0.02 * gg * gg
        ref_p = ref_p - (1e-2 * math.sqrt(1 - 0.98 ** s) / (1 - 0.9 ** s)) * (rm / (rv.sqrt() + 1e-8))
        dg = grad.to(DEV)
        sumsq.zero_()
        Kk.grad_sumsq(dg, sumsq)
        Kk.clip_coef(sumsq, 1.0, 2.0, coef, torch.tensor([4.0], device=DEV))
        Kk.adam_step(dp, dg, m, v, p16, coef, 1e-2, 0.9, 0.98, 1e-8, 0.0, s, zero_grad=True)
        assert float(dg.abs().max()) == 0.0
    torch.cuda.synchronize()
    return {"param_abs": float((dp.cpu() - ref_p).abs().max()),
            "bf16_exact": bool((p16.cpu() == dp.cpu().to(torch.bfloat16)).all())}


# ------------------------------------------------------------------ smoke
def smoke_check():
    """One small invocation of the hot path (Conformer + CTC, head dim 64: fused attention kernels + native layer runtime +
    deferred grouped weight gradients) against the reference's own outputs and the bf16-emulating oracle."""
    r = check_encoder_vs_reference("conformer", fixture="ref_conformer_ctc_dh64")
    print("smoke:", r)
    assert abs(r["train_loss"] - r["ref_loss"]) / r["ref_loss"] < 1e-2, r
    assert r["eval_lengths_equal"] and r["eval_greedy_agree_clear_margin"] == 1.0, r
    assert r["eval_logits_vs_emulation"] < 3e-2 and r["worst_grad_vs_emulation"][1] < 6e-2, r
    return r


# ------------------------------------------------------------------ native layer runtime vs per-kernel composition
def check_native_layer(p_drop=0.0, seed=0, C=64, heads=4, T=37):
    """Same weights, same input: csrc/engine.hip (one call per layer) must reproduce the Python composition
    of the individual kernels (outputs and every gradient), with and without dropout (same seeds -> same masks
    is NOT guaranteed across the two paths, so dropout runs only check finiteness + determinism)."""
    from espresso_amd import functional as F
    from espresso_amd.modules.conformer_layer import ConformerWithRelativePositionalEmbeddingEncoderLayer as Layer

    torch.manual_seed(seed)
    model = build_tiny_model("conformer", embed_dim=C, heads=heads, ffn=2 * C).to(DEV)
    layer = model.encoder.layers[0]
    with torch.no_grad():  # the reference initialises the positional biases to zero: make them count
        layer.self_attn.pos_bias_u.normal_(0, 0.1)
        layer.self_attn.pos_bias_v.normal_(0, 0.1)
    B = 3
    x0 = bf(torch.randn(B * T, C)).to(DEV)
    key_len = torch.tensor([T, T - 7, max(1, T // 3)], dtype=torch.int32, device=DEV)
    model.train()
    outs, grads = [], []
    for native in (False, True):
        Layer.use_native_runtime = native
        for p in layer.parameters():
            p.grad = None
        layer.conv_module.batch_norm.running_mean.zero_()
        layer.conv_module.batch_norm.running_var.fill_(1.0)
        x = x0.clone().requires_grad_(True)
        y = layer(x, B, T, key_len=key_len)
        (y.float() * torch.linspace(-1, 1, C, device=DEV)).sum().backward()
        torch.cuda.synchronize()
        outs.append(y.detach().float().cpu())
        g = {n: p.grad.detach().float().cpu().clone() for n, p in layer.named_parameters()}
        g["__x"] = x.grad.float().cpu()
        g["__rm"] = layer.conv_module.batch_norm.running_mean.detach().cpu().clone()
        grads.append(g)
    Layer.use_native_runtime = True
    res = {"out_abs": float((outs[0] - outs[1]).abs().max())}
    worst = ("", 0.0)
    for n in grads[0]:
        if n.endswith("k_proj.bias"):
            continue  # exactly zero in exact arithmetic (softmax is shift invariant): only rounding noise to compare
        a, b = grads[0][n], grads[1][n]
        e = float((a - b).abs().max() / (a.abs().max() + 1e-6))
        if e > worst[1]:
            worst = (n, e)
    res["worst_grad"] = worst
    return res


# ------------------------------------------------------------------ CTC greedy decoding
def check_ctc_greedy(layer_type="conformer"):
    """HIP decoder vs the reference procedure (max, unique_consecutive, drop blank — ctc_decoder.py:172-188) applied on
    the CPU to (a) the same HIP log-probs -> must be bit-exact, (b) the reference's own fp32 logits -> agreement rate."""
    from espresso_amd.tools.ctc_decoder import CTCDecoder

    g, sd, _, _ = load_fixture(f"ref_{layer_type}_ctc_tiny")
    model = build_tiny_model(layer_type).to(DEV)
    load_ref_state(model, sd)
    model.eval()
    feats = torch.from_numpy(g["feats"]).to(DEV)
    lengths = torch.from_numpy(g["lengths"]).to(DEV)
    sample = {"net_input": {"src_tokens": feats, "src_lengths": lengths}}
    dec = CTCDecoder([model], _Task(40).target_dictionary)
    hyps = dec.generate([model], sample)
    with torch.no_grad():
        out = model(feats, lengths)
        lp = model.get_normalized_probs(out, log_probs=True).float().cpu()  # T x B x V
    ol = g["out::out_lengths"]

    def ref_decode(lprobs_tv):
        scores, toks = lprobs_tv.max(-1)
        seq = toks.unique_consecutive()
        return seq[seq != 0].tolist(), float(scores.sum())

    exact, agree_ref = True, []
    ref_logits = torch.from_numpy(g["out::eval_logits"])
    for b in range(feats.shape[0]):
        want, wscore = ref_decode(lp[: ol[b], b])
        got = hyps[b][0]["tokens"].tolist()
        exact &= got == want and abs(float(hyps[b][0]["score"]) - wscore) < 1e-3
        want_ref, _ = ref_decode(torch.log_softmax(ref_logits[: ol[b], b], -1))
        agree_ref.append(got == want_ref)
    toks, scores, _ = dec.decode([model], sample)
    return {"exact_vs_same_lprobs": bool(exact), "utts_equal_to_reference_fp32_decode": agree_ref, "decode_shape": tuple(toks.shape)}


# ------------------------------------------------------------------ attention encoder-decoder + label-smoothed CE
class _TaskAR:
    feat_dim, feat_in_channels = 80, 1

    def __init__(self, V):
        from espresso_amd.data.asr_dictionary import AsrDictionary

        self.target_dictionary = AsrDictionary.from_symbols([f"t{i}" for i in range(V - 4)], enable_bos=False)
        assert len(self.target_dictionary) == V


def build_tiny_encdec(V=40, embed_dim=64, heads=4, learned_pos=False, ffn=128, dropout=0.0):
    from espresso_amd.models.transformer.speech_transformer_base import SpeechTransformerModelBase
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerConfig

    cfg = SpeechTransformerConfig()
    e, d = cfg.encoder, cfg.decoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = embed_dim, ffn, 2, heads
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, "transformer"
    e.learned_pos = learned_pos
    e.conv_channels = "[64, 64, 16, 16]"
    d.embed_dim, d.ffn_embed_dim, d.layers, d.attention_heads, d.normalize_before = embed_dim, ffn, 2, heads, True
    d.input_dim = d.output_dim = embed_dim
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = dropout
    cfg.layernorm_embedding = True
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    return SpeechTransformerModelBase.build_model(cfg, _TaskAR(V))


def check_legacy_speech_transformer_step(arch="speech_transformer_wsj"):
    """One teacher-forced training step of the argparse preset (absolute encoder positions, no embedding LayerNorm) on the HIP
    path, against the fp32 oracle driven by the same weights (encoder restatement with absolute positions + decoder restatement):
    logits within the bf16 bound, finite non-zero gradients for every parameter."""
    from espresso_amd import functional as F
    from espresso_amd.models.transformer.speech_transformer_legacy import SpeechTransformerModel
    from oracle import torch_ref

    torch.manual_seed(3)
    V = 40
    task = _Task(V)
    model = SpeechTransformerModel.build_model(dict(arch=arch, encoder_layers=2, decoder_layers=1, encoder_embed_dim=128,
                                                    encoder_ffn_embed_dim=256, encoder_attention_heads=2, decoder_attention_heads=2,
                                                    encoder_conv_channels="[64, 64, 16, 16]", dropout=0.0, attention_dropout=0.0,
                                                    activation_dropout=0.0), task).to(DEV)
    B, T, U = 3, 120, 9
    lengths = torch.tensor([120, 100, 64], device=DEV)
    feats = torch.randn(B, T, 80, device=DEV)
    for b in range(B):
        feats[b, lengths[b]:] = 0
    tl = [9, 7, 4]
    target = torch.zeros(B, U, dtype=torch.long, device=DEV)  # pad = 0? use the dictionary's pad
    pad, eos = task.target_dictionary.pad(), task.target_dictionary.eos()
    target.fill_(pad)
    prev = torch.full((B, U), pad, dtype=torch.long, device=DEV)
    for b, n in enumerate(tl):
        toks = torch.randint(4, V, (n - 1,), device=DEV)
        target[b, : n - 1], target[b, n - 1] = toks, eos
        prev[b, 0], prev[b, 1:n] = eos, toks
    model.eval()
    with torch.no_grad():
        lo, _ = model(feats, lengths, prev)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    enc_sd["embed_positions._float_tensor"] = torch.zeros(1)  # marks sinusoidal absolute positions for the restatement
    x, ol = torch_ref.encoder(feats.cpu(), lengths.cpu(), enc_sd, H=2, layer_type="transformer", training=False)
    pad_mask = torch.arange(x.shape[0]).unsqueeze(0) >= ol.unsqueeze(1)
    ref = torch_ref.decoder(prev.cpu(), x, pad_mask if bool(pad_mask.any()) else None, sd, H=2, pad_idx=pad)
    valid = target.ne(pad).cpu()
    res = {"eval_logits_abs_valid": float((lo.float().cpu() - ref)[valid].abs().max()), "ref_logit_scale": float(ref[valid].abs().max())}
    model.train()
    lo, extra = model(feats, lengths, prev)
    loss, nll = F.label_smoothed_ce(extra["_logits_bu"], target.reshape(-1).to(torch.int32).contiguous(), pad, 0.1)
    loss.backward()
    torch.cuda.synchronize()
    res["loss_finite"] = bool(torch.isfinite(loss))
    bad = [n for n, p in model.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all()) or float(p.grad.abs().max()) == 0.0]
    # exactly-zero gradients by construction: key biases (softmax shift invariance), conv biases in front of BatchNorm
    res["params_without_gradient"] = [n for n in bad if not (n.endswith("k_proj.bias") or (".convolutions." in n and n.endswith(".bias")))]
    return res


def _encdec_for(fixture, dropout=0.0):
    d, H, ffn = _fixture_shape(fixture)
    return build_tiny_encdec(embed_dim=d, heads=H, ffn=ffn, dropout=dropout)


def check_encdec_deferred_matches_immediate(fixture="ref_transformer_encdec_dh64"):
    """Encoder-decoder model (learned relative-position tables in the encoder: the layer's `dpe` stays on the main stream in
    deferred mode; decoder layers: their weight gradients as one grouped launch inside the call) with the side work deferred /
    grouped vs immediate (split-K launches joined inside every call, functional.set_backward_deferred(False)): same loss, same
    gradients up to the fp32 summation order.  Two passes each, so halves and scratch tags are in their steady state."""
    from espresso_amd import functional as F

    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    feats, lengths = torch.from_numpy(g["feats"]).to(DEV), torch.from_numpy(g["lengths"]).to(DEV)
    prev, target = torch.from_numpy(g["prev"]).to(DEV), torch.from_numpy(g["target"]).to(DEV)
    out = []
    for deferred in (True, False):
        F.set_backward_deferred(deferred)
        try:
            model = _encdec_for(fixture).to(DEV)
            model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
            model.train()
            for rep in range(2):
                for p in model.parameters():
                    p.grad = None
                F.set_dropout_seed(7)
                lo, extra = model(feats, lengths, prev)
                loss, _ = F.label_smoothed_ce(extra["_logits_bu"], target.reshape(-1).to(torch.int32).contiguous(), 0, 0.1)
                loss.backward()
            torch.cuda.synchronize()
            out.append((float(loss), {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None}))
        finally:
            F.set_backward_deferred(True)
    (la, ga), (lb, gb) = out
    names = [n for n in gb if not (n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias"))]  # (zero up to noise)
    worst = max(((float((ga[n] - gb[n]).abs().max() / (gb[n].abs().max() + 1e-20)), n) for n in names), key=lambda kv: kv[0])
    return {"loss_rel": abs(la - lb) / abs(lb), "worst_grad_rel": worst, "same_params": set(ga) == set(gb), "n_grads": len(ga)}


def check_encdec_vs_reference(fixture="ref_transformer_encdec_tiny"):
    """`ref_transformer_encdec_dh64`: head dim 64 -> native decoder-layer runtime (ea_decoder_layer_fwd/bwd) on the fused
    attention kernels, compared with the reference's own model outputs."""
    from espresso_amd import functional as F

    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    grads = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad::")}
    model = _encdec_for(fixture).to(DEV)
    sd = model.upgrade_state_dict_named(dict(sd), "")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    feats, lengths = torch.from_numpy(g["feats"]).to(DEV), torch.from_numpy(g["lengths"]).to(DEV)
    prev, target = torch.from_numpy(g["prev"]).to(DEV), torch.from_numpy(g["target"]).to(DEV)
    valid = target.ne(0).cpu()
    res = {}
    model.eval()
    with torch.no_grad():
        lo, _ = model(feats, lengths, prev)
    ref = torch.from_numpy(g["out::eval_logits"])
    res["eval_logits_abs_valid"] = float((lo.float().cpu() - ref)[valid].abs().max())
    res["eval_greedy_agree"] = float((lo.float().cpu().argmax(-1) == ref.argmax(-1))[valid].float().mean())
    top2 = ref.topk(2, -1).values
    clear = ((top2[..., 0] - top2[..., 1]) > 0.05) & valid
    res["eval_greedy_agree_clear_margin"] = float((lo.float().cpu().argmax(-1) == ref.argmax(-1))[clear].float().mean())
    model.train()
    lo, extra = model(feats, lengths, prev)
    loss, nll = F.label_smoothed_ce(extra["_logits_bu"], target.reshape(-1).to(torch.int32).contiguous(), 0, 0.1)
    loss.backward()
    torch.cuda.synchronize()
    res["loss"], res["ref_loss"] = float(loss), float(g["out::loss"])
    res["nll"], res["ref_nll"] = float(nll), float(g["out::nll"])
    errs = []
    for n, p in model.named_parameters():
        if n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias"):
            continue
        if n.endswith("attn.k_proj.bias"):
            continue
        r = grads[n]
        errs.append((float((p.grad.float().cpu() - r).abs().max() / (float(r.abs().max()) + 1e-12)), n))
    errs.sort(reverse=True)
    res["worst5"] = [(n, round(e, 4)) for e, n in errs[:5]]
    # ---- the same training pass on the bf16-emulating oracle (rounds where the HIP path stores): tight gradient bounds ----
    from oracle import torch_ref

    d, H, _ = _fixture_shape(fixture)
    sd0 = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    sde = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "version"
               and not k.endswith("_float_tensor") else v.clone()) for k, v in sd0.items()}
    with torch_ref.bf16_emulation(True, flash=(d // H == 64)):
        el = torch_ref.encdec(torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"]), sde, H, 0, training=True)
        eloss, enll = torch_ref.label_smoothed_nll(el.reshape(-1, el.shape[-1]), torch.from_numpy(g["target"]).reshape(-1), 0.1, 0)
        eloss.backward()
    res["emu_loss"] = float(eloss.detach())
    res["train_logits_vs_emulation"] = float((lo.detach().float().cpu() - el.detach())[valid].abs().max())
    errs_emu, l2_emu = [], []
    for n, p in model.named_parameters():
        if (n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias")) or n.endswith("attn.k_proj.bias"):
            continue
        ge = sde[n].grad
        errs_emu.append((float((p.grad.float().cpu() - ge).abs().max() / (float(ge.abs().max()) + 1e-12)), n))
        l2_emu.append((float((p.grad.float().cpu() - ge).norm() / (float(ge.norm()) + 1e-12)), n))
    errs_emu.sort(reverse=True)
    l2_emu.sort(reverse=True)
    res["worst_grad_vs_emulation"] = (errs_emu[0][1], errs_emu[0][0])
    res["worst5_vs_emulation"] = [(n, round(e, 4)) for e, n in errs_emu[:5]]
    res["median_grad_vs_emulation"] = errs_emu[len(errs_emu) // 2][0]
    res["worst_l2_vs_emulation"] = (l2_emu[0][1], l2_emu[0][0])
    res["median_l2_vs_emulation"] = l2_emu[len(l2_emu) // 2][0]
    return res


def check_encdec_dropout_vs_oracle(fixture="ref_transformer_encdec_dh64", p=0.1, seed=99):
    """speech_transformer_base in TRAINING mode with dropout = attention_dropout = activation_dropout = p: loss, logits and
    every gradient vs the oracle running the reference's dropout sites (encoder + decoder embedding, self-attention,
    encoder-decoder attention, FFN: fairseq transformer_decoder.py:324-327, transformer_layer.py:384-529) with the HIP path's
    keep decisions; see check_encoder_dropout_vs_oracle."""
    from espresso_amd import _lib
    from espresso_amd import functional as F
    from oracle import dropout_ref as D
    from oracle import torch_ref

    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    sd0 = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = _encdec_for(fixture, dropout=p).to(DEV)
    missing, unexpected = model.load_state_dict(model.upgrade_state_dict_named(dict(sd0), ""), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    feats, lengths = torch.from_numpy(g["feats"]).to(DEV), torch.from_numpy(g["lengths"]).to(DEV)
    prev, target = torch.from_numpy(g["prev"]).to(DEV), torch.from_numpy(g["target"]).to(DEV)
    valid = target.ne(0).cpu()
    model.train()
    F.set_dropout_seed(seed)
    with F.trace_dropout_seeds() as tr:
        lo, extra = model(feats, lengths, prev)
    loss, nll = F.label_smoothed_ce(extra["_logits_bu"], target.reshape(-1).to(torch.int32).contiguous(), 0, 0.1)
    loss.backward()
    torch.cuda.synchronize()
    lo = lo.detach().float().cpu()
    res = {"loss": float(loss), "sites": [e[0] for e in tr.entries]}
    d, H, _ = _fixture_shape(fixture)
    lib = _lib.lib()
    skip = lambda n: (n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias")) or n.endswith("attn.k_proj.bias")

    def oracle(trace, emulate):
        sde = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "version"
                   and not k.endswith("_float_tensor") else v.clone()) for k, v in sd0.items()}
        plan = D.MaskPlan(trace, lib.ea_layer_dropout_seed)
        with torch_ref.bf16_emulation(emulate, flash=(d // H == 64)), torch_ref.dropout_masks(plan):
            el = torch_ref.encdec(torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"]), sde, H, 0,
                                  training=True)
            eloss, _ = torch_ref.label_smoothed_nll(el.reshape(-1, el.shape[-1]), torch.from_numpy(g["target"]).reshape(-1), 0.1, 0)
            eloss.backward()
        plan.done()
        return float(eloss.detach()), el.detach(), sde, plan

    res["emu_loss"], el, sde, plan = oracle(tr.entries, True)
    res["n_site_masks"] = len(plan.queue)
    res["train_logits_vs_emulation"] = float((lo - el)[valid].abs().max())
    errs = _grad_errors(model.named_parameters(), sde, skip)
    res["worst_grad_vs_emulation"] = (errs[0][1], errs[0][0])
    res["median_grad_vs_emulation"] = errs[len(errs) // 2][0]
    res["fp32_loss"], _, sdf, _ = oracle(tr.entries, False)
    gap = {n: float((sde[n].grad - sdf[n].grad).abs().max() / (float(sdf[n].grad.abs().max()) + 1e-12)) for _, n in errs}
    res["worst_excess_over_bound"] = max(e / max(0.08, gap[n]) for e, n in errs)
    res["oracle_gap_of_worst"] = gap[errs[0][1]]
    gaps = sorted(gap.values())
    res["median_oracle_gap"] = gaps[len(gaps) // 2]
    wrong = [[s_, sd_ + 977 * 64, pp] for s_, sd_, pp in tr.entries]
    res["wrong_mask_loss"], wl, sdw, _ = oracle(wrong, True)
    errsw = _grad_errors(model.named_parameters(), sdw, skip)
    res["wrong_mask_median_grad"] = errsw[len(errsw) // 2][0]
    res["wrong_mask_logits"] = float((lo - wl)[valid].abs().max())
    return res


# ------------------------------------------------------------------ beam search with incremental decoding
def check_ensemble_beam_search_vs_reference(fixture="ref_transformer_encdec_ensemble"):
    """Two independently initialised reference models decoded as an ensemble by the reference's own SequenceGenerator
    (fairseq/sequence_generator.py:837-939) vs the HIP path with the same two state dicts: the reference's per-position scores
    reproduced by force-decoding its best hypotheses through both incremental decoders, scores of the hypotheses both beams hold,
    greedy identity at clear-margin steps."""
    from espresso_amd.sequence_generator import SequenceGenerator

    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    members = []
    for pre in ("sd::", "sd2::"):
        sd = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
        m = _encdec_for(fixture).to(DEV)
        missing, unexpected = m.load_state_dict(m.upgrade_state_dict_named(dict(sd), ""), strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        members.append(m.eval())
    d = _TaskAR(40).target_dictionary
    sample = {"net_input": {"src_tokens": torch.from_numpy(g["feats"]).to(DEV), "src_lengths": torch.from_numpy(g["lengths"]).to(DEV)}}
    res = {}
    for tag, kw in (("b3", dict(beam_size=3, max_len_a=0.0, max_len_b=12)), ("b1", dict(beam_size=1, max_len_a=0.0, max_len_b=12))):
        hyps = SequenceGenerator(members, d, **kw).generate(members, sample)
        top_equal, score_err, shared = [], 0.0, 0
        for b, hl in enumerate(hyps):
            ref, hi = {}, 0
            while f"beam::{tag}::{b}::{hi}::tokens" in g.files:
                ref[tuple(g[f"beam::{tag}::{b}::{hi}::tokens"].tolist())] = float(g[f"beam::{tag}::{b}::{hi}::score"])
                hi += 1
            top_equal.append(tuple(hl[0]["tokens"].tolist()) == tuple(g[f"beam::{tag}::{b}::0::tokens"].tolist()))
            for h in hl:
                k = tuple(h["tokens"].tolist())
                if k in ref:
                    shared += 1
                    score_err = max(score_err, abs(float(h["score"]) - ref[k]))
        res[tag] = {"top1_tokens_equal": top_equal, "shared_hypotheses": shared, "score_abs": score_err}
    ref_best = torch.stack([torch.from_numpy(g[f"beam::b3::{b}::0::tokens"]) for b in range(3)]).to(DEV)
    ref_pos = torch.stack([torch.from_numpy(g[f"beam::b3::{b}::0::pos"]) for b in range(3)])
    L = ref_best.shape[1]
    n_clear = n_match = 0
    with torch.no_grad():
        sts = [m.decoder.init_incremental(m.forward_encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"]), 3, 1)
               for m in members]
        cur = torch.full((3, 1), d.eos(), dtype=torch.long, device=DEV)
        got = []
        for step in range(L):
            par = None if step == 0 else torch.arange(3, device=DEV)
            lp = torch.logsumexp(torch.stack([m.decoder.step(st, cur, step, par).float() for m, st in zip(members, sts)], 0), 0) - math.log(2.0)
            got.append(lp.gather(-1, ref_best[:, step:step + 1]).squeeze(-1).cpu())
            cur = torch.cat([cur, ref_best[:, step:step + 1]], 1)
    res["forced_decode_pos_score_abs"] = float((torch.stack(got, 1) - ref_pos).abs().max())
    return res


def check_beam_search_vs_reference(fixture="ref_transformer_encdec_tiny"):
    """HIP incremental decoder + beam kernels vs what the reference's SequenceGenerator produced with the same weights
    (tests/golden/ref_transformer_encdec_{tiny,dh64}.npz: beam 3, beam 3 + eos_factor 1.5, beam 1)."""
    from espresso_amd.sequence_generator import SequenceGenerator

    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = _encdec_for(fixture).to(DEV)
    model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    model.eval()
    d = _TaskAR(40).target_dictionary
    sample = {"net_input": {"src_tokens": torch.from_numpy(g["feats"]).to(DEV), "src_lengths": torch.from_numpy(g["lengths"]).to(DEV)}}
    res = {}
    for tag, kw in (("b3", dict(beam_size=3, max_len_a=0.0, max_len_b=12)),
                    ("b3_eosf", dict(beam_size=3, max_len_a=0.0, max_len_b=12, eos_factor=1.5)),
                    ("b1", dict(beam_size=1, max_len_a=0.0, max_len_b=12))):
        from espresso_amd.sequence_generator import HipBeamSearch

        class Recording(HipBeamSearch):
            """records, per sentence, the smallest score gap the search ever cut through: between the last candidate inside the
            beam and the first one outside it (all candidates, and non-EOS candidates only).  A search whose gaps all exceed the
            bf16 noise of the scores makes the same decisions as the reference's fp32 run."""

            def __init__(self, eos, beam):
                self.eos, self.beam, self.min_gap, self.bsz0 = eos, beam, None, None

            def step(self, step, lprobs, prev_scores, bsz, beam):
                cs, ct, cb = super().step(step, lprobs, prev_scores, bsz, beam)
                if self.bsz0 is None:
                    self.bsz0, self.min_gap = bsz, [float("inf")] * bsz
                if bsz == self.bsz0 and cs.shape[1] > beam:  # (after the batch shrinks rows no longer map to sentences: stop)
                    c, t = cs.cpu(), ct.cpu()
                    for i in range(bsz):
                        gap = float(c[i, beam - 1] - c[i, beam])
                        live = [float(x) for x, tk in zip(c[i], t[i]) if int(tk) != self.eos and x > -1e30]
                        if len(live) > beam:
                            gap = min(gap, live[beam - 1] - live[beam])
                        if gap == gap:
                            self.min_gap[i] = min(self.min_gap[i], gap)
                return cs, ct, cb

        rec = Recording(d.eos(), kw["beam_size"])
        gen = SequenceGenerator([model], d, search=rec, **kw)
        hyps = gen.generate([model], sample)
        top_equal, any_rank_equal, score_err, clear = [], [], 0.0, []
        for b, hl in enumerate(hyps):
            ref_tokens = []
            hi = 0
            while f"beam::{tag}::{b}::{hi}::tokens" in g.files:
                ref_tokens.append((g[f"beam::{tag}::{b}::{hi}::tokens"].tolist(), float(g[f"beam::{tag}::{b}::{hi}::score"])))
                hi += 1
            top_equal.append(hl[0]["tokens"].tolist() == ref_tokens[0][0])
            # the reference's 1-best is "clear" when its (length-normalised) score leads its own runner-up by more than twice the
            # score tolerance of this comparison (3e-2): then no bf16 realisation may rank another hypothesis first
            margin = ref_tokens[0][1] - ref_tokens[1][1] if len(ref_tokens) > 1 else float("inf")
            clear.append((round(margin, 4), margin > 0.06, top_equal[-1]))
            refset = {tuple(t): s for t, s in ref_tokens}
            any_rank_equal.append(sum(tuple(h["tokens"].tolist()) in refset for h in hl) / len(hl))
            for h in hl:
                k = tuple(h["tokens"].tolist())
                if k in refset:
                    score_err = max(score_err, abs(float(h["score"]) - refset[k]))
        res[tag] = {"top1_tokens_equal": top_equal, "frac_hyps_in_reference_beam": any_rank_equal, "score_abs": score_err,
                    "ref_margin_clear_equal": clear, "min_cut_gap": [round(x, 4) for x in rec.min_gap]}
    # Force-decode the reference's best hypotheses through the INCREMENTAL path (K/V caches, one-query attention) and
    # compare per-position log-probs with the reference generator's positional scores.
    ref_best = [torch.from_numpy(g[f"beam::b3::{b}::0::tokens"]) for b in range(3)]
    ref_pos = torch.stack([torch.from_numpy(g[f"beam::b3::{b}::0::pos"]) for b in range(3)])
    L = ref_best[0].numel()
    toks = torch.stack(ref_best).to(DEV)
    with torch.no_grad():
        enc_out = model.forward_encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"])
        st = model.decoder.init_incremental(enc_out, 3, 1)
        cur = torch.full((3, 1), d.eos(), dtype=torch.long, device=DEV)
        got = []
        for step in range(L):
            lp = model.decoder.step(st, cur, step, None if step == 0 else torch.arange(3, device=DEV))
            got.append(lp.gather(-1, toks[:, step:step + 1]).squeeze(-1).cpu())
            cur = torch.cat([cur, toks[:, step:step + 1]], 1)
    got = torch.stack(got, 1)
    # the last position of the reference hypotheses is the forced EOS at max_len: its reference score is the raw eos log-prob
    res["forced_decode_pos_score_abs"] = float((got - ref_pos).abs().max())
    with torch.no_grad():
        prev3 = torch.cat([torch.full((3, 1), d.eos(), dtype=torch.long, device=DEV), toks[:, :-1]], 1)
        lo3, _ = model(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"], prev3)
        full = torch.log_softmax(lo3.float(), -1).gather(-1, toks.unsqueeze(-1)).squeeze(-1).cpu()
    res["full_forward_vs_ref_per_pos"] = [round(float(x), 3) for x in (full - ref_pos).abs().max(0)[0]]
    res["incr_vs_ref_per_pos"] = [round(float(x), 3) for x in (got - ref_pos).abs().max(0)[0]]
    res["ref_pos_row0"] = [round(float(x), 3) for x in ref_pos[0]]
    res["got_row0"] = [round(float(x), 3) for x in got[0]]
    res["forced_decode_total_abs"] = float((got.sum(1) - ref_pos.sum(1)).abs().max())
    # Greedy (beam 1) token identity, tie-aware: walk the reference's beam-1 hypotheses through the incremental decoder; at
    # every step whose top-2 log-prob margin exceeds the bf16 noise floor (0.05; measured per-position error 0.012) the arg-max
    # must BE the reference's token.  Steps inside the margin have no defined arg-max at this precision.
    ref_b1 = [torch.from_numpy(g[f"beam::b1::{b}::0::tokens"]) for b in range(3)]
    L1 = max(t.numel() for t in ref_b1)
    tk1 = torch.stack([torch.nn.functional.pad(t, (0, L1 - t.numel()), value=d.pad()) for t in ref_b1]).to(DEV)
    n_clear = n_match = 0
    with torch.no_grad():
        st = model.decoder.init_incremental(enc_out, 3, 1)
        cur = torch.full((3, 1), d.eos(), dtype=torch.long, device=DEV)
        for step in range(L1):
            lp = model.decoder.step(st, cur, step, None if step == 0 else torch.arange(3, device=DEV)).float()
            lp[:, d.pad()] = -math.inf  # never selected by the generator (sequence_generator.py:374)
            if step < L1 - 1:
                lp[:, d.eos()] = lp[:, d.eos()] if step >= 1 else -math.inf  # min_len = 1
            t2 = lp.topk(2, -1)
            for bi in range(3):
                if step >= ref_b1[bi].numel() - 1:
                    continue  # the final EOS of these hypotheses is forced at max_len
                if float(t2.values[bi, 0] - t2.values[bi, 1]) > 0.05:
                    n_clear += 1
                    n_match += int(t2.indices[bi, 0]) == int(tk1[bi, step])
            cur = torch.cat([cur, tk1[:, step:step + 1]], 1)
    res["greedy_clear_margin_steps"] = n_clear
    res["greedy_clear_margin_agree"] = n_match / max(1, n_clear)
    # teacher-forced consistency: incremental log-probs == full forward log-probs on the decoded prefix
    best = hyps[0][0]["tokens"].to(DEV)
    prev = torch.cat([torch.tensor([d.eos()], device=DEV), best[:-1]]).unsqueeze(0)
    with torch.no_grad():
        feats1 = sample["net_input"]["src_tokens"][:1]
        lo, extra = model(feats1, sample["net_input"]["src_lengths"][:1], prev)
        lp = torch.log_softmax(lo.float(), -1)[0]
    pos = lp.gather(-1, best.unsqueeze(-1)).squeeze(-1).cpu()
    res["incremental_vs_full_forward_abs"] = float((pos - hyps[0][0]["positional_scores"]).abs().max())
    return res


def check_beam_search_trained(fixture="ref_transformer_encdec_trained"):
    """Beam search on a TRAINED model: tests/golden/ref_transformer_encdec_trained.npz holds the weights of
    the reference's speech_transformer_base after 4 000 updates on the learnable synthetic task (oracle/gen_golden.py
    encdec_trained: the reference's own beam search decodes all 9 held-out utterances exactly) and what the reference's
    SequenceGenerator returned for 9 utterances x {beam 3, beam 3 + eos_factor 1.5, beam 1}.  The HIP generator must return the
    same best hypothesis for every one of the 27 searches, token for token; whole beams are compared wherever the search never
    cut through a gap smaller than the score noise; greedy arg-max identity is counted over ALL decoding steps."""
    from espresso_amd.sequence_generator import HipBeamSearch, SequenceGenerator

    g = np.load(os.path.join(GOLD, fixture + ".npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = build_tiny_encdec(embed_dim=64, heads=4, ffn=128).to(DEV)
    model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    model.eval()
    d = _TaskAR(40).target_dictionary

    class Recording(HipBeamSearch):
        def __init__(self, eos, beam):
            self.eos, self.beam, self.min_gap, self.bsz0 = eos, beam, None, None

        def step(self, step, lprobs, prev_scores, bsz, beam):
            cs, ct, cb = super().step(step, lprobs, prev_scores, bsz, beam)
            if self.bsz0 is None:
                self.bsz0, self.min_gap = bsz, [float("inf")] * bsz
            if bsz == self.bsz0 and cs.shape[1] > beam:
                c, t = cs.cpu(), ct.cpu()
                for i in range(bsz):
                    gap = float(c[i, beam - 1] - c[i, beam])
                    live = [float(x) for x, tk in zip(c[i], t[i]) if int(tk) != self.eos and x > -1e30]
                    if len(live) > beam:
                        gap = min(gap, live[beam - 1] - live[beam])
                    if gap == gap:
                        self.min_gap[i] = min(self.min_gap[i], gap)
            return cs, ct, cb

    top1, beams_equal_defined, n_defined, score_abs, n_steps, n_agree, exact_ref = [], [], 0, 0.0, 0, 0, 0
    for gi in range(int(g["groups"])):
        feats, lengths = torch.from_numpy(g[f"g{gi}::feats"]).to(DEV), torch.from_numpy(g[f"g{gi}::lengths"]).to(DEV)
        sample = {"net_input": {"src_tokens": feats, "src_lengths": lengths}}
        B = feats.shape[0]
        for tag, kw in (("b3", dict(beam_size=3, max_len_a=0.0, max_len_b=12)),
                        ("b3_eosf", dict(beam_size=3, max_len_a=0.0, max_len_b=12, eos_factor=1.5)),
                        ("b1", dict(beam_size=1, max_len_a=0.0, max_len_b=12))):
            rec = Recording(d.eos(), kw["beam_size"])
            hyps = SequenceGenerator([model], d, search=rec, **kw).generate([model], sample)
            for b, hl in enumerate(hyps):
                ref, hi = [], 0
                while f"g{gi}::beam::{tag}::{b}::{hi}::tokens" in g.files:
                    ref.append((g[f"g{gi}::beam::{tag}::{b}::{hi}::tokens"].tolist(), float(g[f"g{gi}::beam::{tag}::{b}::{hi}::score"])))
                    hi += 1
                top1.append(hl[0]["tokens"].tolist() == ref[0][0])
                score_abs = max(score_abs, abs(float(hl[0]["score"]) - ref[0][1]))
                if rec.min_gap[b] > 0.03:
                    n_defined += 1
                    beams_equal_defined.append([h["tokens"].tolist() for h in hl] == [t for t, _ in ref])
                if tag == "b3":
                    tgt = [int(t) for t in g[f"g{gi}::target"][b] if t != 1]
                    exact_ref += int(ref[0][0][:-1] == tgt)
        # greedy identity over ALL steps: the reference's beam-1 tokens walked through the incremental decoder
        ref_b1 = [torch.from_numpy(g[f"g{gi}::beam::b1::{b}::0::tokens"]) for b in range(B)]
        L1 = max(t.numel() for t in ref_b1)
        tk1 = torch.stack([torch.nn.functional.pad(t, (0, L1 - t.numel()), value=d.pad()) for t in ref_b1]).to(DEV)
        with torch.no_grad():
            enc_out = model.forward_encoder(feats, lengths)
            st = model.decoder.init_incremental(enc_out, B, 1)
            cur = torch.full((B, 1), d.eos(), dtype=torch.long, device=DEV)
            for step in range(L1):
                lp = model.decoder.step(st, cur, step, None if step == 0 else torch.arange(B, device=DEV)).float()
                lp[:, d.pad()] = -math.inf
                if step == 0:
                    lp[:, d.eos()] = -math.inf  # min_len = 1
                am = lp.argmax(-1)
                for bi in range(B):
                    if step < ref_b1[bi].numel():
                        n_steps += 1
                        n_agree += int(am[bi]) == int(tk1[bi, step])
                cur = torch.cat([cur, tk1[:, step:step + 1]], 1)
    return {"searches": len(top1), "top1_equal": sum(top1), "defined": n_defined, "defined_beams_equal": sum(beams_equal_defined),
            "top1_score_abs": score_abs, "greedy_steps": n_steps, "greedy_agree": n_agree / max(1, n_steps),
            "reference_decodes_target_exactly": exact_ref}


# ------------------------------------------------------------------ RNN-T loss
def check_rnnt(seed=0):
    from espresso_amd import functional as F
    from oracle import rnnt_ref

    rng = np.random.default_rng(seed)
    B, T, Umax, V = 4, 13, 5, 37
    T_len = [13, 9, 13, 4]
    U_len = [5, 3, 0, 2]
    logits = (rng.standard_normal((B, T, Umax + 1, V)) * 2).astype(np.float32)
    targets = rng.integers(1, V, size=(B, Umax)).astype(np.int32)
    x = torch.from_numpy(logits).to(DEV).requires_grad_(True)
    loss = F.rnnt_loss(x, torch.from_numpy(targets).to(DEV), torch.tensor(T_len, dtype=torch.int32, device=DEV),
                       torch.tensor(U_len, dtype=torch.int32, device=DEV), blank=0)
    loss.sum().backward()
    torch.cuda.synchronize()
    gl = x.grad.cpu().numpy()
    le, ge = 0.0, 0.0
    for b in range(B):
        z = logits[b, : T_len[b], : U_len[b] + 1]
        nll, g = rnnt_ref.rnnt_loss_one(z, targets[b, : U_len[b]].tolist(), want_grad=True)
        le = max(le, abs(float(loss[b]) - nll) / max(1.0, abs(nll)))
        ge = max(ge, float(np.abs(gl[b, : T_len[b], : U_len[b] + 1] - g).max()))
        outside = gl[b].copy()
        outside[: T_len[b], : U_len[b] + 1] = 0
        assert float(np.abs(outside).max()) == 0.0
    return {"loss_rel": le, "grad_abs": ge}


# ------------------------------------------------------------------ validation-time greedy decoder for attention models
def check_simple_greedy_decoder():
    from espresso_amd.tools.simple_greedy_decoder import SimpleGreedyDecoder

    g = np.load(os.path.join(GOLD, "ref_transformer_encdec_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = build_tiny_encdec().to(DEV)
    model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    model.eval()
    d = _TaskAR(40).target_dictionary
    sample = {"net_input": {"src_tokens": torch.from_numpy(g["feats"]).to(DEV), "src_lengths": torch.from_numpy(g["lengths"]).to(DEV)},
              "target": torch.from_numpy(g["target"]).to(DEV)}
    dec = SimpleGreedyDecoder([model], d, for_validation=True)
    tokens, lprobs, _ = dec.decode([model], sample)
    # self-consistency with the teacher-forced forward on the decoded prefix (same check the reference semantics imply)
    prev = torch.cat([torch.full((3, 1), d.eos(), dtype=torch.long, device=DEV), tokens[:, :-1]], 1)
    with torch.no_grad():
        lo, _ = model(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"], prev)
    am = lo.float().argmax(-1)
    U = min(tokens.shape[1], lprobs.shape[1])
    agree = float((am[:, :U] == tokens[:, :U]).float().mean())
    # an ensemble of the model with itself: log of the mean of two identical distributions -> the same tokens and log-probs
    tok2, lp2, _ = SimpleGreedyDecoder([model, model], d, for_validation=True).decode([model, model], sample)
    return {"tokens_shape": tuple(tokens.shape), "lprobs_shape": tuple(lprobs.shape), "argmax_consistency": agree,
            "lprobs_normalised": float(torch.logsumexp(lprobs, -1).abs().max()),
            "ensemble_tokens_equal": bool(tok2.shape == tokens.shape and (tok2 == tokens).all()),
            "ensemble_lprobs_abs": float((lp2 - lprobs).abs().max())}


def _attn_reference(qu, qv, k, v, pp, klen, H, B, T, S, causal):
    """fp32 restatement of fairseq/modules/multihead_attention.py:788-907 on the (already biased / scaled) queries."""
    dh = 64
    f = lambda x, L: x.float().view(B, L, H, dh).permute(0, 2, 1, 3)  # [B][H][L][dh]
    s = f(qu, T) @ f(k, S).transpose(-1, -2)
    if qv is not None:
        raw = f(qv, T) @ pp.float().view(2 * T - 1, H, dh).permute(1, 0, 2).transpose(-1, -2)  # [B][H][T][2T-1]
        ii = torch.arange(T, device=qu.device)[:, None]
        jj = torch.arange(S, device=qu.device)[None, :]
        s = s + torch.gather(raw, 3, ((T - 1) - ii + jj).expand(B, H, T, S))
    jj = torch.arange(S, device=qu.device)
    if klen is not None:
        s = s.masked_fill(jj[None, None, None, :] >= klen.long()[:, None, None, None], float("-inf"))
    if causal:
        ii = torch.arange(T, device=qu.device)
        s = s.masked_fill(jj[None, :] > ii[:, None] + (S - T), float("-inf"))
    p = torch.softmax(s, dim=-1)
    return s, p, f(v, S)


def _flash_impl(general):
    """select the general kernels (flash_attention.hip) or the rel-pos encoder kernels (flash_relpos.hip); returns the previous setting"""
    from espresso_amd import _lib
    return _lib.lib().ea_set_flash_relpos(0 if general else 1)


def check_flash_attention(B=3, H=4, T=150, S=None, relpos=True, causal=False, padded=True, drop_p=0.0, seed=0, general=False):
    """Fused attention forward vs the fp32 restatement (and, with dropout, vs the unfused softmax kernel's mask)."""
    from espresso_amd import kernels as K
    prev = _flash_impl(general)
    try:
        return _check_flash_attention(K, B, H, T, S, relpos, causal, padded, drop_p, seed)
    finally:
        _flash_impl(not prev)


def _check_flash_attention(K, B, H, T, S, relpos, causal, padded, drop_p, seed):
    dev = "cuda:0"
    S = S or T
    dh, C = 64, H * 64
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    qu = (rnd(B * T, C) * 0.35).to(torch.bfloat16)
    qv = (rnd(B * T, C) * 0.35).to(torch.bfloat16) if relpos else None
    kv = rnd(B * S, 2 * C).to(torch.bfloat16)
    k, v = kv[:, :C], kv[:, C:]
    pp = rnd(2 * T - 1, C).to(torch.bfloat16) if relpos else None
    klen = None
    if padded:
        klen = torch.randint(max(1, S // 3), S + 1, (B,), device=dev, generator=g).int()
        klen[0] = S
    out, lse, _ = K.flash_attention_fwd(qu, qv, k, v, pp, klen, H, B, T, S, C, 2 * C, C, causal=causal, drop_p=drop_p, drop_seed=1234,
                                        want_bits=True)
    s, p, vf = _attn_reference(qu, qv, k.contiguous(), v.contiguous(), pp, klen, H, B, T, S, causal)
    lse_ref = torch.logsumexp(s, dim=-1)  # [B][H][T]
    if drop_p > 0:
        # the unfused kernel applies keep(seed, (z*T+i)*S + j) to the same probabilities: reuse its mask
        Sp = (S + 7) // 8 * 8
        ac = torch.zeros(H * B * T, Sp, device=dev)
        ac[:, :S] = s.permute(1, 0, 2, 3).reshape(H * B * T, S)
        kl = klen if klen is not None else torch.full((B,), S, dtype=torch.int32, device=dev)
        P, Pd = K.relpos_softmax_fwd(ac, None, kl, None, H, B, T, S, Sp, 0, Sp, causal=False, drop_p=drop_p, drop_seed=1234)
        keep = (Pd.float() != 0) | (P.float() == 0)
        keep = keep[:, :S].view(H, B, T, S).permute(1, 0, 2, 3)
        p = p * keep / (1.0 - drop_p)
    o_ref = (p @ vf).permute(0, 2, 1, 3).reshape(B * T, C)
    torch.cuda.synchronize()
    lse_got = lse.view(H, B, T).permute(1, 0, 2)
    return {
        "out_abs": float((out.float() - o_ref).abs().max()),
        "out_ref_max": float(o_ref.abs().max()),
        "lse_abs": float((lse_got - lse_ref).abs().max()),
    }


def check_flash_attention_bwd(B=3, H=4, T=150, S=None, relpos=True, causal=False, padded=True, drop_p=0.0, seed=0, general=False):
    """Fused attention backward vs torch autograd through the fp32 restatement."""
    from espresso_amd import kernels as K
    prev = _flash_impl(general)
    try:
        return _check_flash_attention_bwd(K, B, H, T, S, relpos, causal, padded, drop_p, seed)
    finally:
        _flash_impl(not prev)


def _check_flash_attention_bwd(K, B, H, T, S, relpos, causal, padded, drop_p, seed):
    dev = "cuda:0"
    S = S or T
    dh, C = 64, H * 64
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    qu = (rnd(B * T, C) * 0.35).to(torch.bfloat16)
    qv = (rnd(B * T, C) * 0.35).to(torch.bfloat16) if relpos else None
    kv = rnd(B * S, 2 * C).to(torch.bfloat16)
    k, v = kv[:, :C], kv[:, C:]
    pp = rnd(2 * T - 1, C).to(torch.bfloat16) if relpos else None
    dout = rnd(B * T, C).to(torch.bfloat16)
    klen = None
    if padded:
        klen = torch.randint(max(1, S // 3), S + 1, (B,), device=dev, generator=g).int()
        klen[0] = S
    scaling = 0.125
    out, lse, bits = K.flash_attention_fwd(qu, qv, k, v, pp, klen, H, B, T, S, C, 2 * C, C, causal=causal, drop_p=drop_p, drop_seed=77,
                                           want_bits=True)
    dkv = torch.full((B * S, 2 * C), float("nan"), dtype=torch.bfloat16, device=dev)
    dq = torch.full((B * T, C + 8), float("nan"), dtype=torch.bfloat16, device=dev) if relpos else None  # (row pitch != C on purpose)
    t1, t2, dBD = K.flash_attention_bwd(qu, qv, k, v, pp, klen, out, dout, lse, dkv[:, :C], dkv[:, C:], H, B, T, S, C, 2 * C, 2 * C,
                                        ldpp=C, causal=causal, scaling=scaling, drop_p=drop_p, drop_seed=77, keep_bits=bits,
                                        dq=dq, lddq=C + 8)
    # reference
    leaf = lambda x: x.float().clone().requires_grad_(True) if x is not None else None
    qu_r, qv_r, k_r, v_r, pp_r = leaf(qu), leaf(qv), leaf(k.contiguous()), leaf(v.contiguous()), leaf(pp)
    s, p, vf = _attn_reference(qu_r, qv_r, k_r, v_r, pp_r, klen, H, B, T, S, causal)
    if drop_p > 0:
        Sp = (S + 7) // 8 * 8
        ac = torch.zeros(H * B * T, Sp, device=dev)
        ac[:, :S] = s.detach().permute(1, 0, 2, 3).reshape(H * B * T, S)
        kl = klen if klen is not None else torch.full((B,), S, dtype=torch.int32, device=dev)
        P, Pd = K.relpos_softmax_fwd(ac, None, kl, None, H, B, T, S, Sp, 0, Sp, causal=False, drop_p=drop_p, drop_seed=77)
        keep = ((Pd.float() != 0) | (P.float() == 0))[:, :S].view(H, B, T, S).permute(1, 0, 2, 3)
        p = p * keep / (1.0 - drop_p)
    o_ref = (p @ vf).permute(0, 2, 1, 3).reshape(B * T, C)
    (o_ref * dout.float()).sum().backward()
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.float() - b).abs().max() / b.abs().max().clamp_min(1e-6))
    res = {
        "t1": rel(t1, qu_r.grad * scaling),
        "dk": rel(dkv[:, :C], k_r.grad),
        "dv": rel(dkv[:, C:], v_r.grad),
        "finite": bool(torch.isfinite(dkv.float()).all()),
    }
    if relpos:
        res["t2"] = rel(t2, qv_r.grad * scaling)
        res["dq"] = rel(dq[:, :C], (qu_r.grad + qv_r.grad) * scaling)
        res["dq_pad_untouched"] = bool(torch.isnan(dq[:, C:].float()).all())
        R = 2 * T - 1
        d = dBD.float().view(H, B * T, -1)[:, :, :R]                       # [H][(b,i)][r]
        qvh = qv.float().view(B * T, H, dh).permute(1, 0, 2)               # [H][(b,i)][d]
        dpp = torch.bmm(d.transpose(1, 2), qvh).permute(1, 0, 2).reshape(R, C)
        res["dpp"] = rel(dpp, pp_r.grad)
        res["dBD_pad_zero"] = bool((dBD.float().view(H * B * T, -1)[:, R:] == 0).all())
    return res


# ------------------------------------------------------------------ LSTM layer / transducer
def check_lstm_layer(B=5, U=9, I=48, H=64, with_state=False, seed=0):
    """functional.lstm_layer (GEMM + cell kernels, BPTT) vs a torch.nn.LSTMCell loop in fp32 (same bf16-rounded inputs)."""
    from espresso_amd import functional as F
    from espresso_amd.models.speech_lstm import LSTMCellParams

    torch.manual_seed(seed)
    cell = LSTMCellParams(I, H).to(DEV)
    with torch.no_grad():
        for p in cell.parameters():
            p.mul_(3.0)  # U(-0.3, 0.3): gates leave the linear regime
    x0 = bf(torch.randn(U * B, I)).to(DEV)
    h0 = torch.randn(B, H, device=DEV) * 0.5 if with_state else None
    c0 = torch.randn(B, H, device=DEV) * 0.5 if with_state else None
    R = torch.randn(U * B, H, device=DEV)
    x = x0.clone().requires_grad_(True)
    h0g = h0.clone().requires_grad_(True) if with_state else None
    c0g = c0.clone().requires_grad_(True) if with_state else None
    hs, hl, cl = F.lstm_layer(x, cell, B, U, h0g, c0g)
    ((hs.float() * R).sum() + (hl * 0.3).sum() + (cl * 0.2).sum()).backward()
    got = {n: p.grad.detach().float().cpu().clone() for n, p in cell.named_parameters()}
    got["x"] = x.grad.float().cpu()
    if with_state:
        got["h0"], got["c0"] = h0g.grad.cpu(), c0g.grad.cpu()
    # reference
    ref = torch.nn.LSTMCell(I, H).to(DEV)
    with torch.no_grad():
        ref.weight_ih.copy_(bf(cell.weight_ih))
        ref.weight_hh.copy_(bf(cell.weight_hh))
        ref.bias_ih.copy_(cell.bias_ih)
        ref.bias_hh.copy_(cell.bias_hh)
    xr = x0.float().clone().requires_grad_(True)
    h = h0.clone().requires_grad_(True) if with_state else torch.zeros(B, H, device=DEV)
    c = c0.clone().requires_grad_(True) if with_state else torch.zeros(B, H, device=DEV)
    h_in, c_in = h, c
    outs = []
    for t in range(U):
        h, c = ref(xr[t * B:(t + 1) * B], (h, c))
        outs.append(h)
    hr = torch.cat(outs, 0)
    ((hr * R).sum() + (h * 0.3).sum() + (c * 0.2).sum()).backward()
    torch.cuda.synchronize()
    want = {n: p.grad.detach().float().cpu() for n, p in ref.named_parameters()}
    want["x"] = xr.grad.cpu()
    if with_state:
        want["h0"], want["c0"] = h_in.grad.cpu(), c_in.grad.cpu()
    res = {"hs_abs": float((hs.float() - hr).abs().max()), "c_last_abs": float((cl - c).abs().max())}
    for n in want:
        res["grad_" + n] = float((got[n] - want[n]).abs().max() / (want[n].abs().max() + 1e-9))
    return res


def check_lstm_persistent_vs_stepwise(B=5, U=11, I=96, H=256, with_state=False, reverse=False, ragged=False, seed=0):
    """The persistent whole-sequence kernels (csrc/lstm_seq.hip) against the per-step GEMM + cell-kernel path (itself pinned to
    torch.nn.LSTMCell above): same op, same bf16 storage points, only the fp32 summation order of the recurrent product
    differs.  ragged: packed-sequence semantics (frozen rows), as the BiLSTM encoder uses them."""
    from espresso_amd import functional as F
    from espresso_amd import kernels as K
    from espresso_amd.models.speech_lstm import LSTMCellParams

    assert K.lstm_seq_supported(B, H)
    torch.manual_seed(seed)
    cell = LSTMCellParams(I, H).to(DEV)
    with torch.no_grad():
        for p in cell.parameters():
            p.mul_(3.0 * (64.0 / H) ** 0.5 if H > 64 else 3.0)
    x0 = bf(torch.randn(U * B, I)).to(DEV)
    h0 = torch.randn(B, H, device=DEV) * 0.5 if with_state else None
    c0 = torch.randn(B, H, device=DEV) * 0.5 if with_state else None
    R = torch.randn(U * B, H, device=DEV)
    frozen = None
    if ragged:
        lens = torch.randint(1, U + 1, (B,), device=DEV)
        lens[0] = U
        frozen = (torch.arange(U, device=DEV).unsqueeze(1) >= lens.unsqueeze(0)).to(torch.uint8).contiguous()
    out = {}
    for mode in (True, False):
        old = F.set_lstm_persistent(mode)
        try:
            for p in cell.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            h0g = h0.clone().requires_grad_(True) if with_state else None
            c0g = c0.clone().requires_grad_(True) if with_state else None
            if ragged or reverse:
                hs = F.lstm_direction(x, cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, B, U, reverse=reverse, frozen=frozen)
                (hs.float() * R).sum().backward()
                vals = {"hs": hs.detach().float()}
            else:
                hs, hl, cl = F.lstm_layer(x, cell, B, U, h0g, c0g)
                ((hs.float() * R).sum() + (hl * 0.3).sum() + (cl * 0.2).sum()).backward()
                vals = {"hs": hs.detach().float(), "h_last": hl.detach(), "c_last": cl.detach()}
            for n, p in cell.named_parameters():
                vals["grad_" + n] = p.grad.detach().float().clone()
            vals["grad_x"] = x.grad.float()
            if with_state:
                vals["grad_h0"], vals["grad_c0"] = h0g.grad.clone(), c0g.grad.clone()
            out[mode] = vals
        finally:
            F.set_lstm_persistent(old)
    torch.cuda.synchronize()
    res = {"barrier_timeouts": F.lstm_barrier_timeouts()}
    for k in out[True]:
        a_, b_ = out[True][k], out[False][k]
        res[k] = float((a_ - b_).abs().max() / (b_.abs().max() + 1e-9))
    res["finite"] = all(bool(torch.isfinite(v).all()) for v in out[True].values())
    return res


def check_conv1_fused_backward(B=3, T=53, Fd=80, seed=0):
    """First sub-sampler layer: fused BatchNorm-backward + weight-gradient kernel (no dZ tensor) vs the separate kernels, same
    module, same inputs: identical bf16 dZ values by construction, so the gradients agree to fp32 summation order."""
    from espresso_amd import functional as F
    from espresso_amd.modules.speech_convolutions import ConvBNReLU

    torch.manual_seed(seed)
    conv = ConvBNReLU([64, 64, 128, 128], [(3, 3)] * 4, [(1, 1), (2, 2), (1, 1), (2, 2)], in_channels=1).to(DEV)
    with torch.no_grad():
        for bn in conv.batchnorms:
            bn.weight.add_(0.2 * torch.randn_like(bn.weight))
            bn.bias.add_(0.2 * torch.randn_like(bn.bias))
    x = torch.randn(B, T, Fd, device=DEV)
    lens = torch.tensor([T, T - 9, T // 2], device=DEV)[:B]
    out = {}
    for training in (True, False):
        conv.train(training)
        for mode in (True, False):
            old = F.set_conv1_fused_backward(mode)
            try:
                conv.zero_grad()
                y, _, _, _ = conv(x, lens)
                R = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
                (y.float() * R).sum().backward()
                out[(training, mode)] = {n: p.grad.detach().clone() for n, p in conv.named_parameters() if n.split(".")[1] == "0"}
            finally:
                F.set_conv1_fused_backward(old)
    torch.cuda.synchronize()
    res = {}
    for training in (True, False):
        a_, b_ = out[(training, True)], out[(training, False)]
        for n in a_:
            res[("train" if training else "eval") + ":" + n] = float((a_[n] - b_[n]).abs().max() / (b_[n].abs().max() + 1e-12))
    return res


def build_tiny_transducer(V=40, embed_dim=64, heads=4, dropout=0.0):
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerTransducerConfig
    from espresso_amd.models.transformer.speech_transformer_transducer_base import SpeechTransformerTransducerModelBase

    cfg = SpeechTransformerTransducerConfig()
    e, d = cfg.encoder, cfg.decoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = embed_dim, 128, 2, heads
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, "conformer"
    e.conv_channels = "[64, 64, 16, 16]"
    d.embed_dim, d.hidden_size, d.layers, d.residual, d.dropout_in, d.dropout_out = 48, 64, 2, True, dropout, dropout
    cfg.joint_dim = 64
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = dropout
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    return SpeechTransformerTransducerModelBase.build_model(cfg, _Task(V))


def check_transducer_vs_reference():
    """Reference weights -> HIP transducer model: logits (eval / train) and all gradients of sum(logits * R) against what the
    reference's own SpeechTransformerTransducerModelBase produced (tests/golden/ref_conformer_transducer_tiny.npz)."""
    g = np.load(os.path.join(GOLD, "ref_conformer_transducer_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    grads = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad::")}
    model = build_tiny_transducer().to(DEV)
    sd = model.upgrade_state_dict_named(dict(sd), "")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    feats, lengths = torch.from_numpy(g["feats"]).to(DEV), torch.from_numpy(g["lengths"]).to(DEV)
    prev = torch.from_numpy(g["prev"]).to(DEV)
    res = {}
    model.eval()
    with torch.no_grad():
        lo, olen = model(feats, lengths, prev)
    ref = torch.from_numpy(g["out::eval_logits"])
    ol = g["out::out_lengths"]
    valid = torch.zeros(ref.shape[:3], dtype=torch.bool)
    for b in range(ref.shape[0]):
        valid[b, : int(ol[b])] = True
    res["out_lengths_equal"] = olen.cpu().tolist() == ol.tolist()
    res["eval_logits_abs"] = float((lo.float().cpu() - ref)[valid].abs().max())
    res["eval_logits_ref_max"] = float(ref[valid].abs().max())
    model.train()
    lo, _ = model(feats, lengths, prev)
    reft = torch.from_numpy(g["out::train_logits"])
    res["train_logits_abs"] = float((lo.float().cpu() - reft)[valid].abs().max())
    R = torch.from_numpy(g["R"]).to(DEV)
    (lo.float() * R).sum().backward()
    torch.cuda.synchronize()
    # relu(LN(..) + LN(..)) sits on bf16 activations: elements within a bf16 ulp of the kink flip their derivative (about
    # 0.5 % of the lattice), which shows up as zero-mean noise on every upstream gradient.  The check is therefore on the
    # projection onto the reference gradient (scale) and the relative L2 error, max-norm only for the output layer.
    l2, scale, mx = {}, {}, {}
    for n, p in model.named_parameters():
        if n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias"):
            continue
        if n.endswith("attn.k_proj.bias"):
            continue
        r = grads[n]
        a = p.grad.float().cpu()
        l2[n] = float((a - r).norm() / (r.norm() + 1e-12))
        scale[n] = float((a * r).sum() / ((r * r).sum() + 1e-20))
        mx[n] = float((a - r).abs().max() / (float(r.abs().max()) + 1e-12))
    res["worst_l2"] = max(l2.items(), key=lambda kv: kv[1])
    res["worst_scale"] = max(scale.items(), key=lambda kv: abs(kv[1] - 1.0))
    res["fc_out_max"] = max(mx[k] for k in mx if k.startswith("fc_out"))
    # ---- the same pass on the bf16-emulating oracle (rounds where the HIP path stores) ----
    from oracle import torch_ref

    sd0 = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    sde = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "version" else v.clone())
           for k, v in sd0.items()}
    with torch_ref.bf16_emulation(True, flash=False):
        el, _ = torch_ref.transducer(torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"]), sde, H=4, residual=True,
                                     training=True)
        (el * torch.from_numpy(g["R"])).sum().backward()
    res["train_logits_vs_emulation"] = float((lo.detach().float().cpu() - el.detach())[valid].abs().max())
    l2e, mxe = {}, {}
    for n, p in model.named_parameters():
        if (n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias")) or n.endswith("attn.k_proj.bias"):
            continue
        ge = sde[n].grad
        if ge is None:
            continue
        a = p.grad.float().cpu()
        l2e[n] = float((a - ge).norm() / (ge.norm() + 1e-12))
        mxe[n] = float((a - ge).abs().max() / (float(ge.abs().max()) + 1e-12))
    res["n_vs_emulation"] = len(l2e)
    res["worst_l2_vs_emulation"] = max(l2e.items(), key=lambda kv: kv[1])
    res["median_l2_vs_emulation"] = sorted(l2e.values())[len(l2e) // 2]
    # the yardstick: how far the two ORACLE runs (the reference's fp32 gradients of the fixture vs the bf16-emulating restatement)
    # are apart on the same tensors — two bf16 realisations of a ReLU-kink network cannot agree better than bf16 agrees with fp32
    gap = sorted(float((sde[n].grad - grads[n]).norm() / (grads[n].norm() + 1e-12)) for n in l2e)
    res["oracle_gap_median_l2"], res["oracle_gap_worst_l2"] = gap[len(gap) // 2], gap[-1]
    res["worst_max_vs_emulation"] = max(mxe.items(), key=lambda kv: kv[1])
    res["median_max_vs_emulation"] = sorted(mxe.values())[len(mxe) // 2]
    return res


def check_transducer_loss_step():
    """End to end: HIP transducer model + transducer_loss criterion (bf16 logits -> RNN-T kernels) vs the float64 oracle loss
    on the oracle's logits; one backward to check every gradient is finite."""
    from espresso_amd.criterions.transducer_loss import TransducerLossCriterion
    from oracle import rnnt_ref, torch_ref

    g = np.load(os.path.join(GOLD, "ref_conformer_transducer_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = build_tiny_transducer().to(DEV)
    model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    task = _Task(40)
    crit = TransducerLossCriterion(task)
    feats, lengths, prev = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"])
    pad, eos = task.target_dictionary.pad(), task.target_dictionary.eos()
    target = torch.full_like(prev, pad)
    tl = []
    for b in range(prev.shape[0]):
        toks = [int(t) for t in prev[b, 1:] if int(t) != pad]
        tl.append(len(toks))
        target[b, : len(toks)] = torch.tensor(toks)
        target[b, len(toks)] = eos
    sample = {"net_input": {"src_tokens": feats.to(DEV), "src_lengths": lengths.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": int(sum(tl)) + len(tl)}
    model.train()
    loss, sample_size, log = crit(model, sample)
    loss.backward()
    torch.cuda.synchronize()
    lo, ol = torch_ref.transducer(feats, lengths, prev, sd, H=4, pad_idx=pad, residual=True, training=True, update={})
    want = 0.0
    for b in range(prev.shape[0]):
        want += rnnt_ref.rnnt_loss_one(lo[b, : int(ol[b]), : tl[b] + 1].detach().double().numpy(), target[b, : tl[b]].tolist(),
                                       blank=crit.blank_idx)
    finite = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    return {"loss": float(loss), "oracle_loss": float(want), "finite": finite, "sample_size": sample_size}


def _oracle_sd(model, prefix=""):
    """The HIP model's parameters / buffers under the reference's names as an oracle state dict with gradients enabled."""
    sd = {}
    for k, v in model.state_dict().items():
        if not k.startswith(prefix):
            continue
        v = v.detach().float().cpu().clone() if v.is_floating_point() else v.detach().cpu().clone()
        if v.is_floating_point() and "running" not in k and not k.endswith("version") and not k.endswith("_float_tensor"):
            v.requires_grad_(True)
        sd[k[len(prefix):]] = v
    return sd


def _grad_report(named_params, sde, sdf, skip):
    errs = _grad_errors(named_params, sde, skip)
    gap = {n: float((sde[n].grad - sdf[n].grad).abs().max() / (float(sdf[n].grad.abs().max()) + 1e-12)) for _, n in errs}
    gaps = sorted(gap.values())
    ex = max((e / max(0.08, gap[n]), n) for e, n in errs)
    return {"worst_grad_vs_emulation": (errs[0][1], errs[0][0]), "median_grad_vs_emulation": errs[len(errs) // 2][0],
            "worst_excess_over_bound": ex[0], "worst_excess_tensor": (ex[1], dict((n, e) for e, n in errs)[ex[1]], gap[ex[1]]),
            "oracle_gap_of_worst": gap[errs[0][1]],
            "median_oracle_gap": gaps[len(gaps) // 2], "n_grads": len(errs)}


def check_transducer_dropout_vs_oracle(p=0.1, seed=77):
    """speech_transformer_transducer_base (tiny fixture) in TRAINING mode with dropout everywhere (encoder sites, predictor
    dropout_in / dropout_out): logits and the gradients of sum(logits * R) vs the oracle with the HIP path's keep decisions."""
    from espresso_amd import _lib
    from espresso_amd import functional as F
    from oracle import dropout_ref as D
    from oracle import torch_ref

    g = np.load(os.path.join(GOLD, "ref_conformer_transducer_tiny.npz"))
    sd0 = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = build_tiny_transducer(dropout=p).to(DEV)
    missing, unexpected = model.load_state_dict(model.upgrade_state_dict_named(dict(sd0), ""), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    feats, lengths, prev = (torch.from_numpy(g[k]) for k in ("feats", "lengths", "prev"))
    R = torch.from_numpy(g["R"])
    model.train()
    F.set_dropout_seed(seed)
    with F.trace_dropout_seeds() as tr:
        lo, olen = model(feats.to(DEV), lengths.to(DEV), prev.to(DEV))
    (lo.float() * R.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    lo = lo.detach().float().cpu()
    res = {"sites": [e[0] for e in tr.entries]}
    lib = _lib.lib()
    skip = lambda n: (n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias")) or n.endswith("attn.k_proj.bias")

    def oracle(trace, emulate):
        sde = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "version" else v.clone())
               for k, v in sd0.items()}
        plan = D.MaskPlan(trace, lib.ea_layer_dropout_seed)
        with torch_ref.bf16_emulation(emulate, flash=False), torch_ref.dropout_masks(plan):
            el, ol = torch_ref.transducer(feats, lengths, prev, sde, H=4, residual=True, training=True)
            (el * R).sum().backward()
        plan.done()
        return el.detach(), ol, sde, plan

    el, ol, sde, plan = oracle(tr.entries, True)
    res["n_site_masks"] = len(plan.queue)
    valid = torch.zeros(el.shape[:3], dtype=torch.bool)
    for b in range(el.shape[0]):
        valid[b, : int(ol[b])] = True
    res["train_logits_vs_emulation"] = float((lo - el)[valid].abs().max())
    l2 = sorted(((float((p_.grad.float().cpu() - sde[n].grad).norm() / (float(sde[n].grad.norm()) + 1e-12)), n)
                 for n, p_ in model.named_parameters() if not skip(n) and sde[n].grad is not None), reverse=True)
    res["worst_l2_vs_emulation"] = (l2[0][1], l2[0][0])
    res["median_l2_vs_emulation"] = l2[len(l2) // 2][0]
    wl, _, sdw, _ = oracle([[s_, sd_ + 977 * 64, pp] for s_, sd_, pp in tr.entries], True)
    l2w = sorted(float((p_.grad.float().cpu() - sdw[n].grad).norm() / (float(sdw[n].grad.norm()) + 1e-12))
                 for n, p_ in model.named_parameters() if not skip(n) and sdw[n].grad is not None)
    res["wrong_mask_median_l2"] = l2w[len(l2w) // 2]
    res["wrong_mask_logits"] = float((lo - wl)[valid].abs().max())
    return res


def check_fullsize_encdec_vs_oracle(dropout=0.0, seed=0, lens=(330, 211), tl=(24, 15), V=5003):
    """BASELINE config 2 at the recipe's size (examples/asr_librispeech/config/transformer_librispeech.yaml: 12 Transformer
    encoder layers with LEARNED relative-position tables, 6 decoder layers, 512 / 8 / 2048, V = 5003 sentence pieces + specials)
    with random weights on two utterances: eval logits, label-smoothed CE and every gradient vs the pinned oracle (fp32 and
    bf16-emulating); `dropout` > 0: the recipe's training mode with the HIP path's keep decisions fed to the oracle."""
    from espresso_amd import _lib
    from espresso_amd import functional as F
    from espresso_amd.models.transformer.speech_transformer_base import SpeechTransformerModelBase
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerConfig
    from oracle import dropout_ref as D
    from oracle import torch_ref

    torch.manual_seed(seed)
    H = 8
    cfg = SpeechTransformerConfig()
    e, dc = cfg.encoder, cfg.decoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = 512, 2048, 12, H
    e.normalize_before, e.relative_positional_embeddings, e.layer_type, e.learned_pos = True, True, "transformer", True
    e.conv_channels = "[64, 64, 128, 128]"
    dc.embed_dim, dc.ffn_embed_dim, dc.layers, dc.attention_heads, dc.normalize_before = 512, 2048, 6, H, True
    dc.input_dim = dc.output_dim = 512
    cfg.layernorm_embedding = True
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = dropout
    cfg.max_source_positions, cfg.max_target_positions = 3600, 1024
    model = SpeechTransformerModelBase.build_model(cfg, _TaskAR(V))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    g = torch.Generator().manual_seed(seed + 1)
    feats = torch.zeros(len(lens), max(lens), 80)
    for b, n in enumerate(lens):
        feats[b, :n] = torch.randn(n, 80, generator=g)
    lengths = torch.tensor(lens)
    U = max(tl) + 1
    target = torch.zeros(len(lens), U, dtype=torch.long)  # pad = 0, eos = 1 (dictionary without <s>)
    prev = torch.zeros(len(lens), U, dtype=torch.long)
    for b, n in enumerate(tl):
        toks = torch.randint(3, V, (n,), generator=g)
        target[b, :n], target[b, n] = toks, 1
        prev[b, 0], prev[b, 1:n + 1] = 1, toks
    valid = target.ne(0)
    res = {}
    model.eval()
    with torch.no_grad():
        lo, _ = model(feats.to(DEV), lengths.to(DEV), prev.to(DEV))
    hip_eval = lo.float().cpu()
    model.train()
    F.set_dropout_seed(seed + 17)
    with F.trace_dropout_seeds() as tr:
        lo, extra = model(feats.to(DEV), lengths.to(DEV), prev.to(DEV))
    loss, nll = F.label_smoothed_ce(extra["_logits_bu"], target.reshape(-1).to(DEV).to(torch.int32).contiguous(), 0, 0.1)
    loss.backward()
    torch.cuda.synchronize()
    res["hip_loss"] = float(loss.detach())
    skip = lambda n: (n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias")) or n.endswith("attn.k_proj.bias")
    out = {}
    for tag, emu in (("fp32", False), ("emu", True)):
        sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and not k.endswith("version")
                   and not k.endswith("_float_tensor") else v.clone()) for k, v in sd0.items()}
        with torch_ref.bf16_emulation(emu, flash=True):
            with torch.no_grad():
                le = torch_ref.encdec(feats, lengths, prev, sdo, H, 0, training=False)
            plan = D.MaskPlan(tr.entries, _lib.lib().ea_layer_dropout_seed)
            with torch_ref.dropout_masks(plan if dropout > 0 else None):
                lt = torch_ref.encdec(feats, lengths, prev, sdo, H, 0, training=True)
            plan.done()
            oloss, _ = torch_ref.label_smoothed_nll(lt.reshape(-1, lt.shape[-1]), target.reshape(-1), 0.1, 0)
            oloss.backward()
        res[f"{tag}_loss"] = float(oloss.detach())
        res[f"eval_logits_vs_{tag}"] = float((hip_eval - le)[valid].abs().max())
        res["logit_scale"] = float(le[valid].abs().max())
        res["n_site_masks"] = len(plan.queue)
        out[tag] = sdo
    res.update(_grad_report(model.named_parameters(), out["emu"], out["fp32"], skip))
    return res


def check_fullsize_transducer_vs_oracle(dropout=0.0, seed=0, lens=(200, 140), tl=(6, 4), V=5004, layers=16):
    """BASELINE config 4 at the recipe's size (conformer_transducer_librispeech.yaml:66-88: Conformer-16 512 / 8 / 2048, LSTM
    predictor 2 x 512, joint 512, V = 5004 -> the 5056-column logit pitch) with random weights on one short utterance pair: the
    `transducer_loss` criterion's value and every gradient vs the pinned oracle (encoder + predictor + joint restatement, RNN-T
    loss restatement oracle/rnnt_ref.py), fp32 and bf16-emulating; `dropout` > 0: with the HIP path's keep decisions."""
    from espresso_amd import _lib
    from espresso_amd import functional as F
    from espresso_amd.criterions.transducer_loss import TransducerLossCriterion
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerTransducerConfig
    from espresso_amd.models.transformer.speech_transformer_transducer_base import SpeechTransformerTransducerModelBase
    from oracle import dropout_ref as D
    from oracle import rnnt_ref, torch_ref

    torch.manual_seed(seed)
    H = 8
    cfg = SpeechTransformerTransducerConfig()
    e, dc = cfg.encoder, cfg.decoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = 512, 2048, layers, H
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, "conformer"
    e.conv_channels = "[64, 64, 128, 128]"
    dc.embed_dim, dc.hidden_size, dc.layers, dc.dropout_in, dc.dropout_out = 512, 512, 2, dropout, dropout
    cfg.joint_dim = 512
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = dropout
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    task = _Task(V)
    model = SpeechTransformerTransducerModelBase.build_model(cfg, task)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and "weight_g" not in n:
                p.add_(0.1 * torch.randn_like(p))
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    crit = TransducerLossCriterion(task)
    pad, eos = task.target_dictionary.pad(), task.target_dictionary.eos()
    g = torch.Generator().manual_seed(seed + 1)
    feats = torch.zeros(len(lens), max(lens), 80)
    for b, n in enumerate(lens):
        feats[b, :n] = torch.randn(n, 80, generator=g)
    lengths = torch.tensor(lens)
    U1 = max(tl) + 1
    prev = torch.full((len(lens), U1), pad, dtype=torch.long)
    target = torch.full((len(lens), U1), pad, dtype=torch.long)
    for b, n in enumerate(tl):
        toks = torch.randint(5, V, (n,), generator=g)
        prev[b, 0], prev[b, 1:n + 1] = eos, toks
        target[b, :n], target[b, n] = toks, eos
    sample = {"net_input": {"src_tokens": feats.to(DEV), "src_lengths": lengths.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": int(sum(tl)) + len(tl)}
    model.train()
    F.set_dropout_seed(seed + 17)
    with F.trace_dropout_seeds() as tr:
        loss, sample_size, log = crit(model, sample)
    loss.backward()
    torch.cuda.synchronize()
    res = {"hip_loss": float(loss.detach()), "n_seed_draws": len(tr.entries)}
    skip = lambda n: (n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias")) or n.endswith("attn.k_proj.bias")
    out = {}
    for tag, emu in (("fp32", False), ("emu", True)):
        sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and not k.endswith("version")
                   else v.clone()) for k, v in sd0.items()}
        plan = D.MaskPlan(tr.entries, _lib.lib().ea_layer_dropout_seed)
        # (joint_logits_f32: the criterion fuses the output layer with the loss — the lattice logits stay fp32 accumulators)
        with torch_ref.bf16_emulation(emu, flash=True, joint_logits_f32=True), torch_ref.dropout_masks(plan if dropout > 0 else None):
            lt, ol = torch_ref.transducer(feats, lengths, prev, sdo, H=H, pad_idx=pad, residual=False, training=True, update={})
        plan.done()
        oloss = 0.0
        for b in range(len(lens)):
            oloss = oloss + rnnt_ref.rnnt_loss_torch(lt[b, : int(ol[b]), : tl[b] + 1], target[b, : tl[b]].tolist(), crit.blank_idx)
        oloss.backward()
        res[f"{tag}_loss"] = float(oloss.detach())
        res["n_site_masks"] = len(plan.queue)
        out[tag] = sdo
    res.update(_grad_report(model.named_parameters(), out["emu"], out["fp32"], skip))
    # relu(E + D) kinks make every upstream gradient noisy (measured on the tiny fixture): the L2 view as in check_transducer_vs_reference
    l2 = sorted(((float((p_.grad.float().cpu() - out["emu"][n].grad).norm() / (float(out["emu"][n].grad.norm()) + 1e-12)), n)
                 for n, p_ in model.named_parameters() if not skip(n) and p_.grad is not None and out["emu"][n].grad is not None), reverse=True)
    res["worst_l2_vs_emulation"] = (l2[0][1], l2[0][0])
    res["median_l2_vs_emulation"] = l2[len(l2) // 2][0]
    l2g = sorted(float((out["emu"][n].grad - out["fp32"][n].grad).norm() / (float(out["fp32"][n].grad.norm()) + 1e-12))
                 for _, n in l2)
    res["median_l2_oracle_gap"] = l2g[len(l2g) // 2]
    res["worst_l2_oracle_gap"] = l2g[-1]
    return res


def check_joint_fp32_islands(B=3, T=29, U1=7, J=512, seed=0):
    """The fp32-island kernels of the transducer joint (round 6) against plain fp32 torch on the CPU: LayerNorm with an fp32 output
    (`ea_layernorm_fwd_f32out`) and its backward from an fp32 gradient (`ea_layernorm_bwd_f32dy`), relu(E + D) on fp32 operands
    (`ea_joint_add_relu_f32`: equals bf16(relu(fp32 sum)) bit for bit) and the fp32 reductions dE / dD (`ea_joint_reduce_f32`)."""
    from espresso_amd import kernels as K

    g = torch.Generator().manual_seed(seed)
    M = B * T
    x = torch.randn(M, J, generator=g).to(torch.bfloat16)
    gam, bet = 1 + 0.2 * torch.randn(J, generator=g), 0.1 * torch.randn(J, generator=g)
    dy = torch.randn(M, J, generator=g)
    y, mean, rstd = K.layernorm_fwd(x.to(DEV), gam.to(DEV), bet.to(DEV), out_f32=True)
    dg, db = torch.zeros(J, device=DEV), torch.zeros(J, device=DEV)
    dx = K.layernorm_bwd(x.to(DEV), dy.to(DEV), gam.to(DEV), mean, rstd, dg, db)
    xr, gr, br = x.float().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (J,), gr, br, 1e-5)
    (yr * dy).sum().backward()
    res = {"y_dtype_f32": y.dtype == torch.float32, "ln_y_abs": float((y.cpu() - yr.detach()).abs().max()),
           "ln_dx_rel": float((dx.float().cpu() - xr.grad).abs().max() / xr.grad.abs().max()),
           "ln_dg_rel": float((dg.cpu() - gr.grad).abs().max() / gr.grad.abs().max()),
           "ln_db_rel": float((db.cpu() - br.grad).abs().max() / br.grad.abs().max())}
    E = torch.randn(B * T, J, generator=g)
    D = torch.randn(B * U1, J, generator=g)
    Z = K.joint_add_relu(E.to(DEV), D.to(DEV), B, T, U1)
    zr = torch.relu(E.view(B, T, 1, J) + D.view(B, 1, U1, J)).to(torch.bfloat16).reshape(-1, J)
    res["relu_bits_equal"] = bool(torch.equal(Z.cpu(), zr))
    dZ = torch.randn(B * T * U1, J, generator=g).to(torch.bfloat16)
    dE, dD = K.joint_reduce(dZ.to(DEV), B, T, U1, out_f32=True)
    d4 = dZ.float().view(B, T, U1, J)
    res["dE_abs"] = float((dE.cpu() - d4.sum(2).reshape(-1, J)).abs().max())
    res["dD_abs"] = float((dD.cpu() - d4.sum(1).reshape(-1, J)).abs().max())
    res["reduce_dtype_f32"] = dE.dtype == torch.float32 and dD.dtype == torch.float32
    return res


def check_joint_rnnt_fused(B=3, T=37, U1=9, V=40, J=64, seed=0):
    """csrc/joint_rnnt.hip (output layer fused with the RNN-T loss, logits never written) against the unfused kernels of
    csrc/rnnt.hip fed with the SAME logits in fp32 (torch fp32 product of the bf16 operands): per-utterance loss, the gradient of
    the logits (bf16, padded pitch, pad columns and rows outside the lattice zero) on ragged lengths."""
    from espresso_amd import kernels as K

    g = torch.Generator().manual_seed(seed)
    n = B * T * U1
    Z = (torch.randn(n, J, generator=g).clamp_min(0) * 0.7).to(torch.bfloat16).to(DEV)
    W = (torch.randn(V, J, generator=g) * (2.0 / J ** 0.5)).to(torch.bfloat16).to(DEV)
    bias = (torch.randn(V, generator=g) * 0.3).to(DEV)
    Tl = torch.tensor([T] + [max(1, T - 3 - 5 * i) for i in range(1, B)], dtype=torch.int32)
    Ul = torch.tensor([U1 - 1] + [max(0, U1 - 2 - 2 * i) for i in range(1, B)], dtype=torch.int32)
    blank = 0
    tg = torch.randint(1, V, (B, U1 - 1), generator=g, dtype=torch.int32)
    tgd, Tld, Uld = tg.to(DEV), Tl.to(DEV), Ul.to(DEV)
    logits = (Z.float() @ W.float().t() + bias).view(B, T, U1, V).contiguous()
    loss_u, ws_u = K.rnnt_loss_fwd(logits, tgd, Tld, Uld, blank)
    scale = torch.full((1,), 0.37, device=DEV)
    grad_u = K.rnnt_loss_grad(logits, tgd, Tld, Uld, loss_u, ws_u, blank, grad_scale_dev=scale, grad_bf16=False).reshape(n, V)
    assert K.joint_rnnt_supported(Z, W)
    loss_f, ws_f = K.joint_rnnt_loss_fwd(Z, W, bias, tgd, Tld, Uld, B, T, U1, blank)
    Vp = (V + 63) // 64 * 64
    dl = K.joint_rnnt_loss_grad(Z, W, bias, tgd, Tld, Uld, loss_f, ws_f, B, T, U1, blank, Vp, grad_scale_dev=scale)
    torch.cuda.synchronize()
    gu, gf = grad_u.float().cpu(), dl[:, :V].float().cpu()
    # bf16 storage of the fused gradient: half a unit in the last place of each element (2^-9 relative) + fp32 summation order
    tol = gu.abs() * 2.0 ** -8 + 1e-6 * float(gu.abs().max())
    inside = torch.zeros(B, T, U1, dtype=torch.bool)
    for b in range(B):
        inside[b, : int(Tl[b]), : int(Ul[b]) + 1] = True
    return {"loss_rel": float(((loss_f - loss_u).abs() / loss_u.abs().clamp_min(1e-6)).max()), "loss": loss_u.cpu().tolist(),
            "grad_excess": float(((gf - gu).abs() - tol).max()), "grad_scale": float(gu.abs().max()),
            "pad_zero": bool((dl[:, V:] == 0).all()), "outside_zero": bool((dl.view(B, T, U1, Vp)[~inside.to(DEV)] == 0).all()),
            "finite": bool(torch.isfinite(dl.float()).all() and torch.isfinite(loss_f).all())}


def check_transducer_fused_vs_materialised_criterion():
    """`transducer_loss` on the tiny transducer with the fused output layer + loss (default) vs the materialised logits
    (F.set_joint_fused(False)): same weights, same batch -> the same loss up to the bf16 rounding of the logits the unfused path
    stores, gradients within the noise of that rounding."""
    from espresso_amd import functional as F
    from espresso_amd.criterions.transducer_loss import TransducerLossCriterion

    g = np.load(os.path.join(GOLD, "ref_conformer_transducer_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    task = _Task(40)
    feats, lengths, prev = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"])
    pad, eos = task.target_dictionary.pad(), task.target_dictionary.eos()
    target = torch.full_like(prev, pad)
    tl = []
    for b in range(prev.shape[0]):
        toks = [int(t) for t in prev[b, 1:] if int(t) != pad]
        tl.append(len(toks))
        target[b, : len(toks)] = torch.tensor(toks)
        target[b, len(toks)] = eos
    sample = {"net_input": {"src_tokens": feats.to(DEV), "src_lengths": lengths.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": int(sum(tl)) + len(tl)}
    out = {}
    for fused in (True, False):
        old = F.set_joint_fused(fused)
        try:
            model = build_tiny_transducer().to(DEV)
            model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
            model.train()
            loss, _, _ = TransducerLossCriterion(task)(model, sample)
            loss.backward()
            torch.cuda.synchronize()
            out[fused] = (float(loss), {n: p.grad.float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None})
        finally:
            F.set_joint_fused(old)
    (la, ga), (lb, gb) = out[True], out[False]
    names = [n for n in gb if not (n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias")) and not n.endswith("attn.k_proj.bias")]
    l2 = sorted(float((ga[n] - gb[n]).norm() / (gb[n].norm() + 1e-12)) for n in names)
    return {"loss_fused": la, "loss_materialised": lb, "loss_rel": abs(la - lb) / abs(lb), "median_l2": l2[len(l2) // 2], "worst_l2": l2[-1],
            "same_params": set(ga) == set(gb), "n_grads": len(ga)}


def check_transducer_branch_overlap():
    """The transducer's multi-stream schedule (predictor network and the output layer's weight gradient on their own streams,
    functional.set_branch_overlap) against the single-stream schedule: same weights, same batch, same dropout seed -> the
    same loss and the same gradients up to the order of fp32 atomic adds in the split-K reductions.  Three passes each way
    (gradients accumulate), so a missing join between streams shows up as a difference."""
    from espresso_amd import functional as F
    from espresso_amd.criterions.transducer_loss import TransducerLossCriterion

    g = np.load(os.path.join(GOLD, "ref_conformer_transducer_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    task = _Task(40)
    feats, lengths, prev = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"])
    pad, eos = task.target_dictionary.pad(), task.target_dictionary.eos()
    target = torch.full_like(prev, pad)
    tl = []
    for b in range(prev.shape[0]):
        toks = [int(t) for t in prev[b, 1:] if int(t) != pad]
        tl.append(len(toks))
        target[b, : len(toks)] = torch.tensor(toks)
        target[b, len(toks)] = eos
    sample = {"net_input": {"src_tokens": feats.to(DEV), "src_lengths": lengths.to(DEV), "prev_output_tokens": prev.to(DEV)},
              "target": target.to(DEV), "ntokens": int(sum(tl)) + len(tl)}
    out = {}
    for mode in (True, False):
        old = F.set_branch_overlap(mode)
        try:
            model = build_tiny_transducer().to(DEV)
            model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
            crit = TransducerLossCriterion(task)
            model.train()
            losses = []
            for it in range(3):
                F.set_dropout_seed(100 + it)
                loss, _, _ = crit(model, sample)
                loss.backward()
                losses.append(float(loss))
            torch.cuda.synchronize()
            out[mode] = (losses, {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None})
        finally:
            F.set_branch_overlap(old)
    (la, ga), (lb, gb) = out[True], out[False]
    # (the sub-sampler's convolution biases feed BatchNorm: their gradient is zero up to rounding noise, a ratio of two noises)
    names = [n for n in gb if not (n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias"))]
    worst = max(((float((ga[n] - gb[n]).abs().max() / (gb[n].abs().max() + 1e-20)), n) for n in names), key=lambda kv: kv[0])
    return {"loss_rel": max(abs(a - b) / abs(b) for a, b in zip(la, lb)), "worst_grad_rel": worst, "same_params": set(ga) == set(gb),
            "n_grads": len(ga)}


def check_transducer_greedy_decoder():
    """HIP greedy transducer search vs the reference's TransducerGreedyDecoder on the same weights (fixture): token
    alignments identical, summed log-probs within the bf16 tolerance."""
    from espresso_amd.tools.transducer_greedy_decoder import TransducerGreedyDecoder

    g = np.load(os.path.join(GOLD, "ref_conformer_transducer_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = build_tiny_transducer().to(DEV)
    model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    model.eval()
    d = _Task(40).target_dictionary
    sample = {"net_input": {"src_tokens": torch.from_numpy(g["feats"]).to(DEV), "src_lengths": torch.from_numpy(g["lengths"]).to(DEV)}}
    res = {}
    for tag, kw in (("e2", dict(max_num_expansions_per_step=2)), ("e1_eos", dict(max_num_expansions_per_step=1, model_predicts_eos=True))):
        dec = TransducerGreedyDecoder([model], d, print_alignment=True, **kw)
        toks, scores, ali = dec._generate(sample)
        ref_t, ref_s = g[f"out::greedy_{tag}_tokens"], g[f"out::greedy_{tag}_scores"]
        res[tag] = {"tokens_equal": bool((toks.cpu().numpy() == ref_t).all()) if toks.shape == tuple(ref_t.shape) else False,
                    "score_rel": float(np.abs(scores.cpu().numpy() - ref_s).max() / np.abs(ref_s).max()),
                    "agree": float((toks.cpu().numpy() == ref_t).mean()) if toks.shape == tuple(ref_t.shape) else 0.0}
    return res


# ------------------------------------------------------------------ LM fusion
def build_lookahead_from_fixture(g):
    from espresso_amd.data.asr_dictionary import AsrDictionary
    from espresso_amd.models.lstm_lm import LSTMLanguageModelEspresso
    from espresso_amd.models.tensorized_lookahead_language_model import TensorizedLookaheadLanguageModel

    wd = AsrDictionary.from_symbols([str(w) for w in g["words"]], enable_bos=False, add_space=False)
    sd_ = AsrDictionary.from_symbols([str(c) for c in g["chars"]], enable_bos=False)

    class T:
        word_dictionary = target_dictionary = source_dictionary = wd
    lm = LSTMLanguageModelEspresso.build_model(dict(arch="lstm_lm_wsj", decoder_embed_dim=16, decoder_hidden_size=24, decoder_layers=2,
                                                    decoder_out_embed_dim=24, dropout=0.0, share_embed=False, is_wordlm=True), T)
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    missing, unexpected = lm.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    lm = lm.to(DEV).eval()
    return TensorizedLookaheadLanguageModel(lm, sd_, oov_penalty=1e-4, open_vocab=True), wd, sd_


def check_lookahead_lm():
    """Look-ahead word LM on the HIP kernels vs the log-probs the reference's _TensorizedLookaheadLanguageModelDecoder emitted
    along scripted hypotheses (fixture from oracle/gen_golden.py lookahead): OOV path, word ends, leaving the tree, a reorder."""
    g = np.load(os.path.join(GOLD, "ref_lookahead_wordlm_tiny.npz"))
    la, wd, sd_ = build_lookahead_from_fixture(g)
    ref = g["lprobs"]  # [calls][B][Vs]
    last = g["last_tok"]
    orders = g["orders"]
    B = ref.shape[1]
    state = la.init_incremental(B, 1)
    worst, floor_equal = 0.0, True
    for k in range(ref.shape[0]):
        toks = torch.from_numpy(last[k]).to(DEV).view(B, 1)
        parent = torch.from_numpy(orders[k - 1]).to(DEV) if k > 0 else None
        out = la.step(state, toks, k, parent).cpu().numpy()
        fl = ref[k] < -20.0  # clamped "zero" entries: log(1e-10)
        floor_equal = floor_equal and bool(((out < -20.0) == fl).all())
        worst = max(worst, float(np.abs(out - ref[k])[~fl].max()))
    return {"max_abs": worst, "floor_pattern_equal": floor_equal}


def check_lm_fusion_beam_search():
    """Shallow fusion (sequence_generator.py:385-393) with a sub-word LSTM LM: HIP generator vs the reference generator's
    beams (fixture ref_lm_fusion_tiny.npz, model weights from ref_transformer_encdec_tiny.npz)."""
    from espresso_amd.models.lstm_lm import LSTMLanguageModelEspresso
    from espresso_amd.sequence_generator import SequenceGenerator

    g = np.load(os.path.join(GOLD, "ref_transformer_encdec_tiny.npz"))
    gl = np.load(os.path.join(GOLD, "ref_lm_fusion_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = build_tiny_encdec().to(DEV)
    model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    model.eval()
    task = _TaskAR(40)
    d = task.target_dictionary
    lm = LSTMLanguageModelEspresso.build_model(dict(arch="lstm_lm_wsj", decoder_embed_dim=24, decoder_hidden_size=32, decoder_layers=2,
                                                    decoder_out_embed_dim=32, dropout=0.0, share_embed=False), task)
    missing, unexpected = lm.load_state_dict({k[4:]: torch.from_numpy(gl[k]) for k in gl.files if k.startswith("lm::")}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    lm = lm.to(DEV).eval()
    sample = {"net_input": {"src_tokens": torch.from_numpy(g["feats"]).to(DEV), "src_lengths": torch.from_numpy(g["lengths"]).to(DEV)}}
    res = {}
    for tag, kw in (("lm05", dict(beam_size=3, max_len_a=0.0, max_len_b=12, lm_weight=0.5)),
                    ("lm10_eosf", dict(beam_size=3, max_len_a=0.0, max_len_b=12, lm_weight=1.0, eos_factor=1.5))):
        gen = SequenceGenerator([model], d, lm_model=lm, **kw)
        hyps = gen.generate([model], sample)
        in_beam, score_err = [], 0.0
        for b, hl in enumerate(hyps):
            refset, hi = {}, 0
            while f"beam::{tag}::{b}::{hi}::tokens" in gl.files:
                refset[tuple(gl[f"beam::{tag}::{b}::{hi}::tokens"].tolist())] = float(gl[f"beam::{tag}::{b}::{hi}::score"])
                hi += 1
            in_beam.append(sum(tuple(h["tokens"].tolist()) in refset for h in hl) / len(hl))
            for h in hl:
                k = tuple(h["tokens"].tolist())
                if k in refset:
                    score_err = max(score_err, abs(float(h["score"]) - refset[k]))
        res[tag] = {"frac_hyps_in_reference_beam": in_beam, "score_abs": score_err}
    # force-decode the reference's best fused hypothesis: per-position fused scores must match its positional scores
    tag, lmw = "lm05", 0.5
    toks = torch.stack([torch.from_numpy(gl[f"beam::{tag}::{b}::0::tokens"]) for b in range(3)]).to(DEV)
    ref_pos = torch.stack([torch.from_numpy(gl[f"beam::{tag}::{b}::0::pos"]) for b in range(3)])
    with torch.no_grad():
        enc_out = model.forward_encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"])
        st = model.decoder.init_incremental(enc_out, 3, 1)
        lst = lm.init_incremental(3, 1)
        cur = torch.full((3, 1), d.eos(), dtype=torch.long, device=DEV)
        got = []
        for step in range(toks.shape[1]):
            par = None if step == 0 else torch.arange(3, device=DEV)
            lp = model.decoder.step(st, cur, step, par) + lmw * lm.step(lst, cur, step, par)
            got.append(lp.gather(-1, toks[:, step:step + 1]).squeeze(-1).cpu())
            cur = torch.cat([cur, toks[:, step:step + 1]], 1)
    got = torch.stack(got, 1)
    res["forced_decode_pos_score_abs"] = float((got[:, :-1] - ref_pos[:, :-1]).abs().max())  # last = forced EOS at max_len
    return res


def check_label_smoothing_kernel():
    """ea_label_smoothed_ce (uniform / unigram / temporal) vs the reference's label_smoothed_nll_loss outputs (fixture)."""
    from espresso_amd import functional as F

    g = np.load(os.path.join(GOLD, "label_smoothing.npz"))
    res = {}
    for kind in ("uniform", "unigram", "temporal"):
        tgt = torch.from_numpy(g["target" if kind == "uniform" else "target2"]).to(DEV).to(torch.int32)
        x = torch.from_numpy(g["logits"]).to(DEV).requires_grad_(True)
        prior = torch.from_numpy(g["prior"]).to(DEV) if kind == "unigram" else None
        loss, nll = F.label_smoothed_ce(x, tgt, 1, 0.1, kind, prior, tgt.numel())
        loss.backward()
        ref_loss = float(g["loss_0.1"] if kind == "uniform" else g[f"{kind}_loss"])
        ref_grad = torch.from_numpy(g["dlogits" if kind == "uniform" else f"{kind}_dlogits"])
        res[kind] = {"loss_rel": abs(float(loss) - ref_loss) / abs(ref_loss), "grad_abs": float((x.grad.cpu() - ref_grad).abs().max())}
    return res


# ------------------------------------------------------------------ speech_lstm (BASELINE config 1)
def build_tiny_speech_lstm(V=40):
    from espresso_amd.models.speech_lstm import SpeechLSTMModel

    return SpeechLSTMModel.build_model(dict(arch="speech_lstm", dropout=0.0, encoder_conv_channels="[64, 64, 16, 16]",
                                            encoder_rnn_hidden_size=32, encoder_rnn_layers=2, encoder_rnn_residual=True,
                                            decoder_embed_dim=24, decoder_hidden_size=32, decoder_layers=2, decoder_out_embed_dim=48,
                                            attention_dim=40), _TaskAR(V))


def check_speech_lstm_vs_reference():
    """Reference weights -> HIP speech_lstm: logits (eval / train), label-smoothed CE and all gradients vs what the reference's
    SpeechLSTMModel produced (tests/golden/ref_speech_lstm_tiny.npz)."""
    from espresso_amd import functional as F

    g = np.load(os.path.join(GOLD, "ref_speech_lstm_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    grads = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("grad::")}
    model = build_tiny_speech_lstm().to(DEV)
    missing, unexpected = model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    feats, lengths = torch.from_numpy(g["feats"]).to(DEV), torch.from_numpy(g["lengths"]).to(DEV)
    prev, target = torch.from_numpy(g["prev"]).to(DEV), torch.from_numpy(g["target"]).to(DEV)
    valid = target.ne(0).cpu()
    res = {}
    model.eval()
    with torch.no_grad():
        lo, _ = model(feats, lengths, prev)
    ref = torch.from_numpy(g["out::eval_logits"])
    res["eval_logits_abs_valid"] = float((lo.float().cpu() - ref)[valid].abs().max())
    res["logits_ref_max"] = float(ref[valid].abs().max())
    model.train()
    lo, _ = model(feats, lengths, prev)
    res["train_logits_abs_valid"] = float((lo.float().cpu() - torch.from_numpy(g["out::train_logits"]))[valid].abs().max())
    V = lo.shape[-1]
    loss, nll = F.label_smoothed_ce(lo.reshape(-1, V), target.reshape(-1).to(torch.int32).contiguous(), 0, 0.1)
    loss.backward()
    torch.cuda.synchronize()
    res["loss"], res["ref_loss"] = float(loss), float(g["out::loss"])
    l2, scale = {}, {}
    for n, p in model.named_parameters():
        if n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias"):
            continue
        r = grads[n]
        a = p.grad.float().cpu()
        l2[n] = float((a - r).norm() / (r.norm() + 1e-12))
        scale[n] = float((a * r).sum() / ((r * r).sum() + 1e-20))
    # the conv/BatchNorm front-end sits behind the whole recurrent stack: its bf16 gradient noise is bounded separately
    fe = {k: v for k, v in l2.items() if k.startswith("encoder.pre_encoder")}
    rest = {k: v for k, v in l2.items() if not k.startswith("encoder.pre_encoder")}
    res["worst_l2"] = max(rest.items(), key=lambda kv: kv[1])
    res["worst_l2_frontend"] = max(fe.items(), key=lambda kv: kv[1])
    res["worst_scale"] = max(((k, v) for k, v in scale.items() if not k.startswith("encoder.pre_encoder")), key=lambda kv: abs(kv[1] - 1.0))
    # ---- the same pass on the bf16-emulating oracle (rounds where the HIP path stores: hidden states, attention projections,
    # contexts, residual sums, additional_fc output; fp32 gates / cell state / scores / logits) ----
    from oracle import torch_ref

    sde = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone()) for k, v in sd.items()}
    with torch_ref.bf16_emulation(True, flash=False):
        el, _, _ = torch_ref.speech_lstm(torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"]), sde,
                                         enc_residual=True, dec_residual=True, pad_idx=0, training=True, update={})
        le, _ = torch_ref.label_smoothed_nll(el.reshape(-1, el.shape[-1]), torch.from_numpy(g["target"]).reshape(-1), 0.1, 0)
        le.backward()
    res["train_logits_vs_emulation"] = float((lo.detach().float().cpu() - el.detach())[valid].abs().max())
    res["loss_vs_emulation_rel"] = abs(float(loss) - float(le)) / float(le)
    l2e = {}
    for n, p in model.named_parameters():
        if n.startswith("encoder.pre_encoder.convolutions.") and n.endswith(".bias"):
            continue
        r = sde[n].grad
        l2e[n] = float((p.grad.float().cpu() - r).norm() / (r.norm() + 1e-12))
    fe = {k: v for k, v in l2e.items() if k.startswith("encoder.pre_encoder")}
    rest = {k: v for k, v in l2e.items() if not k.startswith("encoder.pre_encoder")}
    res["emu_worst_l2"] = max(rest.items(), key=lambda kv: kv[1])
    res["emu_median_l2"] = float(np.median(list(rest.values())))
    res["emu_worst_l2_frontend"] = max(fe.items(), key=lambda kv: kv[1])
    # how far the emulation itself is from the fp32 fixture (what bf16 storage alone does to these gradients)
    res["emu_vs_fp32_worst_l2"] = max(((n, float((sde[n].grad - grads[n]).norm() / (grads[n].norm() + 1e-12))) for n in rest), key=lambda kv: kv[1])
    return res


def check_speech_lstm_beam_search():
    """Beam search with the attention LSTM decoder (incremental state: (h, c) per layer, input feed, per-sentence encoder keys /
    values addressed through kv_col) vs the reference SequenceGenerator's beams on the same weights."""
    from espresso_amd.sequence_generator import SequenceGenerator

    g = np.load(os.path.join(GOLD, "ref_speech_lstm_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = build_tiny_speech_lstm().to(DEV)
    model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    model.eval()
    d = _TaskAR(40).target_dictionary
    sample = {"net_input": {"src_tokens": torch.from_numpy(g["feats"]).to(DEV), "src_lengths": torch.from_numpy(g["lengths"]).to(DEV)}}
    gen = SequenceGenerator([model], d, beam_size=3, max_len_a=0.0, max_len_b=10)
    hyps = gen.generate([model], sample)
    res = {"in_beam": [], "score_abs": 0.0}
    for b, hl in enumerate(hyps):
        refset, hi = {}, 0
        while f"beam::b3::{b}::{hi}::tokens" in g.files:
            refset[tuple(g[f"beam::b3::{b}::{hi}::tokens"].tolist())] = float(g[f"beam::b3::{b}::{hi}::score"])
            hi += 1
        res["in_beam"].append(sum(tuple(h["tokens"].tolist()) in refset for h in hl) / len(hl))
        for h in hl:
            k = tuple(h["tokens"].tolist())
            if k in refset:
                res["score_abs"] = max(res["score_abs"], abs(float(h["score"]) - refset[k]))
    # force-decode the reference's best hypotheses through the incremental path
    toks = torch.stack([torch.from_numpy(g[f"beam::b3::{b}::0::tokens"]) for b in range(3)]).to(DEV)
    ref_pos = torch.stack([torch.from_numpy(g[f"beam::b3::{b}::0::pos"]) for b in range(3)])
    with torch.no_grad():
        enc_out = model.forward_encoder(sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"])
        st = model.decoder.init_incremental(enc_out, 3, 1)
        cur = torch.full((3, 1), d.eos(), dtype=torch.long, device=DEV)
        got = []
        for step in range(toks.shape[1]):
            lp = model.decoder.step(st, cur, step, None if step == 0 else torch.arange(3, device=DEV))
            got.append(lp.gather(-1, toks[:, step:step + 1]).squeeze(-1).cpu())
            cur = torch.cat([cur, toks[:, step:step + 1]], 1)
    got = torch.stack(got, 1)
    res["forced_decode_pos_score_abs"] = float((got[:, :-1] - ref_pos[:, :-1]).abs().max())
    return res


def check_transducer_beam_search():
    """HIP transducer beam search (modified adaptive expansion search) vs the reference's TransducerBeamSearchDecoder on the
    same weights: n-best token sequences and length-normalised scores for three option sets."""
    from espresso_amd.tools.transducer_beam_search_decoder import TransducerBeamSearchDecoder

    g = np.load(os.path.join(GOLD, "ref_conformer_transducer_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = build_tiny_transducer().to(DEV)
    model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    model.eval()
    d = _Task(40).target_dictionary
    sample = {"net_input": {"src_tokens": torch.from_numpy(g["feats"]).to(DEV), "src_lengths": torch.from_numpy(g["lengths"]).to(DEV)}}
    res = {}
    for tag, kw in (("b3", dict(beam_size=3, max_num_expansions_per_step=2, prefix_alpha=1)),
                    ("b4_beta1_g2", dict(beam_size=4, max_num_expansions_per_step=2, expansion_beta=1, expansion_gamma=2.0, prefix_alpha=2)),
                    ("b2_nonorm", dict(beam_size=2, max_num_expansions_per_step=1, normalize_scores=False))):
        dec = TransducerBeamSearchDecoder([model], d, **kw)
        toks_l, scores_l, _ = dec._generate(sample)
        best_equal, nbest_in_ref, score_abs = [], [], 0.0
        for b in range(len(toks_l)):
            rt, rs = g[f"out::beam_{tag}_{b}_tokens"], g[f"out::beam_{tag}_{b}_scores"]
            strip = lambda row: tuple(int(t) for t in row if int(t) != d.pad())
            refset = {strip(rt[j]): float(rs[j]) for j in range(rt.shape[0])}
            mine = [strip(toks_l[b][j].tolist()) for j in range(toks_l[b].shape[0])]
            # identical 1-best — or, where the reference's own best two hypotheses are closer than the bf16 noise of a score
            # (b2_nonorm, utterance 0: -5.393725 vs -5.393900 on this random-init model), one of those tied hypotheses
            tied = mine[0] in refset and abs(refset[mine[0]] - float(rs[0])) < 2e-3
            best_equal.append(mine[0] == strip(rt[0]) or tied)
            nbest_in_ref.append(sum(m in refset for m in mine) / len(mine))
            for j, m in enumerate(mine):
                if m in refset:
                    score_abs = max(score_abs, abs(float(scores_l[b][j]) - refset[m]))
        res[tag] = {"best_equal": best_equal, "nbest_in_ref": nbest_in_ref, "score_abs": score_abs}
        if tag == "b3":
            # the searches of a batch run as coroutines whose joint / predictor requests are served together: the answers must be
            # what each utterance's search gets when it runs alone (on the same encoder output: the reference's model itself is
            # not invariant to batch padding — the sub-sampler's padded frames feed the next convolution's receptive field)
            same_tokens, sdiff = [], 0.0
            for b in range(len(toks_l)):
                t1, s1, _ = dec._generate(sample, only=[b])  # same encoder pass, this utterance's search alone on the device
                same_tokens.append(torch.equal(t1[b], toks_l[b]))
                sdiff = max(sdiff, float((s1[b] - scores_l[b]).abs().max()))
            res["batched_equals_single"] = {"tokens": same_tokens, "score_abs": sdiff}
    return res


def check_speech_recognize_loop():
    """End-to-end recognition loop of the CLI on raw audio: GPU front-end -> tiny enc-dec model -> beam search -> H-/T- lines,
    WER bookkeeping and the summary line."""
    import io

    from espresso_amd import speech_recognize as sr
    from espresso_amd.sequence_generator import SequenceGenerator
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask

    torch.manual_seed(0)
    d = _TaskAR(40).target_dictionary
    task = SpeechRecognitionEspressoTask.setup_task(SpeechRecognitionEspressoConfig(autoregressive=True), tgt_dict=d)
    model = build_tiny_encdec().to(DEV).eval()
    task.build_frontend(DEV)
    rng = np.random.default_rng(0)
    utts = [f"utt{i}" for i in range(5)]
    waves = [(rng.standard_normal(int(16000 * s)) * 3000).astype(np.float32) for s in (1.2, 0.7, 2.0, 0.9, 1.5)]
    refs = {u: "t1 t2 t3" for u in utts}
    gen = SequenceGenerator([model], d, beam_size=3, max_len_a=0.0, max_len_b=6)
    batches = sr.make_batches(utts, [len(w) for w in waves], max_tokens=400, max_sentences=3)
    buf = io.StringIO()
    scorer, stats = sr.recognize(task, model, gen, (sr.collate(b, utts, waves, DEV) for b in batches), d, refs, out=buf)
    text = buf.getvalue()
    return {"n_batches": len(batches), "H_lines": text.count("\nH-") + text.startswith("H-"), "T_lines": text.count("T-utt"),
            "summary": "Recognized 5 utterances" in text, "wer_reported": "WER" in text, "sentences": stats["sentences"],
            "wer_finite": bool(np.isfinite(scorer.wer()[0]))}


def check_scheduled_sampling():
    """speech_lstm decoder with scheduled sampling: p = 1 reproduces teacher forcing exactly (same kernels, same order); p = 0
    feeds the model's own argmax from step 1 on (checked against an explicit greedy roll-out), and gradients flow."""
    from espresso_amd.models.speech_lstm import ScheduledSamplingRateScheduler

    g = np.load(os.path.join(GOLD, "ref_speech_lstm_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model = build_tiny_speech_lstm().to(DEV)
    model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    feats, lengths, prev = (torch.from_numpy(g[k]).to(DEV) for k in ("feats", "lengths", "prev"))
    model.train()
    lo_tf, _ = model(feats, lengths, prev)
    sch = ScheduledSamplingRateScheduler([0.999999], 1)
    model.decoder.scheduled_sampling_rate_scheduler = sch
    torch.manual_seed(0)
    lo_p1, _ = model(feats, lengths, prev, epoch=1)  # rand < 0.999999 always: gold tokens
    sch.scheduled_sampling_probs = [0.0]
    lo_p0, _ = model(feats, lengths, prev, epoch=1)
    # greedy roll-out with the same weights: token j = argmax of step j-1
    roll = prev.clone()
    for j in range(1, prev.shape[1]):
        sch.scheduled_sampling_probs = [1.0]
        cur, _ = model(feats, lengths, roll)
        roll[:, j] = cur[:, j - 1].argmax(-1)
    sch.scheduled_sampling_probs = [1.0]
    lo_roll, _ = model(feats, lengths, roll)
    sch.scheduled_sampling_probs = [0.0]
    for p_ in model.parameters():
        p_.grad = None
    lo, _ = model(feats, lengths, prev, epoch=1)
    lo.float().square().mean().backward()
    torch.cuda.synchronize()
    finite = all(bool(torch.isfinite(p_.grad).all()) for p_ in model.parameters() if p_.grad is not None)
    has_embed_grad = float(model.decoder.embed_tokens.weight.grad.abs().sum()) > 0
    return {"p1_vs_teacher_forcing": float((lo_p1.float() - lo_tf.float()).abs().max()),
            "p0_vs_rollout": float((lo_p0.float() - lo_roll.float()).abs().max()), "finite": finite, "embed_grad": has_embed_grad}


def check_task_pipeline(tmp_dir):
    """The drop-in boundary end to end on raw audio: data json -> AsrDataset/collater (pinned waveforms) -> task.to_device ->
    GPU front-end in prepare_sample -> `cross_entropy_v2` train step -> valid step with the greedy WER decoder ->
    reduce_metrics; `build_generator` picks the decoder class from the criterion like the reference task."""
    import json

    import torch.nn.functional as TF

    from espresso_amd.data import audio_utils
    from espresso_amd.sequence_generator import SequenceGenerator
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask
    from espresso_amd.tools.ctc_decoder import CTCDecoder
    from espresso_amd.tools.transducer_beam_search_decoder import TransducerBeamSearchDecoder
    from espresso_amd.tools.transducer_greedy_decoder import TransducerGreedyDecoder

    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    d = _TaskAR(40).target_dictionary
    utts = {}
    for i in range(7):
        u = f"utt{i}"
        path = os.path.join(tmp_dir, u + ".wav")
        audio_utils.write_wav(path, rng.standard_normal(int(16000 * rng.uniform(0.6, 1.6))) * 3000)
        utts[u] = {"wave": path, "text": " ".join(f"t{int(k)}" for k in rng.integers(0, 36, size=int(rng.integers(2, 7))))}
    for split in ("train", "valid"):
        with open(os.path.join(tmp_dir, split + ".json"), "w") as f:
            json.dump(utts, f)
    cfg = SpeechRecognitionEspressoConfig(data=tmp_dir, autoregressive=True, criterion_name="cross_entropy_v2")
    task = SpeechRecognitionEspressoTask.setup_task(cfg, tgt_dict=d)
    task.build_frontend(DEV)
    model = build_tiny_encdec().to(DEV)
    crit = task.build_criterion(sentence_avg=False)
    ds = task.load_dataset("train")
    batches = task.get_batches(ds, max_tokens=400, max_sentences=4, seed=1, epoch=1)
    sample = task.to_device(ds.collater([ds[int(i)] for i in batches[0]]), DEV)
    loss, sample_size, log = task.train_step(sample, model, crit)
    # independent value: torch cross entropy on the model's own logits (same mode: the sub-sampler's BatchNorm uses batch statistics)
    with torch.no_grad():
        s2 = task.prepare_sample(sample, train=False)
        logits, _ = model(**s2["net_input"])
        ref = TF.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), s2["target"].reshape(-1), ignore_index=d.pad(), reduction="sum")
        lt, _, _ = crit(model, s2)
    grads_finite = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    task.build_validation_decoder(model)
    vds = task.load_dataset("valid")
    logs = []
    for b in task.get_batches(vds, max_tokens=400, max_sentences=4, shuffle=False):
        _, _, lg = task.valid_step(task.to_device(vds.collater([vds[int(i)] for i in b]), DEV), model, crit)
        logs.append(lg)
    red = task.reduce_metrics(logs)
    n_words = sum(len(v["text"].split()) for v in utts.values())
    gens = {}
    for name, beam, want in (("cross_entropy_v2", 3, SequenceGenerator), ("ctc_loss", 1, CTCDecoder)):
        task.cfg.criterion_name = name
        gens[name] = isinstance(task.build_generator([model], type("A", (), {"beam": beam})()), want)
    tmodel = build_tiny_transducer().to(DEV)
    task.cfg.criterion_name, task.blank_symbol = "transducer_loss", d.bos_word
    gens["transducer_greedy"] = isinstance(task.build_generator([tmodel], type("A", (), {"beam": 1})()), TransducerGreedyDecoder)
    gens["transducer_beam"] = isinstance(task.build_generator([tmodel], type("A", (), {"beam": 4})()), TransducerBeamSearchDecoder)
    return {"loss_vs_torch": abs(float(lt) - float(ref)) / max(1.0, abs(float(ref))), "sample_size": sample_size,
            "ntokens": sample["ntokens"], "grads_finite": grads_finite, "word_count": sum(l["word_count"] for l in logs),
            "char_count": sum(l["char_count"] for l in logs), "n_utts": len(utts),
            "n_words": n_words, "wer": red.get("wer"), "loss_metric": red.get("loss"), "gens": gens,
            "pinned": bool(ds.collater([ds[0]])["wav"].is_pinned())}


def check_global_cmvn_stats(tmp_dir):
    """GPU CMVN statistics tool vs the reference's recipe restated on the host (oracle fbank per utterance, then the pairwise
    sum / unnormalised-variance merge of espresso/tools/compute_global_cmvn_stats.py:96-120)."""
    import io

    from espresso_amd.data import audio_utils
    from espresso_amd.tools import compute_global_cmvn_stats as cg, wav2num_frames as w2n
    from oracle import fbank_ref

    rng = np.random.default_rng(3)
    lines, waves = [], []
    for i, sec in enumerate((1.3, 0.4, 2.2, 0.02, 0.9, 1.7)):
        x = np.round(rng.standard_normal(int(16000 * sec)) * (500 + 700 * i)).astype(np.float32)
        path = os.path.join(tmp_dir, f"c{i}.wav")
        audio_utils.write_wav(path, x)
        lines.append(f"c{i} {path}")
        waves.append(np.clip(x, -32768, 32767))
    scp = os.path.join(tmp_dir, "wav.scp")
    open(scp, "w").write("\n".join(lines) + "\n")
    cg.main(cg.get_parser().parse_args([scp, tmp_dir, "--batch-seconds", "2.5", "--device", DEV]))
    got = np.load(os.path.join(tmp_dir, "gcmvn.npz"))
    total_sum, total_var, total_frames = np.zeros(80), np.zeros(80), 0
    frames = []
    for w in waves:
        feat = fbank_ref.fbank(w).astype(np.float64)
        frames.append(feat.shape[0])
        if feat.shape[0] == 0:
            continue
        cur_sum, cur_frames, cur_var = feat.sum(0), feat.shape[0], np.var(feat, axis=0) * feat.shape[0]
        if total_frames > 0:
            r = total_frames / cur_frames
            total_var = total_var + cur_var + r / (total_frames + cur_frames) * (total_sum / r - cur_sum) ** 2
        else:
            total_var = cur_var
        total_sum, total_frames = total_sum + cur_sum, total_frames + cur_frames
    mean, std = total_sum / total_frames, np.sqrt(total_var / total_frames)
    buf = io.StringIO()
    w2n.main(w2n.get_parser().parse_args([scp]), out=buf)
    nf = [int(l.split()[1]) for l in buf.getvalue().strip().split("\n")]
    return {"mean_abs": float(np.abs(got["mean"] - mean).max()), "std_abs": float(np.abs(got["std"] - std).max()),
            "dtype64": got["mean"].dtype == np.float64, "num_frames_equal": nf == frames}


def check_label_smoothing_known_answers():
    """The reference's own known-answer tests for the CE criterions (tests/test_label_smoothing.py:46-119) through the HIP
    criterions: fixed model probabilities, a batch with padding; nll bookkeeping, padding additivity, reduction, zero epsilon,
    plus the closed-form values."""
    from espresso_amd.criterions.cross_entropy_v2 import CrossEntropyV2Criterion
    from espresso_amd.criterions.label_smoothed_cross_entropy_v2 import LabelSmoothedCrossEntropyV2Criterion
    from espresso_amd.data.asr_dictionary import AsrDictionary

    d = AsrDictionary.from_symbols(["w1", "w2", "w3"], enable_bos=True, add_space=False)  # <s> <pad> </s> <unk> w1 w2 w3
    assert (len(d), d.pad(), d.eos(), d.unk()) == (7, 1, 2, 3)
    pad, eos, w1 = 1, 2, 4
    probs = torch.tensor([[0.05, 0.05, 0.1, 0.05, 0.3, 0.4, 0.05],
                          [0.05, 0.10, 0.2, 0.05, 0.2, 0.3, 0.10],
                          [0.05, 0.15, 0.3, 0.05, 0.1, 0.2, 0.15]])
    task = type("T", (), {"target_dictionary": d})()

    class FixedModel:
        training = False

        def __call__(self, prev_output_tokens=None, **kw):
            B = prev_output_tokens.shape[0]
            return probs.log().unsqueeze(0).expand(B, 3, 7).contiguous().to(DEV), None

    def sample_of(targets):
        L = max(len(t) for t in targets)
        tgt = torch.full((len(targets), L), pad, dtype=torch.long)
        for i, t in enumerate(targets):
            tgt[i, : len(t)] = torch.tensor(t)
        # the model always emits 3 positions; shorter targets are padded to 3 like the reference's collater does
        tgt3 = torch.full((len(targets), 3), pad, dtype=torch.long)
        tgt3[:, :L] = tgt
        return {"net_input": {"prev_output_tokens": tgt3.to(DEV)}, "target": tgt3.to(DEV), "ntokens": int((tgt3 != pad).sum()),
                "id": torch.arange(len(targets))}

    model = FixedModel()
    both = sample_of([[w1, eos], [w1, w1, eos]])
    ce = CrossEntropyV2Criterion(task, sentence_avg=False)
    ls = LabelSmoothedCrossEntropyV2Criterion(task, sentence_avg=False, label_smoothing=0.1)
    ls0 = LabelSmoothedCrossEntropyV2Criterion(task, sentence_avg=False, label_smoothing=0.0)
    nll_loss, nll_ss, nll_log = ce(model, both)
    sm_loss, sm_ss, sm_log = ls(model, both)
    z_loss, _, _ = ls0(model, both)
    l1, _, _ = ls(model, sample_of([[w1, eos]]))
    l2, _, _ = ls(model, sample_of([[w1, w1, eos]]))
    lp = probs.log().double()
    nll_true = -(lp[0, w1] + lp[1, eos]) - (lp[0, w1] + lp[1, w1] + lp[2, eos])
    eps_i = 0.1 / (7 - 1)
    smooth_true = -(lp[0].sum() + lp[1].sum()) - (lp[0].sum() + lp[1].sum() + lp[2].sum())
    sm_true = (1 - 0.1 - eps_i) * nll_true + eps_i * smooth_true
    return {"nll_vs_logging": abs(float(nll_loss) - float(nll_log["loss"])), "nll_vs_smooth_nll": abs(float(nll_loss) - float(sm_log["nll_loss"])),
            "padding_additivity": abs(float(sm_loss) - float(l1) - float(l2)), "zero_eps": abs(float(nll_loss) - float(z_loss)),
            "nll_closed_form": abs(float(nll_loss) - float(nll_true)), "smooth_closed_form": abs(float(sm_loss) - float(sm_true)),
            "sample_sizes": (nll_ss, sm_ss, both["ntokens"])}


# ------------------------------------------------------------------ full-size (BASELINE config 3) property checks
def check_fullsize_ctc(B=24, T=325, V=5004, seed=0):
    """CTC at the headline sizes through size-independent properties: the loss of a batch is the sum of the losses of its halves
    (bit-level independence of utterances), d loss / d logits sums to zero over the vocabulary on valid frames (softmax minus
    posterior) and is exactly zero on padded frames, losses are finite and positive."""
    from espresso_amd import functional as F

    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(B, T, V, generator=g) * 1.5).to(DEV)
    in_len = torch.randint(T // 2, T + 1, (B,), generator=g).to(torch.int32)
    in_len[0] = T
    Lmax = 60
    tgt_len = torch.randint(5, Lmax + 1, (B,), generator=g).to(torch.int32)
    tgt = torch.randint(1, V, (B, Lmax), generator=g).to(torch.int32)

    def run(idx):
        x = logits[idx].reshape(len(idx) * T, V).clone().requires_grad_(True)
        nll, _ = F.ctc_loss(x, tgt[idx].to(DEV), in_len[idx].to(DEV), tgt_len[idx].to(DEV), len(idx), T, 0)
        nll.sum().backward()
        return nll.detach(), x.grad.view(len(idx), T, V)

    allb = list(range(B))
    nll, grad = run(allb)
    n1, g1 = run(allb[: B // 2])
    n2, g2 = run(allb[B // 2:])
    torch.cuda.synchronize()
    valid = (torch.arange(T)[None, :] < in_len[:, None]).to(DEV)
    rowsum = grad.sum(-1)
    return {"finite_positive": bool(torch.isfinite(nll).all() and (nll > 0).all()),
            "halves_nll_abs": float((nll - torch.cat([n1, n2])).abs().max()), "halves_grad_abs": float((grad - torch.cat([g1, g2])).abs().max()),
            "grad_rowsum_abs": float(rowsum[valid].abs().max()), "pad_grad_abs": float(grad[~valid].abs().max())}


def check_fullsize_frontend(seed=0):
    """24 utterances up to 35 s through the fused front-end: every utterance's features are identical to those of the same
    utterance processed alone (batch independence incl. padding), frames beyond an utterance are exactly zero, and global
    CMVN is the affine map it should be."""
    from espresso_amd.data.feature_transforms import GlobalCMVN
    from espresso_amd.data.gpu_frontend import GpuFbankFrontend

    rng = np.random.default_rng(seed)
    secs = np.concatenate([[35.0, 0.03, 1.0], rng.uniform(2.0, 30.0, size=21)])
    wavs = [(rng.standard_normal(int(16000 * s)) * 2500).astype(np.float32) for s in secs]
    ns = [len(w) for w in wavs]
    off = np.zeros(len(ns) + 1, dtype=np.int64)
    off[1:] = np.cumsum(ns)
    wav, offs = torch.from_numpy(np.concatenate(wavs)).to(DEV), torch.from_numpy(off).to(DEV)
    fe = GpuFbankFrontend(DEV)
    feat, lens, frames = fe(wav, offs, ns, train=False)
    worst, pad = 0.0, 0.0
    for b in (0, 1, 2, 7, 23):
        one, l1, _ = fe(torch.from_numpy(wavs[b]).to(DEV), torch.tensor([0, ns[b]], device=DEV), [ns[b]], train=False)
        n = int(l1[0])
        assert n == frames[b]
        if n:
            worst = max(worst, float((feat[b, :n] - one[0, :n]).abs().max()))
        pad = max(pad, float(feat[b, n:].abs().max()) if n < feat.shape[1] else 0.0)
    mean, std = rng.standard_normal(80) * 3 + 5, rng.random(80) + 1.5
    fe2 = GpuFbankFrontend(DEV, cmvn=GlobalCMVN(mean=mean, std=std))
    feat2, _, _ = fe2(wav, offs, ns, train=False)
    m, s_ = torch.tensor(mean, dtype=torch.float32, device=DEV), torch.tensor(std, dtype=torch.float32, device=DEV)
    aff = 0.0
    for b in (0, 7):
        n = frames[b]
        aff = max(aff, float((feat2[b, :n] - (feat[b, :n] - m) / s_).abs().max()))
    return {"batch_independence_abs": worst, "padding_abs": pad, "cmvn_affine_abs": aff, "frames0": frames[0], "frames1": frames[1]}


def check_fullsize_attention(B=7, H=8, T=875, seed=0):
    """Fused rel-pos attention at the longest utterance of the headline workload: with V = 1 every valid query row returns
    exactly 1 (probabilities sum to one whatever the scores), padded keys get no weight, and the result does not depend on
    what the padded keys / values contain."""
    from espresso_amd import kernels as Kk

    g = torch.Generator().manual_seed(seed)
    C = H * 64
    klen = torch.tensor([T, T - 1, 700, 512, 333, 65, 1][:B], dtype=torch.int32, device=DEV)
    qu = (torch.randn(B * T, C, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    qv = (torch.randn(B * T, C, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    kv = torch.randn(B * T, 2 * C, generator=g).to(torch.bfloat16).to(DEV)
    pp = (torch.randn(2 * T - 1, C, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    ones = kv.clone()
    ones[:, C:] = 1.0
    out1, _ = Kk.flash_attention_fwd(qu, qv, ones[:, :C], ones[:, C:], pp, klen, H, B, T, T, C, 2 * C, C)
    out2, _ = Kk.flash_attention_fwd(qu, qv, kv[:, :C], kv[:, C:], pp, klen, H, B, T, T, C, 2 * C, C)
    poisoned = kv.clone().view(B, T, 2 * C)
    for b in range(B):
        poisoned[b, int(klen[b]):] = 77.0
    out3, _ = Kk.flash_attention_fwd(qu, qv, poisoned.view(B * T, 2 * C)[:, :C], poisoned.view(B * T, 2 * C)[:, C:], pp, klen, H, B, T, T, C, 2 * C, C)
    torch.cuda.synchronize()
    return {"ones_abs": float((out1.float() - 1.0).abs().max()), "pad_independence_abs": float((out2.float() - out3.float()).abs().max()),
            "finite": bool(torch.isfinite(out2.float()).all())}


def check_fullsize_encoder_batch_independence(seed=0):
    """The full Conformer-12 + CTC model (config 3, random weights, eval mode): the longest utterance's output frames are the same
    whether it is run alone or at the head of a padded batch of 12 utterances of very different lengths (batch items never mix:
    attention key lengths, BatchNorm running statistics, LayerNorm rows).  Shorter utterances are NOT compared: like the
    reference (fairseq ConvolutionModule gets no padding mask, conformer_layer.py:134-146) the depthwise convolution reads the
    padded frames behind an utterance, so their outputs legitimately depend on the padding length."""
    import bench

    torch.manual_seed(seed)
    task, model, _, _ = bench.build(torch.device(DEV), seed=1)
    model.eval()
    g = torch.Generator().manual_seed(seed)
    lens = [1500, 1333, 1200, 997, 801, 640, 512, 400, 256, 130, 64, 17]
    B, Tm = len(lens), max(lens)
    feats = torch.zeros(B, Tm, 80)
    for b, n in enumerate(lens):
        feats[b, :n] = torch.randn(n, 80, generator=g)
    feats, ln = feats.to(DEV), torch.tensor(lens, device=DEV)
    worst, checked = 0.0, 0
    with torch.no_grad():
        out = model(feats, ln)
        lo, ol = out["encoder_out"][0], out["src_lengths"][0]  # T' x B x V
        for b in (0,):
            one = model(feats[b:b + 1, : lens[b]].contiguous(), ln[b:b + 1])
            n = int(one["src_lengths"][0][0])
            assert n == int(ol[b])
            worst = max(worst, float((lo[:n, b].float() - one["encoder_out"][0][:n, 0].float()).abs().max()))
            checked += n
    scale = float(lo.float().abs().max())
    return {"abs": worst, "scale": scale, "frames_checked": checked, "finite": bool(torch.isfinite(lo.float()).all())}


def check_fullsize_layer_vs_oracle(layer_type="conformer", seed=0, layers=1, lens=(400, 333, 250, 120), tl=(9, 7, 5, 3), V=5004, dropout=0.0):
    """Config-3 LAYER dimensions (embed 512, 8 heads of 64, FFN 2048, depthwise kernel 31, conv front-end 64-64-128-128) in a
    one-layer model with random weights, HIP vs the pinned oracle (oracle/torch_ref.py) on the same weights and inputs: eval
    logits, train-mode CTC loss and every gradient, against the fp32 restatement (north_star's bf16 tolerance) and against its
    bf16-emulating mode (tight).  `layers=12`: the model bench.py times.  `V` = the recipe's 5004 (5000 unigram pieces + specials).
    `dropout` > 0: the train-mode pass runs with dropout = attention_dropout = activation_dropout = that value (the recipe: 0.1) and
    the oracle applies the reference's dropout sites with the HIP path's keep decisions (check_encoder_dropout_vs_oracle)."""
    from espresso_amd import _lib
    from espresso_amd import functional as F
    from oracle import dropout_ref as D
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerConfig
    from espresso_amd.models.transformer.speech_transformer_encoder_model import SpeechTransformerEncoderModel
    from oracle import torch_ref

    torch.manual_seed(seed)
    H = 8
    cfg = SpeechTransformerConfig()
    e = cfg.encoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = 512, 2048, layers, H
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, layer_type
    e.conv_channels = "[64, 64, 128, 128]"
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = dropout
    cfg.layernorm_embedding = True
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    model = SpeechTransformerEncoderModel.build_model(cfg, _Task(V))
    with torch.no_grad():  # biases / norm parameters / positional biases away from their trivial initial values
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd = {k[len("encoder."):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("encoder.")}
    model = model.to(DEV)
    g = torch.Generator().manual_seed(seed + 1)
    lens, tl = list(lens), list(tl)
    feats = torch.zeros(len(lens), max(lens), 80)
    for b, n in enumerate(lens):
        feats[b, :n] = torch.randn(n, 80, generator=g)
    lengths = torch.tensor(lens)
    tgt = torch.full((len(lens), max(tl)), 1, dtype=torch.long)
    for b, n in enumerate(tl):
        tgt[b, :n] = torch.randint(4, V, (n,), generator=g)
    res = {}
    # ---- HIP ----
    model.eval()
    with torch.no_grad():
        out = model(feats.to(DEV), lengths.to(DEV))
    hip_eval = out["encoder_out"][0].float().cpu()
    model.train()
    F.set_dropout_seed(seed + 17)
    with F.trace_dropout_seeds() as tr:
        out = model(feats.to(DEV), lengths.to(DEV))
    B, Tp = out["encoder_padding_mask"][0].shape
    nll, _ = F.ctc_loss(out["_logits_bt"][0], tgt.to(DEV).to(torch.int32).contiguous(), out["src_lengths"][0].to(torch.int32),
                        torch.tensor(tl, dtype=torch.int32, device=DEV), B, Tp, blank=0)
    loss = nll.sum()
    loss.backward()
    torch.cuda.synchronize()
    hip_grads = {n: p.grad.float().cpu() for n, p in model.encoder.named_parameters() if p.grad is not None}
    res["hip_loss"] = float(loss.detach())
    res["n_seed_draws"] = len(tr.entries)
    # ---- oracle: fp32 restatement and bf16 emulation ----
    for tag, emu in (("fp32", False), ("emu", True)):
        sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "version" else v.clone())
               for k, v in sd.items()}
        with torch_ref.bf16_emulation(emu, flash=True):
            with torch.no_grad():
                lo, ol = torch_ref.encoder(feats, lengths, sdo, H=H, layer_type=layer_type, training=False)
            plan = D.MaskPlan(tr.entries, _lib.lib().ea_layer_dropout_seed)
            with torch_ref.dropout_masks(plan if dropout > 0 else None):
                lt, ol = torch_ref.encoder(feats, lengths, sdo, H=H, layer_type=layer_type, training=True)
            plan.done()
            res["n_site_masks"] = len(plan.queue)
            oloss = torch_ref.ctc_loss_sum(lt, tgt, ol, torch.tensor(tl))
            oloss.backward()
        res[f"{tag}_loss"] = float(oloss.detach())
        res[f"eval_logits_vs_{tag}"] = float((hip_eval - lo).abs().max())
        res["logit_scale"] = float(lo.abs().max())
        errs = []
        for n, gh in hip_grads.items():
            if (n.startswith("pre_encoder.convolutions.") and n.endswith(".bias")) or n.endswith("self_attn.k_proj.bias"):
                continue  # true gradient exactly zero (see check_encoder_vs_reference)
            go = sdo[n].grad
            errs.append((float((gh - go).abs().max() / (float(go.abs().max()) + 1e-12)), n))
        errs.sort(reverse=True)
        res[f"worst_grad_vs_{tag}"] = (errs[0][1], errs[0][0])
        res[f"median_grad_vs_{tag}"] = errs[len(errs) // 2][0]
        res["n_grads"] = len(errs)
    return res


def check_scheduled_sampling_transformer():
    """Attention enc-dec (Transformer decoder) with scheduled sampling: p -> 1 reproduces teacher forcing; p = 0 feeds the
    model's own arg-max from step 1 on — the fed tokens equal an explicit greedy roll-out built from full teacher-forced
    passes, logits equal the teacher-forced logits on that sequence, gradients reach the embedding."""
    torch.manual_seed(0)
    g = np.load(os.path.join(GOLD, "ref_transformer_encdec_tiny.npz"))
    model = build_tiny_encdec().to(DEV)
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
    feats, lengths, prev = (torch.from_numpy(g[k]).to(DEV) for k in ("feats", "lengths", "prev"))
    model.train()
    sch = model.decoder.scheduled_sampling_rate_scheduler
    lo_tf, _ = model(feats, lengths, prev)
    sch.scheduled_sampling_probs = [0.999999]
    lo_p1, _ = model(feats, lengths, prev, epoch=1)
    # explicit greedy roll-out with full passes (the sub-sampler's BatchNorm is in training mode in both paths)
    sch.scheduled_sampling_probs = [1.0]
    roll = prev.clone()
    with torch.no_grad():
        for j in range(1, prev.shape[1]):
            cur, _ = model(feats, lengths, roll)
            roll[:, j] = cur[:, j - 1].argmax(-1)
        lo_roll, _ = model(feats, lengths, roll)
    sch.scheduled_sampling_probs = [0.0]
    enc = model.encoder(feats, lengths)
    fed = model.decoder._scheduled_sampling_tokens(prev, enc, 0.0)
    for p_ in model.parameters():
        p_.grad = None
    lo_p0, _ = model(feats, lengths, prev, epoch=1)
    lo_p0.float().square().mean().backward()
    torch.cuda.synchronize()
    finite = all(bool(torch.isfinite(p_.grad).all()) for p_ in model.parameters() if p_.grad is not None)
    sch.scheduled_sampling_probs = [1.0]
    return {"p1_vs_teacher_forcing": float((lo_p1.float() - lo_tf.float()).abs().max()),
            "fed_tokens_equal_rollout": float((fed == roll).float().mean()),
            "p0_vs_rollout_logits": float((lo_p0.float() - lo_roll.float()).abs().max()), "finite": finite,
            "embed_grad": float(model.decoder.embed_tokens.weight.grad.abs().sum()) > 0, "n_sampled": int((fed != prev).sum())}


def check_native_transformer_layer(learned=False, seed=0, C=64, heads=4, T=37):
    """csrc/engine.hip Transformer layer (one call per layer and direction) vs the Python composition of the individual
    kernels: outputs and every gradient (incl. the learned relative table's), same weights, same input, no dropout."""
    from espresso_amd.modules.transformer_layer import TransformerWithRelativePositionalEmbeddingEncoderLayer as Layer

    torch.manual_seed(seed)
    model = build_tiny_model("transformer", embed_dim=C, heads=heads, ffn=2 * C, learned_pos=learned).to(DEV)
    layer = model.encoder.layers[0]
    if not learned:
        with torch.no_grad():
            layer.self_attn.pos_bias_u.normal_(0, 0.1)
            layer.self_attn.pos_bias_v.normal_(0, 0.1)
    B = 3
    x0 = bf(torch.randn(B * T, C)).to(DEV)
    key_len = torch.tensor([T, T - 7, max(1, T // 3)], dtype=torch.int32, device=DEV)
    model.train()
    outs, grads = [], []
    for native in (False, True):
        Layer.use_native_runtime = native
        for p in model.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y = layer(x, B, T, key_len=key_len)
        (y.float() * torch.linspace(-1, 1, C, device=DEV)).sum().backward()
        torch.cuda.synchronize()
        outs.append(y.detach().float().cpu())
        g = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        g["__x"] = x.grad.float().cpu()
        grads.append(g)
    Layer.use_native_runtime = True
    res = {"out_abs": float((outs[0] - outs[1]).abs().max()), "same_params": sorted(grads[0]) == sorted(grads[1]), "n_grads": len(grads[0])}
    worst = ("", 0.0)
    for n in grads[0]:
        if n.endswith("k_proj.bias"):
            continue
        a, b = grads[0][n], grads[1][n]
        e = float((a - b).abs().max() / (a.abs().max() + 1e-6))
        if e > worst[1]:
            worst = (n, e)
    res["worst_grad"] = worst
    res["has_table_grad"] = any("positional" in n or "relative" in n or "embed_positions" in n for n in grads[1]) if learned else True
    return res


def check_native_decoder_layer(seed=0, C=256, heads=4, U=23, S=61):
    """csrc/engine.hip Transformer decoder layer (one call per layer and direction; needs the flat parameter layout) vs the Python
    composition of the individual kernels: output, gradients w.r.t. the layer input, the encoder output and every parameter."""
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerConfig
    from espresso_amd.modules.transformer_decoder_layer import TransformerDecoderLayer as Layer
    from espresso_amd.optim.flat import FlatParams

    torch.manual_seed(seed)
    cfg = SpeechTransformerConfig()
    cfg.encoder.embed_dim = cfg.decoder.embed_dim = C
    cfg.decoder.ffn_embed_dim, cfg.decoder.attention_heads, cfg.decoder.normalize_before = 2 * C, heads, True
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = 0.0
    layer = Layer(cfg).to(DEV)
    with torch.no_grad():
        for p in layer.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1) if p.abs().sum() == 0 else None
    flat = FlatParams(layer, torch.device(DEV))
    B = 3
    x0 = bf(torch.randn(B * U, C)).to(DEV)
    enc0 = bf(torch.randn(B * S, C)).to(DEV)
    enc_len = torch.tensor([S, S - 9, max(1, S // 3)], dtype=torch.int32, device=DEV)
    layer.train()
    res, runs = {}, []
    for native in (False, True):
        Layer.use_native_runtime = native
        flat.g32.zero_()
        x, enc = x0.clone().requires_grad_(True), enc0.clone().requires_grad_(True)
        y = layer(x, enc, enc_len, B, U, S)
        (y.float() * torch.linspace(-1, 1, C, device=DEV)).sum().backward()
        torch.cuda.synchronize()
        g = {n: p.grad.detach().float().cpu().clone() for n, p in layer.named_parameters()}
        g["__x"], g["__enc"] = x.grad.float().cpu(), enc.grad.float().cpu()
        runs.append((y.detach().float().cpu(), g))
    Layer.use_native_runtime = True
    res["native_used"] = getattr(layer, "_ea_binding", None) is not None
    res["out_abs"] = float((runs[0][0] - runs[1][0]).abs().max())
    worst = ("", 0.0)
    for n in runs[0][1]:
        if n.endswith("k_proj.bias"):
            continue  # exactly zero in exact arithmetic (softmax shift invariance)
        a, b = runs[0][1][n], runs[1][1][n]
        e = float((a - b).abs().max() / (a.abs().max() + 1e-6))
        if e > worst[1]:
            worst = (n, e)
    res["worst_grad"] = worst
    res["pad_enc_grad"] = float(runs[1][1]["__enc"].view(B, S, C)[1, S - 9:].abs().max())
    return res


def check_speech_train_cli(tmp_dir):
    """`espresso_amd.speech_train` on a recipe YAML in the reference's hydra layout: raw-audio data json + dictionary file ->
    Conformer + CTC updates with update_freq 2, epoch / best / last checkpoints, validation WER, and resume: a run stopped after 2
    updates (mid-epoch: 5 batches -> 3 updates per epoch) and restarted from checkpoint_last.pt must land where an uninterrupted
    6-update run lands."""
    import contextlib
    import io
    import json

    from espresso_amd import speech_train
    from espresso_amd.data import audio_utils

    rng = np.random.default_rng(5)
    utts = {}
    for i in range(12):
        u = f"utt{i:02d}"
        path = os.path.join(tmp_dir, u + ".wav")
        audio_utils.write_wav(path, rng.standard_normal(int(16000 * rng.uniform(0.6, 1.6))) * 3000)
        utts[u] = {"wave": path, "text": " ".join(f"t{int(k)}" for k in rng.integers(0, 36, size=int(rng.integers(2, 7))))}
    for split in ("train", "valid"):
        with open(os.path.join(tmp_dir, split + ".json"), "w") as f:
            json.dump(utts, f)
    with open(os.path.join(tmp_dir, "dict.txt"), "w") as f:
        f.write("".join(f"t{i} 1\n" for i in range(36)))
    recipe = os.path.join(tmp_dir, "recipe.yaml")
    with open(recipe, "w") as f:
        f.write("""
common: {seed: 1, log_interval: 1}
checkpoint: {save_dir: checkpoints, best_checkpoint_metric: wer, keep_last_epochs: 5}
task:
  _name: speech_recognition_espresso
  data: ???
  dict: ???
  max_source_positions: 3600
  max_target_positions: 200
  autoregressive: false
dataset: {max_tokens: 400, batch_size: 4, required_batch_size_multiple: 1, train_subset: train, valid_subset: valid, curriculum: 1}
criterion: {_name: ctc_loss, zero_infinity: true}
optimization: {max_epoch: 100, clip_norm: 2.0, sentence_avg: true, update_freq: [2], lr: [0.2]}
optimizer: {_name: adam, adam_betas: "(0.9,0.98)", adam_eps: 1e-08, weight_decay: 0.0}
lr_scheduler: {_name: noam, warmup_steps: 25, model_size: "${model.encoder.embed_dim}", final_lr: 1e-6}
model:
  _name: speech_transformer_encoder_model
  encoder:
    conv_channels: "[64, 64, 16, 16]"
    embed_dim: 128
    ffn_embed_dim: 256
    layers: 2
    attention_heads: 2
    normalize_before: true
    relative_positional_embeddings: true
    layer_type: conformer
  attention_dropout: 0.1
  activation_dropout: 0.1
  dropout: 0.1
  layernorm_embedding: true
""")

    def run(save_dir, max_update):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            tr = speech_train.main(["--config", recipe, f"task.data={tmp_dir}", f"task.dict={tmp_dir}/dict.txt",
                                    f"checkpoint.save_dir={save_dir}", f"optimization.max_update={max_update}"])
        lines = [json.loads(l) for l in buf.getvalue().splitlines() if l.startswith("{")]
        return tr, lines

    a_dir, b_dir = os.path.join(tmp_dir, "A"), os.path.join(tmp_dir, "B")
    tr_a, log_a = run(a_dir, 6)
    _, log_b1 = run(b_dir, 2)
    ck_mid = torch.load(os.path.join(b_dir, "checkpoint_last.pt"), map_location="cpu", weights_only=False)
    tr_b, log_b2 = run(b_dir, 6)
    sa = torch.load(os.path.join(a_dir, "checkpoint_last.pt"), map_location="cpu", weights_only=False)
    sb = torch.load(os.path.join(b_dir, "checkpoint_last.pt"), map_location="cpu", weights_only=False)
    diff = max(float((sa["model"][k].double() - sb["model"][k].double()).abs().max()) for k in sa["model"])
    moved = max(float((sa["model"][k].double() - ck_mid["model"][k].double()).abs().max()) for k in sa["model"])
    # train -> recognise round trip: the recognition CLI rebuilds the model from the checkpoint's own cfg (no --model-config)
    from espresso_amd import speech_recognize

    with open(os.path.join(tmp_dir, "wav.scp"), "w") as f:
        f.write("".join(f"{u} {v['wave']}\n" for u, v in utts.items()))
    with open(os.path.join(tmp_dir, "text"), "w") as f:
        f.write("".join(f"{u} {v['text']}\n" for u, v in utts.items()))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        speech_recognize.main(["--path", os.path.join(a_dir, "checkpoint_best.pt"), "--dict", os.path.join(tmp_dir, "dict.txt"),
                               "--wav-scp", os.path.join(tmp_dir, "wav.scp"), "--text", os.path.join(tmp_dir, "text"),
                               "--search", "ctc", "--max-tokens", "400", "--batch-size", "4"])
    rec = buf.getvalue().splitlines()
    la = {l["num_updates"]: l["loss"] for l in log_a if l["kind"] == "train_inner"}
    lb = {l["num_updates"]: l["loss"] for l in log_b1 + log_b2 if l["kind"] == "train_inner"}
    valid = [l for l in log_a if l["kind"] == "valid"]
    return {"files_a": sorted(os.listdir(a_dir)), "files_b": sorted(os.listdir(b_dir)), "param_diff_resumed_vs_straight": diff,
            "param_change_since_resume_point": moved, "loss_a": la, "loss_b": lb,
            "resume": [l for l in log_b2 if l["kind"] == "resume"], "mid_iterator": ck_mid["extra_state"]["train_iterator"],
            "num_updates": (tr_a.num_updates, tr_b.num_updates), "valid": valid,
            "recognize_H_lines": sum(1 for l in rec if l.startswith("H-")), "recognize_summary": [l for l in rec if "WER" in l][:2],
            "hist": sb["optimizer_history"][-1]["num_updates"], "opt_step": sb["last_optimizer_state"]["state"][0]["step"]}


def check_ddp_bucket_accounting(tmp_dir):
    """Data-parallel wrapper armed on ONE rank (RCCL group of size 1, EA_DDP_FORCE=1): for every model family, one update with a
    single micro-batch and one with two (the first under no_sync).  Every parameter must report "gradient complete" exactly
    once per update — through autograd's hook or through the native layer runtime's callback, never both (a double report
    would hand a half-filled bucket to RCCL, which only a multi-rank run could notice) — and every parameter that received a
    gradient must have reported."""
    import json
    import socket

    import torch.distributed as dist

    from espresso_amd import functional as F
    from espresso_amd.data import audio_utils
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask
    from espresso_amd.trainer import Trainer

    rng = np.random.default_rng(11)
    utts = {}
    for i in range(9):
        u = f"utt{i}"
        path = os.path.join(tmp_dir, u + ".wav")
        audio_utils.write_wav(path, rng.standard_normal(int(16000 * rng.uniform(0.6, 1.6))) * 3000)
        utts[u] = {"wave": path, "text": " ".join(f"t{int(k)}" for k in rng.integers(0, 35, size=int(rng.integers(2, 7))))}
    with open(os.path.join(tmp_dir, "train.json"), "w") as f:
        json.dump(utts, f)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {"EA_DDP_FORCE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    dist.init_process_group(backend="nccl", device_id=torch.device(DEV))
    res = {}
    try:
        fams = [  # (family, autoregressive target side, "<s>" = blank in the dictionary, criterion, its options, model)
            ("conformer_ctc", False, True, "ctc_loss", {}, lambda: build_tiny_model("conformer", embed_dim=128, heads=2)),
            ("transformer_learned_ctc", False, True, "ctc_loss", {},
             lambda: build_tiny_model("transformer", embed_dim=128, heads=2, learned_pos=True)),
            ("encdec_lsce", True, False, "label_smoothed_cross_entropy_v2", {"label_smoothing": 0.1},
             lambda: build_tiny_encdec(embed_dim=128, heads=2, learned_pos=True)),
            # the transducer recipes: `autoregressive: true` (input feeding for the predictor) with the blank enabled
            ("transducer", True, True, "transducer_loss", {}, lambda: build_tiny_transducer(embed_dim=128, heads=2)),
            ("speech_lstm", True, False, "label_smoothed_cross_entropy_v2", {"label_smoothing": 0.1}, build_tiny_speech_lstm),
        ]
        for name, ar, blank, crit_name, ckw, make in fams:
            torch.manual_seed(0)
            d = (_Task(40) if blank else _TaskAR(40)).target_dictionary
            task = SpeechRecognitionEspressoTask.setup_task(
                SpeechRecognitionEspressoConfig(data=tmp_dir, autoregressive=ar, criterion_name=crit_name), tgt_dict=d)
            task.build_frontend(DEV)
            ds = task.load_dataset("train")
            # (the transducer data set batches by frames x tokens, speech_recognition.py:454)
            batches = task.get_batches(ds, max_tokens=4000 if crit_name == "transducer_loss" else 400, max_sentences=3, seed=1, epoch=1)
            samples = [task.to_device(ds.collater([ds[int(i)] for i in b]), DEV) for b in batches[:3]]
            model = make()
            crit = task.build_criterion(crit_name, sentence_avg=False, **ckw)
            tr = Trainer(task, model, crit, DEV, lr=1e-3, lr_scheduler=("tri_stage", dict(warmup_steps=5, hold_steps=5, decay_steps=5)))
            tr.reserve(samples[:1])
            fired = []
            orig = tr.ddp.all_reduce_grads

            def spy(orig=orig, tr=tr, fired=fired):
                fired.append(list(tr.ddp._fired))
                # parameters whose gradient is non-zero at this point must have reported
                fired.append([bool(p.grad.abs().sum() > 0) for p in tr.flat.params])
                return orig()

            tr.ddp.all_reduce_grads = spy
            tr.train_step([samples[0]])
            tr.train_step([samples[1], samples[2]])
            torch.cuda.synchronize()
            names = {id(p): n for n, p in tr.model.named_parameters()}
            silent = set()
            for counts, has_grad in (fired[0:2], fired[2:4]):
                silent |= {names[id(p)] for p, c, g in zip(tr.flat.params, counts, has_grad) if g and c == 0}
            res[name] = {"active": tr.ddp.active, "max_fired": tr.ddp.max_fired, "has_grad_but_silent": sorted(silent),
                         "reported": [sum(1 for c in fired[0] if c), sum(1 for c in fired[2] if c)], "n_params": len(tr.flat.params),
                         "native_layers": sum(1 for m in tr.model.modules() if getattr(m, "_ea_binding", None) is not None),
                         "finite": bool(torch.isfinite(tr.flat.p32).all())}
            F.set_grad_ready_callback(None)
    finally:
        F.set_grad_ready_callback(None)
        dist.destroy_process_group()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return res


def check_lstm_lm_training_vs_reference():
    """Language-model TRAINING step (lstm_lm_espresso + criterion `cross_entropy`): logits, summed NLL and every parameter
    gradient vs what the reference's model + fairseq's cross_entropy arithmetic produced (tests/golden/
    ref_lstm_lm_train_tiny.npz, oracle/gen_golden.py lmtrain), with tied and untied output embeddings."""
    from espresso_amd import registry
    from espresso_amd.models.lstm_lm import LSTMLanguageModelEspresso

    g = np.load(os.path.join(GOLD, "ref_lstm_lm_train_tiny.npz"))
    task = _TaskAR(40)
    assert task.target_dictionary.pad() == int(g["pad"]) and task.target_dictionary.eos() == int(g["eos"])
    res = {}
    for tag, share in (("tied", True), ("untied", False)):
        lm = LSTMLanguageModelEspresso.build_model(dict(arch="lstm_lm_wsj", decoder_embed_dim=32 if share else 24, decoder_hidden_size=32,
                                                        decoder_layers=2, decoder_out_embed_dim=32, dropout=0.0, share_embed=share), task)
        sd = {k[len(tag) + 6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "::sd::")}
        missing, unexpected = lm.load_state_dict(sd, strict=False)
        assert not missing and not unexpected, (tag, missing, unexpected)
        lm = lm.to(DEV).train()
        src, target = torch.from_numpy(g[tag + "::src"]).to(DEV), torch.from_numpy(g[tag + "::target"]).to(DEV)
        lens = torch.from_numpy(g[tag + "::lens"]).to(DEV)
        sample = {"net_input": {"src_tokens": src, "src_lengths": lens}, "target": target, "ntokens": int(lens.sum()), "nsentences": src.shape[0]}
        crit = registry.CRITERION_REGISTRY["cross_entropy"](task)
        logits = lm(**sample["net_input"])[0]
        valid = target.ne(task.target_dictionary.pad())
        ref_logits = torch.from_numpy(g[tag + "::logits"]).to(DEV)
        loss, sample_size, _ = crit(lm, sample)
        loss.backward()
        worst, worst_name = 0.0, None
        for n, p in lm.named_parameters():
            rg = torch.from_numpy(g[f"{tag}::grad::{n}"]).to(DEV)
            err = float((p.grad.float() - rg).abs().max() / rg.abs().max().clamp_min(1e-6))
            if err > worst:
                worst, worst_name = err, n
        # the same step on the bf16-emulating oracle (oracle/torch_ref.py lstm_lm, pinned to this fixture in fp32 by the CPU suite)
        from oracle import torch_ref

        sde = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        with torch_ref.bf16_emulation(True, flash=False):
            el = torch_ref.lstm_lm(src.cpu(), sde, pad_idx=int(g["pad"]))
            lp = torch.log_softmax(el.float(), -1)
            le = -(lp.gather(-1, target.cpu().unsqueeze(-1)).squeeze(-1) * valid.cpu()).sum()
            le.backward()
        emu_worst, emu_name = 0.0, None
        for n, p in lm.named_parameters():
            r = sde[n].grad
            err = float((p.grad.float().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-6))
            if err > emu_worst:
                emu_worst, emu_name = err, n
        res[tag] = {"emu_grad_rel_worst": emu_worst, "emu_grad_worst_name": emu_name,
                    "emu_logits_abs": float((logits.float().cpu() - el.detach())[valid.cpu()].abs().max()),
                    "logits_abs": float((logits.float() - ref_logits)[valid].abs().max()), "logits_scale": float(ref_logits[valid].abs().max()),
                    "loss_rel": abs(float(loss) - float(g[tag + "::loss"])) / float(g[tag + "::loss"]), "sample_size": sample_size,
                    "ntokens": sample["ntokens"], "grad_rel_worst": worst, "grad_worst_name": worst_name,
                    "n_params": sum(1 for _ in lm.parameters())}
    return res


def check_lm_train_cli(tmp_dir):
    """The language-model recipe layout (`task: language_modeling_for_asr`, `criterion: cross_entropy`,
    `lr_scheduler: reduce_lr_on_plateau`, `model: lstm_lm_espresso`) through `espresso_amd.speech_train`: binarised train split
    (fairseq mmap format), raw-text valid split, a few epochs on a tiny corpus with a learnable regularity; validation
    perplexity must drop and checkpoints appear."""
    import contextlib
    import io
    import json

    from espresso_amd import speech_train
    from espresso_amd.data.asr_dictionary import AsrDictionary
    from espresso_amd.data.lm_dataset import MMapTokenFile

    rng = np.random.default_rng(3)
    words = [f"w{i}" for i in range(30)]
    with open(os.path.join(tmp_dir, "dict.txt"), "w") as f:
        f.write("".join(f"{w} 1\n" for w in words))
    d = AsrDictionary.load(os.path.join(tmp_dir, "dict.txt"))

    def sentence():  # deterministic successor chain from a random start: learnable by a 1-step context
        n, w = int(rng.integers(3, 9)), int(rng.integers(0, 30))
        out = []
        for _ in range(n):
            out.append(words[w])
            w = (w * 7 + 3) % 30
        return " ".join(out)

    train = [sentence() for _ in range(96)]
    valid = [sentence() for _ in range(24)]
    MMapTokenFile.write(os.path.join(tmp_dir, "train"), [d.encode_line(s, append_eos=True).numpy() for s in train], dtype=np.int32)
    with open(os.path.join(tmp_dir, "valid"), "w") as f:
        f.write("\n".join(valid) + "\n")
    recipe = os.path.join(tmp_dir, "lm.yaml")
    with open(recipe, "w") as f:
        f.write("""
common: {seed: 1, log_interval: 4}
checkpoint: {save_dir: checkpoints, keep_last_epochs: 2}
task:
  _name: language_modeling_for_asr
  data: ???
  dict: ???
  sample_break_mode: eos
  tokens_per_sample: 64
dataset: {max_tokens: 256, batch_size: 16, required_batch_size_multiple: 8, train_subset: train, valid_subset: valid, curriculum: 1}
criterion: {_name: cross_entropy}
optimization: {max_epoch: 12, clip_norm: 1.0, update_freq: [1], lr: [0.05]}
optimizer: {_name: adam, adam_betas: "(0.9,0.999)", adam_eps: 1e-08, weight_decay: 0.0}
lr_scheduler: {_name: reduce_lr_on_plateau, lr_shrink: 0.5}
model:
  _name: lstm_lm_espresso
  dropout: 0.0
  decoder_embed_dim: 64
  decoder_hidden_size: 64
  decoder_layers: 2
  decoder_out_embed_dim: 64
  decoder_rnn_residual: false
  decoder_dropout_in: 0.0
  decoder_dropout_out: 0.0
  share_embed: true
  is_wordlm: false
""")
    buf = io.StringIO()
    save_dir = os.path.join(tmp_dir, "ck")
    with contextlib.redirect_stdout(buf):
        tr = speech_train.main(["--config", recipe, f"task.data={tmp_dir}", f"task.dict={tmp_dir}/dict.txt", f"checkpoint.save_dir={save_dir}"])
    lines = [json.loads(l) for l in buf.getvalue().splitlines() if l.startswith("{")]
    valid_lines = [l for l in lines if l["kind"] == "valid"]
    return {"files": sorted(os.listdir(save_dir)), "valid_loss": [l["loss"] for l in valid_lines], "valid_ppl": [l.get("ppl") for l in valid_lines],
            "num_updates": tr.num_updates, "lr": [l["lr"] for l in lines if l["kind"] == "epoch_end"],
            "train_loss": [l["loss"] for l in lines if l["kind"] == "train_inner"], "setup": [l for l in lines if l["kind"] == "setup"]}


# ------------------------------------------------------------------ training-trajectory parity (+n2)
def check_training_trajectory(steps=None, dropout=0.0):
    """tests/trajectory.py: `steps` Adam updates of the dh-64 Conformer-CTC on the learnable synthetic task, HIP path vs the
    oracle (fp32 and bf16-emulating), same initial weights, batches and order; held-out greedy token error rate at the end."""
    from espresso_amd import functional as F
    from espresso_amd.optim.adam import FlatAdam
    from espresso_amd.optim.flat import FlatParams
    from tests import trajectory as TR

    g, sd, _, _ = load_fixture(TR.FIXTURE)
    d, H, ffn = _fixture_shape(TR.FIXTURE)
    steps = steps or TR.STEPS
    train, heldout = TR.make_batches(TR.TRAIN_BATCHES, seed=0), TR.make_batches(TR.HELDOUT_BATCHES, seed=1)
    model = build_tiny_model("conformer", embed_dim=d, heads=H, ffn=ffn, dropout=dropout).to(DEV)
    load_ref_state(model, sd)
    flat = FlatParams(model, DEV)
    opt = FlatAdam(flat, lr=TR.LR, betas=TR.BETAS, eps=TR.EPS)
    model.train()
    losses, traces = [], []
    for step in range(steps):
        feats, lens, tg = (t.to(DEV) for t in train[step % len(train)])
        F.begin_step(feats.device)
        F.set_dropout_seed(1 + step)  # (trainer.py: seed + num_updates)
        with F.trace_dropout_seeds() as tr:
            out = model(feats, lens)
        traces.append(tr.entries)
        B, Tp = out["encoder_padding_mask"][0].shape
        nll, _ = F.ctc_loss(out["_logits_bt"][0], tg.to(torch.int32).contiguous(), out["src_lengths"][0].to(torch.int32),
                            (tg != 1).sum(-1).to(torch.int32), B, Tp, blank=0)
        loss = nll.sum()
        loss.backward()
        F.end_step()
        opt.clip_and_step(pre_scale=1.0, max_norm=TR.CLIP, denom_dev=torch.full((1,), float(B), device=DEV))
        losses.append(float(loss.detach()) / B)
    model.eval()
    err = tot = 0
    with torch.no_grad():
        for feats, lens, tg in heldout:
            out = model(feats.to(DEV), lens.to(DEV))
            e, t_ = TR.greedy_errors(out["encoder_out"][0].float().cpu(), out["src_lengths"][0].cpu(), tg)
            err += e
            tot += t_
    torch.cuda.synchronize()
    res = {"hip_losses": losses, "hip_ter": err / tot, "tokens": tot}
    from espresso_amd import _lib

    for tag, emu in (("fp32", False), ("emu", True)):
        ol, oe, ot = TR.train_oracle(sd, train, heldout, steps, emu, traces=traces if dropout > 0 else None,
                                     layer_seed_fn=_lib.lib().ea_layer_dropout_seed)
        rel = [abs(a - b) / max(b, 1e-3) for a, b in zip(losses, ol)]
        half = lambda ls: next((i for i, x in enumerate(ls) if x < 0.5 * ls[5]), len(ls))
        res[tag] = {"losses": ol, "ter": oe / ot, "max_rel_first24": max(rel[:24]), "max_rel_all": max(rel),
                    "auc_rel": abs(sum(losses) - sum(ol)) / sum(ol),
                    "half_plateau_step": half(ol), "hip_half_plateau_step": half(losses),
                    "final_loss": sum(ol[-10:]) / 10}
    res["hip_final_loss"] = sum(losses[-10:]) / 10
    return res


def check_encdec_training_trajectory(steps=60):
    """tests/trajectory.py, encoder-decoder edition (+n2 beyond CTC): `steps` Adam updates of the dh-64 Transformer encoder-decoder
    with label-smoothed CE on the learnable synthetic task, teacher-forced, HIP path (native encoder / decoder layers, fused
    attention, LS-CE kernel, FlatAdam) vs the oracle (fp32 and bf16-emulating) from the same weights, batches and order; per-update
    loss per token, and held-out NLL per token / teacher-forced token accuracy (BatchNorm on batch statistics both sides)."""
    from espresso_amd import functional as F
    from espresso_amd.optim.adam import FlatAdam
    from espresso_amd.optim.flat import FlatParams
    from tests import trajectory as TR

    g = np.load(os.path.join(GOLD, TR.ENCDEC_FIXTURE + ".npz"))
    sd0 = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    train, heldout = TR.make_batches(TR.TRAIN_BATCHES, seed=0), TR.make_batches(TR.HELDOUT_BATCHES, seed=1)
    model = _encdec_for(TR.ENCDEC_FIXTURE).to(DEV)
    missing, unexpected = model.load_state_dict(model.upgrade_state_dict_named(dict(sd0), ""), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    flat = FlatParams(model, DEV)
    opt = FlatAdam(flat, lr=TR.LR, betas=TR.BETAS, eps=TR.EPS)
    model.train()

    def run(batch, backward):
        feats, lens, tg = batch
        target, prev = TR.encdec_targets(tg)
        lo, extra = model(feats.to(DEV), lens.to(DEV), prev.to(DEV))
        loss, nll = F.label_smoothed_ce(extra["_logits_bu"], target.reshape(-1).to(torch.int32).to(DEV).contiguous(), TR.ENCDEC_PAD, TR.LS_EPS)
        n = int((target != TR.ENCDEC_PAD).sum())
        if backward:
            loss.backward()
        return lo, loss, nll, n, target

    losses = []
    for step in range(steps):
        F.begin_step(DEV)
        F.set_dropout_seed(1 + step)
        _, loss, _, n, _ = run(train[step % len(train)], True)
        F.end_step()
        opt.clip_and_step(pre_scale=1.0, max_norm=TR.CLIP, denom_dev=torch.full((1,), float(n), device=DEV))
        losses.append(float(loss.detach()) / n)
    nll = tok = hit = 0.0
    with torch.no_grad():
        for batch in heldout:
            lo, _, nl, n, target = run(batch, False)
            valid = target != TR.ENCDEC_PAD
            nll += float(nl)
            tok += n
            hit += int((lo.float().cpu().argmax(-1) == target)[valid].sum())
    torch.cuda.synchronize()
    res = {"hip_losses": losses, "hip_heldout_nll": nll / tok, "hip_heldout_acc": hit / tok, "tokens": int(tok),
           "hip_final_loss": sum(losses[-10:]) / 10}
    for tag, emu in (("fp32", False), ("emu", True)):
        ol, onll, oacc = TR.train_oracle_encdec(sd0, train, heldout, steps, emu)
        rel = [abs(a - b) / max(b, 1e-3) for a, b in zip(losses, ol)]
        res[tag] = {"losses": ol, "heldout_nll": onll, "heldout_acc": oacc, "max_rel_first24": max(rel[:24]), "max_rel_all": max(rel),
                    "auc_rel": abs(sum(losses) - sum(ol)) / sum(ol), "final_loss": sum(ol[-10:]) / 10}
    return res


def check_transducer_training_trajectory(steps=60):
    """tests/trajectory.py, transducer edition (+n2 beyond CTC): `steps` Adam updates of the tiny Conformer transducer (LSTM
    predictor, joint network, RNN-T loss through the `transducer_loss` criterion) on the learnable synthetic task, HIP path vs the
    oracle (fp32 and bf16-emulating) from the same weights, batches and order; per-update loss per sentence and held-out loss."""
    from espresso_amd import functional as F
    from espresso_amd.criterions.transducer_loss import TransducerLossCriterion
    from espresso_amd.optim.adam import FlatAdam
    from espresso_amd.optim.flat import FlatParams
    from tests import trajectory as TR

    g = np.load(os.path.join(GOLD, TR.TD_FIXTURE + ".npz"))
    sd0 = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    train, heldout = TR.make_batches(TR.TRAIN_BATCHES, seed=0), TR.make_batches(4, seed=1)
    model = build_tiny_transducer().to(DEV)
    missing, unexpected = model.load_state_dict(model.upgrade_state_dict_named(dict(sd0), ""), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    crit = TransducerLossCriterion(_Task(40))
    assert (crit.blank_idx, crit.pad_idx, crit.eos_idx) == (TR.TD_BLANK, TR.TD_PAD, TR.TD_EOS)
    flat = FlatParams(model, DEV)
    opt = FlatAdam(flat, lr=TR.LR, betas=TR.BETAS, eps=TR.EPS)
    model.train()

    def sample_of(batch):
        feats, lens, tg = batch
        target, prev, tl = TR.transducer_targets(tg)
        return {"net_input": {"src_tokens": feats.to(DEV), "src_lengths": lens.to(DEV), "prev_output_tokens": prev.to(DEV)},
                "target": target.to(DEV), "ntokens": int(tl.sum()) + len(tl)}

    losses = []
    for step in range(steps):
        F.begin_step(DEV)
        F.set_dropout_seed(1 + step)
        loss, B, _ = crit(model, sample_of(train[step % len(train)]))
        loss.backward()
        F.end_step()
        opt.clip_and_step(pre_scale=1.0, max_norm=TR.CLIP, denom_dev=torch.full((1,), float(B), device=DEV))
        losses.append(float(loss.detach()) / B)
    tot = n = 0.0
    with torch.no_grad():
        for batch in heldout:
            loss, B, _ = crit(model, sample_of(batch))
            tot += float(loss)
            n += B
    torch.cuda.synchronize()
    res = {"hip_losses": losses, "hip_heldout": tot / n, "hip_final_loss": sum(losses[-10:]) / 10}
    for tag, emu in (("fp32", False), ("emu", True)):
        ol, oh = TR.train_oracle_transducer(sd0, train, heldout, steps, emu)
        rel = [abs(a - b) / max(b, 1e-3) for a, b in zip(losses, ol)]
        res[tag] = {"losses": ol, "heldout": oh, "max_rel_first24": max(rel[:24]), "max_rel_all": max(rel),
                    "auc_rel": abs(sum(losses) - sum(ol)) / sum(ol), "final_loss": sum(ol[-10:]) / 10}
    return res


def check_conv_subsample_nondefault_channels(channels=(64, 192, 128, 64), strides=(1, 2, 1, 2)):
    """A sub-sampler whose middle layers the implicit-GEMM kernels accept only partly (192 input channels: forward yes, data
    gradient no; the last layer, 128 -> 64, takes them): forward and backward must pick the same lowering per layer.  Compared with the same stack on the im2col
    lowering (implicit GEMM switched off) — same bf16 operands, so outputs agree to one bf16 step and gradients to 2e-2."""
    from espresso_amd import functional as F
    from espresso_amd.modules.speech_convolutions import ConvBNReLU

    torch.manual_seed(0)
    m = ConvBNReLU(list(channels), [3] * len(channels), list(strides)).to(DEV)
    m.train()
    x = torch.randn(3, 41, 20, device=DEV)
    lens = torch.tensor([41, 33, 17], device=DEV)
    R = None
    out = {}
    for tag, on in (("igemm", True), ("im2col", False)):
        F.set_conv_implicit_gemm(on)
        try:
            for p in m.parameters():
                p.grad = None
            y, _, _, _ = m(x, lens)
            if R is None:
                R = torch.randn_like(y.float())
            (y.float() * R).sum().backward()
            torch.cuda.synchronize()
            out[tag] = (y.float().cpu(), {n: p.grad.float().cpu().clone() for n, p in m.named_parameters()})
        finally:
            F.set_conv_implicit_gemm(True)
    ya, ga = out["igemm"]
    yb, gb = out["im2col"]
    worst = max(((float((ga[n] - gb[n]).abs().max() / (gb[n].abs().max() + 1e-6)), n) for n in ga
                 if not (n.startswith("convolutions.") and n.endswith(".bias"))), key=lambda t: t[0])
    return {"out_rel": float((ya - yb).abs().max() / yb.abs().max()), "worst_grad": (worst[1], worst[0]),
            "finite": bool(all(torch.isfinite(g).all() for g in ga.values()))}
