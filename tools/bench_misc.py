"""Micro-benchmark of the HBM-bound kernels at the headline workload's shapes (run on the GPU box).

Prints us per call and the effective GB/s on the algorithmic bytes of each kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from espresso_amd import kernels as K

DEV = "cuda:0"


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def report(name, us, nbytes):
    print(f"{name:34s} {us:8.1f} us   {nbytes / us / 1e3:8.1f} GB/s   ({nbytes / 1e6:.1f} MB)")


if __name__ == "__main__":
    M, C, F = 6468, 512, 2048
    x = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    dy = torch.randn(M, C, device=DEV).to(torch.bfloat16)
    g = torch.ones(C, device=DEV)
    b = torch.zeros(C, device=DEV)
    dg = torch.zeros(C, device=DEV)
    db = torch.zeros(C, device=DEV)
    y, mean, rstd = K.layernorm_fwd(x, g, b)
    report("ln_fwd 6468x512", timeit(lambda: K.layernorm_fwd(x, g, b)), 2 * M * C * 2)
    report("ln_bwd 6468x512 (+dx_add)", timeit(lambda: K.layernorm_bwd(x, dy, g, mean, rstd, dg, db, dx_add=dy)), 4 * M * C * 2)
    big = torch.randn(M, F, device=DEV).to(torch.bfloat16)
    o512 = torch.zeros(C, device=DEV)
    o2048 = torch.zeros(F, device=DEV)
    lib = K._lib.lib()
    report("colsum 6468x512", timeit(lambda: lib.ea_colsum_bf16(K._p(x), K._p(o512), M, C, C, K._stream())), M * C * 2)
    report("colsum 6468x2048", timeit(lambda: lib.ea_colsum_bf16(K._p(big), K._p(o2048), M, F, F, K._stream())), M * F * 2)
    # split-K wgrad incl. reduce
    for (N_out, K_in) in ((2048, 512), (512, 512), (1536, 512), (1024, 512)):
        dyb = torch.randn(M, N_out, device=DEV).to(torch.bfloat16)
        xb = torch.randn(M, K_in, device=DEV).to(torch.bfloat16)
        dW = torch.zeros(N_out, K_in, device=DEV)
        for sk in (2, 4, 6, 8, 12, 16):
            us = timeit(lambda: K.gemm(dyb, xb, dW, N_out, K_in, M, lda=N_out, ldb=K_in, ldc=K_in, a_kstrided=True,
                                       b_kstrided=True, splitk=sk, accumulate=True), iters=30)
            print(f"wgrad {N_out}x{K_in} sk={sk:2d} (gemm+reduce) {us:8.1f} us  {2.0 * M * N_out * K_in / us / 1e6:7.1f} TF/s")
    # rel-pos softmax at B=21, T=308, H=8
    H, B, T = 8, 21, 308
    Sp, R = (T + 7) // 8 * 8, 2 * T - 1
    Rp = (R + 7) // 8 * 8
    ac = torch.randn(H * B * T, Sp, device=DEV)
    bd = torch.randn(H * B * T, Rp, device=DEV)
    P = torch.empty(H * B * T, Sp, dtype=torch.bfloat16, device=DEV)
    Pd = torch.empty_like(P)
    klen = torch.full((B,), T, dtype=torch.int32, device=DEV)
    us = timeit(lambda: lib.ea_relpos_softmax_fwd(K._p(ac), K._p(bd), K._p(klen), None, K._p(P), K._p(Pd), H, B, T, T, Sp, Rp, Sp, 0,
                                                  1, K.drop_params(0.1)[0], K.drop_params(0.1)[1], K._stream()))
    report("relpos_softmax_fwd", us, H * B * T * (Sp * 4 + Sp * 4 + 2 * Sp * 2))
    # fused attention forward at the headline shape
    Cc = H * 64
    qu = torch.randn(B * T, Cc, device=DEV).to(torch.bfloat16) * 0.3
    qv = torch.randn(B * T, Cc, device=DEV).to(torch.bfloat16) * 0.3
    qkv = torch.randn(B * T, 3 * Cc, device=DEV).to(torch.bfloat16)
    ppj = torch.randn(2 * T - 1, Cc, device=DEV).to(torch.bfloat16)
    for dp in (0.0, 0.1):
        us = timeit(lambda: K.flash_attention_fwd(qu, qv, qkv[:, Cc:], qkv[:, 2 * Cc:], ppj, klen, H, B, T, T, Cc, 3 * Cc, Cc, drop_p=dp, drop_seed=1))
        fl = 2.0 * H * B * T * T * 64 * 2 + 2.0 * H * B * T * (2 * T - 1) * 64
        print(f"flash_attention_fwd p={dp} {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s (useful)")

    out, lse = K.flash_attention_fwd(qu, qv, qkv[:, Cc:], qkv[:, 2 * Cc:], ppj, klen, H, B, T, T, Cc, 3 * Cc, Cc, drop_p=0.1, drop_seed=1)
    dout = torch.randn_like(out)
    dqkv = torch.empty_like(qkv)
    for name, dbg in (("bwd q+kv", 0), ("bwd q+kv, band buffer pre-zeroed", 2)):
        us = timeit(lambda: K.flash_attention_bwd(qu, qv, qkv[:, Cc:], qkv[:, 2 * Cc:], ppj, klen, out, dout, lse, dqkv[:, Cc:], dqkv[:, 2 * Cc:],
                                                  H, B, T, T, Cc, 3 * Cc, 3 * Cc, ldpp=Cc, causal=dbg, scaling=0.125, drop_p=0.1, drop_seed=1))
        print(f"flash_attention {name:24s} {us:8.1f} us")
