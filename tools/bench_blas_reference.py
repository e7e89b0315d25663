"""Diagnostic only (not on the product path): what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on the hot-path shapes,
next to ea_gemm_bf16 on the same operands — a ceiling estimate for the hand-written kernels at this problem size.
    python tools/bench_blas_reference.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from espresso_amd import kernels as K

DEV = "cuda:0"


def timeit(fn, iters=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for M, N, Kd in [(6128, 2048, 512), (6128, 512, 2048), (6128, 512, 512), (6128, 1536, 512), (6128, 1024, 512), (6128, 512, 1024)]:
    a = torch.randn(M, Kd, device=DEV).to(torch.bfloat16)
    w = torch.randn(N, Kd, device=DEV).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    t_ea = timeit(lambda: K.gemm(a, w, c, M, N, Kd, lda=Kd, ldb=Kd, ldc=N))
    wt = w.t()
    t_blas = timeit(lambda: torch.mm(a, wt, out=c))
    fl = 2.0 * M * N * Kd
    print(f"{M}x{N}x{Kd}: ea_gemm_bf16 {t_ea:6.1f} us ({fl / t_ea / 1e6:6.1f} TFLOP/s)   torch.mm {t_blas:6.1f} us ({fl / t_blas / 1e6:6.1f} TFLOP/s)")
