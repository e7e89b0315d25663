"""Isolated cost of the fused epilogues on the FFN GEMMs, and of cold (just-written, larger-than-L2) operands."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from espresso_amd import kernels as K  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, flush=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if flush is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters
    tot = 0.0
    for _ in range(iters):
        flush()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / iters


M, C, F = 6128, 512, 2048
x = torch.randn(M, C, device=DEV).to(torch.bfloat16)
w1 = torch.randn(F, C, device=DEV).to(torch.bfloat16)
w2 = torch.randn(C, F, device=DEV).to(torch.bfloat16)
b1 = torch.randn(F, device=DEV)
b2 = torch.randn(C, device=DEV)
z = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
h = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
y = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
big = torch.empty(300 * 1024 * 1024 // 2, dtype=torch.bfloat16, device=DEV)


def flush():
    big.zero_()  # 300 MB of writes: pushes everything out of L2 and the 256 MB infinity cache


def g1(**kw):
    K.gemm(x, w1, z, M, F, C, lda=C, ldb=C, ldc=F, **kw)


def g2(**kw):
    K.gemm(h, w2, y, M, C, F, lda=F, ldb=F, ldc=C, **kw)


for name, fn in (
    ("ffn W1 plain", lambda: g1()),
    ("ffn W1 +bias", lambda: g1(bias=b1)),
    ("ffn W1 +bias+silu", lambda: g1(bias=b1, act="silu")),
    ("ffn W1 +bias+silu+C2", lambda: g1(bias=b1, act="silu", C2=h, ldc2=F)),
    ("ffn W1 +bias+silu+C2+drop", lambda: g1(bias=b1, act="silu", C2=h, ldc2=F, drop_p=0.1, drop_seed=7)),
    ("ffn W2 plain", lambda: g2()),
    ("ffn W2 +bias+resid+drop", lambda: g2(bias=b2, resid=x, ldr=C, out_scale=0.5, drop_p=0.1, drop_seed=9)),
):
    hot = timeit(fn)
    cold = timeit(fn, iters=10, flush=flush)
    print(f"{name:32s} hot {hot:7.1f} us   cold (after 300 MB flush) {cold:7.1f} us")
