"""Does the isolated GEMM time hold under sustained load?  Times blocks of 200 launches over ~2 s of back-to-back GEMMs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from espresso_amd import kernels as K  # noqa: E402

DEV = "cuda:0"
M, C, F = 6128, 512, 2048
h = torch.randn(M, F, device=DEV).to(torch.bfloat16)
w2 = torch.randn(C, F, device=DEV).to(torch.bfloat16)
y = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
x = torch.randn(M, C, device=DEV).to(torch.bfloat16)
w1 = torch.randn(F, C, device=DEV).to(torch.bfloat16)
z = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)


def g2():
    K.gemm(h, w2, y, M, C, F, lda=F, ldb=F, ldc=C)


def g1():
    K.gemm(x, w1, z, M, F, C, lda=C, ldb=C, ldc=F)


for name, fn in (("ffn W2 (N=512,K=2048)", g2), ("ffn W1 (N=2048,K=512)", g1)):
    evs = []
    for blk in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) * 1e3 / 200 for a, b in evs]
    print(name, "us per launch, blocks of 200:", " ".join(f"{t:.1f}" for t in ts[:4]), "...", " ".join(f"{t:.1f}" for t in ts[-4:]), flush=True)
