"""A few launches of the persistent GEMM (each tile configuration) and of the round-1 kernels on two hot shapes, for --pmc runs."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from espresso_amd import _lib  # noqa: E402
from espresso_amd import kernels as Kk  # noqa: E402
from espresso_amd._lib import EaGemmParams  # noqa: E402

DEV = "cuda:0"
lib = _lib.lib()
M = 6128
for N, K in ((2048, 2048), (512, 2048), (2048, 512)):
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    B = torch.randn(N, K, device=DEV).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    p = EaGemmParams()
    p.A, p.B, p.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    p.M, p.N, p.K, p.batch, p.zdiv = M, N, K, 1, 1
    p.lda, p.ldb, p.ldc = K, K, N
    p.alpha, p.out_scale, p.drop_scale, p.splitk = 1.0, 1.0, 1.0, 1
    for mode in (0, 2, 3, 4):
        lib.ea_set_gemm_persistent(mode)
        for _ in range(3):
            lib.ea_gemm_bf16(ctypes.byref(p), Kk._stream())
        torch.cuda.synchronize()
lib.ea_set_gemm_persistent(1)
