"""Fixed cost vs per-k-step cost of the k-contiguous GEMM kernels on the hot-path shapes (M = 6128): time as a function of K
for N = 512 / 2048, plain and with the FFN-W1 epilogue, for each kernel variant.  Intercept = launch + prologue + epilogue,
slope = one 64-deep k-step.  Pre-built parameter structs, direct ctypes calls (host cost per call ~3 us)."""
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
M = int(os.environ.get("M", 6128))


def params(N, K, epi):
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    B = torch.randn(N, K, device=DEV).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    p = EaGemmParams()
    p.A, p.B, p.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    p.M, p.N, p.K, p.batch, p.zdiv = M, N, K, 1, 1
    p.lda, p.ldb, p.ldc = K, K, N
    p.alpha, p.out_scale, p.drop_scale, p.splitk = 1.0, 1.0, 1.0, 1
    keep = [A, B, C]
    if epi == "w1":
        b = torch.randn(N, device=DEV)
        C2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        p.bias, p.C2, p.ldc2, p.act = b.data_ptr(), C2.data_ptr(), N, 2
        p.drop_thr, p.drop_scale, p.drop_seed = int(0.1 * 2 ** 32), 1 / 0.9, 7
        keep += [b, C2]
    elif epi == "w2":
        b = torch.randn(N, device=DEV)
        R = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        p.bias, p.resid, p.ldr, p.out_scale = b.data_ptr(), R.data_ptr(), N, 0.5
        p.drop_thr, p.drop_scale, p.drop_seed = int(0.1 * 2 ** 32), 1 / 0.9, 9
        keep += [b, R]
    return p, keep


def timeit(p, iters=50):
    st = Kk._stream()
    for _ in range(5):
        lib.ea_gemm_bf16(ctypes.byref(p), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.ea_gemm_bf16(ctypes.byref(p), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


configs = [("old auto", dict(glds=1, variant=0, pk=0)), ("old reg bm64", dict(glds=0, variant=2, pk=0)),
           ("old glds2 bm64", dict(glds=2, variant=2, pk=0)), ("old glds3 bm64", dict(glds=3, variant=2, pk=0)),
           ("old glds2 bm128", dict(glds=2, variant=1, pk=0))]
extra = os.environ.get("EXTRA_CONFIGS")
for N, epi in ((512, None), (512, "w2"), (2048, None), (2048, "w1"), (1536, None), (1024, None)):
    print(f"--- M={M} N={N} epilogue={epi}")
    print(f"{'config':16s}" + "".join(f"  K={k:5d}" for k in (64, 128, 256, 512, 1024, 2048)))
    for name, cfg in configs:
        lib.ea_set_gemm_glds(cfg["glds"])
        lib.ea_set_gemm_variant(cfg["variant"])
        row = []
        for K in (64, 128, 256, 512, 1024, 2048):
            p, keep = params(N, K, epi)
            row.append(timeit(p))
        print(f"{name:16s}" + "".join(f"  {t:7.1f}" for t in row))
lib.ea_set_gemm_glds(1)
lib.ea_set_gemm_variant(0)
