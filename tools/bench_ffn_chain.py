"""Forward FFN GEMM pair chained over 12 layers with per-layer weights and saved activations (the training step's access
pattern: weights and inputs are not cache-resident).  Reports the average W1 / W2 launch time from HIP events per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from espresso_amd import _lib  # noqa: E402
from espresso_amd import kernels as K  # noqa: E402

DEV = "cuda:0"
M, C, F, L = 6128, 512, 2048, 12
g = torch.Generator(device="cpu").manual_seed(0)
W1 = [(torch.randn(F, C, generator=g) * 0.05).to(torch.bfloat16).to(DEV) for _ in range(L)]
W2 = [(torch.randn(C, F, generator=g) * 0.05).to(torch.bfloat16).to(DEV) for _ in range(L)]
B1 = [torch.zeros(F, device=DEV) for _ in range(L)]
B2 = [torch.zeros(C, device=DEV) for _ in range(L)]
Z = [torch.empty(M, F, dtype=torch.bfloat16, device=DEV) for _ in range(L)]
H = [torch.empty(M, F, dtype=torch.bfloat16, device=DEV) for _ in range(L)]
X = [torch.randn(M, C, generator=g).to(torch.bfloat16).to(DEV) for _ in range(L + 1)]


def chain(record=None):
    for l in range(L):
        if record is not None:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
        K.gemm(X[l], W1[l], Z[l], M, F, C, lda=C, ldb=C, ldc=F, bias=B1[l], act="silu", C2=H[l], ldc2=F, drop_p=0.1, drop_seed=l)
        if record is not None:
            e1.record()
        K.gemm(H[l], W2[l], X[l + 1], M, C, F, lda=F, ldb=F, ldc=C, bias=B2[l], resid=X[l], ldr=C, out_scale=0.5, drop_p=0.1, drop_seed=l + 99)
        if record is not None:
            e2.record()
            record.append((e0, e1, e2))


def measure(tag):
    for _ in range(3):
        chain()
    rec = []
    for _ in range(5):
        chain(rec)
    torch.cuda.synchronize()
    t1 = sum(a.elapsed_time(b) for a, b, _ in rec) * 1e3 / len(rec)
    t2 = sum(b.elapsed_time(c) for _, b, c in rec) * 1e3 / len(rec)
    print(f"{tag:40s} W1 {t1:6.1f} us   W2 {t2:6.1f} us", flush=True)


lib = _lib.lib()
for glds, mask, var in ((1, 3, 0), (2, 3, 0), (0, 3, 0), (1, 3, 0)):
    lib.ea_set_gemm_glds(glds)
    lib.ea_set_gemm_xcd_swizzle(mask)
    lib.ea_set_gemm_variant(var)
    measure(f"glds={glds} xcd_mask={mask} tile_variant={var}")
