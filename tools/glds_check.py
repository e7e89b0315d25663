"""A/B of the direct-to-LDS ring GEMM kernel: parity (tests/gpu_checks.check_gemm) and timing on the k-contiguous hot shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from espresso_amd import _lib  # noqa: E402
from tests import gpu_checks as G  # noqa: E402
from tools.bench_gemm import run  # noqa: E402

lib = _lib.lib()
for nst in (2,):
    lib.ea_set_gemm_glds(nst)
    worst = 0.0
    for variant in (1, 2):
        for (M, N, K) in ((200, 130, 64), (333, 257, 128), (129, 64, 1024), (64, 512, 192), (1000, 96, 448), (77, 300, 128), (515, 128, 64),
                          (6128, 512, 2048)):
            for kw in ({}, {"bias": True, "act": "silu", "resid": True}, {"c_f32": True, "splitk": 3}):
                if kw.get("splitk") and K < 192:
                    continue
                worst = max(worst, float(G.check_gemm(M, N, K, False, False, variant=variant, **kw)))
    print("stages", nst, "worst rel err", worst)
M = 6128
for nst, sw, var in ((2, 1, 0), (2, 1, 1), (2, 0, 1), (3, 1, 1)):
    lib.ea_set_gemm_glds(nst)
    lib.ea_set_gemm_xcd_swizzle(sw)
    lib.ea_set_gemm_variant(var)
    print("=== glds stages", nst, "xcd swizzle", sw, "variant", var)
    run("ffn1 fwd", M, 2048, 512)
    run("ffn2 fwd", M, 512, 2048)
    run("qkv fwd", M, 1536, 512)
    run("out/pw2 fwd", M, 512, 512)
    run("pw1 fwd", M, 1024, 512)
    run("fc_out fwd", M, 5004, 512)
    run("square 4096", 4096, 4096, 4096)
