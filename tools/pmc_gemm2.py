"""L2 hit-rate probe: hot k-contiguous shapes with the glds kernel, XCD swizzle off/on (run under rocprofv3 --pmc)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from espresso_amd import _lib  # noqa: E402
from tools.bench_gemm import run  # noqa: E402

lib = _lib.lib()
lib.ea_set_gemm_glds(2)
M = 6128
for sw in (0, 1):
    lib.ea_set_gemm_xcd_swizzle(sw)
    run("ffn1 fwd", M, 2048, 512, iters=2)
    run("ffn2 fwd", M, 512, 2048, iters=2)
torch.cuda.synchronize()
