"""Few launches of the hot-path GEMM shapes for rocprofv3 --pmc passes (run under rocprofv3 on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.bench_gemm import run  # noqa: E402

if __name__ == "__main__":
    M = 6128
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    run("ffn1 fwd N2048 K512", M, 2048, 512, iters=it)
    run("ffn2 fwd N512 K2048", M, 512, 2048, iters=it)
    run("ffn dgrad N512 K2048 (B ks)", M, 512, 2048, b_ks=True, iters=it)
    run("ffn wgrad", 2048, 512, M, a_ks=True, b_ks=True, c_f32=True, splitk=4, iters=it)
    run("square 4096", 4096, 4096, 4096, iters=it)
    torch.cuda.synchronize()
