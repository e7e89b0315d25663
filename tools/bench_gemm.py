"""Micro-benchmark of ea_gemm_bf16 on the hot-path shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from espresso_amd import kernels as K

DEV = "cuda:0"


def run(name, M, N, Kd, a_ks=False, b_ks=False, batch=1, c_f32=False, splitk=1, iters=20):
    A = torch.randn(batch, Kd * M, device=DEV).to(torch.bfloat16)
    B = torch.randn(batch, Kd * N, device=DEV).to(torch.bfloat16)
    C = torch.zeros(batch, M, N, dtype=torch.float32 if c_f32 else torch.bfloat16, device=DEV)
    kw = dict(lda=M if a_ks else Kd, ldb=N if b_ks else Kd, ldc=N, a_kstrided=a_ks, b_kstrided=b_ks, batch=batch,
              zdiv=1, sA=(M * Kd, 0), sB=(N * Kd, 0), sC=(M * N, 0), splitk=splitk)
    for _ in range(3):
        K.gemm(A, B, C, M, N, Kd, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        K.gemm(A, B, C, M, N, Kd, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    tf = 2.0 * M * N * Kd * batch / us / 1e6
    print(f"{name:28s} M={M:7d} N={N:5d} K={Kd:6d} b={batch:4d} ks=({int(a_ks)},{int(b_ks)}) split={splitk:3d}  {us:9.1f} us  {tf:7.1f} TF/s")


if __name__ == "__main__":
  from espresso_amd import _lib
  for variant, sw in ((0, 1), (0, 0)):
    _lib.lib().ea_set_gemm_variant(variant)
    _lib.lib().ea_set_gemm_xcd_swizzle(sw)
    print("=== gemm variant", variant, "xcd swizzle", sw)
    M = 6468
    run("ffn1 fwd", M, 2048, 512)
    run("ffn2 fwd", M, 512, 2048)
    run("qkv fwd", M, 1536, 512)
    run("out/pw2 fwd", M, 512, 512)
    run("fc_out fwd", M, 5004, 512)
    run("ffn dgrad (B ks)", M, 512, 2048, b_ks=True)
    run("ffn dgrad2 (B ks)", M, 2048, 512, b_ks=True)
    for sk in (1, 4, 8, 16):
        run("ffn wgrad", 2048, 512, M, a_ks=True, b_ks=True, c_f32=True, splitk=sk)
    for sk in (1, 8, 16, 32):
        run("proj wgrad 512x512", 512, 512, M, a_ks=True, b_ks=True, c_f32=True, splitk=sk)
    run("attn QK^T", 308, 308, 64, batch=168, c_f32=True)
    run("attn PV", 308, 64, 308, batch=168, b_ks=True)
    run("conv2 fwd (im2col)", 510720, 64, 576)
    run("conv dgrad", 510720, 576, 64, b_ks=True)
    for sk in (64, 200):
        run("conv wgrad", 64, 576, 510720, a_ks=True, b_ks=True, c_f32=True, splitk=sk)
    run("square 4096", 4096, 4096, 4096)
