"""Time of the fused attention kernels at the headline shape, with and without attention dropout (HIP events, per kernel pair).
    python tools/bench_flash.py [B] [T]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from espresso_amd import kernels as K

DEV = "cuda:0"
H = 8
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
T = int(sys.argv[2]) if len(sys.argv) > 2 else 255
C = H * 64
qu = torch.randn(B * T, C, device=DEV).to(torch.bfloat16) * 0.3
qv = torch.randn(B * T, C, device=DEV).to(torch.bfloat16) * 0.3
qkv = torch.randn(B * T, 3 * C, device=DEV).to(torch.bfloat16)
pp = torch.randn(2 * T - 1, C, device=DEV).to(torch.bfloat16)
klen = torch.full((B,), T, dtype=torch.int32, device=DEV)
dqkv = torch.empty_like(qkv)


def run(p, relpos, n=30):
    f = b = 0.0
    for it in range(n + 5):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        out, lse = K.flash_attention_fwd(qu, qv if relpos else None, qkv[:, C:], qkv[:, 2 * C:], pp if relpos else None, klen, H, B, T, T, C,
                                         3 * C, C, drop_p=p, drop_seed=1)
        e[1].record()
        K.flash_attention_bwd(qu, qv if relpos else None, qkv[:, C:], qkv[:, 2 * C:], pp if relpos else None, klen, out, out, lse,
                              dqkv[:, C:], dqkv[:, 2 * C:], H, B, T, T, C, 3 * C, 3 * C, ldpp=C, scaling=0.125, drop_p=p, drop_seed=1)
        e[2].record()
        torch.cuda.synchronize()
        if it >= 5:
            f += e[0].elapsed_time(e[1])
            b += e[1].elapsed_time(e[2])
    return 1e3 * f / n, 1e3 * b / n


for relpos in (True, False):
    for p in (0.1, 0.0):
        f, b = run(p, relpos)
        print(f"relpos={relpos} dropout={p}: fwd {f:.1f} us, bwd (2 kernels) {b:.1f} us  [H={H} B={B} T={T}]")
