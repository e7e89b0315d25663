"""A few launches of the fused attention kernels at the headline shape (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from espresso_amd import kernels as K

DEV = "cuda:0"
H, B, T = 8, 21, 308
C = H * 64
qu = torch.randn(B * T, C, device=DEV).to(torch.bfloat16) * 0.3
qv = torch.randn(B * T, C, device=DEV).to(torch.bfloat16) * 0.3
qkv = torch.randn(B * T, 3 * C, device=DEV).to(torch.bfloat16)
pp = torch.randn(2 * T - 1, C, device=DEV).to(torch.bfloat16)
klen = torch.full((B,), T, dtype=torch.int32, device=DEV)
dqkv = torch.empty_like(qkv)
for _ in range(4):
    out, lse, bits = K.flash_attention_fwd(qu, qv, qkv[:, C:], qkv[:, 2 * C:], pp, klen, H, B, T, T, C, 3 * C, C, drop_p=0.1, drop_seed=1, want_bits=True)
    dout = torch.randn_like(out)
    K.flash_attention_bwd(qu, qv, qkv[:, C:], qkv[:, 2 * C:], pp, klen, out, dout, lse, dqkv[:, C:], dqkv[:, 2 * C:], H, B, T, T, C, 3 * C, 3 * C,
                          ldpp=C, scaling=0.125, drop_p=0.1, drop_seed=1, keep_bits=bits)
torch.cuda.synchronize()
