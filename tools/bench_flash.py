"""Time of the fused attention kernels at the headline shape, both implementations (flash_relpos.hip = the rel-pos encoder
kernels, flash_attention.hip = the general kernels), with and without attention dropout (HIP events around forward and
around the two backward kernels), plus the largest difference between the two implementations' results.
    python tools/bench_flash.py [B] [T] [ragged]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from espresso_amd import kernels as K
from espresso_amd import _lib

DEV = "cuda:0"
H = 8
B = int(sys.argv[1]) if len(sys.argv) > 1 else 21
T = int(sys.argv[2]) if len(sys.argv) > 2 else 308
ragged = len(sys.argv) > 3
PREZERO = 2 if os.environ.get("FLASH_PREZERO") == "1" else 0  # bit 1 of `causal`: dBD outside the band is already zero
C = H * 64
g = torch.Generator(device=DEV).manual_seed(0)
qu = torch.randn(B * T, C, device=DEV, generator=g).to(torch.bfloat16) * 0.3
qv = torch.randn(B * T, C, device=DEV, generator=g).to(torch.bfloat16) * 0.3
qkv = torch.randn(B * T, 3 * C, device=DEV, generator=g).to(torch.bfloat16)
pp = torch.randn(2 * T - 1, C, device=DEV, generator=g).to(torch.bfloat16)
dout = torch.randn(B * T, C, device=DEV, generator=g).to(torch.bfloat16)
klen = torch.full((B,), T, dtype=torch.int32, device=DEV)
if ragged:
    klen = torch.randint(T // 2, T + 1, (B,), device=DEV, generator=g).int()
    klen[0] = T
dqkv = torch.empty_like(qkv)


def once(p, relpos):
    out, lse, bits = K.flash_attention_fwd(qu, qv if relpos else None, qkv[:, C:], qkv[:, 2 * C:], pp if relpos else None, klen, H, B, T, T,
                                           C, 3 * C, C, drop_p=p, drop_seed=1, want_bits=True)
    return out, lse, bits


def run(p, relpos, n=30):
    f = b = 0.0
    for it in range(n + 5):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        out, lse, bits = once(p, relpos)
        e[1].record()
        res = K.flash_attention_bwd(qu, qv if relpos else None, qkv[:, C:], qkv[:, 2 * C:], pp if relpos else None, klen, out, dout, lse,
                                    dqkv[:, C:], dqkv[:, 2 * C:], H, B, T, T, C, 3 * C, 3 * C, ldpp=C, scaling=0.125, drop_p=p, drop_seed=1,
                                    keep_bits=bits, causal=PREZERO)
        e[2].record()
        torch.cuda.synchronize()
        if it >= 5:
            f += e[0].elapsed_time(e[1])
            b += e[1].elapsed_time(e[2])
    return 1e3 * f / n, 1e3 * b / n, (out, lse) + tuple(res) + (dqkv[:, C:].clone(),)


def rel(a, b):
    if a is None:
        return 0.0
    a, b = a.float(), b.float()
    m = torch.isfinite(a) & torch.isfinite(b)
    return float(((a - b).abs() * m).max() / b[m].abs().max().clamp_min(1e-6))


for p in (0.1, 0.0):
    res = {}
    for impl, name in ((1, "relpos kernels "), (0, "general kernels")):
        _lib.lib().ea_set_flash_relpos(impl)
        f, b, outs = run(p, True)
        res[impl] = outs
        print(f"{name} dropout={p}: fwd {f:.1f} us, bwd (Q + KV) {b:.1f} us  [H={H} B={B} T={T} ragged={ragged}]")
    _lib.lib().ea_set_flash_relpos(1)
    names = ["out", "lse", "t1", "t2", "dBD", "dkv"]
    print("   max |relpos - general| / max|general|: " + ", ".join(f"{n} {rel(a, b):.2e}" for n, a, b in zip(names, res[1], res[0])))
f, b, _ = run(0.1, False)
print(f"no positional term dropout=0.1: fwd {f:.1f} us, bwd {b:.1f} us")
