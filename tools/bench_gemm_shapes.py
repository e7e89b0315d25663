"""Per-shape table of the step's forward / data-gradient GEMMs: every hot shape with the epilogue the layer
runtime gives it, isolated on an otherwise idle device —
  * hot  : back-to-back launches on the same operands (operands and outputs cache-resident),
  * cold : after 300 MB of writes that push everything out of the L2s and the 256 MB infinity cache,
for the kernel the dispatcher picks (`auto`) and for every forced variant (tile height 64 / 128 x register-staged / direct-to-LDS
ring of 2, 3, 4 stages), next to the vendor library on the plain product (torch.mm -> hipBLASLt; diagnostic ceiling only, not on
the product path).  The in-step part of profiles/r04_gemm_shapes.txt comes from a kernel trace (tools/gemm_instep_report.py), the
workgroup timing from tools/probes/gemm_timing.hip.
    python tools/bench_gemm_shapes.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from espresso_amd import _lib  # noqa: E402
from espresso_amd import kernels as K  # noqa: E402

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6240
lib = _lib.lib()
big = torch.empty(300 * 1024 * 1024 // 2, dtype=torch.bfloat16, device=DEV)


def timeit(fn, iters=30, cold=False):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    if not cold:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters
    tot = 0.0
    for _ in range(8):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / 8


def bf(*shape):
    return torch.randn(*shape, device=DEV).to(torch.bfloat16)


def cases():
    C, F = 512, 2048
    out = []
    for N, Kd, tag, kw in (
        (F, C, "ffn W1 fwd: bias+silu+drop, 2 outputs", "w1"),
        (F, C, "ffn W2 dgrad: drop * silu'(aux)", "w2d"),
        (C, F, "ffn W2 fwd: bias+drop, 0.5*y + resid", "w2"),
        (C, F, "ffn W1 dgrad: plain", "plain"),
        (3 * C, C, "qkv projection: bias", "bias"),
        (3 * C, C, "qkv projection: bias + query split (in-step form)", "qsplit"),
        (2 * C, C, "conv pointwise 1: plain", "plain"),
        (C, C, "out_proj / pointwise 2: bias+drop+resid", "w2"),
        (C, C, "dgrad of a C x C projection: plain", "plain"),
        (C, 2 * C, "conv pointwise 1 dgrad: plain", "plain"),
        (C, 3 * C, "qkv dgrad: plain", "plain"),
        (C, 2560, "fc0: bias", "bias"),
        (5004 + 52, C, "fc_out (row pitch 5056): bias", "bias"),
    ):
        a, w = bf(M, Kd), bf(N, Kd)
        c = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        args = dict(lda=Kd, ldb=Kd, ldc=N)
        if kw == "w1":
            args.update(bias=torch.randn(N, device=DEV), act="silu", C2=torch.empty_like(c), ldc2=N, drop_p=0.1, drop_seed=7)
        elif kw == "w2d":
            args.update(aux=bf(M, N), ldaux=N, act="silu", drop_p=0.1, drop_seed=7)
        elif kw == "w2":
            args.update(bias=torch.randn(N, device=DEV), resid=bf(M, N), ldr=N, out_scale=0.5, drop_p=0.1, drop_seed=9)
        elif kw == "bias":
            args.update(bias=torch.randn(N, device=DEV))
        elif kw == "qsplit":  # the q third goes to two scaled, position-biased copies instead of C (EaGemmParams.q_u)
            qu, qv = torch.empty(M, C, dtype=torch.bfloat16, device=DEV), torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
            args.update(bias=torch.randn(N, device=DEV), qsplit=(qu, qv, torch.randn(C, device=DEV), torch.randn(C, device=DEV), C, C, 0.125))
        out.append((N, Kd, tag, a, w, c, args))
    return out


W8 = (("256sq", 2), ("128sq/4", 3), ("256x128", 4), ("128x256", 5))
QUICK = os.environ.get("QUICK", "0") == "1"  # skip the forced 4-wave variants (round-4 columns)
print(f"# M = {M}; microseconds per launch, isolated (idle device); TFLOP/s of the auto choice in brackets")
print(f"# w8 columns: 8-wave large-tile kernels (csrc/gemm_w8.hip), forced configuration; ' ' = output bit-identical to the 4-wave kernel, '~' = differs by bf16 rounding only (< 2e-2 relative), '!' = wrong")
print(f"# {'N':>5} {'K':>5}  {'epilogue':44s} {'auto hot':>14s} {'auto cold':>9s} | {'4-wave hot':>10s} | {'vendor hot':>10s} {'cold':>6s} | "
      + " ".join(f"{h:>10s}" for h, _ in W8) + " | "
      + " ".join(f"{h:>7s}" for h in ("64/reg", "64/g2", "64/g3", "64/g4", "128/reg", "128/g2", "128/g3", "128/g4")))


def outputs(c, args):
    outs = [c]
    if "C2" in args:
        outs.append(args["C2"])
    if "qsplit" in args:
        outs += [args["qsplit"][0], args["qsplit"][1]]
    return outs


for N, Kd, tag, a, w, c, args in cases():
    fn = lambda: K.gemm(a, w, c, M, N, Kd, **args)
    lib.ea_set_gemm_variant(0)
    lib.ea_set_gemm_glds(1)
    lib.ea_set_gemm_w8(0)
    for o in outputs(c, args):
        o.fill_(7.0)
    fn()
    torch.cuda.synchronize()
    ref = [o.clone() for o in outputs(c, args)]
    old_hot = timeit(fn)
    w8cells = []
    for name, cfg in W8:
        lib.ea_set_gemm_w8(cfg)
        for o in outputs(c, args):
            o.fill_(7.0)
        fn()
        torch.cuda.synchronize()
        same = all(torch.equal(o, r) for o, r in zip(outputs(c, args), ref))
        if not same:  # (ping-pong kernels add even and odd k-tiles separately: last-bit differences in fp32 -> some bf16 roundings flip)
            worst = max(((o.float() - r.float()).abs() / (r.float().abs() + 1.0)).max().item() for o, r in zip(outputs(c, args), ref))
            flag = "~" if worst < 2e-2 else "!"
        else:
            flag = " "
        w8cells.append(f"{timeit(fn, iters=20):9.1f}{flag}")
    lib.ea_set_gemm_w8(1)
    hot, cold = timeit(fn), timeit(fn, cold=True)
    wt = w.t()
    vfn = lambda: torch.mm(a, wt, out=c)
    vhot, vcold = timeit(vfn), timeit(vfn, cold=True)
    cells = []
    lib.ea_set_gemm_w8(0)
    for variant in (2, 1):
        for g in (0, 2, 3, 4):
            if QUICK:
                continue
            lib.ea_set_gemm_variant(variant)
            lib.ea_set_gemm_glds(g)
            try:
                cells.append(f"{timeit(fn, iters=20):7.1f}")
            except Exception:
                cells.append("    err")
    lib.ea_set_gemm_variant(0)
    lib.ea_set_gemm_glds(1)
    lib.ea_set_gemm_w8(1)
    fl = 2.0 * M * N * Kd
    print(f"  {N:5d} {Kd:5d}  {tag:44s} {hot:7.1f} ({fl / hot / 1e6:4.0f}) {cold:9.1f} | {old_hot:10.1f} | {vhot:10.1f} {vcold:6.1f} | " + " ".join(w8cells)
          + " | " + " ".join(cells), flush=True)
