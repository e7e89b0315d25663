"""Round 5: the transducer joint's three products at the recipe's micro-batch (70 000 lattice rows, V = 5004 at pitch 5056, J = 512)
under the 4-wave kernels and the 8-wave kernels (csrc/gemm_w8.hip in many-round mode, csrc/wgrad_w8.hip).  Isolated, hot, events on
the launch stream.  Output: microseconds per call and TFLOP/s; parity flag = outputs bit-identical to the 4-wave result."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from espresso_amd import _lib  # noqa: E402
from espresso_amd import functional as F  # noqa: E402
from espresso_amd import kernels as K  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.lib()


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    n = int(os.environ.get("N_ROWS", "70000"))
    V, J = 5004, 512
    Vp = (V + 63) // 64 * 64
    g = torch.Generator(device="cpu").manual_seed(0)
    Z = torch.relu(torch.randn(n, J, generator=g)).to(torch.bfloat16).to(DEV)
    W = (torch.randn(V, J, generator=g) * J ** -0.5).to(torch.bfloat16).to(DEV)
    b = torch.randn(V, generator=g).to(DEV)
    dl = torch.zeros(n, Vp, dtype=torch.bfloat16)
    dl[:, :V] = (torch.randn(n, V, generator=g) * 0.05).to(torch.bfloat16)
    dl = dl.to(DEV)
    wt = torch.zeros(J, Vp, dtype=torch.bfloat16, device=DEV)
    wt[:, :V] = W.t()
    fl = 2.0 * n * V * J

    def fwd():
        buf = torch.empty(n, Vp, dtype=torch.bfloat16, device=DEV)
        K.gemm(Z, W, buf, n, V, J, lda=J, ldb=J, ldc=Vp, bias=b)
        return buf

    def dgrad():
        dZ = torch.empty(n, J, dtype=torch.bfloat16, device=DEV)
        K.gemm(dl, wt, dZ, n, J, Vp, lda=Vp, ldb=Vp, ldc=J, aux=Z, ldaux=J, act="relu")
        return dZ

    def wgrad():
        return F._joint_wgrad(dl, Z, n, V, J, Vp)

    print(f"# n = {n} lattice rows, V = {V} (pitch {Vp}), J = {J}; {fl / 1e12:.3f} TFLOP per product")
    for name, fn, modes in (("forward  logits = Z W^T + b", fwd, [0, 1, 2, 4, 5]), ("dgrad    dZ = (dl W) * relu'", dgrad, [0, 1, 2, 3, 4])):
        ref = None
        for m in modes:
            old = lib.ea_set_gemm_w8(m)
            try:
                out = fn()
                torch.cuda.synchronize()
                us = timeit(fn)
            finally:
                lib.ea_set_gemm_w8(old)
            o = out[:, :V] if out.shape[1] == Vp else out
            if ref is None:
                ref = o.clone()
            same = bool((o.view(torch.int16) == ref.view(torch.int16)).all())
            label = {0: "4-wave", 1: "auto", 2: "256x256", 3: "128x128/4", 4: "256x128", 5: "128x256"}[m]
            print(f"{name:32s} {label:10s} {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {'identical' if same else 'DIFFERS'}")
    ref = None
    for m in (0, 1, 2):
        old = lib.ea_set_wgrad_w8(m)
        try:
            dw, db = wgrad()
            torch.cuda.synchronize()
            us = timeit(wgrad)
        finally:
            lib.ea_set_wgrad_w8(old)
        if ref is None:
            ref = (dw.clone(), db.clone())
        dwe = float((dw - ref[0]).abs().max() / ref[0].abs().max())
        dbe = float((db - ref[1]).abs().max() / ref[1].abs().max())
        label = {0: "4-wave", 1: "auto", 2: "8-wave"}[m]
        print(f"{'wgrad    dW = dl^T Z, db (slabs + sum)':32s} {label:10s} {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  rel diff dW {dwe:.1e} db {dbe:.1e}")


if __name__ == "__main__":
    main()
