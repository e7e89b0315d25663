"""Host cost of one kernel launch through the C ABI (enqueue only, tiny problems so that the device drains faster than the host fills):
a LayerNorm launch (plain hipLaunchKernelGGL), a small GEMM through ea_gemm_bf16's dispatcher, an ATen op for comparison."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from espresso_amd import _lib, kernels as K

lib = _lib.lib()
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
M, C = 64, 512
x = torch.randn(M, C, device=dev).bfloat16()
y = torch.empty_like(x)
g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
st = K._stream()
w = torch.randn(512, C, device=dev).bfloat16()
out = torch.empty(M, 512, device=dev, dtype=torch.bfloat16)

def bench(name, fn, n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:40s} host {1e6*(t1-t0)/n:6.2f} us per call   (drain {1e3*(t2-t1):.2f} ms)")

bench("ea_layernorm_fwd (ctypes)", lambda: lib.ea_layernorm_fwd(P(x), P(g), P(b), P(y), P(mean), P(rstd), M, C, 1e-5, None, 0, 0, 1.0, st))
p = _lib.EaGemmParams()
def mk():
    pp = _lib.EaGemmParams()
    pp.A, pp.B, pp.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    pp.M, pp.N, pp.K, pp.lda, pp.ldb, pp.ldc = M, 512, C, C, C, 512
    pp.batch, pp.zdiv, pp.alpha, pp.out_scale, pp.splitk, pp.drop_scale = 1, 1, 1.0, 1.0, 1, 1.0
    return pp
pp = mk()
bench("ea_gemm_bf16 (ctypes)", lambda: lib.ea_gemm_bf16(ctypes.byref(pp), st))
bench("torch add_ (ATen)", lambda: y.add_(1))
bench("python no-op lambda", lambda: None)
fn = lib.ea_layernorm_fwd
args = (P(x), P(g), P(b), P(y), P(mean), P(rstd), M, C, 1e-5, None, 0, 0, 1.0, st)
bench("ea_layernorm_fwd (prebuilt args)", lambda: fn(*args))
side = torch.cuda.Stream()
ev = torch.cuda.Event()
def forkjoin():
    ev.record(); side.wait_event(ev)
bench("event record + wait (torch)", forkjoin)
