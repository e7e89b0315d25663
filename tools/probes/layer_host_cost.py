"""Host cost of ONE native Conformer layer forward call at the transducer recipe's size (C 512, F 2048, H 8, M = B*T ~ 1 500 rows):
Python wrapper vs the ctypes C call vs the HIP launches inside it."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import gpu_checks as G
from espresso_amd import _lib, functional as F

dev = torch.device("cuda:0")
model = G.build_tiny_model("conformer", embed_dim=512, heads=8, ffn=2048, dropout=0.1).to(dev)
layer = model.encoder.layers[0]
B, T = 5, 300
x = torch.randn(B * T, 512, device=dev).bfloat16()
key_len = torch.tensor([300, 280, 250, 200, 150], dtype=torch.int32, device=dev)
lib = _lib.lib()
orig = lib.ea_conformer_layer_fwd_chained
acc = [0.0, 0]
def timed(*a):
    t0 = time.perf_counter(); r = orig(*a); acc[0] += time.perf_counter() - t0; acc[1] += 1; return r
class Proxy:
    def __getattr__(self, n):
        return timed if n == "ea_conformer_layer_fwd_chained" else getattr(lib, n)
_lib._lib = Proxy()
for mode in ("train", "eval"):
    layer.train(mode == "train")
    with torch.no_grad():
        for _ in range(30): layer(x, B, T, key_len=key_len)
        torch.cuda.synchronize(); acc[0] = 0.0; acc[1] = 0
        n = 300
        t0 = time.perf_counter()
        for _ in range(n): layer(x, B, T, key_len=key_len)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{mode}: python call {1e6*(t1-t0)/n:7.1f} us, of which the C call {1e6*acc[0]/acc[1]:7.1f} us; device drained after {1e3*(t2-t1):.2f} ms more "
          f"(GPU time per call ~{1e6*(t2-t0)/n:.0f} us)")
