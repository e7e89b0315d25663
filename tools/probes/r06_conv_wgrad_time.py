"""Round 6: isolated timing of the sub-sampler's weight-gradient kernel (csrc/conv_igemm.hip conv_wgrad_kernel) at the recipe
batch's three shapes (26 000 input frames, 80 mel bins): conv 2 (64 -> 64, stride 2), conv 3 (64 -> 128), conv 4 (128 -> 128, stride 2).
Round 5 trace: 222 us per launch on average.  Prints us per launch and TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from espresso_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
B, T = 24, 1083
for name, (Tin, Fin, Cin, Cout, s) in {"conv2": (T, 80, 64, 64, 2), "conv3": ((T + 1) // 2, 40, 64, 128, 1), "conv4": ((T + 1) // 2, 40, 128, 128, 2)}.items():
    To, Fo = (Tin - 1) // s + 1, (Fin - 1) // s + 1
    X = torch.randn(B, Tin, Fin, Cin, device=dev).to(torch.bfloat16)
    dZ = torch.randn(B, To, Fo, Cout, device=dev).to(torch.bfloat16)
    dW = torch.zeros(Cout, 3, 3, Cin, device=dev)
    ws = torch.empty(lib.ea_conv3x3_wgrad_workspace_bytes(B, Tin, Fin, Cin, Cout, s, s), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: _lib.check(lib.ea_conv3x3_wgrad(X.data_ptr(), dZ.data_ptr(), dW.data_ptr(), ws.data_ptr(), B, Tin, Fin, Cin, Cout, s, s, st), "wgrad")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * B * To * Fo * Cout * 9 * Cin
    print(f"{name}: {us:7.1f} us per call (kernel + slab reduce), {fl / us * 1e-6:6.1f} TFLOP/s, positions {B * To * Fo}")
