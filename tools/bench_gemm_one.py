import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_gemm import run
which = sys.argv[1] if len(sys.argv) > 1 else "ffn1"
if which == "ffn1":
    run("ffn1 fwd", 6468, 2048, 512, iters=5)
elif which == "wgrad":
    run("ffn wgrad", 2048, 512, 6468, a_ks=True, b_ks=True, c_f32=True, splitk=8, iters=5)
elif which == "sq":
    run("square 4096", 4096, 4096, 4096, iters=5)
