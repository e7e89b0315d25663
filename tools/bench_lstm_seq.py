"""Per-step latency of the persistent LSTM recurrence (csrc/lstm_seq.hip): forward + backward of one layer, B rows, U steps.
    python tools/bench_lstm_seq.py [B] [U] [H]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from espresso_amd import kernels as K

DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
U = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
gx = torch.randn(U * B, 4 * H, device=DEV)
w = (torch.randn(4 * H, H, device=DEV) * 0.05).to(torch.bfloat16)
wt = w.t().contiguous()
hs = torch.empty(U * B, H, dtype=torch.bfloat16, device=DEV)
cs = torch.empty(U, B, H, device=DEV)
act = torch.empty(U, B, 4 * H, device=DEV)
hl = torch.empty(B, H, device=DEV)
cnt = torch.zeros(2, dtype=torch.int32, device=DEV)
dhs = torch.randn(U * B, H, device=DEV).to(torch.bfloat16)
dG = torch.empty(U * B, 4 * H, dtype=torch.bfloat16, device=DEV)
for name, fn in (("fwd", lambda: K.lstm_seq_fwd(gx, w, None, None, None, hs, cs, act, hl, cnt, B, U, H)),
                 ("bwd", lambda: K.lstm_seq_bwd(dhs, None, None, act, cs, None, None, wt, dG, None, None, cnt, B, U, H))):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{name}: {t:.1f} us per launch, {t / U:.2f} us per step  [B={B} U={U} H={H}] timeouts={int(cnt[1])}")
