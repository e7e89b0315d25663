"""Which Python lines issue the small ATen kernels of one bench step: a TorchDispatchMode logs every non-view aten op on CUDA tensors
with the innermost repo frame (main thread only: the autograd engine's thread is not covered)."""
import collections, os, sys, traceback
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from espresso_amd.data import synthetic

VIEW = ("view", "reshape", "slice", "select", "permute", "transpose", "expand", "unsqueeze", "squeeze", "detach", "alias", "as_strided", "t.default",
        "_unsafe_view", "lift_fresh", "empty", "narrow", "unbind", "split", "size", "stride", "is_", "_local_scalar", "set_", "resize", "record_stream")
cnt = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        if not any(v in name for v in VIEW):
            def cuda(x):
                return isinstance(x, torch.Tensor) and x.is_cuda
            flat = list(args) + list((kwargs or {}).values())
            if any(cuda(a) or (isinstance(a, (list, tuple)) and any(cuda(b) for b in a)) for a in flat) or cuda(out):
                fr = [f for f in traceback.extract_stack() if ("/espresso_amd/" in f.filename or f.filename.endswith("bench.py")) and "probes" not in f.filename]
                key = (name, f"{os.path.relpath(fr[-1].filename, R)}:{fr[-1].lineno}" if fr else "?")
                cnt[key] += 1
        return out

device = torch.device("cuda:0")
task, model, criterion, trainer = bench.build(device)
batches, n_samples = synthetic.make_batches(20000, max_tokens=26000, max_sentences=24, seed=1)
pad = task.target_dictionary.pad()
samples = [synthetic.make_sample(b, n_samples, bench.VOCAB, pad, device, seed=1) for b in batches[:8]]
task.build_frontend(device, cmvn=bench.estimate_cmvn(task, samples[0], device))
task.begin_epoch(1)
trainer.reserve([max(samples, key=lambda s: s["audio_seconds"]), max(samples, key=lambda s: max(s["num_samples"]))])
for i in range(3):
    trainer.train_step([samples[i]])
torch.cuda.synchronize()
N = 3
with Log():
    for i in range(3, 3 + N):
        trainer.train_step([samples[i]])
torch.cuda.synchronize()
tot = 0
for (name, frame), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    tot += c / N
    print(f"{c / N:6.1f}  {name:34s} {frame}")
print("total per step", tot)
