"""Round 6: which host lines launch the ~125 tiny ATen / runtime kernels of one bench step (copyBuffer, fills, dtype casts, compares)?
torch.profiler over 3 steps with Python stacks; prints, per (op, innermost repo frame), the number of calls per step."""
import collections
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from espresso_amd.data import synthetic  # noqa: E402

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
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for i in range(3, 3 + N):
        trainer.train_step([samples[i]])
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue  # top-level ops only
    frame = next((f for f in (ev.stack or []) if "/espresso_amd/" in f or "/bench.py" in f), "(no repo frame)")
    cnt[(ev.name, frame.split("/repo/")[-1][:110])] += 1
for (name, frame), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c / N:6.1f}  {name:28s} {frame}")
