"""Which lines of espresso_amd issue ATen device ops inside one update step (the "micro-op" launches between the HIP kernels).
    python tools/aten_ops.py [steps]  -> gpurun_out/aten_ops.txt
torch.profiler (CPU + CUDA activities, with_stack) over a few bench steps; every ATen op that launched a device kernel / copy is
attributed to the innermost espresso_amd (or bench.py) frame of its Python stack."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from espresso_amd.data import synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
device = torch.device("cuda:0")
task, model, criterion, trainer = bench.build(device)
batches, n_samples = synthetic.make_batches(20000, max_tokens=26000, max_sentences=24, seed=1)
pad = task.target_dictionary.pad()
samples = [synthetic.make_sample(b, n_samples, bench.VOCAB, pad, device, seed=1) for b in batches[: steps + 3]]
task.build_frontend(device, cmvn=bench.estimate_cmvn(task, samples[0], device))
task.begin_epoch(1)
trainer.reserve([max(samples, key=lambda s: s["audio_seconds"]), max(samples, key=lambda s: max(s["num_samples"]))])
for i in range(3):
    trainer.train_step([samples[i]])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(3, 3 + steps):
        trainer.train_step([samples[i]])
    torch.cuda.synchronize()

by_site = collections.Counter()
by_op = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or not ev.kernels:  # only ops that put something on the device queue themselves
        continue
    if any(ch.kernels for ch in ev.cpu_children if ch.name.startswith("aten::")):
        continue  # a composite: its child is counted
    site = "?"
    for fr in ev.stack or []:
        if "espresso_amd/" in fr or "bench.py" in fr:
            site = fr.split("repo/")[-1]
            break
    by_site[(site, ev.name)] += len(ev.kernels)
    by_op[ev.name] += len(ev.kernels)
out = [f"device launches issued by ATen ops, per step (over {steps} steps)"]
for (site, op), n in sorted(by_site.items(), key=lambda kv: -kv[1]):
    out.append(f"{n / steps:8.1f}  {op:28s} {site}")
out.append("")
for op, n in by_op.most_common():
    out.append(f"{n / steps:8.1f}  {op}")
out.append(f"total {sum(by_op.values()) / steps:.1f} per step")
txt = "\n".join(out)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "aten_ops.txt"), "w").write(txt)
print(txt[:8000])
