"""Which lines of espresso_amd issue ATen device ops inside one update step (the "micro-op" launches between the HIP kernels).
    python tools/aten_ops.py [steps]  -> gpurun_out/aten_ops.txt
A TorchDispatchMode records every ATen call that touches a device tensor (views excluded) together with the innermost
espresso_amd / bench.py frame of its Python stack; autograd runs single-threaded for the measurement so that the ops issued
from inside custom backward functions are seen (and attributed) too."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten
import bench
from espresso_amd.data import synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
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

VIEWS = ("view", "as_strided", "slice", "select", "transpose", "t.", "expand", "alias", "detach", "unsqueeze", "squeeze", "permute",
         "_unsafe_view", "reshape", "unbind", "split", "narrow", "_reshape_alias", "lift_fresh", "is_pinned", "empty", "_local_scalar",
         "record_stream", "set_", "resize_", "sym_")
counts = collections.Counter()


class Tracer(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if any(name.startswith(v) for v in VIEWS):
            return out
        flat, _ = tree_flatten((args, kwargs, out))
        if not any(isinstance(a, torch.Tensor) and a.is_cuda for a in flat):
            return out
        site = "?"
        for fr in reversed(traceback.extract_stack()[:-1]):
            if ("espresso_amd/" in fr.filename or fr.filename.endswith("bench.py")) and "aten_ops.py" not in fr.filename:
                site = f"{fr.filename.split('repo/')[-1].split('espresso_amd/')[-1]}:{fr.lineno} {fr.name}"
                break
        counts[(site, name)] += 1
        return out


torch.autograd.set_multithreading_enabled(False)
with Tracer():
    for i in range(3, 3 + steps):
        trainer.train_step([samples[i]])
torch.cuda.synchronize()
out = [f"ATen device ops per update step (views excluded), {steps} steps, by issuing line"]
for (site, op), n in sorted(counts.items(), key=lambda kv: -kv[1]):
    out.append(f"{n / steps:8.1f}  {op:34s} {site}")
out.append(f"total {sum(counts.values()) / steps:.1f} per step")
txt = "\n".join(out)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "aten_ops.txt"), "w").write(txt)
print(txt[:12000])
