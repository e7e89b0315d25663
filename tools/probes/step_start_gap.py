"""Probe: is the device idle at the start of a step in an UN-profiled run?  CUDA events right behind the fbank kernel and right
in front of the SpecAugment kernel (between them: host-side mask drawing + two small host-to-device copies) and at the end of
each step; prints the device-time between those points and the host lead.  python tools/probes/step_start_gap.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from espresso_amd import kernels as K
from espresso_amd.data import synthetic

device = torch.device("cuda:0")
task, model, criterion, trainer = bench.build(device)
batches, n_samples = synthetic.make_batches(20000, max_tokens=26000, max_sentences=24, seed=1)
pad = task.target_dictionary.pad()
steps = 12
samples = [synthetic.make_sample(b, n_samples, bench.VOCAB, pad, device, seed=1) for b in batches[: steps + 3]]
task.build_frontend(device, cmvn=bench.estimate_cmvn(task, samples[0], device))
task.begin_epoch(1)
trainer.reserve([max(samples, key=lambda s: s["audio_seconds"]), max(samples, key=lambda s: max(s["num_samples"]))])
ev = []
f0, s0 = K.fbank_batch, K.specaugment
def fb(*a, **k):
    r = f0(*a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(("fbank_done", e, time.perf_counter()))
    return r
def sa(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(("specaug_start", e, time.perf_counter()))
    return s0(*a, **k)
K.fbank_batch, K.specaugment = fb, sa
for i in range(3):
    trainer.train_step([samples[i]])
torch.cuda.synchronize()
ev.clear()
t0 = time.perf_counter()
start = torch.cuda.Event(enable_timing=True); start.record()
for i in range(3, 3 + steps):
    trainer.train_step([samples[i]])
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(("step_end", e, time.perf_counter()))
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t0) / steps:.2f} ms per step")
last_fb = None
for name, e, th in ev:
    tg = start.elapsed_time(e)
    if name == "fbank_done":
        last_fb = tg
    if name == "specaug_start":
        print(f"  fbank_done -> specaug_start on the device: {1e3 * (tg - last_fb):7.1f} us ; host was at {1e3 * (th - t0):8.2f} ms, device at {tg:8.2f} ms")
    if name == "step_end":
        print(f"step_end: host {1e3 * (th - t0):8.2f} ms, device {tg:8.2f} ms, host lead {tg - 1e3 * (th - t0):7.2f} ms")
