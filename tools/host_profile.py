"""Host-side cost of one update step: cProfile over a few steps of the bench workload (GPU box).
    python tools/host_profile.py [steps]  -> gpurun_out/host_profile.txt"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from espresso_amd.data import synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(3, 3 + steps):
    trainer.train_step([samples[i]])
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
out = io.StringIO()
out.write(f"host enqueue {1e3 * (t1 - t0) / steps:.2f} ms/step (under cProfile), drained after {1e3 * (t2 - t0) / steps:.2f} ms/step\n")
ps = pstats.Stats(pr, stream=out).sort_stats("cumulative")
ps.print_stats(70)
ps.sort_stats("tottime").print_stats(45)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "host_profile.txt"), "w").write(out.getvalue())
print(out.getvalue()[:6000])
