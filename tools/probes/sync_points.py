"""Where the training step synchronises host and device (torch.cuda sync-debug mode): every synchronising ATen call of a few updates,
counted by the repo frame that issued it.  usage: sync_points.py [transducer|ctc]"""
import collections, os, sys, traceback, warnings
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch

which = sys.argv[1] if len(sys.argv) > 1 else "transducer"
counts = collections.Counter()
armed = [False]

def show(message, category, filename, lineno, file=None, line=None):
    if not armed[0] or "synchroniz" not in str(message):
        return
    fr = [f for f in traceback.extract_stack() if "/espresso_amd/" in f.filename or "/tools/" in f.filename or f.filename.endswith("bench.py")]
    fr = [f for f in fr if "sync_points" not in f.filename]
    key = " <- ".join(f"{os.path.relpath(f.filename, R)}:{f.lineno}" for f in fr[-3:][::-1])
    counts[key] += 1

warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
steps = 4
if which == "transducer":
    import bench_transducer as B
    import time
    orig = time.perf_counter
    # arm after the warm-up: bench_transducer.run() has no hook, so count everything and divide by (steps + warmup)
    armed[0] = True
    B.run(steps=steps, warmup=2)
    n = steps + 2
else:
    sys.argv = ["bench.py", "--steps", str(steps), "--warmup", "2", "--no-cpu-baseline", "--no-decode", "--no-other-configs"]
    armed[0] = True
    import runpy
    runpy.run_path(os.path.join(R, "bench.py"), run_name="__main__")
    n = steps + 2
torch.cuda.set_sync_debug_mode(0)
print(f"synchronising calls over {n} updates (incl. set-up), by issuing frame:")
for k, v in counts.most_common(40):
    print(f"{v:6d}  {v / n:7.2f} per update  {k}")
