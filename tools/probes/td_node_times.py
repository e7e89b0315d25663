"""Config 4 update step: host time of every autograd node class of espresso_amd.functional (forward and backward, inclusive of the C
calls they make), measured in the trainer loop with perf_counter.  The backward nodes run on the autograd engine's device thread,
which cProfile does not see."""
import os, sys, time, collections, inspect
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from espresso_amd import functional as F
import bench_transducer as B

acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(cls, name):
    orig = getattr(cls, name)
    key = f"{cls.__name__}.{name}"
    def f(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); d = time.perf_counter() - t0
        e = acc[key]; e[0] += d; e[1] += 1
        return r
    setattr(cls, name, staticmethod(f))
mods = [F]
for modname in ("espresso_amd.models.speech_lstm", "espresso_amd.criterions.transducer_loss", "espresso_amd.modules.speech_convolutions"):
    try:
        mods.append(__import__(modname, fromlist=["x"]))
    except Exception as e:
        print("skip", modname, e)
for mod in mods:
    for nm, cls in list(vars(mod).items()):
        if inspect.isclass(cls) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function:
            for name in ("forward", "backward"):
                if name in cls.__dict__:
                    wrap(cls, name)
steps, warm = 10, 3
res = B.run(steps=steps, warmup=warm)
n = steps + warm
print({k: round(res[k], 3) for k in ("ms_per_step", "host_enqueue_ms_per_step")})
tot = sum(t for t, c in acc.values()) / n * 1e3
print(f"autograd nodes of espresso_amd.functional: {tot:.2f} ms of host time per update")
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{k:44s} {c/n:7.1f} calls/update {1e6*t/c:8.1f} us each {1e3*t/n:7.2f} ms/update")
