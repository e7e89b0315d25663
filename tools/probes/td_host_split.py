"""Config 4 update step: where the host time of a native layer call goes — the Python wrapper, the two C calls inside it, and the
backward's C call — measured in the real trainer loop (tools/bench_transducer.py) with perf_counter around each."""
import os, sys, time, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from espresso_amd import _lib, functional as F
import bench_transducer as B

lib = _lib.lib()
acc = collections.defaultdict(lambda: [0.0, 0])
class Proxy:
    def __getattr__(self, n):
        fn = getattr(lib, n)
        if not n.startswith("ea_"):
            return fn
        def timed(*a):
            t0 = time.perf_counter(); r = fn(*a); d = time.perf_counter() - t0
            e = acc[n]; e[0] += d; e[1] += 1
            return r
        return timed
_lib._lib = Proxy()
for cls in (F._ConformerLayerNative,):
    for name in ("forward", "backward"):
        orig = getattr(cls, name)
        def wrap(orig=orig, key=f"py:{cls.__name__}.{name}"):
            def f(*a, **k):
                t0 = time.perf_counter(); r = orig(*a, **k); d = time.perf_counter() - t0
                e = acc[key]; e[0] += d; e[1] += 1
                return r
            return staticmethod(f)
        setattr(cls, name, wrap())
steps, warm = 10, 3
res = B.run(steps=steps, warmup=warm)
n = steps + warm
print({k: res[k] for k in ("ms_per_step", "host_enqueue_ms_per_step")})
tot = 0.0
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"{k:46s} {c/n:8.1f} calls/update {1e6*t/c:8.1f} us each {1e3*t/n:8.2f} ms/update")
