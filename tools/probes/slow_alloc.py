"""Probe: which torch.empty / torch.zeros calls of the transducer step take more than 0.5 ms on the host, and what the caching
allocator did (num_alloc_retries, segments) — python tools/probes/slow_alloc.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
import traceback

orig_empty = torch.empty
log = []
def timed_empty(*a, **k):
    t0 = time.perf_counter()
    r = orig_empty(*a, **k)
    dt = time.perf_counter() - t0
    if dt > 5e-4:
        st = torch.cuda.memory_stats()
        fr = traceback.extract_stack(limit=3)[0]
        log.append((round(dt * 1e3, 2), r.numel() * r.element_size() >> 20, f"{os.path.basename(fr.filename)}:{fr.lineno}",
                    st.get("num_device_alloc", -1), st.get("num_device_free", -1), st.get("segment.all.current", -1)))
    return r
torch.empty = timed_empty
import bench_transducer
res = bench_transducer.run(steps=4, warmup=2)
print({k: res[k] for k in ("ms_per_batch", "host_enqueue_ms_per_step")})
for e in log[-40:]:
    print(e)
print("slow calls:", len(log))
