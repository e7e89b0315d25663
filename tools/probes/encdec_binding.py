"""Probe: does the enc-dec model's encoder run its Transformer layers in deferred (grouped weight-gradient) mode?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from espresso_amd import functional as F
calls = {"begin": 0, "deferred": 0, "why": {}}
orig = F._native_bwd_begin
def spy(sh, dev, tag, deferrable):
    r = orig(sh, dev, tag, deferrable)
    calls["begin"] += 1
    calls["deferred"] += r is not None
    k = (tag[0] if isinstance(tag[0], str) else "conformer", bool(deferrable), int(sh.pos_mode), F._defer_enabled)
    calls["why"][k] = calls["why"].get(k, 0) + 1
    return r
F._native_bwd_begin = spy
import bench_encdec
res = bench_encdec.run(steps=2, warmup=1)
print(calls)
