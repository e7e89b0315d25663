"""Data-parallel gradient reduction overlapped with backward, on the flat gradient buffer.

Reference behaviour being replaced: `ddp_backend: legacy_ddp`
(fairseq/distributed/legacy_distributed_data_parallel.py:76-165, called from fairseq/trainer.py:903-907):
copy all grads into one flat buffer, divide by world, ONE all-reduce after backward, copy back.
Result-equivalent MI355X design: gradients already live in one flat buffer (optim/flat.py); it is cut
into buckets in reverse parameter order, each bucket is all-reduced (SUM) by RCCL on a dedicated HIP
stream as soon as autograd has produced every gradient in it, while backward keeps computing
earlier layers.  The 1/world factor is folded into the optimizer's gradient scale, so the reduced
values equal the reference's mean.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets
default to 64 MB so each collective is large enough to stream over all links.

API mirrors the wrappers in fairseq/models/distributed_fairseq_model.py:35-147: `.module`,
`forward`, `no_sync()`, `all_reduce_grads()`.
"""
import contextlib
import os
from typing import List

import torch
import torch.distributed as dist

from .. import functional as _F
from ..optim.flat import FlatParams


class OverlappedDistributedDataParallel(torch.nn.Module):
    def __init__(self, module: torch.nn.Module, flat: FlatParams, process_group=None, bucket_mb: float = 64.0, last_bucket_mb: float = 8.0):
        super().__init__()
        self.module = module
        self.flat = flat
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # EA_DDP_FORCE=1: run the bucket hooks and collectives even with a single rank (exercises the RCCL / stream path on
        # a one-GPU box; the reduction itself is then the identity)
        self.active = self.world_size > 1 or (dist.is_initialized() and os.environ.get("EA_DDP_FORCE") == "1")
        self._rs_ag = os.environ.get("EA_DDP_COLLECTIVE", "all_reduce") == "rs_ag"
        self.accumulate_grads = False
        self._on_gpu = flat.g32.is_cuda
        # (a stream that provably does not share a hardware queue with the compute stream: functional.new_side_stream)
        self.comm_stream = _F.new_side_stream(flat.g32.device) if self._on_gpu else None
        elems = max(1, int(bucket_mb * 1024 * 1024 / 4))
        # the last bucket to complete (sub-sampler, fc0: the end of backward) is all-reduced with nothing left to hide it behind:
        # keep it small (8 MB: ~30 us over seven xGMI links instead of ~0.25 ms for 64 MB)
        self.buckets = flat.slices_in_backward_order(elems, int(last_bucket_mb * 1024 * 1024 / 4) if last_bucket_mb < bucket_mb else 0)
        # map each parameter to the bucket(s) that contain it
        self._param_buckets = {}
        self._bucket_total = [0] * len(self.buckets)
        for p in flat.params:
            o, n = flat.offsets[id(p)], p.numel()
            ids = [i for i, (s, e) in enumerate(self.buckets) if o < e and o + n > s]
            self._param_buckets[id(p)] = ids
            for i in ids:
                self._bucket_total[i] += 1
        self._pending = list(self._bucket_total)
        self._works: List = []
        self._launched = [False] * len(self.buckets)
        # per-step accounting: a parameter must report "gradient complete" at most once per update (autograd's
        # post-accumulate hook OR the native runtime's callback, never both) — a second report would release its bucket
        # to RCCL before the other gradients in it exist, which no single-rank run can notice
        self._index = {id(p): i for i, p in enumerate(flat.params)}
        self._fired = [0] * len(flat.params)
        self.max_fired = 0
        self._hooks = {id(p): self._make_hook(p) for p in flat.params}
        self._native_cache = {}
        if self.active:
            for p in flat.params:
                p.register_post_accumulate_grad_hook(self._hooks[id(p)])
            # gradients written directly by the native layer runtime never pass through autograd's
            # AccumulateGrad, so the runtime reports them here
            _F.set_grad_ready_callback(self._native_ready)

    def _make_hook(self, p):
        ids = self._param_buckets[id(p)]
        k = self._index[id(p)]

        def hook(param):
            if self.accumulate_grads:
                return
            self._fired[k] += 1
            for i in ids:
                self._pending[i] -= 1
                if self._pending[i] == 0:
                    self._launch(i)

        return hook

    def _native_ready(self, params):
        """A native layer call reports ALL of its parameters at once (the same list object every step): the per-bucket
        decrements of such a list are computed once and cached, so a layer costs a handful of dictionary operations instead of
        ~30 hook calls (12 layers x 30 closures were 1.6 ms of host time per step next to a 16.5 ms GPU step)."""
        if self.accumulate_grads:
            return
        key = tuple(map(id, params))  # (the parameters themselves, not the list object: a list mutated in place or a fresh list
        ent = self._native_cache.get(key)  # per call must not meet a stale or a never-matching entry)
        if ent is None:
            counts, idx = {}, []
            for p in params:
                k = self._index.get(id(p))
                if k is None:
                    continue
                idx.append(k)
                for i in self._param_buckets[id(p)]:
                    counts[i] = counts.get(i, 0) + 1
            ent = (tuple(params), sorted(counts.items()), idx, len(params))
            self._native_cache[key] = ent
        fired = self._fired
        for k in ent[2]:
            fired[k] += 1
        pending = self._pending
        for i, n in ent[1]:
            pending[i] -= n
            if pending[i] == 0:
                self._launch(i)

    def _launch(self, i):
        if self._launched[i]:
            return
        self._launched[i] = True
        s, e = self.buckets[i]
        view = self.flat.g32[s:e]
        if self._on_gpu:
            # the bucket's gradients were accumulated on the stream that reported last — and, for models that run independent
            # branches on their own streams (the transducer's predictor network: autograd accumulates its parameters' gradients
            # on that stream), possibly on others: wait for all of them
            cur = torch.cuda.current_stream()
            streams = [cur] + [st for st in _F.python_side_streams(view.device) if st != cur]
            for st in streams:
                self.comm_stream.wait_stream(st)
            with torch.cuda.stream(self.comm_stream):
                n = (e - s) // self.world_size
                if self._rs_ag and n * self.world_size == e - s:
                    # diagnostic (EA_DDP_COLLECTIVE=rs_ag): the same sum as reduce-scatter + all-gather, in place (rank r's shard
                    # is its own slice of the bucket); measured against the all-reduce in profiles/r05_ddp_one_rank_channels.json
                    r = dist.get_rank(self.process_group)
                    shard = view[r * n:(r + 1) * n]
                    self._works.append(dist.reduce_scatter_tensor(shard, view, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True))
                    w = dist.all_gather_into_tensor(view, shard, group=self.process_group, async_op=True)
                else:
                    w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
        else:
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
        self._works.append(w)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        """Accumulate gradients locally (all but the last micro-batch of an update; trainer.py:801-819)."""
        old = self.accumulate_grads
        self.accumulate_grads = True
        try:
            yield
        finally:
            self.accumulate_grads = old

    def all_reduce_grads(self):
        """Finish the step's reduction: launch buckets whose hooks did not all fire (unused parameters,
        dummy batches — trainer.py:873-877), wait for RCCL, re-arm.  Gradients hold the SUM over ranks."""
        if self.active:
            top = max(self._fired) if self._fired else 0
            self.max_fired = max(self.max_fired, top)
            if top > 1:
                bad = [i for i, n in enumerate(self._fired) if n > 1]
                raise RuntimeError(f"{len(bad)} parameter(s) reported their gradient complete more than once in one update "
                                   f"(flat indices {bad[:8]}): buckets were reduced before they were full")
            for i in range(len(self.buckets)):
                if not self._launched[i]:
                    self._launch(i)
            for w in self._works:
                w.wait()
            if self._on_gpu:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._works = []
        self._pending = list(self._bucket_total)
        self._launched = [False] * len(self.buckets)
        self._fired = [0] * len(self._fired)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("module"), name)
