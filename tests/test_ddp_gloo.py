"""world_size=2 `gloo` test of the overlapped data-parallel path on CPU: bucketed all-reduce launched
from post-accumulate-grad hooks over the flat gradient buffer, no_sync accumulation, unused
parameters, and the fused (sample_size, loss, ntokens, nsentences) statistics all-reduce pattern
used by espresso_amd/trainer.py.  Mirrors what fairseq/trainer.py:884-923 + legacy_ddp guarantee:
after the step every rank holds sum_over_ranks(grad)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 32)
        self.b = nn.Linear(32, 8)
        self.unused = nn.Linear(4, 4)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from espresso_amd.distributed.overlapped_ddp import OverlappedDistributedDataParallel
    from espresso_amd.optim.flat import FlatParams

    torch.manual_seed(0)
    model = Tiny()
    ref = Tiny()
    ref.load_state_dict(model.state_dict())
    flat = FlatParams(model, torch.device("cpu"))
    ddp = OverlappedDistributedDataParallel(model, flat, bucket_mb=0.001)  # many small buckets
    assert len(ddp.buckets) > 2
    ok = True
    for step in range(2):
        g = torch.Generator().manual_seed(100 * step + rank)
        xs = [torch.randn(5, 16, generator=g) for _ in range(2)]
        # micro-batch 1 under no_sync, micro-batch 2 with reduction
        with ddp.no_sync():
            ddp(xs[0]).pow(2).sum().backward()
        ddp(xs[1]).pow(2).sum().backward()
        stats = torch.tensor([float(rank + 1), 2.0])
        dist.all_reduce(stats)
        ddp.all_reduce_grads()
        # reference: every rank's data, summed
        ref.zero_grad()
        for r in range(world):
            gr = torch.Generator().manual_seed(100 * step + r)
            for _ in range(2):
                ref(torch.randn(5, 16, generator=gr)).pow(2).sum().backward()
        for (n, p), (_, rp) in zip(model.named_parameters(), ref.named_parameters()):
            want = rp.grad if rp.grad is not None else torch.zeros_like(rp)
            ok &= bool(torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6))
            ok &= p.grad.data_ptr() >= flat.g32.data_ptr()  # still a view of the flat buffer
        ok &= stats.tolist() == [3.0, 4.0]
        flat.zero_grad()
    q.put((rank, ok))
    dist.destroy_process_group()


def test_overlapped_ddp_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_bucket_slices_cover_buffer():
    from espresso_amd.optim.flat import FlatParams

    m = Tiny()
    flat = FlatParams(m, torch.device("cpu"))
    sl = flat.slices_in_backward_order(100)
    assert sl[0][1] == flat.numel and sl[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(sl, sl[1:]))


def _bench_rank(tmp):
    """What a rank started by bench.spawn_ranks sees: the torch.distributed.run environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*); a world-2 gloo group forms from it and reduces."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    with open(os.path.join(tmp, f"rank{rank}.txt"), "w") as f:
        f.write(f"{dist.get_world_size()} {float(t)}")
    dist.destroy_process_group()


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` outside torchrun starts N ranks itself: the launcher used for that."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    bench.spawn_ranks(2, _bench_rank, (str(tmp_path),))
    for r in range(2):
        assert open(tmp_path / f"rank{r}.txt").read() == "2 3.0"


# ---- the native layer runtime's reporting pattern under two ranks --------------------------------------------------------------
class _NativeLike(torch.autograd.Function):
    """y = x W as the HIP layer runtime does it (espresso_amd/functional.py `_ConformerLayerNative`): the weight gradient is
    accumulated straight into W.grad behind autograd's back (backward returns None for W) and reported through
    functional's grad-ready callback — ONE LAYER LATE (deferred side work is joined by the next layer's call) and, for the last
    layer to run, from the autograd engine's end-of-backward callback."""

    @staticmethod
    def forward(ctx, x, k, owner):
        # like the real runtime, the parameters are NOT autograd inputs (the module is passed, not its tensors): no
        # AccumulateGrad node exists for them, so the only completion report is the runtime's own callback
        ctx.save_for_backward(x)
        ctx.owner, ctx.k = owner, k
        return x @ owner.w[k].detach()

    @staticmethod
    def backward(ctx, g):
        from espresso_amd import functional as F

        (x,) = ctx.saved_tensors
        W = ctx.owner.w[ctx.k]
        with torch.no_grad():
            W.grad.add_(x.t() @ g)  # (p.grad is a view of the flat gradient buffer)
        st = ctx.owner.state
        if st["accumulate"]:
            return g @ W.detach().t(), None, None
        if st["pending"] is not None and F._grad_ready_callback is not None:
            F._grad_ready_callback([st["pending"]])  # the PREVIOUS layer's gradient is complete only now
        st["pending"] = W

        def flush():
            if st["pending"] is not None and F._grad_ready_callback is not None:
                F._grad_ready_callback([st["pending"]])
            st["pending"] = None

        torch.autograd.Variable._execution_engine.queue_callback(flush)
        return g @ W.detach().t(), None, None


class NativeStack(nn.Module):
    def __init__(self):
        super().__init__()
        self.w = nn.ParameterList([nn.Parameter(torch.randn(12, 12) * 0.3) for _ in range(4)])
        self.head = nn.Linear(12, 5)  # an ordinary autograd parameter next to the native ones
        self.state = {"pending": None, "accumulate": False}

    def forward(self, x):
        x = x.detach().requires_grad_(True)  # (the real layers sit behind the sub-sampler, whose output requires grad)
        for k in range(len(self.w)):
            x = torch.tanh(_NativeLike.apply(x, k, self))
        return self.head(x)


def _native_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from espresso_amd.distributed.overlapped_ddp import OverlappedDistributedDataParallel
    from espresso_amd.optim.flat import FlatParams

    torch.manual_seed(0)
    model = NativeStack()
    ref_w = [W.detach().clone().requires_grad_(True) for W in model.w]
    ref_head = nn.Linear(12, 5)
    ref_head.load_state_dict(model.head.state_dict())
    flat = FlatParams(model, torch.device("cpu"))
    ddp = OverlappedDistributedDataParallel(model, flat, bucket_mb=0.0005)  # a bucket per parameter or so: order matters
    assert len(ddp.buckets) >= 4
    ok = True
    for step in range(3):
        g = torch.Generator().manual_seed(10 * step + rank)
        xs = [torch.randn(6, 12, generator=g) for _ in range(2)]
        model.state["accumulate"] = True
        with ddp.no_sync():
            ddp(xs[0]).pow(2).sum().backward()
        model.state["accumulate"] = False
        ddp(xs[1]).pow(2).sum().backward()
        assert model.state["pending"] is None  # the end-of-backward callback reported the last layer
        ddp.all_reduce_grads()
        ok &= ddp.max_fired == 1              # every parameter reported exactly once per update
        for W in ref_w:
            W.grad = None
        ref_head.zero_grad()
        for r in range(world):
            gr = torch.Generator().manual_seed(10 * step + r)
            for _ in range(2):
                x = torch.randn(6, 12, generator=gr)
                for W in ref_w:
                    x = torch.tanh(x @ W)
                ref_head(x).pow(2).sum().backward()
        for W, R in zip(model.w, ref_w):
            ok &= bool(torch.allclose(W.grad, R.grad, rtol=1e-5, atol=1e-6))
        ok &= bool(torch.allclose(model.head.weight.grad, ref_head.weight.grad, rtol=1e-5, atol=1e-6))
        flat.zero_grad()
    q.put((rank, ok))
    dist.destroy_process_group()


def test_overlapped_ddp_with_native_runtime_reporting_gloo_world2():
    """the two-rank test drove a plain nn.Linear model only.  Here the gradients are written behind autograd's
    back and reported late / from the end-of-backward callback, exactly as the deferred backward of the HIP layer runtime does;
    after the step both ranks hold the sum over ranks, and no parameter released its bucket twice or early."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_native_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res
