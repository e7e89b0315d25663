"""Worker of tests/test_fairseq_binding.py::test_fairseq_legacy_ddp_reduces_gradients_written_behind_autograd: one rank of a gloo
group that wraps a stand-in model in fairseq's LegacyDistributedDataParallel (run as a script: the reference tree has a `tests`
package of its own, so a spawned child could not import this test module by name).  argv: rank world port; exit code 0 = ok."""
import os
import sys

REF = "/root/reference"


def main(rank, world, port):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "oracle", "ref_stubs"))
    sys.path.insert(1, REF)
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fairseq.distributed.legacy_distributed_data_parallel import LegacyDistributedDataParallel

    class Behind(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, holder):
            ctx.save_for_backward(x)
            ctx.holder = holder
            return x @ holder.W.detach().t()

        @staticmethod
        def backward(ctx, dy):
            (x,) = ctx.saved_tensors
            W = ctx.holder.W
            if W.grad is None:
                W.grad = torch.zeros_like(W)
            W.grad += dy.t() @ x  # in place, nothing returned for W: the layer runtime's pattern
            return dy @ W.detach(), None

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inp = torch.nn.Linear(8, 8)
            self.W = torch.nn.Parameter(torch.randn(4, 8) * 0.3)

        def forward(self, x):
            return Behind.apply(torch.tanh(self.inp(x)), self)

    torch.manual_seed(0)
    model, ref = Model(), Model()
    ref.load_state_dict(model.state_dict())
    ddp = LegacyDistributedDataParallel(model, process_group=dist.group.WORLD, buffer_size=16)  # small buffer: several buckets
    ok = True
    for step in range(2):
        model.zero_grad()
        xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(10 * step + r)) for r in range(world)]
        with ddp.no_sync():
            ddp(xs[rank]).pow(2).sum().backward()          # micro-batch 1: accumulate locally
        ddp(xs[rank] * 0.5).pow(2).sum().backward()        # micro-batch 2
        ddp.all_reduce_grads()                             # what fairseq's trainer calls after backward (trainer.py:884-923)
        ref.zero_grad()
        for r in range(world):
            ref(xs[r]).pow(2).sum().backward()
            ref(xs[r] * 0.5).pow(2).sum().backward()
        for (n, p), (_, rp) in zip(model.named_parameters(), ref.named_parameters()):
            ok = ok and bool(torch.allclose(p.grad, rp.grad / world, rtol=1e-5, atol=1e-6))
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])))
