"""One-process-per-GPU trainer for the hot path: the update step of fairseq/trainer.py:780-1097
(micro-batches with no_sync -> grad sync -> multiply_grads(world/sample_size) -> clip_grad_norm ->
Adam -> lr schedule) re-scheduled for MI355X: gradient buckets are reduced by RCCL while backward is
still running, the per-step scalars (sample_size, loss, ntokens, nsentences) ride ONE small
all-reduce and stay on the device, and scaling + clipping + Adam + bf16 re-cast + grad zeroing are
a single streaming pass over the flat buffers.  No host synchronisation happens inside a step."""
import contextlib

import torch
import torch.distributed as dist

from . import functional as F
from .distributed.overlapped_ddp import OverlappedDistributedDataParallel
from .optim.adam import FlatAdam
from .optim.flat import FlatParams
from .optim.noam_lr_scheduler import NoamSchedule


class Trainer:
    def __init__(self, task, model, criterion, device, clip_norm=2.0, lr=5.0, warmup_steps=25000, adam_betas=(0.9, 0.98),
                 adam_eps=1e-8, weight_decay=0.0, final_lr=1e-6, seed=1, bucket_mb=64.0):
        self.task, self.criterion, self.device = task, criterion, device
        self.model = model.to(device)
        self.flat = FlatParams(self.model, device)
        self.optimizer = FlatAdam(self.flat, lr=lr, betas=adam_betas, eps=adam_eps, weight_decay=weight_decay)
        self.lr_scheduler = NoamSchedule(self.optimizer, lr=lr, warmup_steps=warmup_steps,
                                         model_size=model.cfg.encoder.embed_dim, final_lr=final_lr)
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.ddp = OverlappedDistributedDataParallel(self.model, self.flat, bucket_mb=bucket_mb)
        self.clip_norm = clip_norm
        self.seed = seed
        self.num_updates = 0
        self._stats = torch.zeros(4, dtype=torch.float32, device=device)
        self.last_coef = None

    def train_step(self, samples):
        """One optimizer update over a list of micro-batches (update_freq = len(samples))."""
        F.set_dropout_seed(self.seed + self.num_updates)  # trainer.py:782 _set_seed
        self.model.train()
        self._stats.zero_()
        for i, sample in enumerate(samples):
            last = i == len(samples) - 1
            ctx = contextlib.nullcontext() if last else self.ddp.no_sync()
            with ctx:
                sample = self.task.prepare_sample(sample, train=True)
                loss, sample_size, log = self.criterion(self.ddp, sample)
                loss.backward()
            self._stats[0] += float(sample_size)
            self._stats[1] += loss.detach()
            self._stats[2] += float(log["ntokens"])
            self._stats[3] += float(log["nsentences"])
        if self.ddp.active:
            dist.all_reduce(self._stats)  # C3 + sample_size in one 16-byte collective
        self.ddp.all_reduce_grads()
        self.last_coef = self.optimizer.clip_and_step(pre_scale=1.0, max_norm=self.clip_norm, denom_dev=self._stats[0:1])
        self.num_updates += 1
        self.model.set_num_updates(self.num_updates)
        self.lr_scheduler.step_update(self.num_updates)
        return self._stats

    def reserve(self, samples):
        """Size every grow-only arena (saved activations per layer, scratch, split-K slabs, allocator pools) for the largest
        batch shapes that will be seen: one forward + backward over each given sample with gradient reduction disabled,
        gradients and BatchNorm statistics restored afterwards; no optimizer update.  Called once at start-up (the reference's
        trainer does the same for the CUDA caching allocator with its dummy-batch / OOM-recovery logic, trainer.py:803-855)."""
        bn = {n: b.clone() for n, b in self.model.named_buffers()}
        self.model.train()
        with self.ddp.no_sync():
            for sample in samples:
                sample = self.task.prepare_sample(sample, train=True)
                loss, _, _ = self.criterion(self.ddp, sample)
                loss.backward()
        self.flat.zero_grad()
        with torch.no_grad():
            for n, b in self.model.named_buffers():
                b.copy_(bn[n])
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def valid_step(self, sample):
        return self.task.valid_step(sample, self.model, self.criterion)
