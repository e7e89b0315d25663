"""Training entry point for the reference's recipes on MI355X — the loop of fairseq_cli/train.py:44-560 (what
`fairseq-hydra-train --config-dir examples/asr_librispeech/config --config-name transformer_ctc_librispeech task.data=…`
runs) around the native update step of `espresso_amd/trainer.py`.

    python -m espresso_amd.speech_train --config recipe.yaml task.data=DIR task.dict=DICT [group.key=value …]
    python -m espresso_amd.speech_train DATA --task speech_recognition_espresso --arch speech_conv_lstm_wsj --flag value …   (legacy)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m espresso_amd.speech_train …

Same semantics as the reference's loop: per-epoch batch plan shuffled with `seed + epoch` and dealt round-robin to the ranks
(short ranks run a zero-weight dummy batch), `update_freq` micro-batches per update, validation + checkpoint at epoch ends and every
`save_interval_updates`, `checkpoint_best.pt` by `best_checkpoint_metric` (loss or WER of the greedy validation decoder), epoch-end
`lr_scheduler.step(epoch, val_loss)`, stop on `max_epoch` / `max_update` / `stop_min_lr` / `patience` / `stop_time_hours`, resume
from `checkpoint_last.pt` including the position inside the epoch.  What differs is the mechanics: one process per GPU, collated
batches (pinned raw audio) prefetched by a host thread and copied asynchronously, no host synchronisation inside an update — the
training statistics stay on the device and are read once per `log_interval`."""
import argparse
import inspect
import json
import math
import os
import queue
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

from . import registry
from .checkpoint_utils import CheckpointSaver
from .config import from_legacy_argv, is_legacy_argv, load_config, per_epoch
from .trainer import Trainer


def log(rank, kind, **kv):
    if rank == 0:
        print(json.dumps({"kind": kind, **kv}), flush=True)


class Prefetcher:
    """Collates batches ahead of the training loop (`dataset.num_workers` of the reference: there, DataLoader worker processes
    running get_waveform + fbank on the CPU).  Here a batch is file decoding + packing into a pinned int16 buffer — the features
    are computed on the GPU — and `num_workers` is the number of decoder threads of the library's batch reader
    (csrc/ingest.hip ea_audio_read_batch_i16: one ctypes call per batch, outside the interpreter lock), driven by ONE host
    thread that stays `depth` batches ahead."""

    def __init__(self, dataset, batches, depth=4, num_workers=1):
        src = getattr(dataset, "src", None)
        if getattr(src, "is_wave", False):
            src.lazy = True  # file entries reach the collater unread: it decodes the whole batch in parallel
            src.num_workers = max(1, int(num_workers))
        self.q = queue.Queue(maxsize=max(1, depth))
        self.t = threading.Thread(target=self._run, args=(dataset, batches), daemon=True)
        self.t.start()

    def _run(self, dataset, batches):
        try:
            for b in batches:
                self.q.put(dataset.collater([dataset[int(i)] for i in b]) if len(b) > 0 else {})
        except BaseException as e:  # surface reader errors in the training thread
            self.q.put(e)
        self.q.put(None)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item


def build_task(cfg):
    name = cfg["task"].get("_name", "speech_recognition_espresso")
    cls, dc = registry.TASK_REGISTRY[name], registry.TASK_DATACLASS_REGISTRY[name]
    known = dc.__dataclass_fields__
    kw = {k: v for k, v in cfg["task"].items() if k in known}
    extra = dict(train_subset=cfg["dataset"]["train_subset"], valid_subset=cfg["dataset"]["valid_subset"],
                 criterion_name=cfg["criterion"]["_name"], seed=cfg["common"]["seed"])
    if cfg.get("bpe"):
        extra.update(bpe=cfg["bpe"].get("_name"), sentencepiece_model=cfg["bpe"].get("sentencepiece_model"))
    kw.update({k: v for k, v in extra.items() if k in known})
    if "autoregressive" in known:
        # the reference's task default (espresso/tasks/speech_recognition.py:73-79); the LSTM recipes rely on it
        kw.setdefault("autoregressive", True)
    return cls.setup_task(dc(**kw))


def build_model(cfg, task):
    from . import models  # noqa: F401  (registers the model classes)

    name = cfg["model"]["_name"]
    cls = registry.MODEL_REGISTRY.get(name) or registry.ARCH_MODEL_REGISTRY.get(name)
    if cls is None:
        raise KeyError(f"unknown model {name!r}; registered: {sorted(registry.MODEL_REGISTRY)}")
    cfg_cls = getattr(cls, "config_class", None)
    mcfg = cfg_cls.from_dict(cfg["model"]) if cfg_cls is not None else dict(cfg["model"])
    return cls.build_model(mcfg, task)


def build_criterion(cfg, task):
    from . import criterions  # noqa: F401

    body = dict(cfg["criterion"])
    name = body.pop("_name")
    body.setdefault("sentence_avg", cfg["optimization"]["sentence_avg"])
    sig = inspect.signature(registry.CRITERION_REGISTRY[name].__init__).parameters
    if not any(p.kind == inspect.Parameter.VAR_KEYWORD for p in sig.values()):
        body = {k: v for k, v in body.items() if k in sig}
    return task.build_criterion(name, **body)


def _plan(task, ds, cfg, epoch, world, rank, train):
    d = cfg["dataset"]
    if train:
        return task.get_batches(ds, max_tokens=d["max_tokens"], max_sentences=d["batch_size"], max_positions=task.max_positions(),
                                seed=cfg["common"]["seed"], epoch=epoch, num_shards=world, shard_id=rank,
                                shuffle=epoch > d["curriculum"], bsz_mult=d["required_batch_size_multiple"])
    return task.get_batches(ds, max_tokens=d["max_tokens_valid"] or d["max_tokens"], max_sentences=d["batch_size_valid"] or d["batch_size"],
                            max_positions=task.max_positions(), seed=cfg["common"]["seed"], epoch=1, num_shards=world, shard_id=rank,
                            shuffle=False, bsz_mult=d["required_batch_size_multiple"])


def validate(cfg, trainer, task, subsets, device, world, rank):
    """fairseq_cli/train.py:452-520: every rank scores its shard, the logging outputs are summed over ranks, the task reduces
    them (adds wer / cer); returns [score of `best_checkpoint_metric` per subset] and the stats of the first subset."""
    scores, first = [], None
    for subset in subsets:
        ds = task.dataset(subset)
        logs = []
        for sample in Prefetcher(ds, [b for b in _plan(task, ds, cfg, 1, world, rank, train=False) if len(b) > 0],
                                 num_workers=cfg["dataset"].get("num_workers", 1) or 1):
            _, _, lg = trainer.valid_step(task.to_device(sample, device))
            logs.append({k: float(v) for k, v in lg.items() if isinstance(v, (int, float)) or torch.is_tensor(v) and v.numel() == 1})
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, logs)
            logs = [lg for part in gathered for lg in part]
        stats = task.reduce_metrics(logs, trainer.criterion)
        stats["num_updates"] = trainer.num_updates
        log(rank, "valid", subset=subset, **{k: (round(v, 4) if isinstance(v, float) else v) for k, v in stats.items()})
        scores.append(stats.get(cfg["checkpoint"]["best_checkpoint_metric"]))
        first = first or stats
    trainer.model.train()
    return scores, first


def update_group_sizes(n_batches, skip, update_freq):
    """Micro-batches per update for the rest of an epoch (fairseq's GroupedIterator over the epoch iterator, iterators.py:
    574-620): chunks of `update_freq`, the last one possibly shorter.  `skip` batches were consumed before a restart; saves
    happen after whole updates, so `skip` is a chunk boundary of the uninterrupted run and the chunks line up."""
    left = max(0, n_batches - skip)
    return [min(update_freq, left - s) for s in range(0, left, update_freq)]


class EarlyStop:
    """fairseq_cli/train.py:205-233 (`patience` validations without improvement)."""

    def __init__(self, cfg):
        self.patience, self.maximize = cfg["checkpoint"]["patience"], cfg["checkpoint"]["maximize_best_checkpoint_metric"]
        self.best, self.runs = None, 0

    def __call__(self, score):
        if score is None or self.patience <= 0:
            return False
        if self.best is None or (score > self.best if self.maximize else score < self.best):
            self.best, self.runs = score, 0
            return False
        self.runs += 1
        return self.runs >= self.patience


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", help="recipe YAML (hydra layout of the reference's examples/*/config/*.yaml)")
    ap.add_argument("--config-dir")
    ap.add_argument("--config-name")
    ap.add_argument("overrides", nargs="*", help="group.key=value")
    argv = list(sys.argv[1:] if argv is None else argv)
    if is_legacy_argv(argv):  # `fairseq_cli/train.py DATA --task … --arch … --flag value` of the WSJ / SWBD recipes
        cfg = from_legacy_argv(argv)
    else:
        args = ap.parse_args(argv)
        path = args.config or os.path.join(args.config_dir, args.config_name + ("" if args.config_name.endswith(".yaml") else ".yaml"))
        cfg = load_config(path, args.overrides)

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if not torch.cuda.is_available():
        raise RuntimeError("espresso_amd.speech_train needs an MI355X (HIP) device; there is no CPU training path")
    torch.cuda.set_device(local)
    device = torch.device(f"cuda:{local}")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)
    seed = int(cfg["common"]["seed"])
    np.random.seed(seed)
    torch.manual_seed(seed)

    task = build_task(cfg)
    valid_subsets = [s for s in str(cfg["dataset"]["valid_subset"]).split(",") if s]
    if not cfg["dataset"]["disable_validation"]:
        for s in valid_subsets:
            task.load_dataset(s)
    train_ds = task.load_dataset(cfg["dataset"]["train_subset"])
    task.build_frontend(device)
    model = build_model(cfg, task)
    criterion = build_criterion(cfg, task)
    trainer = Trainer.from_cfg(cfg, task, model, criterion, device)
    if not cfg["dataset"]["disable_validation"]:
        task.build_validation_decoder(model)  # the reference attaches it in build_model (speech_recognition.py:497-517)
    log(rank, "setup", model=cfg["model"]["_name"], criterion=cfg["criterion"]["_name"], world_size=world,
        num_params=sum(p.numel() for p in model.parameters()), train_utts=len(train_ds))

    saver = CheckpointSaver(cfg["checkpoint"])
    itr_state = saver.restore(trainer)
    epoch, skip = 1, 0
    if itr_state is not None:
        epoch, skip = int(itr_state["epoch"]), int(itr_state["iterations_in_epoch"])
        if itr_state.get("end_of_epoch", False):
            epoch, skip = epoch + 1, 0
        log(rank, "resume", epoch=epoch, iterations_in_epoch=skip, num_updates=trainer.num_updates)

    opt = cfg["optimization"]
    max_epoch = opt["max_epoch"] or math.inf
    max_update = opt["max_update"] or math.inf
    early_stop = EarlyStop(cfg)
    t_start = time.time()
    reserved = False
    should_stop = trainer.num_updates >= max_update
    interval = torch.zeros(4, dtype=torch.float32, device=device)  # [sample_size, loss, ntokens, nsentences] since the last log line
    n_interval, t_interval = 0, time.time()
    def validate_and_save(end_of_epoch, n_batches_done):
        """fairseq_cli/train.py:363-434.  Returns (valid_losses, should_stop); `valid_losses` is [None] whenever validation did
        not run in THIS call (train.py:422), so the epoch-end plateau schedule never sees a stale loss."""
        n = trainer.num_updates
        stop = n >= max_update
        if opt["stop_time_hours"] > 0:
            over = (time.time() - t_start) / 3600.0 > opt["stop_time_hours"]
            if world > 1:  # clocks differ by rank; every rank must take the same branch (collectives follow)
                flag = torch.tensor([float(over)], device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                over = bool(flag.item() > 0)
            stop = stop or over
        ck, d = cfg["checkpoint"], cfg["dataset"]
        do_save = ((end_of_epoch and epoch % ck["save_interval"] == 0) or stop
                   or (ck["save_interval_updates"] > 0 and n > 0 and n % ck["save_interval_updates"] == 0 and n >= d["validate_after_updates"]))
        do_validate = (((not end_of_epoch and do_save) or (end_of_epoch and epoch % d["validate_interval"] == 0) or stop
                        or (d["validate_interval_updates"] > 0 and n > 0 and n % d["validate_interval_updates"] == 0))
                       and not d["disable_validation"] and n >= d["validate_after_updates"])
        losses = [None]
        if do_validate:
            losses, _ = validate(cfg, trainer, task, valid_subsets, device, world, rank)
        stop |= early_stop(losses[0])
        if do_save or stop:
            # EpochBatchIterator.state_dict (fairseq/data/iterators.py:421-436): a finished epoch is written as the START of the next
            state = {"version": 2, "epoch": epoch + 1 if end_of_epoch else epoch,
                     "iterations_in_epoch": 0 if end_of_epoch else n_batches_done, "shuffle": True}
            files = saver.save(trainer, epoch, end_of_epoch, state, losses[0], is_master=rank == 0)
            if files:
                log(rank, "checkpoint", files=[os.path.basename(f) for f in files], num_updates=n, score=losses[0])
            if world > 1:
                dist.barrier()
        return losses, stop

    while epoch <= max_epoch and not should_stop:
        batches = _plan(task, train_ds, cfg, epoch, world, rank, train=True)
        if not reserved:  # size the arenas once for the largest batch and the longest utterance of the plan (no update)
            sizes = train_ds.num_tokens_vec(np.arange(len(train_ds)))
            full = [b for b in batches if len(b) > 0]
            by_m = max(full, key=lambda b: int(sizes[b].sum()))
            by_t = max(full, key=lambda b: int(sizes[b].max()))
            picks = [by_m] if by_m is by_t else [by_m, by_t]
            trainer.reserve([task.to_device(train_ds.collater([train_ds[int(i)] for i in b]), device) for b in picks])
            reserved = True
        task.begin_epoch(epoch, model)
        trainer.lr_step_begin_epoch(epoch)
        uf = int(per_epoch(opt["update_freq"], epoch))
        dummy_src = next((b for b in batches if len(b) > 0), None)
        todo = batches[skip:]
        done = skip
        stream = iter(Prefetcher(train_ds, [b if len(b) > 0 else dummy_src for b in todo], depth=cfg["dataset"].get("data_buffer_size", 4) or 4,
                                 num_workers=cfg["dataset"].get("num_workers", 1) or 1))
        for size in update_group_sizes(len(batches), skip, uf):
            group = []
            for b in batches[done:done + size]:
                sample = task.to_device(next(stream), device)
                if len(b) == 0:
                    sample["_dummy"] = True
                group.append(sample)
            done += size
            stats = trainer.train_step(group)
            if stats is None:  # out of memory in forward / backward: the update was skipped (fairseq/trainer.py:842-857)
                continue
            interval += stats
            n_interval += 1
            n = trainer.num_updates
            if n % cfg["common"]["log_interval"] == 0:
                ss, ls, nt, ns = interval.tolist()  # the only host<-device read of the training loop
                gnorm = float(trainer.last_coef[1]) if trainer.last_coef is not None else None
                trainer.check_grad_norm_consistency()  # (fairseq/trainer.py:1451-1488; gathered inside the statistics all-reduce)
                if gnorm is not None and not math.isfinite(gnorm):
                    # fairseq/trainer.py:949-958 (fp32): a non-finite gradient norm is fatal.  The fused Adam kernel leaves the
                    # parameters untouched for such a step; it is detected here, at the loop's only host<-device read.
                    raise FloatingPointError(f"gradients are Nan/Inf (update {n}, epoch {epoch})")
                dt = time.time() - t_interval
                log(rank, "train_inner", epoch=epoch, num_updates=n, loss=round(ls / max(ss, 1.0) / math.log(2), 4), ntokens=nt,
                    nsentences=ns, sample_size=ss, lr=trainer.get_lr(), gnorm=gnorm, ups=round(n_interval / max(dt, 1e-9), 2))
                interval.zero_()
                n_interval, t_interval = 0, time.time()
            end = done >= len(batches)
            if not end:
                _, should_stop = validate_and_save(False, done)
                if should_stop:
                    break
        if not should_stop:
            valid_losses, should_stop = validate_and_save(True, done)
            lr = trainer.lr_step(epoch, valid_losses[0])
            log(rank, "epoch_end", epoch=epoch, num_updates=trainer.num_updates, lr=lr)
            if lr <= opt["stop_min_lr"]:
                should_stop = True
            epoch, skip = epoch + 1, 0
    torch.cuda.synchronize()
    log(rank, "done", num_updates=trainer.num_updates, wall_s=round(time.time() - t_start, 1))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return trainer


if __name__ == "__main__":
    main(sys.argv[1:])
