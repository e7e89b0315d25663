"""Checkpoint naming / retention / resume rules of fairseq/checkpoint_utils.py:34-310 for the training loop
(`espresso_amd/speech_train.py`): `checkpoint{epoch}.pt` at epoch ends, `checkpoint_{epoch}_{updates}.pt` every
`save_interval_updates`, `checkpoint_best.pt` on a new best validation score, `checkpoint.best_{metric}_{score}.pt` when
`keep_best_checkpoints > 0`, `checkpoint_last.pt` always; old files pruned by `keep_interval_updates` / `keep_last_epochs` /
`keep_best_checkpoints`.  The file content is `Trainer.state_dict()` (the reference's top-level keys)."""
import collections
import os
import re
import shutil

import numpy as np

from .data.data_utils import numpy_seed


def checkpoint_paths(path, pattern=r"checkpoint(\d+)\.pt"):
    """Checkpoints in `path` matching `pattern`, sorted by the first group in descending order (checkpoint_utils.py:493-516)."""
    pt = re.compile(pattern)
    entries = []
    for i, f in enumerate(os.listdir(path) if os.path.isdir(path) else []):
        m = pt.fullmatch(f)
        if m is not None:
            idx = float(m.group(1)) if len(m.groups()) > 0 else i
            entries.append((idx, f))
    return [os.path.join(path, x[1]) for x in sorted(entries, reverse=True)]


class CheckpointSaver:
    """State of `save_checkpoint.best` lives on the instance instead of a function attribute."""

    def __init__(self, cfg):
        self.cfg = cfg  # the `checkpoint` group
        self.best = None

    def is_better(self, a, b):
        return a >= b if self.cfg["maximize_best_checkpoint_metric"] else a <= b

    def save(self, trainer, epoch, end_of_epoch, itr_state, val_loss, is_master=True):
        cfg = self.cfg
        prev_best = self.best if self.best is not None else val_loss
        if val_loss is not None:
            self.best = (max if cfg["maximize_best_checkpoint_metric"] else min)(val_loss, prev_best)
        if cfg["no_save"]:
            return []
        if is_master:
            os.makedirs(cfg["save_dir"], exist_ok=True)
        updates = trainer.num_updates
        conds = collections.OrderedDict()
        conds[f"checkpoint{epoch}.pt"] = end_of_epoch and not cfg["no_epoch_checkpoints"] and epoch % cfg["save_interval"] == 0
        conds[f"checkpoint_{epoch}_{updates}.pt"] = (not end_of_epoch and cfg["save_interval_updates"] > 0
                                                     and updates % cfg["save_interval_updates"] == 0)
        conds["checkpoint_best.pt"] = val_loss is not None and (self.best is None or self.is_better(val_loss, self.best))
        metric = cfg["best_checkpoint_metric"]
        best_pat = r"checkpoint\.best_{}_(\d+\.?\d*)\.pt".format(re.escape(metric))
        if val_loss is not None and cfg["keep_best_checkpoints"] > 0:
            worst_best = self.best
            kept = checkpoint_paths(cfg["save_dir"], pattern=best_pat)
            if kept:
                p = kept[-1] if cfg["maximize_best_checkpoint_metric"] else kept[0]
                worst_best = float(p.rsplit("_")[-1].replace(".pt", ""))
            with numpy_seed(epoch, updates, val_loss):  # random digits resolve ties, as the reference
                rand_sfx = np.random.randint(0, cfg["keep_best_checkpoints"])
            conds["checkpoint.best_{}_{:.3f}{}.pt".format(metric, val_loss, rand_sfx)] = (
                worst_best is None or self.is_better(val_loss, worst_best))
        conds["checkpoint_last.pt"] = not cfg["no_last_checkpoints"]
        extra_state = {"train_iterator": itr_state, "val_loss": val_loss}
        if self.best is not None:
            extra_state["best"] = self.best
        files = [os.path.join(cfg["save_dir"], fn) for fn, c in conds.items() if c]
        if files and is_master:
            trainer.save_checkpoint(files[0], extra_state)
            for cp in files[1:]:
                shutil.copyfile(files[0], cp)
        if is_master:
            self._prune(end_of_epoch, best_pat)
        return files

    def _prune(self, end_of_epoch, best_pat):
        cfg = self.cfg
        if not end_of_epoch and cfg["keep_interval_updates"] > 0:
            for old in checkpoint_paths(cfg["save_dir"], pattern=r"checkpoint_\d+_(\d+)\.pt")[cfg["keep_interval_updates"]:]:
                if os.path.lexists(old):
                    os.remove(old)
        if cfg["keep_last_epochs"] > 0:
            for old in checkpoint_paths(cfg["save_dir"], pattern=r"checkpoint(\d+)\.pt")[cfg["keep_last_epochs"]:]:
                if os.path.lexists(old):
                    os.remove(old)
        if cfg["keep_best_checkpoints"] > 0:
            kept = checkpoint_paths(cfg["save_dir"], pattern=best_pat)
            if not cfg["maximize_best_checkpoint_metric"]:
                kept = kept[::-1]
            for old in kept[cfg["keep_best_checkpoints"]:]:
                if os.path.lexists(old):
                    os.remove(old)

    def restore(self, trainer):
        """checkpoint_utils.load_checkpoint :203-310: `restore_file` relative to `save_dir` unless it is a path; returns the
        training-iterator state to resume from (or None)."""
        cfg = self.cfg
        rf = cfg["restore_file"]
        path = rf if os.path.sep in rf or os.path.isabs(rf) else os.path.join(cfg["save_dir"], rf)
        extra = trainer.load_checkpoint(path, reset_optimizer=cfg["reset_optimizer"], reset_lr_scheduler=cfg["reset_lr_scheduler"])
        if extra is None:
            return None
        if "best" in extra and not cfg["reset_optimizer"] and not cfg["reset_meters"]:
            self.best = extra["best"]
        if cfg["reset_dataloader"]:
            return None
        return extra.get("train_iterator")
