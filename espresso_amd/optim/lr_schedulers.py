"""The other learning-rate schedules the reference's recipes name (host-side scalars; the optimizer kernel reads the value):

* `tri_stage` — fairseq/optim/lr_scheduler/tri_stage_lr_scheduler.py:60-175 (Conformer / Transformer enc-dec recipes):
  linear warm-up from `init_lr_scale*lr`, hold, exponential decay to `final_lr_scale*lr`, then constant.
* `polynomial_decay_v2` — espresso/optim/lr_scheduler/polynomial_decay_schedule.py:13-17 over
  fairseq/optim/lr_scheduler/polynomial_decay_schedule.py:39-89 (Transformer-transducer recipe): the fairseq schedule without the
  per-epoch reset.
* `reduce_lr_on_plateau_v2` — espresso/optim/lr_scheduler/reduce_lr_on_plateau_v2.py:36-66 over
  fairseq/optim/lr_scheduler/reduce_lr_on_plateau.py:60-143 (LSTM recipes): shrink on validation-loss plateaus starting at
  `start_reduce_lr_epoch`, floor `final_lr_scale*lr`, optional linear warm-up per update."""
import math

from ..registry import register_lr_scheduler


@register_lr_scheduler("tri_stage")
class TriStageLRSchedule:
    def __init__(self, optimizer, lr=1e-3, warmup_steps=0, hold_steps=0, decay_steps=0, phase_ratio=None, init_lr_scale=0.01,
                 final_lr_scale=0.01, max_update=0):
        self.optimizer = optimizer
        self.peak_lr, self.init_lr, self.final_lr = lr, init_lr_scale * lr, final_lr_scale * lr
        if phase_ratio is not None:
            assert max_update > 0 and sum(phase_ratio) == 1, "phase ratios must add up to 1"
            warmup_steps, hold_steps, decay_steps = (int(max_update * r) for r in phase_ratio)
        assert warmup_steps + hold_steps + decay_steps > 0, "please specify steps or phase_ratio"
        self.warmup_steps, self.hold_steps, self.decay_steps = warmup_steps, hold_steps, decay_steps
        self.warmup_rate = (self.peak_lr - self.init_lr) / warmup_steps if warmup_steps != 0 else 0
        self.decay_factor = -math.log(final_lr_scale) / decay_steps
        self.lr = self.init_lr
        optimizer.set_lr(self.lr)

    def step_update(self, num_updates):
        n = num_updates
        if n < self.warmup_steps:
            self.lr = self.init_lr + self.warmup_rate * n
        elif n < self.warmup_steps + self.hold_steps:
            self.lr = self.peak_lr
        elif n <= self.warmup_steps + self.hold_steps + self.decay_steps:
            self.lr = self.peak_lr * math.exp(-self.decay_factor * (n - self.warmup_steps - self.hold_steps))
        else:
            self.lr = self.final_lr
        self.optimizer.set_lr(self.lr)
        return self.lr

    def step(self, epoch, val_loss=None):  # no change at epoch boundaries
        return self.lr


@register_lr_scheduler("polynomial_decay_v2")
class PolynomialDecayV2LRSchedule:
    def __init__(self, optimizer, lr=1e-3, warmup_updates=0, end_learning_rate=0.0, power=1.0, total_num_update=1000000):
        assert total_num_update > 0
        self.optimizer, self.lr = optimizer, lr
        self.warmup_updates, self.end_learning_rate, self.power, self.total_num_update = warmup_updates, end_learning_rate, power, total_num_update
        self.warmup_factor = 1.0 / warmup_updates if warmup_updates > 0 else 1
        optimizer.set_lr(self.warmup_factor * lr)

    def step_update(self, num_updates):
        if self.warmup_updates > 0 and num_updates <= self.warmup_updates:
            self.warmup_factor = num_updates / float(self.warmup_updates)
            lr = self.warmup_factor * self.lr
        elif num_updates >= self.total_num_update:
            lr = self.end_learning_rate
        else:
            pct = 1 - (num_updates - self.warmup_updates) / (self.total_num_update - self.warmup_updates)
            lr = (self.lr - self.end_learning_rate) * pct ** self.power + self.end_learning_rate
        self.optimizer.set_lr(lr)
        return lr

    def step_begin_epoch(self, epoch):  # v2: the schedule is a function of the update count only
        pass

    def step(self, epoch, val_loss=None):
        return self.optimizer.get_lr()


@register_lr_scheduler("reduce_lr_on_plateau_v2")
class ReduceLROnPlateauLRScheduleV2:
    """Plateau logic of torch.optim.lr_scheduler.ReduceLROnPlateau (mode min/max, relative threshold, cooldown 0) restated on the
    scalar learning rate of the flat optimizer."""

    def __init__(self, optimizer, lr=1e-3, lr_shrink=0.1, lr_threshold=1e-4, lr_patience=0, warmup_updates=0, warmup_init_lr=-1,
                 start_reduce_lr_epoch=0, final_lr_scale=0.01, maximize_best_checkpoint_metric=False):
        self.optimizer, self.base_lr = optimizer, lr
        self.factor, self.threshold, self.patience = lr_shrink, lr_threshold, lr_patience
        self.start_epoch, self.min_lr, self.maximize = start_reduce_lr_epoch, final_lr_scale * lr, maximize_best_checkpoint_metric
        self.best = -math.inf if self.maximize else math.inf
        self.num_bad_epochs, self.last_epoch = 0, 0
        self.warmup_updates = warmup_updates
        if warmup_init_lr < 0:
            warmup_init_lr = 0 if warmup_updates > 0 else lr
        self.warmup_init_lr = warmup_init_lr
        self.lr_step = (lr - warmup_init_lr) / warmup_updates if warmup_updates > 0 else 0.0
        self.warmup_end = warmup_updates <= 0
        optimizer.set_lr(lr if self.warmup_end else warmup_init_lr)

    def _better(self, a):
        if self.maximize:
            return a > self.best * (1.0 + self.threshold)
        return a < self.best * (1.0 - self.threshold)

    def step(self, epoch, val_loss=None):
        if epoch < self.start_epoch:
            self.last_epoch = epoch
            self.optimizer.set_lr(self.base_lr)
            return self.optimizer.get_lr()
        if val_loss is not None and self.warmup_end:
            self.last_epoch += 1
            if self._better(float(val_loss)):
                self.best, self.num_bad_epochs = float(val_loss), 0
            else:
                self.num_bad_epochs += 1
            if self.num_bad_epochs > self.patience:
                new = max(self.optimizer.get_lr() * self.factor, self.min_lr)
                if self.optimizer.get_lr() - new > 1e-8:
                    self.optimizer.set_lr(new)
                self.num_bad_epochs = 0
        else:
            self.last_epoch = epoch
        return self.optimizer.get_lr()

    def step_update(self, num_updates):
        if self.warmup_updates > 0:
            if num_updates <= self.warmup_updates:
                self.optimizer.set_lr(self.warmup_init_lr + num_updates * self.lr_step)
            elif not self.warmup_end:
                self.warmup_end = True
        return self.optimizer.get_lr()

    def state_dict(self):
        return {"best": self.best, "last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.best = sd["best"]
        self.last_epoch = sd.get("last_epoch", self.last_epoch)


@register_lr_scheduler("reduce_lr_on_plateau")
class ReduceLROnPlateauLRSchedule(ReduceLROnPlateauLRScheduleV2):
    """fairseq/optim/lr_scheduler/reduce_lr_on_plateau.py:60-143 (the language-model recipe): the v2 schedule without a start
    epoch or a learning-rate floor."""

    def __init__(self, optimizer, lr=1e-3, lr_shrink=0.1, lr_threshold=1e-4, lr_patience=0, warmup_updates=0, warmup_init_lr=-1,
                 maximize_best_checkpoint_metric=False):
        super().__init__(optimizer, lr=lr, lr_shrink=lr_shrink, lr_threshold=lr_threshold, lr_patience=lr_patience,
                         warmup_updates=warmup_updates, warmup_init_lr=warmup_init_lr, start_reduce_lr_epoch=0, final_lr_scale=0.0,
                         maximize_best_checkpoint_metric=maximize_best_checkpoint_metric)
