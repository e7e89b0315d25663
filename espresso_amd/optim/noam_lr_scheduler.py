"""`noam` schedule — espresso/optim/lr_scheduler/noam_lr_scheduler.py:32-78:
lr = lr0 * d^-0.5 * min(n^-0.5, n * warmup^-1.5), n = num_updates + 1, floored at final_lr after warm-up."""
from ..registry import register_lr_scheduler


@register_lr_scheduler("noam")
class NoamSchedule:
    def __init__(self, optimizer, lr=5.0, warmup_steps=25000, model_size=512, final_lr=1e-6):
        self.optimizer = optimizer
        self.lr0, self.warmup_steps, self.model_size, self.final_lr = lr, warmup_steps, model_size, final_lr
        self.step_update(0)

    def step_update(self, num_updates):
        n = num_updates + 1
        lr = self.lr0 * self.model_size ** -0.5 * min(n ** -0.5, n * self.warmup_steps ** -1.5)
        if n > self.warmup_steps:
            lr = max(lr, self.final_lr)
        self.optimizer.set_lr(lr)
        return lr
