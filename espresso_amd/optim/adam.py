"""Adam on the flat buffers with fused gradient scaling/clipping — results equal to
fairseq/optim/adam.py:215-240 applied after Trainer's multiply_grads + clip_grad_norm
(fairseq/trainer.py:918-951, fairseq/utils.py:347-397)."""
import torch

from .. import kernels as K
from ..registry import register_optimizer
from .flat import FlatParams


@register_optimizer("adam")
class FlatAdam:
    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        dev = flat.p32.device
        self.exp_avg = torch.zeros_like(flat.p32)
        self.exp_avg_sq = torch.zeros_like(flat.p32)
        self.step_count = 0
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._coef = torch.zeros(2, dtype=torch.float32, device=dev)

    def set_lr(self, lr):
        self.lr = lr

    def get_lr(self):
        return self.lr

    def clip_and_step(self, pre_scale=1.0, max_norm=0.0, denom_dev=None):
        """grad *= pre_scale/denom ; grad *= min(1, max_norm/(|grad|+1e-6)) ; Adam ; grads re-zeroed.
        Returns the device tensor coef = [applied scale, grad norm] (read it lazily for logging)."""
        f = self.flat
        self._sumsq.zero_()
        K.grad_sumsq(f.g32, self._sumsq)
        K.clip_coef(self._sumsq, pre_scale, max_norm, self._coef, denom_dev)
        self.step_count += 1
        K.adam_step(f.p32, f.g32, self.exp_avg, self.exp_avg_sq, f.p16, self._coef, self.lr, self.betas[0], self.betas[1],
                    self.eps, self.weight_decay, self.step_count, zero_grad=True)
        return self._coef

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr}

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr = sd.get("lr", self.lr)
