"""`transducer_loss` criterion — interface and bookkeeping of espresso/criterions/transducer_loss.py:44-192: targets are
`target[:, :-1]` (EOS stripped; the whole target with `task.include_eos_in_transducer_loss`) as int32, blank = "<s>" index, per-utterance loss from the RNN-T kernel, sum reduction,
sentence_avg sample size.  The arithmetic is csrc/rnnt.hip (the reference calls torchaudio.functional.rnnt_loss)."""
import math

import torch

from .. import functional as F
from ..registry import register_criterion


@register_criterion("transducer_loss")
class TransducerLossCriterion:
    def __init__(self, task, sentence_avg=True, print_training_sample_interval=500):
        self.task = task
        d = task.target_dictionary
        if hasattr(task, "blank_symbol"):
            if task.blank_symbol is None:  # a task set up for another criterion: its dictionary has no "<s>" to use as blank
                raise ValueError(f"{type(self).__name__} needs task.blank_symbol (task.criterion_name must name this criterion)")
            self.blank_idx = d.index(task.blank_symbol)
        else:
            self.blank_idx = d.bos()
        self.pad_idx, self.eos_idx = d.pad(), d.eos()
        self.sentence_avg = sentence_avg
        self.include_eos = bool(getattr(getattr(task, "cfg", None), "include_eos_in_transducer_loss", False))  # II("task.…") :40

    def __call__(self, model, sample, reduce=True):
        return self.forward(model, sample, reduce)

    def forward(self, model, sample, reduce=True):
        # (lazy_joint: the model returns the joint network's branches + output layer instead of the (B, T', U+1, V) logits, and the
        # loss below runs the output layer itself, fused with the log-sum-exp / gradient — the logits never reach HBM)
        inner = getattr(model, "module", model)  # (the data-parallel wrapper passes keyword arguments through)
        kw = {"lazy_joint": True} if getattr(inner, "supports_lazy_joint", False) else {}
        net_output, encoder_out_lengths = model(**sample["net_input"], **kw)
        if self.include_eos:
            target = sample["target"]
            target_lengths = sample["target"].ne(self.pad_idx).sum(-1)
        else:
            target = sample["target"][:, :-1].contiguous() if sample["target"].size(1) > 1 else sample["target"]
            target_lengths = (sample["target"].ne(self.pad_idx) & sample["target"].ne(self.eos_idx)).sum(-1)
        loss_fn = F.joint_rnnt_loss if isinstance(net_output, F.LazyJointLogits) else F.rnnt_loss
        loss = loss_fn(net_output, target.to(torch.int32).contiguous(), encoder_out_lengths.to(torch.int32).contiguous(),
                       target_lengths.to(torch.int32).contiguous(), blank=self.blank_idx).sum()
        nsentences = sample["target"].size(0)
        sample_size = nsentences if self.sentence_avg else sample["ntokens"]
        return loss, sample_size, {"loss": loss.detach(), "ntokens": sample["ntokens"], "nsentences": nsentences,
                                   "sample_size": sample_size}

    @staticmethod
    def reduce_metrics(logging_outputs):
        loss_sum = float(sum(float(l.get("loss", 0)) for l in logging_outputs))
        ntokens = sum(l.get("ntokens", 0) for l in logging_outputs)
        sample_size = sum(l.get("sample_size", 0) for l in logging_outputs)
        out = {"loss": loss_sum / max(sample_size, 1) / math.log(2), "ntokens": ntokens, "sample_size": sample_size}
        if sample_size != ntokens:
            out["nll_loss"] = loss_sum / max(ntokens, 1) / math.log(2)
        return out

    @staticmethod
    def logging_outputs_can_be_summed():
        return True
