"""`ctc_loss` criterion — interface and bookkeeping of espresso/criterions/ctc_loss.py:40-169
(blank = "<s>" index, targets with pad/eos removed, reduction sum, zero_infinity, sentence_avg
sample size, same logging keys); the arithmetic is the HIP CTC path (functional.ctc_loss)."""
import math

import torch

from .. import functional as F
from ..registry import register_criterion


@register_criterion("ctc_loss")
class CtcLossCriterion:
    def __init__(self, task, sentence_avg=True, zero_infinity=True, print_training_sample_interval=500):
        self.task = task
        d = task.target_dictionary
        if hasattr(task, "blank_symbol"):
            if task.blank_symbol is None:  # a task set up for another criterion: its dictionary has no "<s>" to use as blank
                raise ValueError(f"{type(self).__name__} needs task.blank_symbol (task.criterion_name must name this criterion)")
            self.blank_idx = d.index(task.blank_symbol)
        else:
            self.blank_idx = d.bos()
        self.pad_idx = d.pad()
        self.eos_idx = d.eos()
        self.sentence_avg = sentence_avg
        self.zero_infinity = zero_infinity
        self.print_interval = print_training_sample_interval
        self.num_updates = -1
        self.epoch = 0

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __call__(self, model, sample, reduce=True):
        return self.forward(model, sample, reduce)

    def forward(self, model, sample, reduce=True):
        net_output = model(**sample["net_input"], **({"epoch": self.epoch} if False else {}))
        input_lengths = net_output["src_lengths"][0]
        target = sample["target"]
        pad_mask = (target != self.pad_idx) & (target != self.eos_idx)
        target_lengths = pad_mask.sum(-1)
        lg = net_output.get("_logits_bt")
        if lg:
            logits = lg[0]
            B, Tp = net_output["encoder_padding_mask"][0].shape
        else:
            e = net_output["encoder_out"][0]
            Tp, B, V = e.shape
            logits = e.transpose(0, 1).reshape(B * Tp, V)
        # left-aligned labels per row (pad/eos only trail the labels in AsrDataset.collater output)
        nll, lprobs = F.ctc_loss(logits, target.to(torch.int32).contiguous(), input_lengths.to(torch.int32).contiguous(),
                                 target_lengths.to(torch.int32).contiguous(), B, Tp, blank=self.blank_idx,
                                 zero_infinity=self.zero_infinity)
        if self.zero_infinity:
            nll = torch.where(torch.isinf(nll), torch.zeros_like(nll), nll)
        loss = nll.sum()
        ntokens = sample["ntokens"] if "ntokens" in sample else int(target_lengths.sum())
        nsentences = target.size(0)
        sample_size = nsentences if self.sentence_avg else ntokens
        logging_output = {
            "loss": loss.detach(),
            "ntokens": ntokens,
            "nsentences": nsentences,
            "sample_size": sample_size,
        }
        return loss, sample_size, logging_output

    @staticmethod
    def reduce_metrics(logging_outputs):
        """Aggregate like espresso/criterions/ctc_loss.py:133-160: loss in base 2 per sample_size."""
        loss_sum = float(sum(float(l.get("loss", 0)) for l in logging_outputs))
        ntokens = sum(l.get("ntokens", 0) for l in logging_outputs)
        nsentences = sum(l.get("nsentences", 0) for l in logging_outputs)
        sample_size = sum(l.get("sample_size", 0) for l in logging_outputs)
        out = {"loss": loss_sum / max(sample_size, 1) / math.log(2), "ntokens": ntokens, "nsentences": nsentences,
               "sample_size": sample_size}
        if sample_size != ntokens:
            out["nll_loss"] = loss_sum / max(ntokens, 1) / math.log(2)
        return out

    @staticmethod
    def logging_outputs_can_be_summed():
        return True
