"""`label_smoothed_cross_entropy_v2` — interface and bookkeeping of
espresso/criterions/label_smoothed_cross_entropy_v2.py:158-243 on the fused HIP kernel
(log-sum-exp + target gather + smoothing term + gradient in one pass per row; pad rows zeroed).
Smoothing types (:49-119): `uniform`, `unigram` (dictionary counts + pseudo count, :152-155) and `temporal`
(neighbouring targets 5:2, the WSJ recipe) are all evaluated inside the kernel."""
import math

import torch

from .. import functional as F
from ..registry import register_criterion
from . import forward_accepts_epoch


@register_criterion("label_smoothed_cross_entropy_v2")
class LabelSmoothedCrossEntropyV2Criterion:
    def __init__(self, task, sentence_avg=True, label_smoothing=0.1, smoothing_type="uniform", unigram_pseudo_count=1.0, **unused):
        if smoothing_type not in ("uniform", "unigram", "temporal"):
            raise ValueError("Unsupported smoothing type: {}".format(smoothing_type))
        self.smoothing_type = smoothing_type
        self.unigram_tensor = None
        if smoothing_type == "unigram":
            u = torch.tensor(task.target_dictionary.count, dtype=torch.float32) + unigram_pseudo_count
            self.unigram_tensor = u / u.sum()
        self.task = task
        self.sentence_avg = sentence_avg
        self.eps = label_smoothing
        self.padding_idx = task.target_dictionary.pad()
        self.epoch = 1

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __call__(self, model, sample, reduce=True):
        return self.forward(model, sample, reduce)

    def forward(self, model, sample, reduce=True):
        if forward_accepts_epoch(model):  # scheduled sampling is epoch-driven (label_smoothed_cross_entropy_v2.py:173)
            net_output = model(**sample["net_input"], epoch=self.epoch)
        else:
            net_output = model(**sample["net_input"])
        logits3 = net_output[0]
        logits = net_output[1].get("_logits_bu") if isinstance(net_output[1], dict) else None
        if logits is None:
            logits = logits3.reshape(-1, logits3.shape[-1])
        target = sample["target"].reshape(-1).to(torch.int32).contiguous()
        prior = None
        if self.unigram_tensor is not None:
            if self.unigram_tensor.device != logits.device:
                self.unigram_tensor = self.unigram_tensor.to(logits.device)
            prior = self.unigram_tensor
        loss, nll = F.label_smoothed_ce(logits, target, self.padding_idx, self.eps, self.smoothing_type, prior,
                                        sample["target"].shape[1])
        sample_size = sample["target"].size(0) if self.sentence_avg else sample["ntokens"]
        logging_output = {"loss": loss.detach(), "nll_loss": nll.detach(), "ntokens": sample["ntokens"],
                          "nsentences": sample["target"].size(0), "sample_size": sample_size}
        return loss, sample_size, logging_output

    @staticmethod
    def reduce_metrics(logging_outputs):
        loss_sum = float(sum(float(l.get("loss", 0)) for l in logging_outputs))
        nll_sum = float(sum(float(l.get("nll_loss", 0)) for l in logging_outputs))
        ntokens = sum(l.get("ntokens", 0) for l in logging_outputs)
        sample_size = sum(l.get("sample_size", 0) for l in logging_outputs)
        return {"loss": loss_sum / max(sample_size, 1) / math.log(2), "nll_loss": nll_sum / max(ntokens, 1) / math.log(2),
                "ppl": 2 ** (nll_sum / max(ntokens, 1) / math.log(2)), "ntokens": ntokens, "sample_size": sample_size}

    @staticmethod
    def logging_outputs_can_be_summed():
        return True
