"""`cross_entropy_v2` — espresso/criterions/cross_entropy_v2.py:33-105: token-level NLL over the non-pad targets with
the model called as `model(**net_input, epoch=epoch)` (scheduled sampling is epoch-driven) and a randomly sampled
REF/PRD pair logged every `print_training_sample_interval` updates.  The loss/gradient run in the fused HIP kernel of
`label_smoothed_cross_entropy_v2` with ε = 0 (log-sum-exp + target gather + gradient in one pass, pad rows zeroed)."""
import logging
import math

import numpy as np
import torch

from .. import functional as F
from ..data.data_utils import numpy_seed
from ..registry import register_criterion
from . import forward_accepts_epoch

logger = logging.getLogger(__name__)


@register_criterion("cross_entropy_v2")
class CrossEntropyV2Criterion:
    def __init__(self, task, sentence_avg=True, print_training_sample_interval=500, **unused):
        self.task = task
        self.sentence_avg = sentence_avg
        self.dictionary = task.target_dictionary
        self.padding_idx = self.dictionary.pad()
        self.print_interval = print_training_sample_interval
        self.epoch = 1
        self.prev_num_updates = -1

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
        loss, _ = F.label_smoothed_ce(logits, target, self.padding_idx, 0.0, "uniform", None, sample["target"].shape[1])
        sample_size = sample["target"].size(0) if self.sentence_avg else sample["ntokens"]
        logging_output = {"loss": loss.detach(), "ntokens": sample["ntokens"], "nsentences": sample["target"].size(0),
                          "sample_size": sample_size}
        n = getattr(model, "num_updates", None)
        if (n is not None and getattr(model, "training", False) and n // self.print_interval > (n - 1) // self.print_interval
                and n != self.prev_num_updates):
            self.prev_num_updates = n
            self._log_sample(sample, logits3, n)
        return loss, sample_size, logging_output

    def _log_sample(self, sample, logits3, num_updates):
        """One randomly chosen utterance of the batch, greedy token predictions vs the reference text (:68-88)."""
        target = sample["target"]
        pred = logits3.detach().argmax(-1).cpu()
        with numpy_seed(num_updates):
            i = np.random.randint(0, len(sample["id"]))
        length = int(target[i].ne(self.padding_idx).sum())
        ref_one = sample["text"][i] if sample.get("text") is not None else self.dictionary.string(target[i].cpu())
        pred_one = self.dictionary.wordpiece_decode(self.dictionary.string(
            pred[i][:length], extra_symbols_to_ignore=getattr(self.task, "extra_symbols_to_ignore", None)))
        logger.info("sample REF: " + ref_one)
        logger.info("sample PRD: " + pred_one)

    @staticmethod
    def reduce_metrics(logging_outputs):
        loss_sum = float(sum(float(l.get("loss", 0)) for l in logging_outputs))
        ntokens = sum(l.get("ntokens", 0) for l in logging_outputs)
        sample_size = sum(l.get("sample_size", 0) for l in logging_outputs)
        out = {"loss": loss_sum / max(sample_size, 1) / math.log(2), "ntokens": ntokens, "sample_size": sample_size}
        nll = loss_sum / max(ntokens, 1) / math.log(2) if sample_size != ntokens else out["loss"]
        out["nll_loss"], out["ppl"] = nll, 2 ** nll
        return out

    @staticmethod
    def logging_outputs_can_be_summed():
        return True
