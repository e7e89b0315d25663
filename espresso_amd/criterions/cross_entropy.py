"""`cross_entropy` — fairseq/criterions/cross_entropy.py:26-99 (the language-model recipe's criterion): summed NLL over the
non-pad targets, sample size = tokens (sentences with `sentence_avg`), `loss` / `ppl` in base 2.  The arithmetic is the fused
HIP label-smoothing kernel with ε = 0, as for `cross_entropy_v2`."""
import torch

from .. import functional as F
from ..registry import register_criterion
from .cross_entropy_v2 import CrossEntropyV2Criterion


@register_criterion("cross_entropy")
class CrossEntropyCriterion(CrossEntropyV2Criterion):
    def __init__(self, task, sentence_avg=False, **unused):
        super().__init__(task, sentence_avg=sentence_avg, print_training_sample_interval=1 << 62)

    def forward(self, model, sample, reduce=True):
        logits3 = model(**sample["net_input"])[0]
        logits = logits3.reshape(-1, logits3.shape[-1])
        target = sample["target"].reshape(-1).to(torch.int32).contiguous()
        loss, _ = F.label_smoothed_ce(logits, target, self.padding_idx, 0.0, "uniform", None, sample["target"].shape[1])
        sample_size = sample["target"].size(0) if self.sentence_avg else sample["ntokens"]
        return loss, sample_size, {"loss": loss.detach(), "ntokens": sample["ntokens"], "nsentences": sample["target"].size(0),
                                   "sample_size": sample_size}
