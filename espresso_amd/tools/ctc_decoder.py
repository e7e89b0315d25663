"""CTCDecoder — greedy CTC decoding with the generator API of espresso/tools/ctc_decoder.py:18-188
(`decode` for validation WER, `generate` for speech_recognize.py).  The per-utterance Python loop of the
reference (`_generate_one_example`: max over V, unique_consecutive, drop blank) is ONE batched HIP launch
pair (ea_ctc_greedy_decode).  The Flashlight/KenLM lexicon path (`lm_model`, external C++ library) is out
of scope."""
from typing import Dict, List

import torch

from .. import kernels as K


class CTCDecoder:
    def __init__(self, models, dictionary, blank=None, print_alignment=False, **kwargs):
        self.model = models[0] if isinstance(models, (list, tuple)) else models
        self.pad = dictionary.pad()
        self.blank = dictionary.bos() if blank is None else blank
        self.vocab_size = len(dictionary)
        self.print_alignment = print_alignment

    def cuda(self):
        self.model.cuda()
        return self

    @torch.no_grad()
    def _generate(self, sample):
        net_input = sample["net_input"]
        net_output = self.model(**net_input)
        lprobs = self.model.get_normalized_probs(net_output, log_probs=True)  # T x B x V view of [B][T][V]
        Tp, B, V = lprobs.shape
        enc_len = net_output["src_lengths"][0].to(torch.int32).contiguous()
        flat = lprobs.transpose(0, 1).reshape(B * Tp, V)
        tokens, out_len, score, align = K.ctc_greedy_decode(flat, enc_len, B, Tp, V, self.blank, self.pad)
        return tokens, out_len, score, align

    @torch.no_grad()
    def decode(self, models, sample, **kwargs):
        """(tokens B x U padded with pad, scores B, None) — the validation-time API (ctc_decoder.py:79-100)."""
        tokens, out_len, score, _ = self._generate(sample)
        U = max(1, int(out_len.max()))
        return tokens[:, :U].to(torch.long), score, None

    @torch.no_grad()
    def generate(self, models, sample, **kwargs) -> List[List[Dict[str, torch.Tensor]]]:
        tokens, out_len, score, align = self._generate(sample)
        tokens, out_len, score, align = tokens.cpu(), out_len.cpu(), score.cpu(), align.cpu()
        out = []
        for b in range(tokens.shape[0]):
            n = int(out_len[b])
            out.append([{
                "tokens": tokens[b, :n].to(torch.long),
                "score": score[b],
                "attention": None,
                "alignment": align[b, :n].to(torch.long) if self.print_alignment else None,
            }])
        return out
