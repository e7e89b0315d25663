"""SimpleGreedyDecoder — teacher-free arg-max decoding used for validation WER of attention models, with the semantics of
espresso/tools/simple_greedy_decoder.py:75-166: decode up to max(T', target length) steps (validation) or
max_len_a*src_len+max_len_b steps, stop when every hypothesis has produced EOS, force EOS / uniform log-probs after the first
EOS of a row, and return (tokens without the leading EOS, per-step log-probs for the target positions, None).
Model compute per step = decoder.step() on the HIP incremental path; arg-max = the CTC decoder's row arg-max kernel."""
import math

import torch

from .. import kernels as K


class SimpleGreedyDecoder:
    def __init__(self, models, dictionary, max_len_a=0, max_len_b=200, max_len=0, temperature=1.0, eos=None,
                 for_validation=True, **unused):
        # an ensemble decodes with the log of the mean member probability (the reference wraps the list in fairseq's EnsembleModel:
        # espresso/tools/simple_greedy_decoder.py:47-52, fairseq/sequence_generator.py:837-939)
        self.models = list(models) if isinstance(models, (list, tuple)) else [models]
        self.model = self.models[0]
        self.pad, self.unk = dictionary.pad(), dictionary.unk()
        self.eos = dictionary.eos() if eos is None else eos
        self.vocab_size = len(dictionary)
        self.max_len_a, self.max_len_b = max_len_a, max_len_b
        self.max_len = max_len or self.model.max_decoder_positions()
        assert temperature == 1.0
        self.for_validation = for_validation

    @torch.no_grad()
    def decode(self, models, sample, bos_token=None, **kwargs):
        net_input = sample["net_input"]
        src_tokens = net_input["src_tokens"]
        bsz, src_len = src_tokens.shape[:2]
        dev = src_tokens.device
        encoder_outs = [m.forward_encoder(net_input["src_tokens"], net_input["src_lengths"]) for m in self.models]
        encoder_out = encoder_outs[0]
        target = sample.get("target")
        assert target is not None or not self.for_validation
        max_enc = encoder_out["encoder_padding_mask"][0].shape[1]
        max_len = (max(max_enc, target.size(1)) if self.for_validation
                   else min(int(self.max_len_a * src_len + self.max_len_b), self.max_len - 1))
        max_len = min(max_len, min(m.max_decoder_positions() for m in self.models) - 1)
        tokens = torch.full((bsz, max_len + 2), self.pad, dtype=torch.long, device=dev)
        tokens[:, 0] = self.eos if bos_token is None else bos_token
        lprobs = (torch.full((bsz, target.size(1), self.vocab_size), -math.log(self.vocab_size), device=dev)
                  if self.for_validation else None)
        states = [m.decoder.init_incremental(eo, bsz, 1) for m, eo in zip(self.models, encoder_outs)]
        ident = torch.arange(bsz, device=dev)
        ones = torch.ones(bsz, dtype=torch.int32, device=dev)
        for step in range(max_len + 1):
            is_eos = tokens[:, step].eq(self.eos)
            if step > 0 and bool(is_eos.all()):
                tokens = tokens[:, : step + 1]
                break
            parent = None if step == 0 else ident
            lp = self.models[0].decoder.step(states[0], tokens[:, : step + 1], step, parent)  # fp32 [B][V]
            if len(self.models) > 1:
                every = [lp] + [m.decoder.step(st, tokens[:, : step + 1], step, parent) for m, st in zip(self.models[1:], states[1:])]
                lp = torch.logsumexp(torch.stack(every, 0), dim=0) - math.log(len(every))
            best, _, _, _ = K.ctc_greedy_decode(lp, ones, bsz, 1, self.vocab_size, blank=-1, pad=self.pad, want_align=False)
            tokens[:, step + 1] = best[:, 0].long()
            if step > 0:
                lp[is_eos, :] = -math.log(self.vocab_size)
                tokens[is_eos, step + 1] = self.eos
            if self.for_validation and step < target.size(1):
                lprobs[:, step, :] = lp
        return tokens[:, 1:], lprobs, None
