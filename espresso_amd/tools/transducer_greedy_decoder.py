"""Greedy transducer search — espresso/tools/transducer_greedy_decoder.py:92-251 (+ the option handling of
transducer_base_decoder.py:17-110).  For every encoder frame up to `max_num_expansions_per_step` non-blank symbols are
emitted greedily; a row that emits blank (or has run out of frames) keeps its predictor state
(`masked_copy_cached_state`).  Returns (tokens [B][T*(E+1)], summed log-prob scores [B], alignments or None) exactly
like the reference; `generate()` wraps them in fairseq's hypothesis dicts.

Compute: encoder, predictor step (LSTM cell kernels), joint step (add-relu kernel + MFMA GEMM) and log-softmax run on the
HIP kernels; the per-expansion bookkeeping is a handful of tiny tensor ops on the device, with the one host
synchronisation per expansion that the reference has too (`blank_mask.all()`)."""
from typing import Optional

import torch

from .. import kernels as K


class TransducerGreedyDecoder:
    def __init__(self, models, dictionary, max_len=0, max_num_expansions_per_step=2, temperature=1.0, eos=None, bos=None,
                 blank=None, model_predicts_eos=False, symbols_to_strip_from_output=None, lm_model=None, lm_weight=1.0,
                 print_alignment=False, **kwargs):
        self.model = models[0]
        self.eos = dictionary.eos() if eos is None else eos
        self.bos = dictionary.eos() if bos is None else bos
        self.blank = dictionary.bos() if blank is None else blank
        self.model_predicts_eos = model_predicts_eos
        strip = {self.eos, self.bos, self.blank}
        self.symbols_to_strip_from_output = strip.union(symbols_to_strip_from_output) if symbols_to_strip_from_output else strip
        self.vocab_size = len(dictionary)
        self.beam_size = 1
        self.max_len = max_len
        assert max_num_expansions_per_step > 0, "--max-num-expansions-per-step must be at least 1"
        self.max_num_expansions_per_step = max_num_expansions_per_step
        assert temperature > 0, "--temperature must be greater than 0"
        self.temperature = temperature
        self.print_alignment = print_alignment
        self.model.eval()
        self.lm_model, self.lm_weight = lm_model, lm_weight
        if lm_model is not None:
            nlm = len(lm_model.decoder.dictionary)
            assert nlm in (self.vocab_size, self.vocab_size - 1)
            self.no_blank_in_lm = nlm == self.vocab_size - 1
            lm_model.eval()

    @torch.no_grad()
    def decode(self, models, sample, **kwargs):
        return self._generate(sample, **kwargs)

    @torch.no_grad()
    def generate(self, models, sample, **kwargs):
        tokens, scores, alignments = self._generate(sample, bos_token=kwargs.get("bos_token", None))
        return [[{"tokens": tokens[i], "score": scores[i], "attention": None,
                  "alignment": alignments[i] if (self.print_alignment and alignments is not None) else None}]
                for i in range(tokens.size(0))]

    @torch.no_grad()
    def _generate(self, sample, bos_token: Optional[int] = None):
        net_input = sample["net_input"]
        model = self.model
        enc = model.encoder(net_input["src_tokens"], net_input["src_lengths"])
        x = enc["_x_bt"][0]
        enc_len = enc["src_lengths"][0]
        bsz = enc_len.shape[0]
        Tp = x.shape[0] // bsz
        dev = x.device
        max_enc = int(enc_len.max())
        max_len = min(max_enc, self.max_len) if self.max_len > 0 else max_enc
        Ex = self.max_num_expansions_per_step
        E = model.joint_encoder_branch(x).view(bsz, Tp, -1)
        tokens = torch.full((bsz, max_len, Ex + 1), self.blank, dtype=torch.long, device=dev)
        scores = torch.zeros((bsz, max_len, Ex + 1), dtype=torch.float32, device=dev)
        prev = torch.full((bsz,), self.bos if bos_token is None else bos_token, dtype=torch.long, device=dev)
        state = model.decoder.init_state(bsz, dev)
        lm_state = self.lm_model.decoder.init_state(bsz, dev) if self.lm_model is not None else None
        nonblank = None
        if self.lm_model is not None:
            nonblank = torch.ones(self.vocab_size, dtype=torch.bool, device=dev)
            nonblank[self.blank] = False
        V = self.vocab_size
        for step in range(max_len):
            blank_mask = step >= enc_len  # B
            k = 0
            while not bool(blank_mask.all()) and k < Ex + 1:
                dec_out, new_state = model.decoder.advance(prev, state)
                logits = model.joint_step(E[:, step].contiguous(), dec_out)[:, :V]
                if self.temperature != 1.0:
                    logits = logits / self.temperature
                lprobs = K.log_softmax(logits, bsz, V, logits.stride(0))
                if self.lm_model is not None:
                    lm_prev = torch.where(prev > self.blank, prev - 1, prev) if self.no_blank_in_lm else prev
                    lm_feat, new_lm_state = self.lm_model.decoder.advance(lm_prev, lm_state)
                    lm_logits = self.lm_model.decoder.output_layer(lm_feat)
                    lm_lprobs = K.log_softmax(lm_logits, bsz, lm_logits.shape[1], lm_logits.stride(0))
                    nb = lprobs[:, nonblank]
                    if not self.no_blank_in_lm:
                        lm_lprobs = lm_lprobs[:, nonblank]
                    fused = nb + self.lm_weight * lm_lprobs
                    # keep the non-blank probability mass unchanged after adding the LM score (:201-210)
                    fused = fused + (nb.exp().sum(1).log() - fused.exp().sum(1).log()).unsqueeze(1)
                    lprobs[:, nonblank] = fused
                if self.model_predicts_eos:
                    lprobs[:, self.blank] = torch.logaddexp(lprobs[:, self.blank], lprobs[:, self.eos])
                    lprobs[:, self.eos] = float("-inf")
                if k < Ex:
                    sc, tk = lprobs.max(-1)
                    sc = sc.masked_fill(blank_mask, 0.0)
                    blank_mask = blank_mask | (tk == self.blank)
                    tk = tk.masked_fill(blank_mask, self.blank)
                    scores[:, step, k] = sc
                    tokens[:, step, k] = tk
                    prev = torch.where(blank_mask, prev, tk)
                else:
                    # add the score of the closing blank if the frame has not emitted one yet
                    scores[:, step, k] = torch.where(blank_mask, scores[:, step, k], lprobs[:, self.blank])
                    blank_mask = torch.ones_like(blank_mask)
                # rows that emitted blank keep the predictor (and LM) state they had before this expansion
                keep = blank_mask.unsqueeze(1)
                state = {n: [torch.where(keep, o, nw) for o, nw in zip(state[n], new_state[n])] for n in state}
                if self.lm_model is not None:
                    lm_state = {n: [torch.where(keep, o, nw) for o, nw in zip(lm_state[n], new_lm_state[n])] for n in lm_state}
                k += 1
        alignments = tokens if self.print_alignment else None
        return tokens.view(bsz, -1), scores.view(bsz, -1).sum(-1), alignments
