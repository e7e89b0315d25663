"""TEST INFRASTRUCTURE ONLY — torch restatement of the per-step logit rules (fairseq/sequence_generator.py:395-424) and of
BeamSearch.step (fairseq/search.py:103-144), plus the scripted decoder of the reference's own known-answer test
(tests/utils.py:67-165, 556-600), adapted to espresso_amd's decoder protocol.  Used to run the SequenceGenerator host logic on
the CPU and as the checker of the HIP beam kernels."""
import math

import torch


class TorchRefSearch:
    def mask(self, lprobs, pad, unk, eos, unk_penalty, only_eos, forbid_eos, eos_factor):
        lprobs = lprobs.clone()
        lprobs[lprobs != lprobs] = -math.inf
        lprobs[:, pad] = -math.inf
        lprobs[:, unk] -= unk_penalty
        if only_eos:
            lprobs[:, :eos] = -math.inf
            lprobs[:, eos + 1:] = -math.inf
        elif eos_factor is not None:
            dis = lprobs[:, eos] < eos_factor * lprobs.max(dim=1)[0]
            lprobs[dis, eos] = -math.inf
        if forbid_eos:
            lprobs[:, eos] = -math.inf
        return lprobs

    def step(self, step, lprobs, prev_scores, bsz, beam):
        V = lprobs.shape[1]
        lp = lprobs.view(bsz, beam, V)
        if step == 0:
            lp = lp[:, ::beam, :].contiguous()
        else:
            lp = lp + prev_scores.view(bsz, beam, 1)
        k = min(beam * 2, lp.view(bsz, -1).size(1) - 1)
        sc, idx = torch.topk(lp.view(bsz, -1), k=k)
        return sc, idx.fmod(V), torch.div(idx, V, rounding_mode="trunc")


class ScriptedDecoder:
    """Emits the probabilities scripted in tests/utils.py:81-160 regardless of the hypotheses (row = beam slot)."""

    def __init__(self, beam_probs, vocab, eos):
        self.beam_probs, self.vocab, self.eos = beam_probs, vocab, eos

    def init_incremental(self, encoder_out, bsz, beam):
        return {}

    def step(self, st, tokens, step, parent):
        N = tokens.shape[0]
        probs = torch.zeros(N, self.vocab)
        if step < len(self.beam_probs):
            probs[:, self.eos:] = self.beam_probs[step][:N]
        else:
            probs[:, self.eos] = 1.0
        return probs.log().to(tokens.device)


class ScriptedModel:
    def __init__(self, decoder):
        self.decoder = decoder

    def forward_encoder(self, src_tokens, src_lengths):
        return {"encoder_out": [src_tokens]}

    def max_decoder_positions(self):
        return 100


class DummyDict:
    """tests/utils.py:34 dummy_dictionary(vocab_size=2): <s>=0 <pad>=1 </s>=2 <unk>=3, words 4,5."""

    def __init__(self, n=6):
        self.n = n

    def __len__(self):
        return self.n

    def pad(self):
        return 1

    def eos(self):
        return 2

    def unk(self):
        return 3


def scripted_setup():
    unk = 0.0
    T = torch.FloatTensor
    beam_probs = [
        T([[0.0, unk, 0.9, 0.1], [0.0, unk, 0.9, 0.1], [0.0, unk, 0.7, 0.3], [0.0, unk, 0.7, 0.3]]),
        T([[1.0, unk, 0.0, 0.0], [0.0, unk, 0.9, 0.1], [0.25, unk, 0.35, 0.4], [0.00, unk, 0.10, 0.9]]),
        T([[0.0, unk, 0.1, 0.9], [0.6, unk, 0.2, 0.2], [0.60, unk, 0.4, 0.00], [0.01, unk, 0.0, 0.99]]),
        T([[1.0, unk, 0.0, 0.0], [1.0, unk, 0.0, 0.0], [0.1, unk, 0.5, 0.4], [1.0, unk, 0.0, 0.0]]),
    ]
    d = DummyDict()
    model = ScriptedModel(ScriptedDecoder(beam_probs, len(d), d.eos()))
    src_tokens = torch.LongTensor([[4, 5, 2], [4, 5, 2]])
    src_lengths = torch.LongTensor([2, 2])
    return d, 4, 5, {"net_input": {"src_tokens": src_tokens, "src_lengths": src_lengths}}, model
