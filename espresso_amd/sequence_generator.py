"""Batched beam search with incremental decoder state, LM fusion hook and eos_factor — the semantics of
fairseq/sequence_generator.py:212-620 (`_generate`), :657-766 (`finalize_hypos`), :768-785 (`is_finished`) as patched by
the reference (lm_model / lm_weight / eos_factor, :40-41,116-137,385-410) and fairseq/search.py:103-144 (BeamSearch.step).

What runs where on the MI355X path:
  * model compute per step  -> decoder.step(): HIP kernels (one-query attention against K/V caches, GEMMs, log-softmax)
  * cache reorder           -> fused with the K/V append (ea_kv_append_reorder), encoder K/V are never copied: each
                               hypothesis carries the row of its sentence (`kv_row`), the reference's beam-aware dedup
  * logit rules + top-k     -> ea_beam_mask_rows + ea_beam_topk
  * hypothesis bookkeeping  -> this file (tiny integer tensors; same order of operations as the reference so that
                               finalisation order, score normalisation and tie handling are identical)

Decoder protocol (implemented by models.transformer.speech_transformer_base.SpeechTransformerDecoderBase and by the scripted
test decoder): `init_incremental(encoder_out, bsz, beam)` -> state; `step(state, tokens[:, :step+1], step, parent)` -> fp32
log-probs [N][V] for the N live hypotheses, where `parent[n]` is the previous-step hypothesis each one continues.
"""
import math
from typing import Dict, List, Optional

import torch

from . import kernels as K


class HipBeamSearch:
    """Logit rules + candidate selection on the GPU (ea_beam_mask_rows / ea_beam_topk)."""

    def mask(self, lprobs, pad, unk, eos, unk_penalty, only_eos, forbid_eos, eos_factor):
        return K.beam_mask_rows(lprobs, pad, unk, eos, unk_penalty, only_eos, forbid_eos, eos_factor)

    def step(self, step, lprobs, prev_scores, bsz, beam):
        """-> (cand_scores, cand_tokens, cand_beams) each [bsz][k], k = min(2*beam, V*beams_used - 1)."""
        V = lprobs.shape[1]
        used = 1 if step == 0 else beam
        k = min(2 * beam, V * used - 1)
        cs, ct, cb = K.beam_topk(lprobs, None if step == 0 else prev_scores, bsz, beam, used, k)
        return cs, ct.long(), cb.long()


class SequenceGenerator:
    def __init__(self, models, tgt_dict, beam_size=1, max_len_a=0, max_len_b=200, max_len=0, min_len=1, normalize_scores=True,
                 len_penalty=1.0, unk_penalty=0.0, temperature=1.0, match_source_len=False, lm_model=None, lm_weight=1.0,
                 eos_factor=None, eos=None, search=None, **unused):
        # an ensemble (`--path a:b`) decodes with the log of the MEAN probability of its members
        # (fairseq/sequence_generator.py:837-939 EnsembleModel.forward_decoder: logsumexp of the log-probabilities - log n)
        self.models = list(models) if isinstance(models, (list, tuple)) else [models]
        self.model = self.models[0]
        self.tgt_dict = tgt_dict
        self.pad, self.unk = tgt_dict.pad(), tgt_dict.unk()
        self.eos = tgt_dict.eos() if eos is None else eos
        self.vocab_size = len(tgt_dict)
        self.beam_size = min(beam_size, self.vocab_size - 1)
        self.max_len_a, self.max_len_b, self.min_len = max_len_a, max_len_b, min_len
        mp = min((m.max_decoder_positions() if hasattr(m, "max_decoder_positions") else 1024) for m in self.models)
        self.max_len = max_len or mp
        self.normalize_scores, self.len_penalty, self.unk_penalty = normalize_scores, len_penalty, unk_penalty
        self.temperature = temperature
        assert temperature == 1.0, "temperature != 1 is not used by the ASR recipes"
        self.match_source_len = match_source_len
        self.lm_model, self.lm_weight = lm_model, lm_weight
        self.eos_factor = eos_factor
        assert eos_factor is None or eos_factor >= 1.0, "--eos-factor must be >= 1.0 if set"
        self.search = search if search is not None else HipBeamSearch()

    @torch.no_grad()
    def generate(self, models, sample, **kwargs) -> List[List[Dict[str, torch.Tensor]]]:
        return self._generate(sample, **kwargs)

    def forward(self, sample, **kwargs):
        return self._generate(sample, **kwargs)

    @torch.no_grad()
    def _generate(self, sample, prefix_tokens=None, constraints=None, bos_token=None):
        assert prefix_tokens is None and constraints is None, "prefix / constrained decoding: not on the ASR path"
        net_input = sample["net_input"]
        src_tokens = net_input["src_tokens"]
        src_lengths = net_input["src_lengths"] if "src_lengths" in net_input else (
            (src_tokens.ne(self.eos) & src_tokens.ne(self.pad)).long().sum(dim=1))
        bsz, src_len = src_tokens.shape[:2]
        beam = self.beam_size
        dev = src_tokens.device
        if self.match_source_len:
            max_len = int(src_lengths.max())
        else:
            max_len = min(int(self.max_len_a * src_len + self.max_len_b), self.max_len - 1)
        assert self.min_len <= max_len, "min_len cannot be larger than max_len, please adjust these!"

        states = [m.decoder.init_incremental(m.forward_encoder(net_input["src_tokens"], net_input.get("src_lengths")), bsz, beam)
                  for m in self.models]
        lm_state = self.lm_model.init_incremental(bsz, beam) if self.lm_model is not None else None

        scores = torch.zeros(bsz * beam, max_len + 1, dtype=torch.float32, device=dev)
        tokens = torch.full((bsz * beam, max_len + 2), self.pad, dtype=torch.long, device=dev)
        tokens[:, 0] = self.eos if bos_token is None else bos_token
        cands_to_ignore = torch.zeros(bsz, beam, dtype=torch.bool, device=dev)
        finalized: List[List[Dict[str, torch.Tensor]]] = [[] for _ in range(bsz)]
        finished = [False] * bsz
        num_remaining = bsz
        cand_size = 2 * beam
        bbsz_offsets = (torch.arange(0, bsz, device=dev) * beam).unsqueeze(1)
        cand_offsets = torch.arange(0, cand_size, device=dev)
        parent: Optional[torch.Tensor] = None

        for step in range(max_len + 1):
            lprobs = self.models[0].decoder.step(states[0], tokens[:, : step + 1], step, parent)  # fp32 [N][V]
            if len(self.models) > 1:
                every = [lprobs] + [m.decoder.step(st, tokens[:, : step + 1], step, parent) for m, st in zip(self.models[1:], states[1:])]
                lprobs = torch.logsumexp(torch.stack(every, 0), dim=0) - math.log(len(every))
            if self.lm_model is not None:
                lm_lprobs = self.lm_model.step(lm_state, tokens[:, : step + 1], step, parent)
                lprobs = lprobs + self.lm_weight * lm_lprobs
            only_eos = step >= max_len
            forbid_eos = (not only_eos) and step < self.min_len
            lprobs = self.search.mask(lprobs.contiguous(), self.pad, self.unk, self.eos, self.unk_penalty, only_eos, forbid_eos,
                                      None if only_eos else self.eos_factor)
            prev = scores[:, step - 1].contiguous() if step > 0 else None
            cand_scores, cand_indices, cand_beams = self.search.step(step, lprobs, prev, bsz, beam)
            kc = cand_scores.shape[1]
            if kc < cand_size:  # tiny vocabularies: pad the candidate list (never selected: -inf)
                padn = cand_size - kc
                cand_scores = torch.cat([cand_scores, cand_scores.new_full((bsz, padn), -math.inf)], 1)
                cand_indices = torch.cat([cand_indices, cand_indices.new_full((bsz, padn), self.pad)], 1)
                cand_beams = torch.cat([cand_beams, cand_beams.new_zeros((bsz, padn))], 1)
            cand_bbsz_idx = cand_beams + bbsz_offsets

            eos_mask = cand_indices.eq(self.eos) & cand_scores.ne(-math.inf)
            eos_mask[:, :beam][cands_to_ignore] = False
            eos_bbsz_idx = torch.masked_select(cand_bbsz_idx[:, :beam], mask=eos_mask[:, :beam])
            finalized_sents: List[int] = []
            if eos_bbsz_idx.numel() > 0:
                eos_scores = torch.masked_select(cand_scores[:, :beam], mask=eos_mask[:, :beam])
                finalized_sents = self._finalize(step, eos_bbsz_idx, eos_scores, tokens, scores, finalized, finished, beam, max_len)
                num_remaining -= len(finalized_sents)
            assert num_remaining >= 0
            if num_remaining == 0:
                break
            assert step < max_len, f"{step} < {max_len}"

            batch_idxs = None
            if finalized_sents:
                new_bsz = bsz - len(finalized_sents)
                batch_mask = torch.ones(bsz, dtype=torch.bool, device=dev)
                batch_mask[finalized_sents] = False
                batch_idxs = torch.arange(bsz, device=dev).masked_select(batch_mask)
                eos_mask = eos_mask[batch_idxs]
                cand_beams = cand_beams[batch_idxs]
                bbsz_offsets = bbsz_offsets[:new_bsz]
                cand_bbsz_idx = cand_beams + bbsz_offsets
                cand_scores = cand_scores[batch_idxs]
                cand_indices = cand_indices[batch_idxs]
                src_lengths = src_lengths[batch_idxs]
                cands_to_ignore = cands_to_ignore[batch_idxs]
                scores = scores.view(bsz, -1)[batch_idxs].view(new_bsz * beam, -1)
                tokens = tokens.view(bsz, -1)[batch_idxs].view(new_bsz * beam, -1)
                bsz = new_bsz

            eos_mask[:, :beam] = ~((~cands_to_ignore) & (~eos_mask[:, :beam]))
            active_mask = eos_mask.long() * cand_size + cand_offsets[: eos_mask.size(1)]
            new_cands_to_ignore, active_hypos = torch.topk(active_mask, k=beam, dim=1, largest=False)
            cands_to_ignore = new_cands_to_ignore.ge(cand_size)[:, :beam]
            assert (~cands_to_ignore).any(dim=1).all()
            active_bbsz_idx = torch.gather(cand_bbsz_idx, 1, active_hypos).view(-1)
            tokens[:, : step + 1] = torch.index_select(tokens[:, : step + 1], 0, active_bbsz_idx)
            tokens.view(bsz, beam, -1)[:, :, step + 1] = torch.gather(cand_indices, 1, active_hypos)
            if step > 0:
                scores[:, :step] = torch.index_select(scores[:, :step], 0, active_bbsz_idx)
            scores.view(bsz, beam, -1)[:, :, step] = torch.gather(cand_scores, 1, active_hypos)
            # parent of every surviving hypothesis in the PREVIOUS numbering (before sentences were removed)
            if batch_idxs is not None:
                corr = batch_idxs - torch.arange(batch_idxs.numel(), device=dev)
                parent = (active_bbsz_idx.view(-1, beam) + corr.unsqueeze(-1) * beam).view(-1)
            else:
                parent = active_bbsz_idx

        for sent in range(len(finalized)):
            sc = torch.tensor([float(e["score"]) for e in finalized[sent]])
            _, order = torch.sort(sc, descending=True)
            finalized[sent] = [finalized[sent][int(i)] for i in order]
        return finalized

    def _finalize(self, step, bbsz_idx, eos_scores, tokens, scores, finalized, finished, beam, max_len):
        """Store hypotheses that just produced EOS; return the (current-numbering) sentences that became finished."""
        tokens_clone = tokens.index_select(0, bbsz_idx)[:, 1: step + 2].clone()
        tokens_clone[:, step] = self.eos
        pos_scores = scores.index_select(0, bbsz_idx)[:, : step + 1].clone()
        pos_scores[:, step] = eos_scores
        pos_scores[:, 1:] = pos_scores[:, 1:] - pos_scores[:, :-1]
        eos_scores = eos_scores.clone()
        if self.normalize_scores:
            eos_scores /= (step + 1) ** self.len_penalty
        cum_unfin, prev = [], 0
        for f in finished:
            if f:
                prev += 1
            else:
                cum_unfin.append(prev)
        unfin = (bbsz_idx // beam).tolist()
        sents = [u + cum_unfin[u] for u in unfin]
        tokens_clone, pos_scores, eos_scores = tokens_clone.cpu(), pos_scores.cpu(), eos_scores.cpu()
        for i, s in enumerate(sents):
            if len(finalized[s]) < beam:
                finalized[s].append({"tokens": tokens_clone[i], "score": eos_scores[i], "attention": None,
                                     "alignment": torch.empty(0), "positional_scores": pos_scores[i]})
        newly = []
        for s, u in sorted(set(zip(sents, unfin))):
            if not finished[s] and (len(finalized[s]) == beam or step == max_len):
                finished[s] = True
                newly.append(u)
        return newly
