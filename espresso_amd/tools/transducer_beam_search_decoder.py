"""Transducer beam search — the modified Adaptive Expansion Search of
espresso/tools/transducer_beam_search_decoder.py:21-601 (Kim et al. 2020, adapted from ESPnet) with the hypothesis
bookkeeping of espresso/tools/transducer_utils.py:17-757: per utterance and encoder frame, prefix search (scores of
hypotheses that are prefixes of longer ones are merged with logaddexp), up to `max_num_expansions_per_step` rounds of
top-(beam + beta) expansions with optional prune-by-value (gamma), blank hypotheses set aside, non-blank hypotheses pushed
through the predictor and, after the last round, closed with their blank probability; final scores normalised by length.

Layout of the work: everything that is per-hypothesis *bookkeeping* (scores, token sequences, lengths, emission counts) is a
handful of tiny host tensors per utterance — selection uses the same torch.topk / argsort calls as the reference, so ties break
the same way.  Everything that is *model compute* (predictor LSTM step, joint step, log-softmax, optional LM step) runs on the
HIP kernels and is BATCHED ACROSS THE UTTERANCES of the sample: each utterance's search is a coroutine that yields its next
compute request (joint over these hypotheses at this frame / advance these hypotheses / raw token log-probabilities for the prefix
merge); the driver serves all pending requests of one kind with ONE launch sequence and ONE device-to-host transfer and resumes
the coroutines.  Predictor / LM outputs AND their (h, c) states live in append-only device pools shared by the whole batch;
hypotheses carry slot numbers, so select / merge / pad / top-k move small host index tensors and launch nothing."""
from collections import defaultdict
from typing import List, Optional

import torch
import torch.nn.functional as TF

from .. import kernels as K


class _Pool:
    """Append-only device store of rows ([cap][width], any dtype) shared by every utterance of a batch: hypotheses refer to rows by
    slot number (the reference `index_select_`s / pads / concatenates `[n][L][H]` histories and `[L][n][H]` states instead)."""

    def __init__(self, width, device, dtype=torch.bfloat16, cap=4096):
        self.buf = torch.zeros(cap, width, dtype=dtype, device=device)
        self.n, self.device = 0, device

    def put(self, rows):
        m = rows.shape[0]
        if self.n + m > self.buf.shape[0]:
            grown = torch.zeros(max(2 * self.buf.shape[0], self.n + m), self.buf.shape[1], dtype=self.buf.dtype, device=self.device)
            grown[: self.n] = self.buf[: self.n]
            self.buf = grown
        self.buf[self.n:self.n + m] = rows
        self.n += m
        return self.n - m

    def get(self, slots):
        idx = slots if torch.is_tensor(slots) else torch.tensor(slots, dtype=torch.long)
        return self.buf.index_select(0, idx.reshape(-1).to(self.device))


class _StatePools:
    """(h16, h32, c) of every layer of an LSTM predictor / LM in row pools; slot 0 is the zero state."""

    def __init__(self, dec, device):
        z = dec.init_state(1, device)
        self.pools = {k: [_Pool(t.shape[1], device, t.dtype) for t in v] for k, v in z.items()}
        for k, v in z.items():
            for pool, t in zip(self.pools[k], v):
                pool.put(t)

    def get(self, slots):
        idx = slots.to(self.pools["c"][0].device)
        return {k: [p.buf.index_select(0, idx) for p in v] for k, v in self.pools.items()}

    def put(self, state):
        base = None
        for k, v in state.items():
            for pool, t in zip(self.pools[k], v):
                base = pool.put(t)
        return base


class _Hist:
    """Per-hypothesis output history: slots [n][L] (host i64) into a shared _Pool; position lens-1 is the newest."""

    def __init__(self, pool, slots):
        self.pool, self.slots, self.device = pool, slots, pool.device

    def select(self, index):
        return _Hist(self.pool, self.slots[index])

    def padded(self, L):
        return self if L == self.slots.shape[1] else _Hist(self.pool, TF.pad(self.slots, (0, L - self.slots.shape[1])))

    def last_slots(self, lens):
        return self.slots[torch.arange(self.slots.shape[0]), lens - 1]

    def set_last(self, lens, base):
        n = self.slots.shape[0]
        self.slots[torch.arange(n), lens - 1] = base + torch.arange(n)

    def slots_at(self, ri, ci):
        return self.slots[ri, ci]

    # row-valued forms of the three accessors above (single-batch use and the dense-semantics test)
    def last(self, lens):
        return self.pool.get(self.last_slots(lens))

    def put(self, lens, rows):
        self.set_last(lens, self.pool.put(rows))

    def rows(self, ri, ci):
        return self.pool.get(self.slots_at(ri, ci))

    def at(self, i, k):
        return self.pool.buf[int(self.slots[i, k])]


class _Hyps:
    """Batch of hypotheses of ONE utterance, host tensors only: scores [n] f32, seqs [n][L] i64 (pad-filled), lens [n], nemit [n],
    prev [n], sslot [n] (row of the predictor state in the shared state pools), dec = _Hist of predictor-output slots by sequence
    position; lm_scores / lm_sslot / lm_dec likewise when an LM is fused."""

    def __init__(self, scores, seqs, lens, nemit, prev, sslot, dec, lm_scores=None, lm_sslot=None, lm_dec=None):
        self.scores, self.seqs, self.lens, self.nemit, self.prev = scores, seqs, lens, nemit, prev
        self.sslot, self.dec = sslot, dec
        self.lm_scores, self.lm_sslot, self.lm_dec = lm_scores, lm_sslot, lm_dec

    def size(self):
        return int(self.scores.numel())

    def select(self, index):
        """transducer_utils.py:62-102 index_select_ (returns a new batch)."""
        return _Hyps(self.scores[index], self.seqs[index], self.lens[index], self.nemit[index], self.prev[index],
                     self.sslot[index], self.dec.select(index),
                     None if self.lm_scores is None else self.lm_scores[index],
                     None if self.lm_sslot is None else self.lm_sslot[index],
                     None if self.lm_dec is None else self.lm_dec.select(index))

    def sort_by_length(self, descending=True):
        return self if self.size() == 0 else self.select(self.lens.argsort(descending=descending))

    def sort_by_score(self, descending=True, normalize=False):
        if self.size() == 0:
            return self
        s = self.scores / self.nemit if normalize else self.scores
        return self.select(s.argsort(descending=descending))

    def keep_top_k(self, k, normalize=False):
        """transducer_utils.py:163-201."""
        if k > self.size():
            return self.sort_by_score(True, normalize)
        s = self.scores / self.nemit if normalize else self.scores
        _, idx = torch.topk(s, k, largest=True, sorted=True)
        return self.select(idx)

    def masked(self, mask):
        return self.select(mask.nonzero(as_tuple=False).view(-1))

    def last_slots(self):
        """Slots of the predictor (and LM) output at the last non-blank position of every hypothesis (transducer_utils.py:386-417)."""
        return self.dec.last_slots(self.lens), (None if self.lm_dec is None else self.lm_dec.last_slots(self.lens))

    @staticmethod
    def combine(a, b, pad):
        """transducer_utils.py:492-637: concatenation along the hypothesis axis after padding the position axis."""
        if b.size() == 0:
            return a
        if a.size() == 0:
            return b
        L = max(a.seqs.shape[1], b.seqs.shape[1])
        pseq = lambda s: TF.pad(s, (0, L - s.shape[1]), value=pad)
        cat_hist = lambda x, y: None if x is None else _Hist(x.pool, torch.cat((x.padded(L).slots, y.padded(L).slots)))
        cat = lambda x, y: None if x is None else torch.cat((x, y))
        return _Hyps(torch.cat((a.scores, b.scores)), torch.cat((pseq(a.seqs), pseq(b.seqs))), torch.cat((a.lens, b.lens)),
                     torch.cat((a.nemit, b.nemit)), torch.cat((a.prev, b.prev)), torch.cat((a.sslot, b.sslot)),
                     cat_hist(a.dec, b.dec), cat(a.lm_scores, b.lm_scores), cat(a.lm_sslot, b.lm_sslot), cat_hist(a.lm_dec, b.lm_dec))


class TransducerBeamSearchDecoder:
    def __init__(self, models, dictionary, beam_size=1, max_len=0, max_num_expansions_per_step=2, expansion_beta=0,
                 expansion_gamma=None, prefix_alpha=None, normalize_scores=True, temperature=1.0, eos=None, bos=None, blank=None,
                 pad=None, model_predicts_eos=False, symbols_to_strip_from_output=None, lm_model=None, lm_weight=1.0,
                 print_alignment=False, **kwargs):
        self.model = models[0]
        self.eos = dictionary.eos() if eos is None else eos
        self.bos = dictionary.eos() if bos is None else bos
        self.blank = dictionary.bos() if blank is None else blank
        self.pad = dictionary.pad() if pad is None else pad
        self.model_predicts_eos = model_predicts_eos
        strip = {self.eos, self.bos, self.blank}
        self.symbols_to_strip_from_output = strip.union(symbols_to_strip_from_output) if symbols_to_strip_from_output else strip
        self.vocab_size = len(dictionary)
        self.beam_size = min(beam_size, self.vocab_size - (1 if self.pad != self.blank else 0))
        self.max_len = max_len
        assert max_num_expansions_per_step > 0, "--max-num-expansions-per-step must be at least 1"
        self.max_num_expansions_per_step = max_num_expansions_per_step
        assert expansion_beta >= 0, "--expansion-beta must be non-negative"
        assert expansion_gamma is None or expansion_gamma > 0.0, "--expansion-gamma must be greater than 0.0"
        assert prefix_alpha is None or prefix_alpha > 0, "--prefix-alpha must be None or at least 1"
        self.expansion_beta, self.expansion_gamma, self.prefix_alpha = expansion_beta, expansion_gamma, prefix_alpha
        self.normalize_scores = normalize_scores
        assert temperature > 0, "--temperature must be greater than 0"
        self.temperature = temperature
        assert not print_alignment, "alignments are produced by the greedy decoder only"
        self.model.eval()
        self.lm_model, self.lm_weight = lm_model, lm_weight
        if lm_model is not None:
            nlm = len(lm_model.decoder.dictionary)
            assert nlm in (self.vocab_size, self.vocab_size - 1)
            self.no_blank_in_lm = nlm == self.vocab_size - 1
            lm_model.eval()

    # ------------------------------------------------------------------ API of the reference
    @torch.no_grad()
    def decode(self, models, sample, **kwargs):
        tokens_list, scores_list, _ = self._generate(sample, **kwargs)
        L = max(t.shape[1] for t in tokens_list)
        tokens = torch.stack([TF.pad(t[0], (0, L - t.shape[1]), value=self.pad) for t in tokens_list])
        return tokens, torch.stack([s[0] for s in scores_list]), None

    @torch.no_grad()
    def generate(self, models, sample, **kwargs):
        tokens_list, scores_list, _ = self._generate(sample, bos_token=kwargs.get("bos_token", None))
        out = []
        for toks, scs in zip(tokens_list, scores_list):
            out.append([{"tokens": toks[j][toks[j] != self.pad], "score": scs[j], "attention": None, "alignment": None}
                        for j in range(toks.shape[0])])
        return out

    @torch.no_grad()
    def _generate(self, sample, bos_token: Optional[int] = None, only: Optional[List[int]] = None):
        """only: search just these utterances of the batch (same encoder pass; the others return None) — lets a test compare a
        search that shared its launches with the other utterances against the same search running alone."""
        net_input = sample["net_input"]
        enc = self.model.encoder(net_input["src_tokens"], net_input["src_lengths"])
        x = enc["_x_bt"][0]
        enc_len = enc["src_lengths"][0].tolist()
        bsz = len(enc_len)
        Tp = x.shape[0] // bsz
        dev = x.device
        self._E = self.model.joint_encoder_branch(x).contiguous()  # [bsz * T'][J], row = utterance * T' + frame
        self._Tp = Tp
        # Predictor / LM states live in append-only row pools that are released when a group of searches ends: frames x
        # expansions x beam rows per utterance, (layers x H x 10 + H x 2) bytes per row.  The utterances of a batch are therefore
        # searched in groups whose estimated pool size stays under `state_pool_budget_bytes` (one group for ordinary batches; the
        # encoder pass above is shared), instead of letting a long or large batch grow the pools without bound.
        want = [i for i in range(bsz) if (only is None or i in only)]
        done = [(None, None)] * bsz
        threads = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            for group in self._pool_groups(want, enc_len):
                self._search_group(group, enc_len, bos_token, done, dev)
        finally:
            torch.set_num_threads(threads)
            self._E = self._dec_pool = self._dec_state = self._lm_pool = self._lm_state = None
        return [d[0] for d in done], [d[1] for d in done], None

    state_pool_budget_bytes = 4 << 30

    def _pool_row_bytes(self):
        dec = self.model.decoder
        n = sum(t.shape[1] * t.element_size() for v in dec.init_state(1, "cpu").values() for t in v)
        n += 2 * (dec.hidden_size if not hasattr(dec, "additional_fc") else dec.additional_fc.weight.shape[0])
        if self.lm_model is not None:
            lmd = self.lm_model.decoder
            n += sum(t.shape[1] * t.element_size() for v in lmd.init_state(1, "cpu").values() for t in v)
            n += 2 * (lmd.hidden_size if not hasattr(lmd, "additional_fc") else lmd.additional_fc.weight.shape[0])
        return n

    def _pool_groups(self, want, enc_len):
        """Utterance indices in batch order, cut so that the estimated pool rows of a group fit the budget (at least one each)."""
        per_row = self._pool_row_bytes()
        cap = max(1, int(self.state_pool_budget_bytes // per_row))
        groups, cur, rows = [], [], 0
        for i in want:
            r = max(1, int(enc_len[i])) * self.max_num_expansions_per_step * self.beam_size
            if cur and rows + r > cap:
                groups.append(cur)
                cur, rows = [], 0
            cur.append(i)
            rows += r
        if cur:
            groups.append(cur)
        return groups

    def _search_group(self, group, enc_len, bos_token, done, dev):
        """The coroutine driver for one group of utterances; fills done[i] = (tokens, scores)."""
        dec = self.model.decoder
        Hd = dec.hidden_size if not hasattr(dec, "additional_fc") else dec.additional_fc.weight.shape[0]
        self._dec_pool, self._dec_state = _Pool(Hd, dev), _StatePools(dec, dev)
        self._lm_pool = self._lm_state = None
        if self.lm_model is not None:
            lmd = self.lm_model.decoder
            Hl = lmd.hidden_size if not hasattr(lmd, "additional_fc") else lmd.additional_fc.weight.shape[0]
            self._lm_pool, self._lm_state = _Pool(Hl, dev), _StatePools(lmd, dev)
        # the search's bookkeeping is many small host-tensor operations: run them on one thread (with the intra-op pool of a
        # 128-core host every `seqs[index]` / pad / topk above the parallel grain fans out and synchronises: 0.75 ms per select)
        searches = {i: self._one(i, int(enc_len[i]), bos_token) for i in group}
        pending = {}
        for i, g in searches.items():
            try:
                pending[i] = next(g)
            except StopIteration as e:  # (an utterance without frames)
                done[i] = e.value
        while pending:
            kinds = defaultdict(list)
            for i, req in pending.items():
                kinds[req[0]].append((i, req))
            answers = {}
            for kind, items in kinds.items():
                answers.update(self._serve(kind, items))
            nxt = {}
            for i, ans in answers.items():
                try:
                    nxt[i] = searches[i].send(ans)
                except StopIteration as e:
                    done[i] = e.value
            pending = nxt

    # ------------------------------------------------------------------ model compute on the device, batched over utterances
    def _joint_lprobs(self, frame_rows, dec_slots):
        """log-softmax(joint(E[frame_rows[r]], predictor output dec_slots[r]) / temperature), r = 0..N-1 -> device f32 [N][V]."""
        V = self.vocab_size
        dev = self._E.device
        E_rows = self._E.index_select(0, frame_rows.to(dev))
        logits = self.model.joint_step(E_rows, self._dec_pool.get(dec_slots))[:, :V]
        if self.temperature != 1.0:
            logits = logits / self.temperature
        return K.log_softmax(logits, logits.shape[0], V, logits.stride(0))

    def _serve(self, kind, items):
        """items: [(utterance, request)] of one kind -> {utterance: answer}; one launch sequence and one device-to-host copy."""
        if kind == "advance":
            self._advance_many([req[1] for _, req in items])
            return {i: None for i, _ in items}
        sizes = [int(req[2].numel()) for _, req in items]
        rows = torch.cat([torch.full((n,), int(req[1]), dtype=torch.long) for n, (_, req) in zip(sizes, items)])
        lp = self._joint_lprobs(rows, torch.cat([req[2] for _, req in items]))
        if kind == "token":  # RAW acoustic log-probability of one given token per row (prefix merge)
            tok = torch.cat([req[3] for _, req in items]).to(lp.device)
            vals = lp.gather(1, tok.unsqueeze(1)).squeeze(1).cpu().split(sizes)
            return {i: v for (i, _), v in zip(items, vals)}
        if kind == "blank":  # blank log-probability that closes the non-blank hypotheses of the last expansion round
            vals = lp[:, self.blank].cpu().split(sizes)
            return {i: v for (i, _), v in zip(items, vals)}
        assert kind == "lprobs"
        lm_pad = None
        if self.lm_model is not None:  # shallow fusion that keeps the non-blank mass (:289-321)
            n, V = lp.shape
            lm_rows = self._lm_pool.get(torch.cat([req[3] for _, req in items]))
            lm_logits = self.lm_model.decoder.output_layer(lm_rows.contiguous())
            lm_lp = K.log_softmax(lm_logits, n, lm_logits.shape[1], lm_logits.stride(0))
            nb = torch.ones(V, dtype=torch.bool, device=lp.device)
            nb[self.blank] = False
            lp_nb = lp[:, nb]
            if not self.no_blank_in_lm:
                lm_lp = lm_lp[:, nb]
            fused = lp_nb + self.lm_weight * lm_lp
            fused = fused + (lp_nb.exp().sum(1).log() - fused.exp().sum(1).log()).unsqueeze(1)
            lp[:, nb] = fused
            lm_pad = torch.cat((lm_lp[:, : self.blank], lm_lp.new_zeros(n, 1), lm_lp[:, self.blank:]), 1).cpu().split(sizes)
        if self.model_predicts_eos:
            lp[:, self.blank] = torch.logaddexp(lp[:, self.blank], lp[:, self.eos])
            lp[:, self.eos] = float("-inf")
        host = lp.cpu().split(sizes)
        return {i: (host[k], None if lm_pad is None else lm_pad[k]) for k, (i, _) in enumerate(items)}

    def _lm_tokens(self, tokens):
        return torch.where(tokens > self.blank, tokens - 1, tokens) if self.no_blank_in_lm else tokens

    def _advance_many(self, batches):
        """Push the newest token of every hypothesis of every given batch through the predictor (and LM) in one call; the outputs
        go to position lens-1 of the histories, the new states to fresh slots."""
        dev = self._dec_pool.device
        tok = torch.cat([h.prev for h in batches]).to(dev)
        dec_out, new = self.model.decoder.advance(tok, self._dec_state.get(torch.cat([h.sslot for h in batches])))
        base, sbase = self._dec_pool.put(dec_out), self._dec_state.put(new)
        if self.lm_model is not None:
            lm_out, lm_new = self.lm_model.decoder.advance(self._lm_tokens(tok), self._lm_state.get(torch.cat([h.lm_sslot for h in batches])))
            lbase, lsbase = self._lm_pool.put(lm_out), self._lm_state.put(lm_new)
        off = 0
        for h in batches:
            n = h.size()
            h.dec.set_last(h.lens, base + off)
            h.sslot = sbase + off + torch.arange(n)
            if self.lm_model is not None:
                h.lm_dec.set_last(h.lens, lbase + off)
                h.lm_sslot = lsbase + off + torch.arange(n)
            off += n

    # ------------------------------------------------------------------ the search of one utterance (a coroutine: see _generate)
    def _one(self, utt, enc_len, bos_token):
        max_len = min(enc_len, self.max_len) if self.max_len > 0 else enc_len
        bos = self.bos if bos_token is None else bos_token
        hyps = _Hyps(torch.zeros(1), torch.full((1, 1), bos, dtype=torch.long), torch.ones(1, dtype=torch.long),
                     torch.zeros(1, dtype=torch.long), torch.full((1,), bos, dtype=torch.long), torch.zeros(1, dtype=torch.long),
                     _Hist(self._dec_pool, torch.zeros(1, 1, dtype=torch.long)))
        if self.lm_model is not None:
            hyps.lm_scores, hyps.lm_sslot = torch.zeros(1), torch.zeros(1, dtype=torch.long)
            hyps.lm_dec = _Hist(self._lm_pool, torch.zeros(1, 1, dtype=torch.long))
        yield ("advance", hyps)
        nxt = hyps
        for step in range(max_len):
            nxt = nxt.sort_by_length(descending=True)
            frame = utt * self._Tp + step  # row of this utterance's frame in the joint's encoder branch
            hyps = yield from self._prefix_search_and_merge(nxt, frame)
            blanks = None
            for exp_idx in range(self.max_num_expansions_per_step):
                d_last, lm_last = hyps.last_slots()
                lprobs, lm_pad = yield ("lprobs", frame, d_last, lm_last)
                kexp = self._select_k_expansions(hyps, lprobs, lm_pad)
                bmask = kexp.prev == self.blank
                kb = kexp.masked(bmask)
                blanks = kb if blanks is None else _Hyps.combine(blanks, kb, self.pad)
                knb = kexp.masked(~bmask)
                if knb.size() == 0:  # every candidate emitted blank: early exit of the expansions
                    nxt = blanks.keep_top_k(self.beam_size, self.normalize_scores)
                    break
                yield ("advance", knb)
                if exp_idx < self.max_num_expansions_per_step - 1:
                    hyps = knb
                else:
                    # last round: close the non-blank hypotheses with their blank probability, merge, prune
                    lp_blank = yield ("blank", frame, knb.last_slots()[0])
                    knb.scores = knb.scores + lp_blank
                    knb.prev = torch.full_like(knb.prev, self.blank)
                    knb.nemit = knb.nemit + 1
                    nxt = _Hyps.combine(blanks, knb, self.pad).keep_top_k(self.beam_size, self.normalize_scores)
        nxt.scores = nxt.scores / (nxt.lens - 1)
        nxt = nxt.sort_by_score(descending=True)
        return nxt.seqs[:, 1:], nxt.scores

    def _select_k_expansions(self, hyps, lprobs, lm_pad):
        """transducer_utils.py:639-710."""
        V = lprobs.shape[1]
        tot = lprobs + hyps.scores.unsqueeze(-1)
        k = min(self.beam_size + self.expansion_beta, V - (1 if self.pad != self.blank else 0))
        scores, indices = torch.topk(tot, k=k)
        n = hyps.size()
        rep = torch.arange(n).repeat_interleave(k)
        kexp = hyps.select(rep)
        kexp.scores = scores.reshape(-1).clone()
        if lm_pad is not None:
            kexp.lm_scores = kexp.lm_scores + lm_pad.gather(1, indices).reshape(-1)
        self._append_tokens(kexp, indices.reshape(-1))
        if self.expansion_gamma is not None:
            keep = scores >= (scores[:, :1] - self.expansion_gamma)
            if not bool(keep.all()):
                kexp = kexp.masked(keep.reshape(-1))
        return kexp.keep_top_k(k, self.normalize_scores)

    def _append_tokens(self, h, tokens):
        """transducer_utils.py:278-343: non-blank tokens extend their sequence (position axis grows when a longest hypothesis
        is extended); every candidate counts one more emission."""
        h.prev = tokens.clone()
        bmask = tokens == self.blank
        if bool(bmask.all()):
            return
        max_length = int(h.lens.max())
        if bool((tokens[h.lens == max_length] != self.blank).any()):
            h.seqs = TF.pad(h.seqs, (0, 1), value=self.pad)
            h.dec = h.dec.padded(h.dec.slots.shape[1] + 1)
            if h.lm_dec is not None:
                h.lm_dec = h.lm_dec.padded(h.lm_dec.slots.shape[1] + 1)
        h.seqs.scatter_(1, h.lens.unsqueeze(1), tokens.masked_fill(bmask, self.pad).unsqueeze(1))
        h.lens = h.lens + (~bmask).long()
        h.nemit = h.nemit + 1

    def _prefix_search_and_merge(self, hyps, frame):
        """:417-601 — `hyps` sorted by non-increasing length; the score of a hypothesis that is a prefix of a longer one (at
        most `prefix_alpha` tokens shorter) is added (logaddexp) to the longer one after extending it token by token.
        (A generator like `_one`: the joint evaluations it needs are requests served together with the other utterances'.)"""
        n = hyps.size()
        lens = hyps.lens
        lens_l, seqs_l = lens.tolist(), hyps.seqs.tolist()  # host bookkeeping on plain lists
        merge = [[False] * n for _ in range(n)]
        any_merge = False
        for j in range(n - 1):
            for i in range(j + 1, n):
                li = lens_l[i]
                ok = li < lens_l[j] and seqs_l[i][:li] == seqs_l[j][:li]
                if ok and self.prefix_alpha is not None:
                    ok = li + self.prefix_alpha >= lens_l[j]
                merge[i][j] = ok
                any_merge |= ok
        if not any_merge:
            return hyps
        if self.lm_model is None:
            # every (predictor output, next token) pair the merges need, through the joint in ONE call; only the m requested
            # log-probabilities come back to the host (the reference evaluates the joint once per extension step, :487-505)
            ri, ci, toks, spans = [], [], [], []
            for j in range(n - 1):
                for i in range(j + 1, n):
                    if not merge[i][j]:
                        continue
                    li, lj = lens_l[i], lens_l[j]
                    a = len(toks)
                    ri.append(i), ci.append(li - 1), toks.append(seqs_l[j][li])
                    for k in range(li, lj - 1):
                        ri.append(j), ci.append(k), toks.append(seqs_l[j][k + 1])
                    spans.append((i, j, a, len(toks)))
            vals = yield ("token", frame, hyps.dec.slots_at(ri, ci), torch.tensor(toks, dtype=torch.long))
            for i, j, a, b in spans:
                score = float(hyps.scores[i]) + vals[a]
                for k in range(a + 1, b):
                    score += vals[k]
                hyps.scores[j] = torch.logaddexp(hyps.scores[j], torch.tensor(float(score)))
            return hyps
        E_t = self._E[frame]
        for j in range(n - 1):
            for i in range(j + 1, n):
                if not merge[i][j]:
                    continue
                li, lj = lens_l[i], lens_l[j]
                # first extension uses hypothesis i's newest predictor output, the following ones hypothesis j's own history
                lp, lm_lp = self._row_lprobs(E_t, hyps.dec.at(i, li - 1), None if hyps.lm_dec is None else hyps.lm_dec.at(i, li - 1))
                tok = int(hyps.seqs[j, li])
                score = float(hyps.scores[i]) + lp[tok]
                lm_score = None
                if self.lm_model is not None:
                    loc, scale = self._lm_terms(lp, lm_lp, tok)
                    lm_score = float(hyps.lm_scores[i]) + loc
                    score += self.lm_weight * loc + scale
                for k in range(li, lj - 1):
                    lp, lm_lp = self._row_lprobs(E_t, hyps.dec.at(j, k), None if hyps.lm_dec is None else hyps.lm_dec.at(j, k))
                    tok = int(hyps.seqs[j, k + 1])
                    score += lp[tok]
                    if self.lm_model is not None:
                        loc, scale = self._lm_terms(lp, lm_lp, tok)
                        lm_score += loc
                        score += self.lm_weight * loc + scale
                hyps.scores[j] = torch.logaddexp(hyps.scores[j], torch.tensor(float(score)))
                if self.lm_model is not None:
                    hyps.lm_scores[j] = torch.logaddexp(hyps.lm_scores[j], torch.tensor(float(lm_score)))
        return hyps
        yield  # (makes this function a generator also on the paths that request nothing)

    def _row_lprobs(self, E_t, dec_row, lm_row):
        """RAW acoustic log-probs of one hypothesis position (the prefix search applies the LM terms itself, :487-505)."""
        V = self.vocab_size
        logits = self.model.joint_step(E_t.unsqueeze(0).contiguous(), dec_row.unsqueeze(0).contiguous())[:, :V]
        if self.temperature != 1.0:
            logits = logits / self.temperature
        lp = K.log_softmax(logits, 1, V, logits.stride(0))[0].cpu()
        lm_lp = None
        if self.lm_model is not None:
            lm_logits = self.lm_model.decoder.output_layer(lm_row.unsqueeze(0).contiguous())
            lm_lp = K.log_softmax(lm_logits, 1, lm_logits.shape[1], lm_logits.stride(0))[0].cpu()
        return lp, lm_lp

    def _lm_terms(self, lp, lm_lp, tok):
        lm_tok = tok - 1 if (self.no_blank_in_lm and tok > self.blank) else tok
        loc = float(lm_lp[lm_tok])
        nb = torch.ones(self.vocab_size, dtype=torch.bool)
        nb[self.blank] = False
        lp_nb = lp[nb]
        lm_nb = lm_lp if self.no_blank_in_lm else lm_lp[nb]
        fused = lp_nb + self.lm_weight * lm_nb
        return loc, float(lp_nb.exp().sum().log() - fused.exp().sum().log())
