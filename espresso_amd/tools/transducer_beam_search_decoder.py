"""Transducer beam search — the modified Adaptive Expansion Search of
espresso/tools/transducer_beam_search_decoder.py:21-601 (Kim et al. 2020, adapted from ESPnet) with the hypothesis
bookkeeping of espresso/tools/transducer_utils.py:17-757: per utterance and encoder frame, prefix search (scores of
hypotheses that are prefixes of longer ones are merged with logaddexp), up to `max_num_expansions_per_step` rounds of
top-(beam + beta) expansions with optional prune-by-value (gamma), blank hypotheses set aside, non-blank hypotheses pushed
through the predictor and, after the last round, closed with their blank probability; final scores normalised by length.

Layout of the work: everything that is per-hypothesis *bookkeeping* (scores, token sequences, lengths, emission counts) is a
handful of tiny host tensors — selection uses the same torch.topk / argsort calls as the reference, so ties break the
same way; everything that is *model compute* (predictor LSTM step for all surviving hypotheses at once, joint step,
log-softmax, optional LM step) runs on the HIP kernels with the predictor / LM state and the per-position predictor outputs
kept on the device and re-indexed by the surviving-hypothesis index lists."""
from typing import List, Optional

import torch
import torch.nn.functional as TF

from .. import kernels as K


class _Pool:
    """Append-only device store of predictor (or LM) output rows of ONE utterance's search: hypotheses refer to rows by slot
    number, so selecting / merging / padding hypotheses moves small host index tensors instead of re-copying `[n][L][H]`
    histories on the device at every expansion (the reference's `index_select_` / `pad` / `cat` on the full tensors)."""

    def __init__(self, width, device, cap=2048):
        self.buf = torch.zeros(cap, width, dtype=torch.bfloat16, device=device)
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


class _Hist:
    """Per-hypothesis output history: slots [n][L] (host i64) into a shared _Pool; position lens-1 is the newest."""

    def __init__(self, pool, slots):
        self.pool, self.slots, self.device = pool, slots, pool.device

    def select(self, index):
        return _Hist(self.pool, self.slots[index])

    def padded(self, L):
        return self if L == self.slots.shape[1] else _Hist(self.pool, TF.pad(self.slots, (0, L - self.slots.shape[1])))

    def last(self, lens):
        return self.pool.get(self.slots[torch.arange(self.slots.shape[0]), lens - 1])

    def put(self, lens, rows):
        n = self.slots.shape[0]
        base = self.pool.put(rows)
        self.slots[torch.arange(n), lens - 1] = base + torch.arange(n)

    def rows(self, ri, ci):
        return self.pool.get(self.slots[ri, ci])

    def at(self, i, k):
        return self.pool.buf[int(self.slots[i, k])]


class _Hyps:
    """Batch of hypotheses of ONE utterance.  Host: scores [n] f32, seqs [n][L] i64 (pad-filled), lens [n], nemit [n], prev [n],
    lm_scores [n] or None, dec = _Hist of predictor outputs by sequence position.  Device: state (predictor (h16, h32, c) lists
    with n rows) and the pool behind `dec`; lm_state / lm_dec likewise."""

    def __init__(self, scores, seqs, lens, nemit, prev, state, dec, lm_scores=None, lm_state=None, lm_dec=None):
        self.scores, self.seqs, self.lens, self.nemit, self.prev = scores, seqs, lens, nemit, prev
        self.state, self.dec = state, dec
        self.lm_scores, self.lm_state, self.lm_dec = lm_scores, lm_state, lm_dec

    def size(self):
        return int(self.scores.numel())

    def select(self, index):
        """transducer_utils.py:62-102 index_select_ (returns a new batch)."""
        di = index.to(self.dec.device)
        sel_state = lambda st: None if st is None else {k: [t.index_select(0, di) for t in v] for k, v in st.items()}
        return _Hyps(self.scores[index], self.seqs[index], self.lens[index], self.nemit[index], self.prev[index],
                     sel_state(self.state), self.dec.select(index),
                     None if self.lm_scores is None else self.lm_scores[index], sel_state(self.lm_state),
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

    def last_dec(self):
        """Predictor (and LM) output at the last non-blank position of every hypothesis (transducer_utils.py:386-417)."""
        return self.dec.last(self.lens), (None if self.lm_dec is None else self.lm_dec.last(self.lens))

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
        cat_state = lambda x, y: None if x is None else {k: [torch.cat((u, v), 0) for u, v in zip(x[k], y[k])] for k in x}
        return _Hyps(torch.cat((a.scores, b.scores)), torch.cat((pseq(a.seqs), pseq(b.seqs))), torch.cat((a.lens, b.lens)),
                     torch.cat((a.nemit, b.nemit)), torch.cat((a.prev, b.prev)), cat_state(a.state, b.state),
                     cat_hist(a.dec, b.dec),
                     None if a.lm_scores is None else torch.cat((a.lm_scores, b.lm_scores)), cat_state(a.lm_state, b.lm_state),
                     cat_hist(a.lm_dec, b.lm_dec))


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
    def _generate(self, sample, bos_token: Optional[int] = None):
        net_input = sample["net_input"]
        enc = self.model.encoder(net_input["src_tokens"], net_input["src_lengths"])
        x = enc["_x_bt"][0]
        enc_len = enc["src_lengths"][0].tolist()
        bsz = len(enc_len)
        Tp = x.shape[0] // bsz
        E = self.model.joint_encoder_branch(x).view(bsz, Tp, -1)
        toks, scs = [], []
        # the search's bookkeeping is many small host-tensor operations: run them on one thread (with the intra-op pool of a
        # 128-core host every `seqs[index]` / pad / topk above the parallel grain fans out and synchronises: 0.75 ms per select)
        threads = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            for i in range(bsz):
                t, s = self._one(E[i], int(enc_len[i]), bos_token)
                toks.append(t)
                scs.append(s)
        finally:
            torch.set_num_threads(threads)
        return toks, scs, None

    # ------------------------------------------------------------------ model compute on the device
    def _lprobs(self, E_t, dec_rows, lm_rows):
        """Joint + log-softmax (+ LM shallow fusion that keeps the non-blank mass, :289-321).  Returns (lprobs [n][V] f32 on
        the HOST, lm_lprobs padded to V on the host or None)."""
        n = dec_rows.shape[0]
        V = self.vocab_size
        logits = self.model.joint_step(E_t.unsqueeze(0).expand(n, -1).contiguous(), dec_rows.contiguous())[:, :V]
        if self.temperature != 1.0:
            logits = logits / self.temperature
        lprobs = K.log_softmax(logits, n, V, logits.stride(0))
        lm_pad = None
        if self.lm_model is not None:
            lm_logits = self.lm_model.decoder.output_layer(lm_rows.contiguous())
            lm_lp = K.log_softmax(lm_logits, n, lm_logits.shape[1], lm_logits.stride(0))
            nb = torch.ones(V, dtype=torch.bool, device=lprobs.device)
            nb[self.blank] = False
            lp_nb = lprobs[:, nb]
            if not self.no_blank_in_lm:
                lm_lp = lm_lp[:, nb]
            fused = lp_nb + self.lm_weight * lm_lp
            fused = fused + (lp_nb.exp().sum(1).log() - fused.exp().sum(1).log()).unsqueeze(1)
            lprobs[:, nb] = fused
            lm_pad = torch.cat((lm_lp[:, : self.blank], lm_lp.new_zeros(n, 1), lm_lp[:, self.blank:]), 1).cpu()
        if self.model_predicts_eos:
            lprobs[:, self.blank] = torch.logaddexp(lprobs[:, self.blank], lprobs[:, self.eos])
            lprobs[:, self.eos] = float("-inf")
        return lprobs.cpu(), lm_pad

    def _lm_tokens(self, tokens):
        return torch.where(tokens > self.blank, tokens - 1, tokens) if self.no_blank_in_lm else tokens

    def _advance(self, hyps):
        """Push the newest token of every hypothesis through the predictor (and LM); store the outputs at position lens-1."""
        dev = hyps.dec.device
        tok = hyps.prev.to(dev)
        dec_out, hyps.state = self.model.decoder.advance(tok, hyps.state)
        hyps.dec.put(hyps.lens, dec_out)
        if self.lm_model is not None:
            lm_out, hyps.lm_state = self.lm_model.decoder.advance(self._lm_tokens(tok), hyps.lm_state)
            hyps.lm_dec.put(hyps.lens, lm_out)

    # ------------------------------------------------------------------ the search (one utterance)
    def _one(self, E, enc_len, bos_token):
        dev = E.device
        max_len = min(enc_len, self.max_len) if self.max_len > 0 else enc_len
        bos = self.bos if bos_token is None else bos_token
        dec = self.model.decoder
        Hd = dec.hidden_size if not hasattr(dec, "additional_fc") else dec.additional_fc.weight.shape[0]
        hyps = _Hyps(torch.zeros(1), torch.full((1, 1), bos, dtype=torch.long), torch.ones(1, dtype=torch.long),
                     torch.zeros(1, dtype=torch.long), torch.full((1,), bos, dtype=torch.long), dec.init_state(1, dev),
                     _Hist(_Pool(Hd, dev), torch.zeros(1, 1, dtype=torch.long)))
        if self.lm_model is not None:
            lmd = self.lm_model.decoder
            Hl = lmd.hidden_size if not hasattr(lmd, "additional_fc") else lmd.additional_fc.weight.shape[0]
            hyps.lm_scores, hyps.lm_state = torch.zeros(1), lmd.init_state(1, dev)
            hyps.lm_dec = _Hist(_Pool(Hl, dev), torch.zeros(1, 1, dtype=torch.long))
        self._advance(hyps)
        nxt = hyps
        for step in range(max_len):
            nxt = nxt.sort_by_length(descending=True)
            E_t = E[step]
            hyps = self._prefix_search_and_merge(nxt, E_t)
            blanks = None
            for exp_idx in range(self.max_num_expansions_per_step):
                d_last, lm_last = hyps.last_dec()
                lprobs, lm_pad = self._lprobs(E_t, d_last, lm_last)
                kexp = self._select_k_expansions(hyps, lprobs, lm_pad)
                bmask = kexp.prev == self.blank
                kb = kexp.masked(bmask)
                blanks = kb if blanks is None else _Hyps.combine(blanks, kb, self.pad)
                knb = kexp.masked(~bmask)
                if knb.size() == 0:  # every candidate emitted blank: early exit of the expansions
                    nxt = blanks.keep_top_k(self.beam_size, self.normalize_scores)
                    break
                self._advance(knb)
                if exp_idx < self.max_num_expansions_per_step - 1:
                    hyps = knb
                else:
                    # last round: close the non-blank hypotheses with their blank probability, merge, prune
                    d_last, _ = knb.last_dec()
                    n = knb.size()
                    logits = self.model.joint_step(E_t.unsqueeze(0).expand(n, -1).contiguous(), d_last.contiguous())[:, : self.vocab_size]
                    if self.temperature != 1.0:
                        logits = logits / self.temperature
                    lp = K.log_softmax(logits, n, self.vocab_size, logits.stride(0))
                    knb.scores = knb.scores + lp[:, self.blank].cpu()
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

    def _prefix_search_and_merge(self, hyps, E_t):
        """:417-601 — `hyps` sorted by non-increasing length; the score of a hypothesis that is a prefix of a longer one (at
        most `prefix_alpha` tokens shorter) is added (logaddexp) to the longer one after extending it token by token."""
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
            vals = self._token_lprobs(E_t, hyps.dec.rows(ri, ci), toks)
            for i, j, a, b in spans:
                score = float(hyps.scores[i]) + vals[a]
                for k in range(a + 1, b):
                    score += vals[k]
                hyps.scores[j] = torch.logaddexp(hyps.scores[j], torch.tensor(float(score)))
            return hyps
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

    def _token_lprobs(self, E_t, dec_rows, toks):
        """RAW acoustic log-probability of token toks[r] given predictor output dec_rows[r], r = 0..m-1 -> host f32 [m]."""
        V, m = self.vocab_size, dec_rows.shape[0]
        logits = self.model.joint_step(E_t.unsqueeze(0).expand(m, -1).contiguous(), dec_rows.contiguous())[:, :V]
        if self.temperature != 1.0:
            logits = logits / self.temperature
        lp = K.log_softmax(logits, m, V, logits.stride(0))
        idx = torch.tensor(toks, dtype=torch.long).to(lp.device)
        return lp.gather(1, idx.unsqueeze(1)).squeeze(1).cpu()

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
