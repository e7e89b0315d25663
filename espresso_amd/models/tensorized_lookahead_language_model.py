"""Look-ahead word language model for sub-word beam search — espresso/models/tensorized_lookahead_language_model.py:17-304
(Hori et al. 2018, Eqn. 15, adapted to sentences that end with <space> before <eos>).

Host side: state bookkeeping only.  Per step the word LSTM LM advances for the hypotheses that just closed a word
(`ea_lstm_cell_fwd` with frozen rows = the reference's masked_copy_cached_state), its distribution is turned into a prefix
sum over the lexically sorted word vocabulary (`ea_softmax_cumsum`), every hypothesis moves along the prefix tree
(`ea_lookahead_advance`) and the sub-word log-probabilities of all four cases are produced by `ea_lookahead_logprobs`."""
import torch

from .. import _lib
from .. import kernels as K
from ..tools.tensorized_prefix_tree import TensorizedPrefixTree, tokenize


class TensorizedLookaheadLanguageModel:
    def __init__(self, word_lm, subword_dict, oov_penalty: float = 1e-4, open_vocab: bool = True):
        self.word_lm = word_lm
        self.lm_decoder = word_lm.decoder
        self.decoder = self  # fairseq-style access: model.decoder.dictionary
        self.dictionary = self.lm_decoder.dictionary
        self.oov_penalty, self.open_vocab = float(oov_penalty), bool(open_vocab)
        wd = self.dictionary
        self.word_pad_idx, self.word_eos_idx, self.word_unk_idx = wd.pad(), wd.eos(), wd.unk()
        self.subword_space_idx, self.subword_pad_idx, self.subword_eos_idx = subword_dict.space(), subword_dict.pad(), subword_dict.eos()
        self.subword_vocab_size = len(subword_dict)
        nls = getattr(subword_dict, "non_lang_syms", None)
        self.tree = TensorizedPrefixTree.build(wd, subword_dict, lambda x: tokenize(x, non_lang_syms=nls).split(" "))
        assert self.tree.max_out_degree() <= self.subword_vocab_size

    def eval(self):
        self.word_lm.eval()
        return self

    def max_positions(self):
        return int(1e5)

    def init_incremental(self, bsz, beam):
        dev = self.lm_decoder.embed_tokens.weight.device
        n = bsz * beam
        return {"lstm": self.lm_decoder.init_state(n, dev), "cumsum": None, "nodes": None, "lp_eos": None}

    @torch.no_grad()
    def step(self, state, tokens, step, parent=None):
        """tokens [N][step+1] sub-word history; returns fp32 sub-word log-probs [N][Vs] (already log-probs: :268-274)."""
        dev = tokens.device
        N = tokens.shape[0]
        prev = tokens[:, -1].to(torch.int32).contiguous()
        Vw = len(self.dictionary)
        children, prev_sub, word_idx, word_set = self.tree.device_tensors(dev)
        lib = _lib.lib()
        p, st = K._p, K._stream()
        if state["cumsum"] is None:  # first step: the word history is <eos>, every hypothesis sits at the root
            w = torch.full((N,), self.word_eos_idx, dtype=torch.long, device=dev)
            feat, state["lstm"] = self.lm_decoder.advance(w, state["lstm"])
            logits = self.lm_decoder.output_layer(feat)
            state["cumsum"] = torch.empty(N, Vw, dtype=torch.float32, device=dev)
            state["lp_eos"] = torch.empty(N, dtype=torch.float32, device=dev)
            _lib.check(lib.ea_softmax_cumsum(p(logits), logits.stride(0), None, p(state["cumsum"]), p(state["lp_eos"]), N, Vw,
                                             self.word_eos_idx, st), "ea_softmax_cumsum")
            state["nodes"] = torch.full((N,), self.tree.root_id, dtype=torch.int32, device=dev)
        else:
            if parent is not None:
                idx = parent.to(torch.int32).contiguous()
                state["lstm"] = self.lm_decoder.reorder_state(state["lstm"], idx)
                state["cumsum"] = K.gather_rows(state["cumsum"], idx)
                state["lp_eos"] = K.gather_rows(state["lp_eos"].view(-1, 1), idx).view(-1)
                state["nodes"] = K.gather_rows(state["nodes"].view(-1, 1).view(torch.float32), idx).view(torch.int32).view(-1)
            nodes = state["nodes"]
            w = word_idx[nodes.long()].long()
            w = torch.where(w < 0, torch.full_like(w, self.word_unk_idx), w)
            space = prev == self.subword_space_idx
            frozen = (~space).to(torch.uint8).contiguous()  # the word LM only advances where a word was just closed
            feat, state["lstm"] = self.lm_decoder.advance(w, state["lstm"], keep_row=frozen)
            logits = self.lm_decoder.output_layer(feat)
            _lib.check(lib.ea_softmax_cumsum(p(logits), logits.stride(0), p(space.to(torch.uint8).contiguous()), p(state["cumsum"]),
                                             p(state["lp_eos"]), N, Vw, self.word_eos_idx, st), "ea_softmax_cumsum")
            _lib.check(lib.ea_lookahead_advance(p(nodes), p(prev), p(children), p(prev_sub), N, children.shape[1],
                                                self.subword_space_idx, self.tree.root_id, st), "ea_lookahead_advance")
        out = torch.empty(N, self.subword_vocab_size, dtype=torch.float32, device=dev)
        _lib.check(lib.ea_lookahead_logprobs(p(state["nodes"]), p(prev), p(state["cumsum"]), p(state["lp_eos"]), p(children), p(prev_sub),
                                             p(word_idx), p(word_set), p(out), N, Vw, self.subword_vocab_size, children.shape[1],
                                             self.oov_penalty, int(self.open_vocab), self.word_unk_idx, self.subword_space_idx,
                                             self.subword_eos_idx, self.subword_pad_idx, st), "ea_lookahead_logprobs")
        return out

    def shrink(self, state, keep_rows):
        idx = keep_rows.to(torch.int32).contiguous()
        state["lstm"] = self.lm_decoder.reorder_state(state["lstm"], idx)
        state["cumsum"] = K.gather_rows(state["cumsum"], idx)
        state["lp_eos"] = K.gather_rows(state["lp_eos"].view(-1, 1), idx).view(-1)
        state["nodes"] = K.gather_rows(state["nodes"].view(-1, 1).view(torch.float32), idx).view(torch.int32).view(-1)
