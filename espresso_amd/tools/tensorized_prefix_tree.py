"""Tensorized lexical prefix tree — same tensors as espresso/tools/tensorized_prefix_tree.py:14-108 built on top of
espresso/tools/lexical_prefix_tree.py:11-67 (node 0 = "none"/OOV sink, node 1 = root, nodes numbered in pre-order with the
children visited in ascending sub-word id; `word_set_idx` = (first-1, last) of the word-id range under a node, the root
covers (0, len(word_dict)-1); words containing an unknown sub-word are skipped; the word dictionary is assumed to be in
lexical order so that every prefix owns a contiguous id range).

The trie is built directly in flat arrays (edge dictionary keyed by (parent, sub-word)) and then renumbered."""
from typing import Callable, List

import numpy as np
import torch


from ..data.encoders import tokenize  # noqa: F401  (espresso/tools/utils.py:tokenize)


class TensorizedPrefixTree:
    none_id, root_id = 0, 1

    def __init__(self, children, prev_subword_idx, word_idx, word_set_idx):
        self.children, self.prev_subword_idx, self.word_idx, self.word_set_idx = children, prev_subword_idx, word_idx, word_set_idx
        self._dev = {}

    def max_out_degree(self) -> int:
        return self.children.shape[1]

    def device_tensors(self, device):
        """int32 device copies in the layout csrc/lookahead.hip reads."""
        t = self._dev.get(device)
        if t is None:
            t = tuple(x.to(device=device, dtype=torch.int32).contiguous()
                      for x in (self.children, self.prev_subword_idx, self.word_idx, self.word_set_idx))
            self._dev[device] = t
        return t

    @staticmethod
    def build(word_dict, subword_dict, subword_tokenizer: Callable[[str], List[str]] = None):
        special = {word_dict.pad(), word_dict.eos(), word_dict.unk()}
        assert 0 in special
        edges = {}                      # (parent tmp id, sub-word id) -> tmp id ; tmp id 0 is the root
        w_end, lo, hi = [-1], [None], [None]
        for widx in range(len(word_dict)):
            if widx in special:
                continue
            word = word_dict[widx]
            subs = subword_tokenizer(word) if subword_tokenizer is not None else list(word)
            ids = [subword_dict.index(s) for s in subs]
            if any(i == subword_dict.unk() for i in ids):
                continue
            cur = 0
            for k, sidx in enumerate(ids):
                nxt = edges.get((cur, sidx))
                if nxt is None:
                    nxt = len(w_end)
                    edges[(cur, sidx)] = nxt
                    w_end.append(-1)
                    lo.append(widx - 1)
                    hi.append(widx)
                else:
                    lo[nxt] = min(lo[nxt], widx - 1)
                    hi[nxt] = max(hi[nxt], widx)
                if k == len(ids) - 1:
                    w_end[nxt] = widx
                cur = nxt
        kids = [[] for _ in w_end]
        for (par, sidx), child in edges.items():
            kids[par].append((sidx, child))
        for k in kids:
            k.sort()
        # pre-order numbering, children ascending by sub-word id; final ids start at 1 (0 is the none node)
        order, stack = [], [0]
        while stack:
            cur = stack.pop()
            order.append(cur)
            for _, child in reversed(kids[cur]):
                stack.append(child)
        new_id = {tmp: i + 1 for i, tmp in enumerate(order)}
        n_nodes = len(order) + 1
        D = max(len(k) for k in kids)
        children = np.zeros((n_nodes, D), dtype=np.int64)
        prev_sub = np.full((n_nodes,), subword_dict.pad(), dtype=np.int64)
        word_idx = np.full((n_nodes,), -1, dtype=np.int64)
        word_set = np.full((n_nodes, 2), word_dict.pad(), dtype=np.int64)
        for tmp in order:
            nid = new_id[tmp]
            for i, (sidx, child) in enumerate(kids[tmp]):
                children[nid, i] = new_id[child]
                prev_sub[new_id[child]] = sidx
            word_idx[nid] = w_end[tmp]
            word_set[nid] = (0, len(word_dict) - 1) if lo[tmp] is None else (lo[tmp], hi[tmp])
        return TensorizedPrefixTree(torch.from_numpy(children), torch.from_numpy(prev_sub), torch.from_numpy(word_idx),
                                    torch.from_numpy(word_set))
