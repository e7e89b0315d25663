"""AsrDictionary — symbol/index layout of espresso/data/asr_dictionary.py:18-142 on top of the
fairseq Dictionary contract (fairseq/data/dictionary.py): with a blank symbol the specials are
<s>=0 (blank), <pad>=1, </s>=2, <unk>=3, <space> appended; without: <pad>=0, </s>=1, <unk>=2."""
from typing import List

import torch


class AsrDictionary:
    def __init__(self, bos="<s>", pad="<pad>", eos="</s>", unk="<unk>", space="<space>", enable_bos=False,
                 extra_special_symbols=None):
        self.bos_word, self.unk_word, self.pad_word, self.eos_word, self.space_word = bos, unk, pad, eos, space
        self.symbols: List[str] = []
        self.count: List[int] = []
        self.indices = {}
        if enable_bos:
            self.bos_index = self.add_symbol(bos)  # count 1, as the reference's add_symbol default (asr_dictionary.py:43)
        self.pad_index = self.add_symbol(pad, n=0)
        self.eos_index = self.add_symbol(eos, n=0)
        self.unk_index = self.add_symbol(unk, n=0)
        if not enable_bos:
            self.bos_index = self.eos_index  # reference: bos() falls back when <s> is not a symbol
        if extra_special_symbols:
            for s in extra_special_symbols:
                self.add_symbol(s, n=0)
        self.nspecial = len(self.symbols)
        self.space_index = -1
        self.non_lang_syms = None
        self.tokenizer = None
        self.bpe = None

    def __len__(self):
        return len(self.symbols)

    def __getitem__(self, idx):
        return self.symbols[idx] if idx < len(self.symbols) else self.unk_word

    def __contains__(self, sym):
        return sym in self.indices

    def index(self, sym):
        return self.indices.get(sym, self.unk_index)

    def add_symbol(self, word, n=1, overwrite=False):
        if word in self.indices and not overwrite:
            idx = self.indices[word]
            self.count[idx] += n
            return idx
        idx = len(self.symbols)
        self.indices[word] = idx
        self.symbols.append(word)
        self.count.append(n)
        return idx

    def bos(self):
        return self.bos_index

    def pad(self):
        return self.pad_index

    def eos(self):
        return self.eos_index

    def unk(self):
        return self.unk_index

    def space(self):
        return self.space_index

    @classmethod
    def load(cls, f, enable_bos=False, f_non_lang_syms=None):
        """Load from a `<symbol> <count>` text file (espresso/data/asr_dictionary.py:91-123)."""
        d = cls(enable_bos=enable_bos)
        lines = open(f, encoding="utf-8").read().splitlines() if isinstance(f, str) else list(f)
        for line in lines:
            if not line.strip():
                continue
            word, *rest = line.rstrip().rsplit(" ", 1)
            d.add_symbol(word, n=int(rest[0]) if rest else 1)
        d.space_index = d.indices.get(d.space_word, -1)
        d.non_lang_syms = None
        if f_non_lang_syms is not None:
            d.non_lang_syms = [x.rstrip() for x in open(f_non_lang_syms, encoding="utf-8") if x.strip()]
            for sym in d.non_lang_syms:
                assert d.index(sym) != d.unk(), "{} in {} is not in the dictionary".format(sym, f_non_lang_syms)
        return d

    def build_bpe(self, name=None, sentencepiece_model=None):
        """espresso/data/asr_dictionary.py:118-128 — `characters_asr` gets this dictionary's space / non-language symbols."""
        from .encoders import build_bpe

        self.bpe = build_bpe(name, self, sentencepiece_model)
        return self.bpe

    def wordpiece_encode(self, x: str) -> str:
        if self.tokenizer is not None:
            x = self.tokenizer.encode(x)
        if self.bpe is not None:
            x = self.bpe.encode(x)
        return x

    def wordpiece_decode(self, x: str) -> str:
        if self.bpe is not None:
            x = self.bpe.decode(x)
        if self.tokenizer is not None:
            x = self.tokenizer.decode(x)
        return x

    def save(self, f):
        """`<symbol> <count>` lines for the non-special symbols (fairseq/data/dictionary.py:save)."""
        with open(f, "w", encoding="utf-8") as fd:
            for s, c in zip(self.symbols[self.nspecial:], self.count[self.nspecial:]):
                print(f"{s} {c}", file=fd)

    @classmethod
    def from_symbols(cls, symbols, enable_bos=False, add_space=True):
        """add_space=False: word-level dictionaries have no <space> symbol (space_index = -1, asr_dictionary.py:86)."""
        d = cls(enable_bos=enable_bos)
        for s in symbols:
            d.add_symbol(s)
        if add_space:
            d.space_index = d.add_symbol(d.space_word) if d.space_word not in d.indices else d.indices[d.space_word]
        else:
            d.space_index = d.indices.get(d.space_word, -1)
        d.non_lang_syms = None
        return d

    def string(self, tensor, bpe_symbol=None, escape_unk=False, extra_symbols_to_ignore=None):
        ignore = set(extra_symbols_to_ignore or [])
        ignore.add(self.eos())
        ignore.add(self.pad())
        if torch.is_tensor(tensor) and tensor.dim() == 2:
            return "\n".join(self.string(t, bpe_symbol, escape_unk, extra_symbols_to_ignore) for t in tensor)
        toks = [self[int(i)] for i in tensor if int(i) not in ignore and int(i) != (self.bos_index if self.bos_word in self.indices else -1)]
        return " ".join(toks)

    def encode_line(self, line, append_eos=True):
        words = line.strip().split()
        ids = [self.index(w) for w in words]
        if append_eos:
            ids.append(self.eos_index)
        return torch.tensor(ids, dtype=torch.long)
