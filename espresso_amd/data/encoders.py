"""Target-side text encoders used by the ASR dictionary (espresso/data/asr_dictionary.py:118-142).

`characters_asr` (espresso/data/encoders/characters_asr.py:19-41): characters separated by blanks, blanks become
`<space>`, non-language symbols stay whole, sentences end with `<space>`.  `sentencepiece` wraps a trained model file
(fairseq/data/encoders/sentencepiece_bpe.py) — the LibriSpeech recipes use a 5000-piece unigram model."""
from typing import List, Optional


def tokenize(sent: str, space: str = "<space>", non_lang_syms: Optional[List[str]] = None) -> str:
    """espresso/tools/utils.py:36-58.  At every position the non-language symbols are tried in the order given (the
    reference's regex alternation), then a single character is consumed."""
    sent = " ".join(sent.strip().split())
    syms = [s for s in (non_lang_syms or []) if s]
    out, i = [], 0
    while i < len(sent):
        for s in syms:
            if sent.startswith(s, i):
                out.append(s)
                i += len(s)
                break
        else:
            out.append(space if sent[i] == " " else sent[i])
            i += 1
    return " ".join(out)


class CharactersAsr:
    def __init__(self, space_symbol="<space>", ends_with_space=True, non_lang_syms: Optional[List[str]] = None):
        self.space_symbol, self.ends_with_space, self.non_lang_syms = space_symbol, ends_with_space, non_lang_syms

    def encode(self, x: str) -> str:
        y = tokenize(x, space=self.space_symbol, non_lang_syms=self.non_lang_syms)
        return y + " " + self.space_symbol if self.ends_with_space else y

    def decode(self, x: str) -> str:
        return x.replace(" ", "").replace(self.space_symbol, " ").strip()


class SentencepieceBPE:
    """encode: pieces joined by blanks; decode: drop blanks, U+2581 -> blank (sentencepiece_bpe.py:40-48)."""

    def __init__(self, sentencepiece_model: str):
        import sentencepiece as spm

        self.sp = spm.SentencePieceProcessor()
        self.sp.Load(sentencepiece_model)

    def encode(self, x: str) -> str:
        return " ".join(self.sp.EncodeAsPieces(x))

    def decode(self, x: str) -> str:
        return x.replace(" ", "").replace("▁", " ").strip()


def build_bpe(name: Optional[str], dictionary=None, sentencepiece_model: Optional[str] = None):
    if name in (None, "", "none"):
        return None
    if name == "characters_asr":
        return CharactersAsr(space_symbol=dictionary.space_word if dictionary is not None else "<space>",
                             non_lang_syms=getattr(dictionary, "non_lang_syms", None))
    if name == "sentencepiece":
        if not sentencepiece_model:
            raise ValueError("--bpe sentencepiece needs --sentencepiece-model")
        return SentencepieceBPE(sentencepiece_model)
    raise ValueError(f"unknown bpe '{name}' (characters_asr | sentencepiece)")
