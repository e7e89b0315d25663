"""WER / CER scorer with the counting rules of espresso/tools/wer.py:16-220: character-level counts on the token strings,
word-level counts after `dictionary.wordpiece_decode`, optional non-language-symbol and regex word filters, accumulated with
espresso/tools/utils.py edit_distance (corr > sub > ins > del backtrace priority)."""
import re
from collections import Counter, OrderedDict

from . import utils as speech_utils


def wordpiece_decode(x: str, space_word="<space>") -> str:
    """AsrDictionary.wordpiece_decode: character/word-piece tokens -> words (espresso/data/asr_dictionary.py)."""
    if space_word in x.split():
        return x.replace(" ", "").replace(space_word, " ").strip()
    return x.replace(" ", "").replace("▁", " ").strip()


class Scorer(object):
    def __init__(self, dictionary, wer_output_filter=None):
        self.dictionary = dictionary
        self.word_filters = []
        if wer_output_filter:
            with open(wer_output_filter, "r", encoding="utf-8") as f:
                for line in f:
                    line = line.strip()
                    if line.startswith("#!") or line == "":
                        continue
                    m = re.match(r"s/(.+)/(.*)/g", line) or re.match(r"s:(.+):(.*):g", line)
                    if m is not None:
                        self.word_filters.append([m.group(1), m.group(2)])
        self.reset()

    def reset(self):
        self.char_counter = Counter()
        self.word_counter = Counter()
        self.char_results = OrderedDict()
        self.results = OrderedDict()

    def _decode(self, s):
        d = self.dictionary
        if getattr(d, "bpe", None) is not None or getattr(d, "tokenizer", None) is not None:
            return d.wordpiece_decode(s)
        return wordpiece_decode(s, getattr(self.dictionary, "space_word", "<space>"))

    def add_prediction(self, utt_id, pred):
        assert isinstance(utt_id, str) and isinstance(pred, str)
        assert utt_id not in self.char_results, "Duplicated utterance id detected: {}".format(utt_id)
        self.char_results[utt_id] = pred + "\n"
        self.results[utt_id] = self._decode(pred) + "\n"

    def add_evaluation(self, utt_id, ref, pred):
        assert isinstance(utt_id, str) and isinstance(ref, str) and isinstance(pred, str)
        non_lang_syms = getattr(self.dictionary, "non_lang_syms", None)
        if non_lang_syms:
            ref = " ".join(x for x in ref.strip().split() if x not in non_lang_syms)
            pred = " ".join(x for x in pred.strip().split() if x not in non_lang_syms)
        _, _, counter = speech_utils.edit_distance(ref.strip().split(), pred.strip().split())
        self.char_counter += counter
        ref_words, pred_words = self._decode(ref), self._decode(pred)
        for pattern, repl in self.word_filters:
            ref_words, pred_words = re.sub(pattern, repl, ref_words), re.sub(pattern, repl, pred_words)
        _, _, counter = speech_utils.edit_distance(ref_words.split(), pred_words.split())
        self.word_counter += counter

    @staticmethod
    def _rates(c):
        """(error rate, substitution, insertion, deletion rates) in percent — the 4-tuple of espresso/tools/wer.py:117-149."""
        assert c["words"] > 0
        n = c["words"]
        return (float(c["sub"] + c["ins"] + c["del"]) / n * 100, float(c["sub"]) / n * 100, float(c["ins"]) / n * 100,
                float(c["del"]) / n * 100)

    def cer(self):
        return self._rates(self.char_counter)

    def wer(self):
        return self._rates(self.word_counter)

    def tot_word_error(self):
        c = self.word_counter
        return c["sub"] + c["ins"] + c["del"]

    def tot_char_error(self):
        c = self.char_counter
        return c["sub"] + c["ins"] + c["del"]

    def tot_word_count(self):
        return self.word_counter["words"]

    def tot_char_count(self):
        return self.char_counter["words"]

    def summary_lines(self):
        """The two result lines of espresso/speech_recognize.py:360-377."""
        return ["WER={:.2f}%, Sub={:.2f}%, Ins={:.2f}%, Del={:.2f}%, #words={:d}".format(*self.wer(), self.tot_word_count()),
                "CER={:.2f}%, Sub={:.2f}%, Ins={:.2f}%, Del={:.2f}%, #chars={:d}".format(*self.cer(), self.tot_char_count())]

    def print_stats(self):
        c, w = self.char_counter, self.word_counter
        return ("CER: {:.2f}%, WER: {:.2f}% ({} words: sub {} ins {} del {})".format(
            self.cer()[0], self.wer()[0], w["words"], w["sub"], w["ins"], w["del"]))
