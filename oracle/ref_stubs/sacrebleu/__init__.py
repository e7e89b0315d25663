__version__ = "2.0.0"


class BLEU:
    TOKENIZERS = {"none": None, "13a": None, "intl": None, "zh": None, "char": None}
