class BLEU:
    TOKENIZERS = {"none": None, "13a": None, "intl": None, "zh": None, "char": None}
