_TOKENIZERS = {"none": None, "13a": None}
