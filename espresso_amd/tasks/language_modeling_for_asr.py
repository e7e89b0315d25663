"""`language_modeling_for_asr` task — espresso/tasks/language_modeling_for_asr.py:27-117 over
fairseq/tasks/language_modeling.py:37-262 for what the LM recipe (`lstm_lm_librispeech.yaml`) uses: AsrDictionary, one
`<data>/<split>` token file per subset, "future" targets, `sample_break_mode` eos / none.  The LSTM LM it trains is the one
decode-time fusion consumes (`models/lstm_lm.py`).  Same interface subset as the speech-recognition task, so the training
entry point (`speech_train.py`) drives both."""
from dataclasses import dataclass
from typing import Optional

import torch

from .. import registry
from ..data.asr_dictionary import AsrDictionary
from ..data.lm_dataset import MonolingualDataset, load_token_file
from .speech_recognition import SpeechRecognitionEspressoTask


@dataclass
class LanguageModelingForASRConfig:
    data: Optional[str] = None
    dict: Optional[str] = None
    sample_break_mode: str = "none"
    tokens_per_sample: int = 1024
    output_dictionary_size: int = -1
    seed: int = 1
    train_subset: str = "train"
    valid_subset: str = "valid"
    criterion_name: str = "cross_entropy"


@registry.register_task("language_modeling_for_asr", dataclass=LanguageModelingForASRConfig)
class LanguageModelingForASRTask:
    def __init__(self, cfg, dictionary):
        if cfg.output_dictionary_size >= 0:
            raise NotImplementedError("output_dictionary_size (truncated output vocabulary) is not used by the ASR recipes")
        self.cfg, self.dictionary = cfg, dictionary
        self.datasets, self.criterion, self.epoch = {}, None, 1
        self.blank_symbol = None
        self.decoder_for_validation = None

    @classmethod
    def setup_task(cls, cfg, dictionary=None):
        if dictionary is None:
            import os

            path = cfg.dict if cfg.dict is not None else os.path.join(cfg.data.split(":")[0], "dict.txt")
            dictionary = AsrDictionary.load(path)
        return cls(cfg, dictionary)

    @property
    def target_dictionary(self):
        return self.dictionary

    @property
    def source_dictionary(self):
        return self.dictionary

    def load_dataset(self, split, epoch=1, combine=False, **unused):
        import os

        paths = self.cfg.data.split(":")
        prefix = os.path.join(paths[(epoch - 1) % len(paths)], split)
        tokens = load_token_file(prefix, self.dictionary)
        self.datasets[split] = MonolingualDataset(tokens, self.dictionary, self.cfg.sample_break_mode, self.cfg.tokens_per_sample,
                                                  shuffle=True)
        return self.datasets[split]

    def dataset(self, split):
        return self.datasets[split]

    # batch plan / device transfer: same rules as the speech task (fairseq_task.get_batch_iterator)
    get_batches = SpeechRecognitionEspressoTask.get_batches
    to_device = SpeechRecognitionEspressoTask.to_device

    def max_positions(self):
        return int(self.cfg.tokens_per_sample)

    def build_model(self, model_cfg, model_name="lstm_lm_espresso"):
        return registry.MODEL_REGISTRY[model_name].build_model(model_cfg, self)

    def build_criterion(self, name=None, **kwargs):
        name = name or self.cfg.criterion_name
        self.cfg.criterion_name = name
        self.criterion = registry.CRITERION_REGISTRY[name](self, **kwargs)
        return self.criterion

    def build_frontend(self, device, cmvn=None):
        return None

    def build_validation_decoder(self, model):
        return None

    def begin_epoch(self, epoch, model=None):
        self.epoch = epoch

    def prepare_sample(self, sample, train=True):
        return sample

    def valid_step(self, sample, model, criterion):
        model.eval()
        with torch.no_grad():
            return criterion(model, sample)

    def reduce_metrics(self, logging_outputs, criterion=None):
        crit = criterion if criterion is not None else self.criterion
        return dict(crit.reduce_metrics(logging_outputs))
