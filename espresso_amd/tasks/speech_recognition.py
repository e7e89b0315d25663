"""`speech_recognition_espresso` task — the drop-in boundary with the contract of
espresso/tasks/speech_recognition.py:272-687 for the parts the hot path touches: dictionaries
(blank handling :336-354), feat_dim / feat_in_channels, build_model / build_criterion through the
registries, train_step / valid_step, max_positions.  Datasets hand over RAW waveforms; the task's
`prepare_sample` hook runs the fused GPU front-end (fbank + CMVN + SpecAugment + padding) right
before `model(**net_input)`, so `net_input` keeps the reference's keys (src_tokens, src_lengths)."""
from dataclasses import dataclass, field
from typing import Optional

import torch

from .. import registry
from ..data.asr_dictionary import AsrDictionary
from ..data.feature_transforms import AdaptiveSpecAugmentTransform, GlobalCMVN
from ..data.gpu_frontend import GpuFbankFrontend


@dataclass
class SpeechRecognitionEspressoConfig:
    data: Optional[str] = None
    dict: Optional[str] = None
    max_source_positions: int = 3600
    max_target_positions: int = 200
    autoregressive: bool = False
    is_wordlm: bool = False
    global_cmvn_stats_path: Optional[str] = None
    specaugment_config: Optional[str] = None
    feat_dim: int = 80
    feat_in_channels: int = 1
    seed: int = 1


@registry.register_task("speech_recognition_espresso", dataclass=SpeechRecognitionEspressoConfig)
class SpeechRecognitionEspressoTask:
    def __init__(self, cfg, tgt_dict, feat_dim=80, word_dict=None):
        self.cfg = cfg
        self.tgt_dict = tgt_dict
        self.word_dict = word_dict
        self.feat_dim = feat_dim
        self.feat_in_channels = cfg.feat_in_channels
        # CTC / transducer (non-autoregressive target side) use "<s>" as the blank symbol
        self.blank_symbol = tgt_dict.bos_word if not cfg.autoregressive else None
        self.frontend = None
        self.epoch = 1

    @classmethod
    def setup_task(cls, cfg, tgt_dict=None):
        if tgt_dict is None:
            tgt_dict = AsrDictionary.load(cfg.dict, enable_bos=not cfg.autoregressive)
        return cls(cfg, tgt_dict, feat_dim=cfg.feat_dim)

    @property
    def target_dictionary(self):
        return self.tgt_dict

    @property
    def source_dictionary(self):
        return None

    @property
    def word_dictionary(self):
        return self.word_dict

    def max_positions(self):
        return (self.cfg.max_source_positions, self.cfg.max_target_positions)

    def build_model(self, model_cfg, model_name="speech_transformer_encoder_model"):
        return registry.MODEL_REGISTRY[model_name].build_model(model_cfg, self)

    def build_criterion(self, name="ctc_loss", **kwargs):
        return registry.CRITERION_REGISTRY[name](self, **kwargs)

    def build_frontend(self, device, cmvn: Optional[GlobalCMVN] = None):
        specaug = None
        if self.cfg.specaugment_config:
            sc = self.cfg.specaugment_config
            sc = eval(sc) if isinstance(sc, str) else sc  # the reference evals the same string (speech_recognition.py:206-210)
            specaug = AdaptiveSpecAugmentTransform.from_config_dict(sc)
        if cmvn is None and self.cfg.global_cmvn_stats_path:
            cmvn = GlobalCMVN(self.cfg.global_cmvn_stats_path)
        self.frontend = GpuFbankFrontend(device, num_mel_bins=self.feat_dim, cmvn=cmvn, specaug=specaug, seed=self.cfg.seed)
        return self.frontend

    def begin_epoch(self, epoch, model=None):
        self.epoch = epoch

    def prepare_sample(self, sample, train=True):
        """Run the GPU front-end when the batch carries raw audio (`wav`, `wav_offsets`, `num_samples`)."""
        if "wav" not in sample:
            return sample
        feat, lengths, _ = self.frontend(sample["wav"], sample["wav_offsets"], sample["num_samples"], train=train,
                                         epoch=self.epoch, indices=sample.get("id_list"))
        out = dict(sample)
        out["net_input"] = dict(sample.get("net_input", {}))
        out["net_input"]["src_tokens"] = feat
        out["net_input"]["src_lengths"] = lengths.to(torch.long)
        return out

    def train_step(self, sample, model, criterion, update_num=0):
        model.train()
        sample = self.prepare_sample(sample, train=True)
        loss, sample_size, logging_output = criterion(model, sample)
        loss.backward()
        return loss, sample_size, logging_output

    def valid_step(self, sample, model, criterion):
        model.eval()
        with torch.no_grad():
            sample = self.prepare_sample(sample, train=False)
            loss, sample_size, logging_output = criterion(model, sample)
        return loss, sample_size, logging_output
