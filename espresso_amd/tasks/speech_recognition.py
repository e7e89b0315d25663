"""`speech_recognition_espresso` task — the drop-in boundary with the contract of
espresso/tasks/speech_recognition.py:272-687 for the parts the hot path touches: dictionaries
(blank handling :336-354), feat_dim / feat_in_channels, build_model / build_criterion through the
registries, train_step / valid_step, max_positions.  Datasets hand over RAW waveforms; the task's
`prepare_sample` hook runs the fused GPU front-end (fbank + CMVN + SpecAugment + padding) right
before `model(**net_input)`, so `net_input` keeps the reference's keys (src_tokens, src_lengths)."""
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
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
    # the rest of espresso/tasks/speech_recognition.py:30-124 that the path consumes
    non_lang_syms: Optional[str] = None
    word_dict: Optional[str] = None
    wer_output_filter: Optional[str] = None
    criterion_name: Optional[str] = None  # the reference interpolates criterion._name here (speech_recognition.py:121)
    include_eos_in_transducer_loss: bool = False
    prepend_bos_as_input_feeding: bool = False
    batch_based_on_both_src_tgt: bool = False
    required_seq_len_multiple: int = 1
    train_subset: str = "train"
    valid_subset: str = "valid"
    bpe: Optional[str] = None
    sentencepiece_model: Optional[str] = None


def uses_blank(cfg) -> bool:
    """`<s>` is enabled and reserved as the blank for `ctc_loss` / `transducer_loss`: decided by the criterion's name alone, as
    the reference does (espresso/tasks/speech_recognition.py:324, 345-347); `autoregressive` plays no part in it."""
    return getattr(cfg, "criterion_name", None) in ("transducer_loss", "ctc_loss")


@registry.register_task("speech_recognition_espresso", dataclass=SpeechRecognitionEspressoConfig)
class SpeechRecognitionEspressoTask:
    def __init__(self, cfg, tgt_dict, feat_dim=80, word_dict=None):
        self.cfg = cfg
        self.tgt_dict = tgt_dict
        self.word_dict = word_dict
        self.feat_dim = feat_dim
        self.feat_in_channels = cfg.feat_in_channels
        # CTC / transducer use "<s>" as the blank symbol
        self.blank_symbol = tgt_dict.bos_word if uses_blank(cfg) else None
        self.frontend = None
        self.epoch = 1
        self.datasets = {}
        self.criterion = None
        self.decoder_for_validation = None
        # symbols never scored by WER (speech_recognition.py:332-334: blank in addition to </s> and <pad>)
        self.extra_symbols_to_ignore = {tgt_dict.pad()}
        if self.blank_symbol is not None:
            self.extra_symbols_to_ignore.add(tgt_dict.index(self.blank_symbol))

    @classmethod
    def setup_task(cls, cfg, tgt_dict=None):
        if tgt_dict is None:
            tgt_dict = cls.load_dictionary(cfg.dict, enable_bos=uses_blank(cfg), non_lang_syms=cfg.non_lang_syms)
        word_dict = None
        if getattr(cfg, "word_dict", None):
            word_dict = cls.load_dictionary(cfg.word_dict, enable_bos=False)
        if getattr(cfg, "bpe", None):
            tgt_dict.build_bpe(cfg.bpe, getattr(cfg, "sentencepiece_model", None))
        return cls(cfg, tgt_dict, feat_dim=cfg.feat_dim, word_dict=word_dict)

    @classmethod
    def load_dictionary(cls, filename, enable_bos=False, non_lang_syms=None):
        """speech_recognition.py:297-306: `<s>` is enabled (and doubles as the blank) for CTC / transducer targets."""
        return AsrDictionary.load(filename, enable_bos=enable_bos, f_non_lang_syms=non_lang_syms)

    def build_bpe(self, name=None, sentencepiece_model=None):
        return self.tgt_dict.build_bpe(name, sentencepiece_model)

    def load_dataset(self, split: str, epoch=1, combine=False, pin_memory=True):
        """speech_recognition.py:414-469: `<data>/<split>.json`; shuffling only for the training split."""
        from ..data.asr_dataset import get_asr_dataset_from_json

        train = split == getattr(self.cfg, "train_subset", "train")
        transducer = getattr(self.cfg, "criterion_name", None) == "transducer_loss"
        ds = get_asr_dataset_from_json(
            self.cfg.data, split, self.tgt_dict, combine=combine, shuffle=train,
            pad_to_multiple=getattr(self.cfg, "required_seq_len_multiple", 1), autoregressive=self.cfg.autoregressive,
            # :449-454: the transducer recipes feed <s> ... when </s> is part of the loss, and batch by frames x tokens
            prepend_bos_as_input_feeding=(getattr(self.cfg, "prepend_bos_as_input_feeding", False)
                                          or (transducer and getattr(self.cfg, "include_eos_in_transducer_loss", False))),
            batch_based_on_both_src_tgt=getattr(self.cfg, "batch_based_on_both_src_tgt", False) or transducer,
            pin_memory=pin_memory and torch.cuda.is_available())
        self.datasets[split] = ds
        self.feat_dim = ds.src.feat_dim
        if train and ds.tgt is not None:  # :462-469: counts of </s> and <unk> from the training targets (unigram label smoothing)
            self.tgt_dict.count[self.tgt_dict.eos()] = len(ds.tgt)
            unk = self.tgt_dict.unk()
            self.tgt_dict.count[unk] = int(sum(int((ds.tgt[i][0] == unk).sum()) for i in range(len(ds.tgt))))
        return ds

    def dataset(self, split):
        return self.datasets[split]

    def get_batches(self, dataset, max_tokens=None, max_sentences=None, max_positions=None, seed=1, epoch=1, num_shards=1,
                    shard_id=0, shuffle=True, bsz_mult=1) -> List[np.ndarray]:
        """Batch plan of fairseq's `get_batch_iterator` + `EpochBatchIterator` (fairseq/tasks/fairseq_task.py:285-306,
        fairseq/data/iterators.py: shuffle with `seed + epoch`, then rank `shard_id` takes every `num_shards`-th
        batch; short ranks get empty batches so that every rank runs the same number of steps)."""
        from ..data.data_utils import batch_by_size, numpy_seed

        with numpy_seed(seed):
            indices = dataset.ordered_indices()
        if max_positions is not None:
            indices, _ = dataset.filter_indices_by_size(indices, max_positions)
        batches = batch_by_size(indices, dataset.num_tokens_vec(indices), max_tokens=max_tokens, max_sentences=max_sentences,
                                bsz_mult=bsz_mult)
        if shuffle:
            with numpy_seed(seed + epoch):
                np.random.shuffle(batches)
        if num_shards > 1:
            n = (len(batches) + num_shards - 1) // num_shards
            mine = batches[shard_id::num_shards]
            batches = mine + [np.zeros(0, dtype=np.int64)] * (n - len(mine))
        return batches

    def to_device(self, sample, device):
        """Asynchronous H2D of a collated batch (the raw-audio buffer is pinned by the collater)."""
        def mv(x):
            if torch.is_tensor(x):
                return x.to(device, non_blocking=True)
            if isinstance(x, dict):
                return {k: mv(v) for k, v in x.items()}
            return x
        return {k: mv(v) for k, v in sample.items()}

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

    def build_criterion(self, name=None, **kwargs):
        name = name or getattr(self.cfg, "criterion_name", None) or "ctc_loss"
        self.cfg.criterion_name = name
        self.criterion = registry.CRITERION_REGISTRY[name](self, **kwargs)
        return self.criterion

    def build_generator(self, models, args=None, seq_gen_cls=None, extra_gen_cls_kwargs=None, lm_model=None):
        """speech_recognition.py:526-596: transducer greedy / beam search, CTC greedy, or the attention beam search,
        chosen from the criterion the task was configured with.  `args`: namespace with the generation options."""
        g = lambda k, d=None: getattr(args, k, d) if args is not None else d  # noqa: E731
        extra = dict(extra_gen_cls_kwargs or {})
        if g("print_alignment", False):
            extra["print_alignment"] = True
        crit = getattr(self.cfg, "criterion_name", "ctc_loss")
        if crit == "transducer_loss":
            from ..tools.transducer_beam_search_decoder import TransducerBeamSearchDecoder
            from ..tools.transducer_greedy_decoder import TransducerGreedyDecoder

            if seq_gen_cls is None:
                seq_gen_cls = TransducerGreedyDecoder if g("beam", 1) == 1 else TransducerBeamSearchDecoder
            include_eos = getattr(self.cfg, "include_eos_in_transducer_loss", False)
            kw = dict(temperature=g("temperature", 1.0), max_num_expansions_per_step=g("transducer_max_num_expansions_per_step", 20),
                      bos=self.tgt_dict.bos() if include_eos else self.tgt_dict.eos(),
                      blank=self.tgt_dict.index(self.blank_symbol), model_predicts_eos=include_eos)
            if seq_gen_cls is not TransducerGreedyDecoder:
                kw.update(beam_size=g("beam", 1), normalize_scores=not g("unnormalized", False),
                          expansion_beta=g("transducer_expansion_beta", 0), expansion_gamma=g("transducer_expansion_gamma", None),
                          prefix_alpha=g("transducer_prefix_alpha", None))
            if lm_model is not None:
                kw.update(lm_model=lm_model, lm_weight=g("lm_weight", 0.0))
            kw.update(extra)
            return seq_gen_cls(models, self.tgt_dict, **kw)
        if crit == "ctc_loss":
            from ..tools.ctc_decoder import CTCDecoder

            return (seq_gen_cls or CTCDecoder)(models, self.tgt_dict, beam_size=g("beam", 1), **extra)
        from ..sequence_generator import SequenceGenerator

        kw = dict(beam_size=g("beam", 5), max_len_a=g("max_len_a", 0.0), max_len_b=g("max_len_b", 200), min_len=g("min_len", 1),
                  normalize_scores=not g("unnormalized", False), len_penalty=g("lenpen", 1.0), unk_penalty=g("unkpen", 0.0),
                  temperature=g("temperature", 1.0), lm_model=lm_model, lm_weight=g("lm_weight", 0.0), eos_factor=g("eos_factor", None))
        kw.update(extra)
        return (seq_gen_cls or SequenceGenerator)(models, self.tgt_dict, **kw)

    def build_validation_decoder(self, model):
        """The greedy decoder the reference attaches for WER during validation (speech_recognition.py:497-517)."""
        crit = getattr(self.cfg, "criterion_name", "ctc_loss")
        if crit == "ctc_loss":
            from ..tools.ctc_decoder import CTCDecoder

            self.decoder_for_validation = CTCDecoder([model], self.tgt_dict)
        elif crit == "transducer_loss":
            from ..tools.transducer_greedy_decoder import TransducerGreedyDecoder

            include_eos = getattr(self.cfg, "include_eos_in_transducer_loss", False)
            self.decoder_for_validation = TransducerGreedyDecoder(
                [model], self.tgt_dict, max_num_expansions_per_step=20,
                bos=self.tgt_dict.bos() if include_eos else self.tgt_dict.eos(), blank=self.tgt_dict.index(self.blank_symbol),
                model_predicts_eos=include_eos)
        else:
            from ..tools.simple_greedy_decoder import SimpleGreedyDecoder

            self.decoder_for_validation = SimpleGreedyDecoder([model], self.tgt_dict, for_validation=True)
        return self.decoder_for_validation

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
        if self.criterion is not None and hasattr(self.criterion, "set_epoch"):
            self.criterion.set_epoch(epoch)  # speech_recognition.py:609-613

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
            if self.decoder_for_validation is not None:
                (logging_output["word_error"], logging_output["word_count"], logging_output["char_error"],
                 logging_output["char_count"]) = self._inference_with_wer(self.decoder_for_validation, sample, model)
        return loss, sample_size, logging_output

    def _inference_with_wer(self, decoder, sample, model):
        """speech_recognition.py:663-687."""
        from ..tools.wer import Scorer

        scorer = Scorer(self.tgt_dict, wer_output_filter=getattr(self.cfg, "wer_output_filter", None))
        tokens, _, _ = decoder.decode([model], sample)
        pred = tokens.cpu()
        assert pred.size(0) == sample["target"].size(0)
        for i in range(pred.size(0)):
            utt_id = sample["utt_id"][i]
            ref_tokens = self.tgt_dict.wordpiece_encode(sample["text"][i])
            pred_tokens = self.tgt_dict.string(pred[i], extra_symbols_to_ignore=self.extra_symbols_to_ignore)
            scorer.add_evaluation(utt_id, ref_tokens, pred_tokens)
        return scorer.tot_word_error(), scorer.tot_word_count(), scorer.tot_char_error(), scorer.tot_char_count()

    def reduce_metrics(self, logging_outputs, criterion=None):
        """Sum over the data-parallel workers' logging outputs; adds wer / cer (speech_recognition.py:615-629)."""
        crit = criterion if criterion is not None else self.criterion
        r = crit.reduce_metrics(logging_outputs) if crit is not None and hasattr(crit, "reduce_metrics") else None
        out = dict(r) if r is not None else {}  # (a fairseq criterion logs into fairseq's aggregators itself and returns None)
        tot = {k: sum(log.get(k, 0) for log in logging_outputs) for k in ("word_error", "word_count", "char_error", "char_count")}
        if tot["word_count"] > 0:
            out["wer"] = float(tot["word_error"]) / tot["word_count"] * 100
        if tot["char_count"] > 0:
            out["cer"] = float(tot["char_error"]) / tot["char_count"] * 100
        return out
