"""ASR datasets and the collater — the data side of the drop-in boundary.

Mirrors espresso/data/feat_text_dataset.py (AudioFeatDataset :36-165, AsrTextDataset :333-395),
espresso/data/asr_dataset.py (collate :17-136, AsrDataset :139-450) and the json loader
espresso/tasks/speech_recognition.py:127-269.  One deliberate difference: a `wave`/`command` source is NOT turned
into features on the host.  `AudioWaveDataset` yields raw int16-scale samples and the collater packs them as
`wav` (concatenated fp32, pinned), `wav_offsets`, `num_samples`, `id_list`; the task's `prepare_sample` runs the fused
HIP front-end (fbank + CMVN + SpecAugment + padding) on the GPU and fills `net_input["src_tokens"]` there.
Every other key of the batch (`id`, `utt_id`, `nsentences`, `ntokens`, `net_input.src_lengths`,
`net_input.prev_output_tokens`, `target`, `text`), the descending-length sort and the sizes used for batching are the
reference's.  Pre-computed features (`feat` entries, Kaldi ark) go through `AudioFeatDataset` unchanged."""
import itertools
import json
import os
import re
from collections import OrderedDict
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from ..tools.utils import collate_frames
from . import audio_utils, kaldi_io
from .data_utils import collate_tokens

FRAME_LEN, FRAME_SHIFT = 400, 160  # 25 ms / 10 ms at 16 kHz (espresso/tools/utils.py:478-486)


def samples_to_frames(n: int) -> int:
    return 0 if n < FRAME_LEN else 1 + (n - FRAME_LEN) // FRAME_SHIFT


class _AudioBase:
    def check_index(self, i):
        if i < 0 or i >= self.size:
            raise IndexError("index out of range")

    def filter_and_reorder(self, indices):
        indices = np.array(indices)
        assert all(indices < len(self.utt_ids)) and all(indices >= 0)
        assert len(np.unique(indices)) == len(indices), "Duplicate elements in indices."
        self.utt_ids = [self.utt_ids[i] for i in indices]
        self.rxfiles = [self.rxfiles[i] for i in indices]
        self.sizes = self.sizes[indices]
        self.size = len(self.utt_ids)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.size


class AudioWaveDataset(_AudioBase):
    """Raw waveforms: WAV paths, `cmd |` pipes, or in-memory float arrays.  `sizes` are frame counts (what
    `--max-tokens` budgets), from `utt2num_frames` when given, else from the WAV headers."""

    is_wave = True

    def __init__(self, utt_ids: List[str], rxfiles: Sequence[Union[str, np.ndarray]], utt2num_frames: Optional[List[int]] = None,
                 feat_dim: int = 80, sample_rate: int = 16000):
        assert len(utt_ids) == len(rxfiles)
        self.utt_ids, self.rxfiles, self.size = list(utt_ids), list(rxfiles), len(utt_ids)
        self.feat_dim, self.sample_rate, self.epoch = feat_dim, sample_rate, 1
        self._nsamp = {}  # sample counts read from the file headers (lazy mode needs them before the files are decoded)
        if utt2num_frames is not None and len(utt2num_frames) > 0:
            assert len(utt2num_frames) == self.size
            sizes = utt2num_frames
        else:
            sizes = [samples_to_frames(len(r) if isinstance(r, np.ndarray) else audio_utils.num_samples(r)) for r in self.rxfiles]
        self.sizes = np.array(sizes, dtype=np.int32)

    # lazy mode (set by the training loop): file entries are returned as `audio_utils.LazyWave` placeholders and the collater
    # decodes the whole batch in parallel into its pinned int16 buffer; `num_workers` = decoder threads (dataset.num_workers)
    lazy = False
    num_workers = 1

    def num_samples(self, i) -> int:
        n = self._nsamp.get(i)
        if n is None:
            r = self.rxfiles[i]
            n = len(r) if isinstance(r, np.ndarray) else audio_utils.num_samples(r)
            self._nsamp[i] = n
        return n

    def __getitem__(self, i):
        self.check_index(i)
        r = self.rxfiles[i]
        if isinstance(r, np.ndarray):
            return r.astype(np.float32, copy=False)
        if self.lazy and audio_utils._is_file(r):
            return audio_utils.LazyWave(r, self.num_samples(i))
        x, sr = audio_utils.get_waveform(r)
        assert sr == self.sample_rate, f"{self.utt_ids[i]}: {sr} Hz, the front-end tables are built for {self.sample_rate} Hz"
        return x


class AudioFeatDataset(_AudioBase):
    """Pre-computed features: Kaldi `ark:offset` entries or in-memory `(T, F)` arrays."""

    is_wave = False

    def __init__(self, utt_ids: List[str], rxfiles: Sequence[Union[str, np.ndarray]], utt2num_frames: Optional[List[int]] = None):
        assert len(utt_ids) == len(rxfiles)
        self.utt_ids, self.rxfiles, self.size, self.epoch = list(utt_ids), list(rxfiles), len(utt_ids), 1
        first = self.rxfiles[0]
        self.feat_dim = first.shape[1] if isinstance(first, np.ndarray) else kaldi_io.read_mat_shape(first)[1]
        if utt2num_frames is not None and len(utt2num_frames) > 0:
            assert len(utt2num_frames) == self.size
            sizes = utt2num_frames
        else:
            sizes = [r.shape[0] if isinstance(r, np.ndarray) else kaldi_io.read_mat_shape(r)[0] for r in self.rxfiles]
        self.sizes = np.array(sizes, dtype=np.int32)

    def __getitem__(self, i) -> torch.Tensor:
        self.check_index(i)
        r = self.rxfiles[i]
        return torch.from_numpy(np.ascontiguousarray(r if isinstance(r, np.ndarray) else kaldi_io.read_mat(r))).float()


class AsrTextDataset:
    """Tokenised + tensorised transcripts, both forms kept (feat_text_dataset.py:333-395)."""

    def __init__(self, utt_ids: List[str], texts: List[str], dictionary=None, append_eos=True):
        assert len(utt_ids) == len(texts)
        self.utt_ids, self.texts, self.dictionary, self.append_eos = list(utt_ids), list(texts), dictionary, append_eos
        self.size = len(self.utt_ids)
        if dictionary is not None:
            sizes = [len(dictionary.wordpiece_encode(t).split()) + (1 if append_eos else 0) for t in texts]
        else:
            sizes = [len(t.split()) for t in texts]
        self.sizes = np.array(sizes, dtype=np.int32)

    def filter_and_reorder(self, indices):
        indices = np.array(indices)
        assert all(indices < self.size) and all(indices >= 0)
        assert len(np.unique(indices)) == len(indices), "Duplicate elements in indices."
        self.utt_ids = [self.utt_ids[i] for i in indices]
        self.texts = [self.texts[i] for i in indices]
        self.sizes = self.sizes[indices]
        self.size = len(self.utt_ids)

    def __getitem__(self, i):
        if i < 0 or i >= self.size:
            raise IndexError("index out of range")
        item = None
        if self.dictionary is not None:
            item = self.dictionary.encode_line(self.dictionary.wordpiece_encode(self.texts[i]), append_eos=self.append_eos).long()
        return item, self.texts[i]

    def __len__(self):
        return self.size


def collate(samples, pad_idx, eos_idx, left_pad_source=False, left_pad_target=False, input_feeding=True, maybe_bos_idx=None,
            pad_to_length=None, pad_to_multiple=1, pin_memory=False, wave_workers=1, sample_rate=16000):
    """espresso/data/asr_dataset.py:17-136.  `source` is either a `(T, F)` feature tensor (reference path) or a 1-D
    float waveform (raw-audio path, see the module docstring)."""
    if len(samples) == 0:
        return {}
    is_wave = isinstance(samples[0]["source"], (np.ndarray, audio_utils.LazyWave)) and samples[0]["source"].ndim == 1

    def merge_tokens(key, move_eos_to_beginning=False, pad_to=None):
        return collate_tokens([s[key] for s in samples], pad_idx, eos_idx, left_pad_target, move_eos_to_beginning,
                              pad_to_length=pad_to, pad_to_multiple=pad_to_multiple)

    id = torch.LongTensor([s["id"] for s in samples])
    if is_wave:
        src_lengths = torch.IntTensor([samples_to_frames(len(s["source"])) for s in samples])
    else:
        src_frames = collate_frames([s["source"] for s in samples], 0.0, left_pad_source,
                                    pad_to_length=pad_to_length["source"] if pad_to_length is not None else None,
                                    pad_to_multiple=pad_to_multiple)
        if pad_to_length is not None:
            src_lengths = torch.IntTensor([s["source"].ne(0.0).any(dim=1).int().sum() for s in samples])
        else:
            src_lengths = torch.IntTensor([s["source"].size(0) for s in samples])
    src_lengths, sort_order = src_lengths.sort(descending=True)
    order = sort_order.tolist()
    id = id.index_select(0, sort_order)
    utt_id = [samples[i]["utt_id"] for i in order]

    prev_output_tokens = target = None
    if samples[0].get("target", None) is not None:
        tgt_pad = pad_to_length["target"] if pad_to_length is not None else None
        target = merge_tokens("target", pad_to=tgt_pad).index_select(0, sort_order)
        ntokens = sum(int(s["target"].ne(pad_idx).sum()) for s in samples)
        if samples[0].get("prev_output_tokens", None) is not None:
            prev_output_tokens = merge_tokens("prev_output_tokens")
        elif input_feeding:
            # shifted targets: </s> moved to the front, or <s> prepended when the dictionary has one (:89-103)
            prev_output_tokens = merge_tokens("target", move_eos_to_beginning=(maybe_bos_idx is None), pad_to=tgt_pad)
            if maybe_bos_idx is not None:
                bos = prev_output_tokens.new_full((len(samples), 1), maybe_bos_idx)
                prev_output_tokens = torch.cat([bos, prev_output_tokens], dim=1)
    else:
        ntokens = int(src_lengths.sum())

    text = None
    if samples[0].get("text", None) is not None:
        text = [samples[i]["text"] for i in order]

    net_input = {"src_lengths": src_lengths}
    batch = {"id": id, "utt_id": utt_id, "nsentences": len(samples), "ntokens": ntokens, "net_input": net_input,
             "target": target, "text": text}
    if is_wave:
        lens = [len(samples[i]["source"]) for i in order]
        offsets = np.zeros(len(order) + 1, dtype=np.int64)
        offsets[1:] = np.cumsum(lens)
        lazy = [isinstance(samples[i]["source"], audio_utils.LazyWave) for i in order]
        if all(lazy):
            # files not read yet: int16 staging buffer (half the bytes of the fp32 path over PCIe; the fbank kernel converts),
            # every file decoded straight into its slot, `wave_workers` files at a time, outside the interpreter lock
            wav = torch.empty(int(offsets[-1]), dtype=torch.int16, pin_memory=pin_memory)
            rates = audio_utils.read_batch_i16([samples[i]["source"].path for i in order], wav.numpy(), offsets, wave_workers)
            assert all(r == sample_rate for r in rates), f"sample rates {sorted(set(rates))}: the front-end tables are built for {sample_rate} Hz"
        else:
            wav = torch.empty(int(offsets[-1]), dtype=torch.float32, pin_memory=pin_memory)
            wnp = wav.numpy()
            for k, i in enumerate(order):
                src = samples[i]["source"]
                wnp[offsets[k]:offsets[k + 1]] = audio_utils.get_waveform(src.path)[0] if lazy[k] else src
        batch.update(wav=wav, wav_offsets=torch.from_numpy(offsets), num_samples=lens, id_list=id.tolist(),
                     audio_seconds=float(offsets[-1]) / 16000.0)
    else:
        net_input["src_tokens"] = src_frames.index_select(0, sort_order)
    if prev_output_tokens is not None:
        net_input["prev_output_tokens"] = prev_output_tokens.index_select(0, sort_order)
    return batch


class AsrDataset:
    """A pair of audio / transcript datasets (espresso/data/asr_dataset.py:139-450)."""

    def __init__(self, src, src_sizes, tgt=None, tgt_sizes=None, dictionary=None, left_pad_source=False, left_pad_target=False,
                 shuffle=True, input_feeding=True, prepend_bos_as_input_feeding=False, pad_to_multiple=1,
                 batch_based_on_both_src_tgt=False, pin_memory=False):
        self.src, self.tgt, self.dictionary = src, tgt, dictionary
        self.src_sizes = np.array(src_sizes)
        self.tgt_sizes = np.array(tgt_sizes) if tgt_sizes is not None else None
        self.left_pad_source, self.left_pad_target = left_pad_source, left_pad_target
        self.shuffle, self.input_feeding = shuffle, input_feeding
        self.prepend_bos_as_input_feeding = prepend_bos_as_input_feeding
        self.pad_to_multiple, self.batch_based_on_both_src_tgt = pad_to_multiple, batch_based_on_both_src_tgt
        self.pin_memory = pin_memory
        self.epoch = 1
        if self.tgt is not None:
            self._match_src_tgt()

    def _match_src_tgt(self):
        """Keep the utterances present on both sides, in the source's order (:259-280)."""
        tgt_pos = {u: i for i, u in enumerate(self.tgt.utt_ids)}
        src_idx = [i for i, u in enumerate(self.src.utt_ids) if u in tgt_pos]
        tgt_idx = [tgt_pos[self.src.utt_ids[i]] for i in src_idx]
        self.src.filter_and_reorder(src_idx)
        self.tgt.filter_and_reorder(tgt_idx)
        self.src_sizes, self.tgt_sizes = np.array(self.src.sizes), np.array(self.tgt.sizes)
        assert self.src.utt_ids == self.tgt.utt_ids

    def __getitem__(self, index):
        tgt_item, text_item = self.tgt[index] if self.tgt is not None else (None, None)
        return {"id": index, "utt_id": self.src.utt_ids[index], "source": self.src[index], "target": tgt_item, "text": text_item}

    def __len__(self):
        return len(self.src)

    def collater(self, samples, pad_to_length=None):
        bos = self.dictionary.bos() if self.prepend_bos_as_input_feeding else None
        return collate(samples, pad_idx=self.dictionary.pad(), eos_idx=self.dictionary.eos(), left_pad_source=self.left_pad_source,
                       left_pad_target=self.left_pad_target, input_feeding=self.input_feeding, maybe_bos_idx=bos,
                       pad_to_length=pad_to_length, pad_to_multiple=self.pad_to_multiple, pin_memory=self.pin_memory,
                       wave_workers=getattr(self.src, "num_workers", 1), sample_rate=getattr(self.src, "sample_rate", 16000))

    def num_tokens(self, index):
        if self.batch_based_on_both_src_tgt and self.tgt_sizes is not None:
            return self.src_sizes[index] * self.tgt_sizes[index]
        return self.src_sizes[index]

    def num_tokens_vec(self, indices):
        sizes = self.src_sizes[indices]
        if self.batch_based_on_both_src_tgt and self.tgt_sizes is not None:
            return sizes * self.tgt_sizes[indices]
        return sizes

    def size(self, index):
        return (self.src_sizes[index], self.tgt_sizes[index] if self.tgt_sizes is not None else 0)

    def ordered_indices(self):
        """Random permutation, then stable sorts by target and source length (:392-409); the caller seeds numpy."""
        indices = np.random.permutation(len(self)).astype(np.int64) if self.shuffle else np.arange(len(self), dtype=np.int64)
        if self.tgt_sizes is not None:
            indices = indices[np.argsort(self.tgt_sizes[indices], kind="mergesort")]
        return indices[np.argsort(self.src_sizes[indices], kind="mergesort")]

    def filter_indices_by_size(self, indices, max_sizes):
        """fairseq/data/data_utils.py:filter_paired_dataset_indices_by_size -> (kept, ignored)."""
        if max_sizes is None:
            return indices, []
        max_src, max_tgt = (max_sizes, max_sizes) if isinstance(max_sizes, (int, float)) else max_sizes
        ok = self.src_sizes[indices] <= max_src if max_src is not None else np.ones(len(indices), dtype=bool)
        if self.tgt_sizes is not None and max_tgt is not None:
            ok &= self.tgt_sizes[indices] <= max_tgt
        return indices[ok], indices[~ok].tolist()

    def set_epoch(self, epoch):
        self.epoch = epoch
        self.src.set_epoch(epoch)

    @property
    def supports_prefetch(self):
        return False

    def prefetch(self, indices):
        pass


def get_asr_dataset_from_json(data_path, split, tgt_dict, combine=False, shuffle=True, pad_to_multiple=1, autoregressive=True,
                              prepend_bos_as_input_feeding=False, batch_based_on_both_src_tgt=False, pin_memory=False):
    """Parse `<split>.json` (`<split>1.json`, ... when `combine`) packed by espresso/tools/asr_prep_json.py: per
    utterance one of `feat` / `wave` / `command`, optional `text` and `utt2num_frames`
    (espresso/tasks/speech_recognition.py:127-269).  CMVN / SpecAugment are not dataset transforms here — the task's
    GPU front-end applies them (same order: CMVN, then SpecAugment on the training split only)."""
    utt_ids, audios, texts, utt2num_frames = [], [], [], []
    kinds = set()
    for k in itertools.count():
        path = os.path.join(data_path, "{}.json".format(split + (str(k) if k > 0 else "")))
        if not os.path.isfile(path):
            if k > 0:
                break
            raise FileNotFoundError("Dataset not found: {}".format(path))
        with open(path, "rb") as f:
            loaded = json.load(f, object_pairs_hook=OrderedDict)
        for utt_id, val in loaded.items():
            kind = next((x for x in ("feat", "wave", "command") if x in val), None)
            if kind is None:
                raise KeyError(f"'feat', 'wave' or 'command' should be present as a field for the entry {utt_id} in {path}")
            kinds.add("feat" if kind == "feat" else "wave")
            utt_ids.append(utt_id)
            audios.append(val[kind])
            if "text" in val:
                texts.append(val["text"])
            if "utt2num_frames" in val:
                utt2num_frames.append(int(val["utt2num_frames"]))
        if not combine:
            break
    assert len(kinds) == 1, "feature and waveform entries cannot be mixed in one dataset"
    assert len(utt2num_frames) == 0 or len(utt_ids) == len(utt2num_frames)
    if "feat" in kinds:
        assert all(re.search(r":\d+$", a.strip()) for a in audios[:1]), "feat entries must be `file.ark:offset`"
        src = AudioFeatDataset(utt_ids, audios, utt2num_frames=utt2num_frames)
    else:
        src = AudioWaveDataset(utt_ids, audios, utt2num_frames=utt2num_frames)
    tgt = None
    if len(texts) > 0:
        assert len(utt_ids) == len(texts) and tgt_dict is not None
        tgt = AsrTextDataset(utt_ids, texts, tgt_dict, append_eos=autoregressive)
    return AsrDataset(src, src.sizes, tgt, tgt.sizes if tgt is not None else None, tgt_dict, left_pad_source=False,
                      left_pad_target=False, shuffle=shuffle, input_feeding=autoregressive,
                      prepend_bos_as_input_feeding=prepend_bos_as_input_feeding, pad_to_multiple=pad_to_multiple,
                      batch_based_on_both_src_tgt=batch_based_on_both_src_tgt, pin_memory=pin_memory)
