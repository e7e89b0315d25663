"""Seeded synthetic LibriSpeech-shaped workload: utterance durations
clip(lognormal(ln 12.3 s, 0.45), 1, 35), int16-scale low-pass noise waveforms, uniform targets of
length round(4.5 * seconds) in [4, V), batched exactly like the recipe
(examples/asr_librispeech/config/transformer_ctc_librispeech.yaml:27-31: max_tokens 26000 frames,
batch_size 24, length-sorted via AsrDataset.ordered_indices espresso/data/asr_dataset.py:392-408)."""
from typing import List

import numpy as np
import torch

from .data_utils import batch_by_size, collate_tokens

SAMPLE_RATE = 16000


def durations(n_utts: int, seed: int = 1, median_s: float = 12.3, sigma: float = 0.45) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return np.clip(rng.lognormal(mean=np.log(median_s), sigma=sigma, size=n_utts), 1.0, 35.0)


def num_frames(n_samples: np.ndarray) -> np.ndarray:
    return np.where(n_samples < 400, 0, 1 + (n_samples - 400) // 160)


def make_batches(n_utts=20000, max_tokens=26000, max_sentences=24, seed=1, shuffle=True, median_s=12.3, sigma=0.45) -> List[np.ndarray]:
    """Batches of utterance indices; also returns per-utterance sample counts.  (Decode workload of SURVEY §8d: dev-other-like
    durations, mean 6.4 s: median_s=5.35, sigma=0.6.)"""
    dur = durations(n_utts, seed, median_s, sigma)
    n_samples = (dur * SAMPLE_RATE).astype(np.int64)
    frames = num_frames(n_samples)
    # reference ordering: shuffle, then stable sort by target length, then by source length (descending batches
    # are produced by the collater); a plain length sort reproduces the bucket shapes.
    order = np.argsort(frames, kind="mergesort")
    batches = batch_by_size(order, frames[order], max_tokens=max_tokens, max_sentences=max_sentences, bsz_mult=1)
    if shuffle:
        rng = np.random.default_rng(seed + 1)
        rng.shuffle(batches)
    return batches, n_samples


def waveform(n: int, rng: np.random.Generator) -> np.ndarray:
    """0.1 * 2^15 * (white noise through y[t] = 0.95 y[t-1] + x[t]) as float32."""
    from scipy.signal import lfilter

    x = rng.standard_normal(n).astype(np.float32)
    y = lfilter([1.0], [1.0, -0.95], x).astype(np.float32)
    y /= max(1e-6, float(np.std(y)))
    return (0.1 * 32768.0 * y).astype(np.float32)


def make_sample(batch: np.ndarray, n_samples: np.ndarray, vocab: int, pad_idx: int, device, seed=1):
    """Device-resident raw-audio batch in AsrDataset.collater layout (sorted by length, descending)."""
    batch = sorted(batch.tolist(), key=lambda i: -int(n_samples[i]))
    wavs, tgts = [], []
    for i in batch:
        rng = np.random.default_rng(seed * 1000003 + int(i))
        n = int(n_samples[i])
        wavs.append(waveform(n, rng))
        L = max(1, int(round(4.5 * n / SAMPLE_RATE)))
        tgts.append(torch.from_numpy(rng.integers(4, vocab, size=L).astype(np.int64)))
    lens = [len(w) for w in wavs]
    offsets = np.zeros(len(wavs) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    wav = torch.from_numpy(np.concatenate(wavs)).to(device)
    target = collate_tokens(tgts, pad_idx).to(device)
    return {
        "id": torch.tensor(batch, dtype=torch.long),
        "id_list": batch,
        "wav": wav,
        "wav_offsets": torch.from_numpy(offsets).to(device),
        "num_samples": lens,
        "target": target,
        "ntokens": int(sum(len(t) for t in tgts)),
        "nsentences": len(batch),
        "net_input": {},
        "audio_seconds": float(sum(lens)) / SAMPLE_RATE,
    }
