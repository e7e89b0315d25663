"""Waveform ingestion for the raw-audio datasets — the `wave` / `command` entries of the data json
(espresso/data/feat_text_dataset.py:133-150 -> fairseq/data/audio/audio_utils.py:get_waveform with
`normalization=False`: samples stay at int16 scale, which is what the Kaldi-compatible fbank expects).
PCM WAV is decoded with the standard library (no libsndfile in this image); mono = channel 0 as in
espresso/tools/utils.py:438-440."""
import io
import subprocess
import wave
from typing import Tuple, Union

import numpy as np


def _decode(w: "wave.Wave_read") -> Tuple[np.ndarray, int]:
    n, ch, sw, sr = w.getnframes(), w.getnchannels(), w.getsampwidth(), w.getframerate()
    raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32)
    elif sw == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) * (1.0 / 65536.0)
    elif sw == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) * 256.0
    else:
        raise ValueError(f"unsupported PCM sample width {sw}")
    if ch > 1:
        x = x.reshape(-1, ch)[:, 0].copy()
    return x, sr


def get_waveform(source: Union[str, bytes, io.BytesIO]) -> Tuple[np.ndarray, int]:
    """`source`: a WAV path, WAV bytes, or a shell command ending with `|` that writes a WAV to stdout."""
    if isinstance(source, str) and source.rstrip().endswith("|"):
        source = subprocess.run(source.rstrip()[:-1], shell=True, stdout=subprocess.PIPE, check=True).stdout
    if isinstance(source, (bytes, bytearray)):
        source = io.BytesIO(source)
    with wave.open(source, "rb") as w:
        return _decode(w)


def read_wav(path: str) -> np.ndarray:
    return get_waveform(path)[0]


def num_samples(source: str) -> int:
    """Sample count from the header only (espresso/tools/wav2num_frames.py reads durations the same way)."""
    if source.rstrip().endswith("|"):
        return len(get_waveform(source)[0])
    with wave.open(source, "rb") as w:
        return w.getnframes()


def write_wav(path: str, samples: np.ndarray, sample_rate: int = 16000):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(np.clip(np.round(samples), -32768, 32767).astype("<i2").tobytes())
