"""Waveform ingestion for the raw-audio datasets — the `wave` / `command` entries of the data json
(espresso/data/feat_text_dataset.py:133-150 -> fairseq/data/audio/audio_utils.py:get_waveform with
`normalization=False`: samples stay at int16 scale, which is what the Kaldi-compatible fbank expects).
WAV and FLAC FILES are decoded by the library's own readers (csrc/ingest.hip: ea_audio_probe / ea_audio_read_i16 /
ea_audio_read_batch_i16 — no libsndfile in this image, and the batch call decodes `dataset.num_workers` files at a time without the
interpreter lock); WAV bytes and `cmd |` pipes go through the standard library.  Mono = channel 0 as in
espresso/tools/utils.py:438-440."""
import ctypes
import io
import subprocess
import wave
from typing import List, Sequence, Tuple, Union

import numpy as np


class LazyWave:
    """A waveform file that has not been read yet: what `AudioWaveDataset` hands the collater in lazy mode, so that the files of
    a whole batch are decoded in parallel straight into the pinned staging buffer (`read_batch_i16`)."""

    __slots__ = ("path", "num_samples")
    ndim = 1

    def __init__(self, path: str, num_samples: int):
        self.path, self.num_samples = path, int(num_samples)

    def __len__(self):
        return self.num_samples


def _is_file(source) -> bool:
    return isinstance(source, str) and not source.rstrip().endswith("|")


def probe(path: str) -> Tuple[int, int, int, int]:
    """(samples per channel, sample rate, channels, bits per sample) from the file header (WAV or FLAC)."""
    from .. import _lib

    n, sr, ch, bits = ctypes.c_long(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    rc = _lib.lib().ea_audio_probe(path.encode(), ctypes.byref(n), ctypes.byref(sr), ctypes.byref(ch), ctypes.byref(bits))
    if rc != 0:
        raise OSError(f"{path}: cannot read audio header (code {rc})")
    return n.value, sr.value, ch.value, bits.value


def read_i16(path: str) -> Tuple[np.ndarray, int]:
    """Channel 0 of a WAV / FLAC file as int16 samples."""
    from .. import _lib

    n = probe(path)[0]
    out = np.empty(n, dtype=np.int16)
    sr = ctypes.c_int(0)
    got = _lib.lib().ea_audio_read_i16(path.encode(), out.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(sr))
    if got != n:
        raise OSError(f"{path}: audio decode failed (code {got}, expected {n} samples)")
    return out, sr.value


def read_batch_i16(paths: Sequence[str], dst: np.ndarray, offsets: np.ndarray, num_threads: int = 1) -> List[int]:
    """File i -> dst[offsets[i] : offsets[i + 1]] (int16), `num_threads` files at a time, outside the interpreter lock.
    Returns the sample rates; raises if any file fails or its length differs from its slot."""
    from .. import _lib

    n = len(paths)
    assert dst.dtype == np.int16 and offsets.dtype == np.int64 and len(offsets) == n + 1
    arr = (ctypes.c_char_p * n)(*[p.encode() for p in paths])
    lengths = np.zeros(n, dtype=np.int64)
    rates = np.zeros(n, dtype=np.int32)
    bad = _lib.lib().ea_audio_read_batch_i16(arr, n, dst.ctypes.data_as(ctypes.c_void_p), offsets.ctypes.data_as(ctypes.c_void_p),
                                             int(num_threads), lengths.ctypes.data_as(ctypes.c_void_p),
                                             rates.ctypes.data_as(ctypes.c_void_p))
    want = offsets[1:] - offsets[:-1]
    if bad or not np.array_equal(lengths, want):
        i = int(np.nonzero(lengths != want)[0][0])
        raise OSError(f"{paths[i]}: audio decode failed (code / samples {int(lengths[i])}, expected {int(want[i])})")
    return rates.tolist()


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
    """`source`: a WAV / FLAC path, WAV bytes, or a shell command ending with `|` that writes a WAV to stdout.
    Returns fp32 samples at int16 scale (`normalization=False`) and the sample rate.
    FLAC wider than 16 bits is TRUNCATED to its top 16 bits by the int16 reader (LibriSpeech and every corpus of the recipes is
    16-bit; a 24-bit WAV keeps its low-order bits as a fraction, as soundfile * 2^15 does in the reference)."""
    if _is_file(source):
        bits = probe(source)[3]
        with open(source, "rb") as f:
            is_flac = f.read(4) == b"fLaC"
        if bits == 16 or is_flac:  # (wider PCM WAV keeps its low-order bits as a fraction: the standard-library path below)
            x, sr = read_i16(source)
            return x.astype(np.float32), sr
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
    return probe(source)[0]


def write_wav(path: str, samples: np.ndarray, sample_rate: int = 16000):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(np.clip(np.round(samples), -32768, 32767).astype("<i2").tobytes())
