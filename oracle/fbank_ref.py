"""TEST INFRASTRUCTURE ONLY — numpy restatement of torchaudio.compliance.kaldi.fbank with the
arguments the reference uses (espresso/tools/utils.py:438-440: num_mel_bins=80, sample_frequency,
everything else default).  torchaudio (pinned only as ">= 0.10.0", README.md:16) is NOT installable
in this environment and nothing in the reference's tests pins its output:  **parity unpinned** against torchaudio itself
at this boundary.  The restatement follows the published Kaldi semantics (SURVEY.md Appendix
A.1): snip_edges framing, per-frame DC removal, pre-emphasis 0.97 with replicate padding, povey
window, 512-point rFFT power spectrum, 80 triangular mel filters 20 Hz..Nyquist (mel = 1127 ln(1+f/700)),
log floored at float32 eps.  It is pinned against the closed-form frame count of espresso/tools/utils.py:457-486
(num_samples_to_num_frames) and, as an independent third-party check, against the Kaldi-compatible numpy front-end of
`transformers.audio_utils` (what that library's feature extractors run in place of torchaudio.compliance.kaldi.fbank):
max abs difference 1.4e-4 on log-mel values up to 22 (tests/test_oracle.py)."""
import math

import numpy as np

EPS = np.finfo(np.float32).eps


def mel_banks(num_bins=80, padded=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0):
    num_fft_bins = padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float32)[:, None]
    left = np.float32(mel_low) + b * np.float32(delta)
    center = np.float32(mel_low) + (b + np.float32(1.0)) * np.float32(delta)
    right = np.float32(mel_low) + (b + np.float32(2.0)) * np.float32(delta)
    freqs = np.float32(fft_bin_width) * np.arange(num_fft_bins, dtype=np.float32)
    mel = (np.float32(1127.0) * np.log(np.float32(1.0) + freqs / np.float32(700.0))).astype(np.float32)[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return np.maximum(np.float32(0), np.minimum(up, down)).astype(np.float32)  # (num_bins, padded/2)


def povey_window(n=400):
    i = np.arange(n, dtype=np.float64)
    hann = 0.5 - 0.5 * np.cos(2.0 * math.pi * i / (n - 1))
    return (hann.astype(np.float32) ** np.float32(0.85)).astype(np.float32)


def fbank(waveform, num_mel_bins=80, sample_frequency=16000.0, frame_length=25.0, frame_shift=10.0, preemph=0.97):
    """waveform: float32 1-D (int16 scale).  Returns float32 (M, num_mel_bins)."""
    wav = np.asarray(waveform, dtype=np.float32)
    wl = int(sample_frequency * frame_length * 0.001)
    ws = int(sample_frequency * frame_shift * 0.001)
    padded = 1 << (wl - 1).bit_length()
    n = wav.shape[0]
    if n < wl:
        return np.zeros((0, num_mel_bins), dtype=np.float32)
    m = 1 + (n - wl) // ws
    idx = np.arange(wl)[None, :] + ws * np.arange(m)[:, None]
    frames = wav[idx].astype(np.float32)
    frames = frames - frames.mean(axis=1, keepdims=True, dtype=np.float32)
    prev = np.concatenate([frames[:, :1], frames[:, :-1]], axis=1)
    frames = frames - np.float32(preemph) * prev
    frames = frames * povey_window(wl)[None, :]
    frames = np.pad(frames, ((0, 0), (0, padded - wl)))
    spec = np.fft.rfft(frames.astype(np.float32), axis=1)
    power = (np.abs(spec).astype(np.float32)) ** np.float32(2.0)
    bank = mel_banks(num_mel_bins, padded, sample_frequency)
    bank = np.pad(bank, ((0, 0), (0, 1)))  # zero Nyquist column
    mel = power.astype(np.float32) @ bank.T
    return np.log(np.maximum(mel, EPS)).astype(np.float32)


def global_cmvn(x, mean, std):
    """fairseq/data/audio/feature_transforms/global_cmvn.py:26-29 (float64 stats) then .float()
    (espresso/data/feat_text_dataset.py:160)."""
    y = np.subtract(x, mean)
    y = np.divide(y, std)
    return y.astype(np.float32)


def _resize_rows_linear(x, h_out):
    """cv2.resize(x, dsize=(x.shape[1], h_out), interpolation=cv2.INTER_LINEAR) for a float32 [h_in][w] image whose width is
    kept: OpenCV's pixel-centre convention, fy = (y + 0.5) * h_in / h_out - 0.5 (double product, float coefficient), rows
    clamped at both ends.  PARITY UNPINNED: cv2 is not installed in this image (the reference's own time warp cannot run
    here either); this restates OpenCV's documented algorithm (modules/imgproc/src/resize.cpp, linear, float)."""
    h_in = x.shape[0]
    out = np.empty((h_out, x.shape[1]), dtype=np.float32)
    scale = float(h_in) / float(h_out)
    for y in range(h_out):
        fy = np.float32((y + 0.5) * scale - 0.5)
        sy = int(np.floor(fy))
        a = np.float32(fy - np.float32(sy))
        if sy < 0:
            sy, a = 0, np.float32(0.0)
        if sy >= h_in - 1:
            sy, a = h_in - 1, np.float32(0.0)
        out[y] = x[sy] * (np.float32(1.0) - a) + x[min(sy + 1, h_in - 1)] * a
    return out


def time_warp(spec, w0, w):
    """espresso/data/feature_transforms/adaptive_specaugment.py:100-109: frames [0, w0) resized to w0 + w rows, the rest to
    the remaining rows (so the length is preserved), concatenated."""
    n = spec.shape[0]
    return np.concatenate((_resize_rows_linear(spec[:w0], w0 + w), _resize_rows_linear(spec[w0:], n - w0 - w)), axis=0)


def specaugment_apply(spec, freq_masks, time_masks, mask_value=None):
    """Mask fill of espresso/data/feature_transforms/adaptive_specaugment.py:111-134 given drawn masks."""
    out = spec.copy()
    mv = spec.mean() if mask_value is None else mask_value
    for f0, f in freq_masks:
        if f != 0:
            out[:, f0:f0 + f] = mv
    for t0, t in time_masks:
        if t != 0:
            out[t0:t0 + t, :] = mv
    return out
