"""Host-built constant tables for the fbank kernel (csrc/fbank.hip): povey window, FFT twiddles and
the sparse triangular mel bank, following Kaldi's feature-window / mel-computations semantics as
exposed by torchaudio.compliance.kaldi.fbank defaults (the call at espresso/tools/utils.py:438-440;
SURVEY.md Appendix A.1): window_type='povey' = hann(periodic=False)^0.85, padded FFT 512,
low_freq 20 Hz, high_freq = Nyquist, mel(f) = 1127 ln(1 + f/700), 80 bins, Nyquist column zero."""
import math

import numpy as np
import torch


def mel_scale(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def build_mel_bank(num_bins=80, padded=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0):
    """Dense float32 [num_bins][padded//2] triangular filters, computed in float32 like torchaudio."""
    num_fft_bins = padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = np.float32(sample_freq / padded)
    # scalars in double precision (python floats in torchaudio), tensors in float32
    mel_low_d = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high_d = 1127.0 * math.log(1.0 + high_freq / 700.0)
    mel_low = np.float32(mel_low_d)
    delta = np.float32((mel_high_d - mel_low_d) / (num_bins + 1))
    b = np.arange(num_bins, dtype=np.float32)[:, None]
    left = mel_low + b * delta
    center = mel_low + (b + np.float32(1.0)) * delta
    right = mel_low + (b + np.float32(2.0)) * delta
    mel = mel_scale(fft_bin_width * np.arange(num_fft_bins, dtype=np.float32)).astype(np.float32)[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    return np.maximum(np.float32(0.0), np.minimum(up, down)).astype(np.float32)


def build_tables(device, num_bins=80, frame_len=400, padded=512, sample_freq=16000.0):
    win = torch.hann_window(frame_len, periodic=False, dtype=torch.float32).pow(0.85)
    k = np.arange(padded // 2, dtype=np.float64)
    tw = np.stack([np.cos(2 * math.pi * k / padded), -np.sin(2 * math.pi * k / padded)], axis=1).astype(np.float32)
    bank = build_mel_bank(num_bins, padded, sample_freq)
    start, length, woff, weights = [], [], [], []
    for m in range(num_bins):
        nz = np.nonzero(bank[m])[0]
        if len(nz) == 0:
            start.append(0); length.append(0); woff.append(len(weights))
            continue
        s, e = int(nz[0]), int(nz[-1]) + 1
        start.append(s); length.append(e - s); woff.append(len(weights))
        weights.extend(bank[m, s:e].tolist())
    t = lambda a, dt: torch.tensor(np.asarray(a), dtype=dt, device=device)
    return {
        "window": win.to(device),
        "twiddle": torch.from_numpy(tw).to(device).contiguous(),
        "mel_start": t(start, torch.int32),
        "mel_len": t(length, torch.int32),
        "mel_woff": t(woff, torch.int32),
        "mel_w": t(weights, torch.float32),
    }
