"""Batched on-GPU front-end: ragged waveforms -> padded (B, T, 80) features in one launch.

Replaces the per-utterance CPU pipeline of espresso/data/feat_text_dataset.py:128-161
(get_waveform -> torchaudio fbank -> numpy_seed(seed, epoch, index) -> GlobalCMVN -> AdaptiveSpecAugment
-> .float()) and the padding of espresso/tools/utils.py:97-113.  The dataset hands over raw int16-scale
float waveforms; mask positions are drawn on the host from the reference's RNG stream."""
from typing import Optional, Sequence

import numpy as np
import torch

from .. import kernels as K
from .fbank_tables import build_tables
from .feature_transforms import AdaptiveSpecAugmentTransform, GlobalCMVN, numpy_seed_value


def apply_time_warp(feat: torch.Tensor, i0: torch.Tensor, i1: torch.Tensor, frac: torch.Tensor) -> torch.Tensor:
    """feat [B][T][F]; i0, i1 int32 [B][T], frac fp32 [B][T] from AdaptiveSpecAugmentTransform.warp_indices:
    out[b, t] = feat[b, i0[b, t]] * (1 - frac[b, t]) + feat[b, i1[b, t]] * frac[b, t]   (adaptive_specaugment.py:94-109).
    Plain torch ops (the recipes never warp: W = 0); the mask value of the masks that follow is the mean of the UNWARPED
    features, as in the reference (the fbank kernel's per-utterance sums are taken before this step)."""
    F_ = feat.shape[-1]
    a = torch.gather(feat, 1, i0.long().unsqueeze(-1).expand(-1, -1, F_))
    b = torch.gather(feat, 1, i1.long().unsqueeze(-1).expand(-1, -1, F_))
    w = frac.unsqueeze(-1)
    return a * (1.0 - w) + b * w


class GpuFbankFrontend:
    def __init__(self, device, num_mel_bins=80, sample_rate=16000, cmvn: Optional[GlobalCMVN] = None,
                 specaug: Optional[AdaptiveSpecAugmentTransform] = None, seed=1):
        self.device = device
        self.nmel = num_mel_bins
        self.frame_len = int(sample_rate * 0.025)
        self.frame_shift = int(sample_rate * 0.010)
        self.tables = build_tables(device, num_mel_bins, self.frame_len, 512, float(sample_rate))
        self.cmvn_mean = self.cmvn_std = None
        if cmvn is not None:
            self.cmvn_mean = torch.tensor(cmvn.mean, dtype=torch.float32, device=device)
            self.cmvn_std = torch.tensor(cmvn.std, dtype=torch.float32, device=device)
        self.specaug = specaug
        self.seed = seed

    def num_frames(self, n_samples: int) -> int:
        return 0 if n_samples < self.frame_len else 1 + (n_samples - self.frame_len) // self.frame_shift

    def draw_specaug(self, n_frames: Sequence[int], epoch: int, indices: Sequence[int]):
        """Host side of SpecAugment: per-utterance mask lists from np.random seeded like
        `numpy_seed(seed, epoch, index)`; global RNG state is restored afterwards."""
        sa = self.specaug
        nf, nt = sa.freq_mask_n, sa.max_time_masks()
        fmask = np.zeros((len(n_frames), max(nf, 1), 2), dtype=np.int32)
        tmask = np.zeros((len(n_frames), max(nt, 1), 2), dtype=np.int32)
        self._warp = None
        if sa.time_warp_w > 0:  # (W = 0 in every recipe) source rows of the time warp, identity for the unwarped / padded frames
            Tm = max(n_frames) if len(n_frames) else 0
            ident = np.arange(Tm, dtype=np.int32)
            self._warp = [np.tile(ident, (len(n_frames), 1)), np.tile(ident, (len(n_frames), 1)),
                          np.zeros((len(n_frames), Tm), dtype=np.float32)]
        state = np.random.get_state()
        try:
            for b, (m, idx) in enumerate(zip(n_frames, indices)):
                np.random.seed(numpy_seed_value(self.seed, epoch, idx))
                warp, fm, tm = sa.draw_masks(int(m), self.nmel, with_warp=True)
                if warp is not None:
                    i0, i1, fr = sa.warp_indices(int(m), *warp)
                    self._warp[0][b, :int(m)], self._warp[1][b, :int(m)], self._warp[2][b, :int(m)] = i0, i1, fr
                for i, e in enumerate(fm):
                    fmask[b, i] = e
                for i, e in enumerate(tm):
                    tmask[b, i] = e
        finally:
            np.random.set_state(state)
        return fmask, tmask

    def _to_device_async(self, arr: np.ndarray) -> torch.Tensor:
        src = torch.from_numpy(arr)
        if torch.device(self.device).type != "cuda":
            return src.to(self.device)
        pinned = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
        pinned.copy_(src)
        return pinned.to(self.device, non_blocking=True)

    def __call__(self, wav: torch.Tensor, offsets: torch.Tensor, n_samples: Sequence[int], train=False, epoch=1,
                 indices: Optional[Sequence[int]] = None):
        """wav: device fp32 concatenated samples; offsets: device int64 [B+1]; n_samples: host lengths.
        Returns (feat fp32 [B][Tmax][nmel], lengths int32 [B] on device, host frame counts)."""
        B = len(n_samples)
        frames = [self.num_frames(int(n)) for n in n_samples]
        Tmax = max(frames) if frames else 0
        use_sa = train and self.specaug is not None
        feat, out_len, utt_sum = K.fbank_batch(wav, offsets, B, self.tables, self.cmvn_mean, self.cmvn_std, Tmax,
                                               nmel=self.nmel, frame_len=self.frame_len, frame_shift=self.frame_shift,
                                               want_sum=use_sa)
        if use_sa:
            fmask, tmask = self.draw_specaug(frames, epoch, indices if indices is not None else list(range(B)))
            # pinned staging: a pageable host-to-device copy drains the stream first, which would stall the GPU at every
            # step start and stop the host from running ahead (the pinned caching allocator recycles the buffers safely)
            fm, tm = self._to_device_async(fmask), self._to_device_async(tmask)
            if self._warp is not None:
                feat = apply_time_warp(feat, *(self._to_device_async(a) for a in self._warp))
            mv = self.specaug.mask_value
            K.specaugment(feat, out_len, utt_sum, fm, tm, use_mean=mv is None, mask_value=0.0 if mv is None else float(mv))
        return feat, out_len, frames
