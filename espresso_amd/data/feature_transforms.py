"""Audio feature transforms with the reference's registry names and parameter-drawing order.

`global_cmvn`  — fairseq/data/audio/feature_transforms/global_cmvn.py:8-29 (stats from an .npz).
`adaptive_specaugment` — espresso/data/feature_transforms/adaptive_specaugment.py:15-136.  On the
MI355X path the *arithmetic* (mask fill) runs inside the front-end kernel; what must stay on the host
for parity is the numpy RNG call sequence that decides the mask positions.
`draw_masks` reproduces that sequence exactly and returns index lists for ea_specaugment."""
import math
from typing import Optional

import numpy as np

from ..registry import register_audio_feature_transform


def numpy_seed_value(seed, *addl_seeds):
    """The integer fairseq.data.data_utils.numpy_seed feeds to np.random.seed (data_utils.py:126-140)."""
    if len(addl_seeds) > 0:
        seed = int(hash((seed, *addl_seeds)) % 1e6)
    return seed


@register_audio_feature_transform("global_cmvn")
class GlobalCMVN:
    @classmethod
    def from_config_dict(cls, config=None):
        _config = {} if config is None else config
        return cls(_config.get("stats_npz_path"))

    def __init__(self, stats_npz_path=None, mean=None, std=None):
        if stats_npz_path is not None:
            stats = np.load(stats_npz_path)
            mean, std = stats["mean"], stats["std"]
        self.mean, self.std = np.asarray(mean), np.asarray(std)


@register_audio_feature_transform("adaptive_specaugment")
class AdaptiveSpecAugmentTransform:
    @classmethod
    def from_config_dict(cls, config=None):
        c = {} if config is None else config
        return cls(c.get("time_warp_W", 0), c.get("freq_mask_N", 0), c.get("freq_mask_F", 0), c.get("time_mask_N", 0),
                   c.get("time_mask_T", 0), c.get("time_mask_p", 0.0), c.get("time_mask_pm", None),
                   c.get("time_mask_ps", None), c.get("mask_value", None))

    def __init__(self, time_warp_w=0, freq_mask_n=0, freq_mask_f=0, time_mask_n=0, time_mask_t=0, time_mask_p=0.0,
                 time_mask_pm: Optional[float] = None, time_mask_ps: Optional[float] = None,
                 mask_value: Optional[float] = 0.0):
        assert mask_value is None or isinstance(mask_value, (int, float))
        self.time_warp_w, self.freq_mask_n, self.freq_mask_f = time_warp_w, freq_mask_n, freq_mask_f
        self.time_mask_n, self.time_mask_t, self.time_mask_p = time_mask_n, time_mask_t, time_mask_p
        self.time_mask_pm, self.time_mask_ps, self.mask_value = time_mask_pm, time_mask_ps, mask_value

    def max_time_masks(self):
        return self.time_mask_n if self.time_mask_pm is None else 20

    def draw_warp(self, num_frames: int):
        """The two draws of the time warp (adaptive_specaugment.py:94-99): split frame w0 and shift w, or None when the
        utterance is not warped (W == 0 or 2 W >= num_frames: no draw is consumed then)."""
        W = self.time_warp_w
        if W <= 0 or 2 * W >= num_frames:
            return None
        w0 = np.random.randint(W, num_frames - W)
        w = np.random.randint(-W + 1, W)
        return int(w0), int(w)

    @staticmethod
    def warp_indices(num_frames: int, w0: int, w: int):
        """Source rows of the warped spectrogram (adaptive_specaugment.py:100-109): frames [0, w0) are resized to w0 + w rows,
        frames [w0, num_frames) to the remaining num_frames - w0 - w rows, each with cv2.resize(..., INTER_LINEAR) along time
        (pixel-centre aligned: source position (y + 0.5) * h_in / h_out - 0.5, clamped at both ends; the frequency axis keeps
        its size, i.e. is copied).  Returns (i0 int32 [num_frames], i1 int32 [num_frames], frac float32 [num_frames]):
        out[y] = in[i0[y]] * (1 - frac[y]) + in[i1[y]] * frac[y]."""
        i0 = np.empty(num_frames, dtype=np.int32)
        i1 = np.empty(num_frames, dtype=np.int32)
        fr = np.empty(num_frames, dtype=np.float32)
        for src0, h_in, dst0, h_out in ((0, w0, 0, w0 + w), (w0, num_frames - w0, w0 + w, num_frames - w0 - w)):
            y = np.arange(h_out, dtype=np.float64)
            f = ((y + 0.5) * (float(h_in) / float(h_out)) - 0.5).astype(np.float32)  # OpenCV: double product, float coefficient
            s = np.floor(f).astype(np.int64)
            a = (f - s.astype(np.float32)).astype(np.float32)
            low = s < 0
            s[low], a[low] = 0, 0.0
            high = s >= h_in - 1
            s[high], a[high] = h_in - 1, 0.0
            i0[dst0:dst0 + h_out] = src0 + s
            i1[dst0:dst0 + h_out] = src0 + np.minimum(s + 1, h_in - 1)
            fr[dst0:dst0 + h_out] = a
        return i0, i1, fr

    def draw_masks(self, num_frames: int, num_freqs: int, with_warp: bool = False):
        """Consume np.random exactly like the reference's __call__ and return
        (freq_masks [(f0, f)], time_masks [(t0, t)]) — zero-width entries are kept (they consume RNG
        draws but mask nothing).  with_warp: -> (warp, freq_masks, time_masks), warp = (w0, w) or None, drawn first."""
        if with_warp:
            if num_frames == 0 or num_freqs < self.freq_mask_f:
                return None, [], []
            warp = self.draw_warp(num_frames)
            fm, tm = self.draw_masks(num_frames, num_freqs)
            return warp, fm, tm
        fm, tm = [], []
        if num_frames == 0 or num_freqs < self.freq_mask_f:
            return fm, tm
        for _ in range(self.freq_mask_n):
            f = np.random.randint(0, self.freq_mask_f)
            f0 = np.random.randint(0, num_freqs - f)
            fm.append((int(f0), int(f)))
        max_t = (min(self.time_mask_t, math.floor(num_frames * self.time_mask_p)) if self.time_mask_ps is None
                 else math.floor(num_frames * self.time_mask_ps))
        if max_t < 1:
            return fm, tm
        n = self.time_mask_n if self.time_mask_pm is None else min(20, math.floor(num_frames * self.time_mask_pm))
        for _ in range(n):
            t = np.random.randint(0, max_t)
            t0 = np.random.randint(0, num_frames - t)
            tm.append((int(t0), int(t)))
        return fm, tm
