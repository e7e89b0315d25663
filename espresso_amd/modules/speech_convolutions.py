"""ConvBNReLU — storage and output-length arithmetic of espresso/modules/speech_convolutions.py:22-129;
the arithmetic runs in functional.conv_subsample (channels-last HIP path)."""
import torch
import torch.nn as nn

from .. import functional as F
from .params import BatchNormParams, ConvParams


class ConvBNReLU(nn.Module):
    def __init__(self, out_channels, kernel_sizes, strides, in_channels=1, apply_batchnorm=True):
        super().__init__()
        assert apply_batchnorm, "the MI355X path implements the recipes' BatchNorm variant"
        assert in_channels == 1, "fbank input has one channel (task.feat_in_channels == 1)"
        self.out_channels, self.kernel_sizes, self.strides, self.in_channels = out_channels, kernel_sizes, strides, in_channels
        self.convolutions = nn.ModuleList()
        self.batchnorms = nn.ModuleList()
        for i, co in enumerate(out_channels):
            ks = kernel_sizes[i]
            ks = tuple(ks) if isinstance(ks, (list, tuple)) else (ks, ks)
            assert ks == (3, 3), "3x3 kernels (all recipes)"
            self.convolutions.append(ConvParams(in_channels if i == 0 else out_channels[i - 1], co, ks))
            self.batchnorms.append(BatchNormParams(co))

    def _stride2(self, s):
        if isinstance(s, (list, tuple)):
            return (s[0], s[1] if len(s) > 1 else s[0])
        return (s, s)

    def output_lengths(self, in_lengths):
        out = in_lengths
        if isinstance(out, torch.Tensor):
            # ceil(ceil(x / a) / b) == ceil(x / (a b)): one add + one floor-divide on the device instead of three tiny launches
            # per convolution (speech_convolutions.py:60-67 computes the same numbers layer by layer)
            total = 1
            for stride in self.strides:
                total *= self._stride2(stride)[0]
            return out if total == 1 else torch.div(out + (total - 1), total, rounding_mode="floor")
        for stride in self.strides:
            s = self._stride2(stride)[0]
            if isinstance(out, torch.Tensor):
                out = torch.div(out + s - 1, s, rounding_mode="floor")
            else:
                out = (out + s - 1) // s
        return out

    def output_feat_dim(self, feat_dim):
        f = feat_dim
        for stride in self.strides:
            s = self._stride2(stride)[1]
            f = (f + s - 1) // s
        return f * self.out_channels[-1]

    def forward(self, src, src_lengths, p_drop=0.0):
        """src fp32 [B][T][F]; returns (x bf16 [B*T'][F'*C] (feature index f*C + c), lengths', padding_mask [B][T'])."""
        B, T, _ = src.shape
        x_lengths = self.output_lengths(src_lengths)
        Tp = (T + 0)
        for stride in self.strides:
            s = self._stride2(stride)[0]
            Tp = (Tp + s - 1) // s
        padding_mask = torch.arange(Tp, device=src.device).unsqueeze(0) >= x_lengths.unsqueeze(1)
        row_zero = padding_mask.reshape(-1).to(torch.uint8).contiguous()
        params, bufs = [], []
        for conv, bn in zip(self.convolutions, self.batchnorms):
            params += [conv.weight, conv.bias, bn.weight, bn.bias]
            bufs += [bn.running_mean, bn.running_var]
        x = F.conv_subsample(src.contiguous().float(), row_zero, [self._stride2(s) for s in self.strides], params, bufs,
                             p_drop=p_drop, training=self.training, bn_eps=self.batchnorms[0].eps,
                             bn_momentum=self.batchnorms[0].momentum)
        if self.training and not getattr(self, "_counters_managed", False):
            for bn in self.batchnorms:
                bn.num_batches_tracked += 1
        return x, x_lengths, padding_mask, row_zero
