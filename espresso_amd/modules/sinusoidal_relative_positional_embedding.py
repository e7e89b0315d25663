"""Sinusoidal relative positional table — values identical to
espresso/modules/sinusoidal_relative_positional_embedding.py:46-71 (get_embedding) scaled by d^-0.5
(relative_positional_embedding.py:28-34), sliced as in :112-124: rows run from relative position
-(L-1) to +(L-1) ("positive when the key is to the right of the query").

The table is a constant: it is built once on the host with the reference's exact fp32 formula and
cached on the device as bf16 (the MFMA operand of pos_proj)."""
import math

import torch


def get_embedding(seq_len: int, embedding_dim: int) -> torch.Tensor:
    half_dim = embedding_dim // 2
    emb = math.log(10000) / (half_dim - 1)
    emb = torch.exp(torch.arange(half_dim, dtype=torch.float) * -emb)
    emb = torch.arange(seq_len, dtype=torch.float).unsqueeze(1) * emb.unsqueeze(0)
    emb_pos = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1).view(seq_len, -1)
    emb_neg = torch.cat([torch.sin(-emb), torch.cos(-emb)], dim=1).view(seq_len, -1)
    if embedding_dim % 2 == 1:
        emb_pos = torch.cat([emb_pos, torch.zeros(seq_len, 1)], dim=1)
        emb_neg = torch.cat([emb_neg, torch.zeros(seq_len, 1)], dim=1)
    emb_neg = torch.flip(emb_neg, [0])
    emb_pos = emb_pos[1:]
    return torch.cat([emb_neg, emb_pos], dim=0)  # (2*seq_len-1, dim)


class SinusoidalRelativePositionalEmbedding:
    """Not an nn.Module: it owns no parameters or persistent buffers (the reference registers only a
    dummy `_float_tensor` buffer, which the encoder model adds for state_dict parity)."""

    learnable = False

    def __init__(self, embedding_dim, init_size=1024, scale_embedding=True):
        self.embedding_dim = embedding_dim
        self.embedding_scale = embedding_dim ** -0.5 if scale_embedding else 1.0
        self.init_size = init_size
        self._cache = {}

    def table(self, seq_len: int, device) -> torch.Tensor:
        """bf16 [2*seq_len-1][dim] for keys of length seq_len: a contiguous row slice around the centre of one cached
        table per device (the values depend only on the offset, exactly how the reference slices its larger table,
        :112-124) — no per-length rebuild / host-to-device copy inside the training loop."""
        ent = self._cache.get(device)
        if ent is None or ent[0] < seq_len:
            size = max(self.init_size, seq_len) if ent is None else max(2 * ent[0], seq_len)
            full = self.embedding_scale * get_embedding(size, self.embedding_dim)
            ent = (size, full.to(device=device, dtype=torch.bfloat16).contiguous())
            self._cache[device] = ent
        size, full = ent
        return full[(size - 1) - (seq_len - 1): (size - 1) + seq_len]
