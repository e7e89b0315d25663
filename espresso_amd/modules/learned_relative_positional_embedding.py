"""Learned relative positional table — espresso/modules/learned_relative_positional_embedding.py:15-80: an
`nn.Embedding(2*max_size-1, dim)` (init N(0, dim^-0.5)) indexed at `max_positions//2 - L + 1 .. max_positions//2 + L - 1`
for keys of length L, i.e. a contiguous row slice around the centre.  The multi-head attention uses the slice directly
(no pos_proj, no pos_bias_u/v: fairseq/modules/multihead_attention.py:150-166, 806-815).  `dim` may be the head dim
(`share_learned_relative_positional_embeddings_across_heads`), in which case the slice is tiled over the heads."""
import torch
import torch.nn as nn


class LearnedRelativePositionalEmbedding(nn.Module):
    learnable = True

    def __init__(self, embedding_dim, max_size=1024):
        super().__init__()
        self.embedding_dim, self.max_size = embedding_dim, max_size
        self.max_positions = 2 * max_size - 1
        self.weight = nn.Parameter(torch.empty(self.max_positions, embedding_dim))
        nn.init.normal_(self.weight, mean=0, std=embedding_dim ** -0.5)

    def table(self, seq_len: int, device=None, num_heads: int = 1, embed_dim: int = None) -> torch.Tensor:
        """fp32 [2*seq_len-1][embed_dim] slice with autograd back to `weight`."""
        assert seq_len <= self.max_size, f"sequence of {seq_len} keys exceeds the learned table ({self.max_size})"
        start = self.max_positions // 2 - seq_len + 1
        t = self.weight[start: start + 2 * seq_len - 1]
        if embed_dim is not None and self.embedding_dim != embed_dim:
            assert self.embedding_dim * num_heads == embed_dim
            t = t.repeat(1, num_heads)  # shared across heads
        return t
