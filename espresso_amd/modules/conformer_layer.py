"""Conformer encoder layer with Transformer-XL relative positions — parameter layout and forward
order of espresso/modules/conformer_with_relative_positional_embedding_encoder_layer.py:19-145
(ffn1 -> self-attention -> conv module -> ffn2 -> final LayerNorm), computed by three fused
autograd nodes + one LayerNorm on the HIP kernels."""
import math

import torch
import torch.nn as nn

from .. import functional as F
from .params import BatchNormParams, ConvParams, LayerNormParams, LinearParams


def _pe_table(pe, T, device, num_heads, embed_dim):
    """bf16 constant slice (sinusoidal) or fp32 autograd slice (learned) of the relative table for T keys."""
    if pe is None:
        return None
    if getattr(pe, "learnable", False):
        return pe.table(T, device, num_heads=num_heads, embed_dim=embed_dim)
    return pe.table(T, device)


class FeedForwardModule(nn.Module):
    """fairseq/modules/conformer_layer.py:104-146 storage (layer_norm, w_1, w_2)."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.layer_norm = LayerNormParams(dim)
        self.w_1 = LinearParams(dim, hidden, init="torch")
        self.w_2 = LinearParams(hidden, dim, init="torch")


class MultiheadAttentionParams(nn.Module):
    """fairseq/modules/multihead_attention.py:66-217 storage in the reference's registration order
    (k_proj, v_proj, q_proj, out_proj, pos_bias_u, pos_bias_v, pos_proj)."""

    def __init__(self, embed_dim, num_heads, relpos=True, kdim=None, vdim=None, positional_embedding=None):
        """relpos: sinusoidal relative positions (pos_bias_u / pos_bias_v / pos_proj).  positional_embedding: a LEARNED
        relative table, registered as `positional_embedding` like the reference's nn.Embedding attribute (:149) and used
        without biases or projection (:150-166)."""
        super().__init__()
        if positional_embedding is not None and getattr(positional_embedding, "learnable", False):
            self.positional_embedding = positional_embedding
            relpos = False
        self.embed_dim, self.num_heads = embed_dim, num_heads
        kdim = embed_dim if kdim is None else kdim
        vdim = embed_dim if vdim is None else vdim
        same = kdim == embed_dim and vdim == embed_dim
        g = 1 / math.sqrt(2) if same else 1.0
        self.k_proj = LinearParams(kdim, embed_dim, init="torch")
        self.v_proj = LinearParams(vdim, embed_dim, init="torch")
        self.q_proj = LinearParams(embed_dim, embed_dim, init="torch")
        self.out_proj = LinearParams(embed_dim, embed_dim, init="torch")
        nn.init.xavier_uniform_(self.k_proj.weight, gain=g)
        nn.init.xavier_uniform_(self.v_proj.weight, gain=g)
        nn.init.xavier_uniform_(self.q_proj.weight, gain=g)
        nn.init.xavier_uniform_(self.out_proj.weight)
        nn.init.constant_(self.out_proj.bias, 0.0)
        if relpos:
            self.pos_bias_u = nn.Parameter(torch.empty(embed_dim))
            self.pos_bias_v = nn.Parameter(torch.empty(embed_dim))
            nn.init.xavier_uniform_(self.pos_bias_u.data.view(num_heads, -1))
            nn.init.xavier_uniform_(self.pos_bias_v.data.view(num_heads, -1))
            self.pos_proj = LinearParams(embed_dim, embed_dim, bias=False, init="xavier", gain=1 / math.sqrt(2))
        else:
            self.pos_bias_u = self.pos_bias_v = self.pos_proj = None

    def fused_qkv(self):
        """(w [3C][C] fp32 with grad, b [3C], w16 bf16) in (q, k, v) order."""
        ws = [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight]
        bs = [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias]
        w = torch.cat(ws, 0)
        b = torch.cat(bs, 0)
        sh = [getattr(p, "_ea_bf16", None) for p in ws]
        if all(s is not None for s in sh):
            a, bq, c = sh
            n = a.numel()
            if (bq.data_ptr() == a.data_ptr() + 2 * n and c.data_ptr() == bq.data_ptr() + 2 * n
                    and getattr(ws[0], "_ea_fused_qkv", None) is not None):
                w16 = ws[0]._ea_fused_qkv
            else:
                w16 = torch.cat(sh, 0)
        else:
            w16 = F.bf16_weight(w)
        return w, b, w16


class ConvolutionModule(nn.Module):
    """fairseq/modules/conformer_layer.py:21-77 storage."""

    def __init__(self, dim, kernel_size):
        super().__init__()
        assert (kernel_size - 1) % 2 == 0
        self.layer_norm = LayerNormParams(dim)
        self.pointwise_conv1 = ConvParams(dim, 2 * dim, 1, bias=False)
        self.depthwise_conv = ConvParams(dim, dim, kernel_size, groups=dim, bias=False)
        self.batch_norm = BatchNormParams(dim)
        self.pointwise_conv2 = ConvParams(dim, dim, 1, bias=False)


class ConformerWithRelativePositionalEmbeddingEncoderLayer(nn.Module):
    def __init__(self, cfg, positional_embedding=None):
        super().__init__()
        self.cfg = cfg
        d = cfg.encoder.embed_dim
        self.embed_dim = d
        self.num_heads = cfg.encoder.attention_heads
        self.positional_embedding = [positional_embedding]  # not registered here (sinusoidal: shared constant; learned: owned by self_attn)
        self.ffn1 = FeedForwardModule(d, cfg.encoder.ffn_embed_dim)
        self.self_attn = MultiheadAttentionParams(d, self.num_heads, relpos=positional_embedding is not None,
                                                  positional_embedding=positional_embedding)
        self.self_attn_layer_norm = LayerNormParams(d)
        self.conv_module = ConvolutionModule(d, cfg.encoder.depthwise_conv_kernel_size)
        self.ffn2 = FeedForwardModule(d, cfg.encoder.ffn_embed_dim)
        self.final_layer_norm = LayerNormParams(d)

    use_native_runtime = True  # whole layer per C-ABI call (csrc/engine.hip); False: per-kernel composition

    def forward(self, x, B, T, key_len=None, attn_mask=None, chain_next=None):
        """x: bf16 [B*T][C] (batch-major rows).  key_len: int32 [B] valid lengths or None.
        attn_mask: fp32 additive [T][T] or None (already -1e8 / -1e4 filled as in the reference :107-110).
        chain_next: the native Conformer layer that is the ONLY consumer of this layer's output (F.conformer_layer_native)."""
        cfg = self.cfg
        tr = self.training
        p_drop = cfg.dropout if tr else 0.0
        p_act = cfg.activation_dropout if tr else 0.0
        p_att = cfg.attention_dropout if tr else 0.0
        if (self.use_native_runtime and self.positional_embedding[0] is not None
                and not getattr(self.positional_embedding[0], "learnable", False)):
            y = F.conformer_layer_native(x, self, key_len, attn_mask, self.positional_embedding[0].table(T, x.device), B, T,
                                         p_drop, p_act, p_att, tr, next_module=chain_next)
            if tr and not getattr(self, "_counters_managed", False):
                self.conv_module.batch_norm.num_batches_tracked += 1
            return y
        f = self.ffn1
        x = F.ffn_module(x, f.layer_norm.weight, f.layer_norm.bias, f.w_1.weight, f.w_1.bias, f.w_2.weight, f.w_2.bias,
                         act="silu", p_act=p_act, p_out=p_drop, out_scale=0.5)
        a = self.self_attn
        pe = self.positional_embedding[0]
        wqkv, bqkv, wqkv16 = a.fused_qkv()
        x = F.relpos_mhsa(x, self.self_attn_layer_norm.weight, self.self_attn_layer_norm.bias, wqkv, bqkv,
                          a.out_proj.weight, a.out_proj.bias, a.pos_bias_u, a.pos_bias_v,
                          a.pos_proj.weight if a.pos_proj is not None else None,
                          _pe_table(pe, T, x.device, self.num_heads, self.embed_dim), key_len, attn_mask, B, T, self.num_heads,
                          p_attn=p_att, p_out=p_drop, wqkv16=wqkv16)
        c = self.conv_module
        x = F.conv_module(x, c.layer_norm.weight, c.layer_norm.bias, c.pointwise_conv1.weight, c.depthwise_conv.weight,
                          c.batch_norm.weight, c.batch_norm.bias, c.pointwise_conv2.weight, c.batch_norm.running_mean,
                          c.batch_norm.running_var, B, T, p_out=p_drop, bn_eps=c.batch_norm.eps,
                          bn_momentum=c.batch_norm.momentum, training=tr)
        if tr and not getattr(self, "_counters_managed", False):
            c.batch_norm.num_batches_tracked += 1
        f = self.ffn2
        x = F.ffn_module(x, f.layer_norm.weight, f.layer_norm.bias, f.w_1.weight, f.w_1.bias, f.w_2.weight, f.w_2.bias,
                         act="silu", p_act=p_act, p_out=p_drop, out_scale=0.5)
        return F.layer_norm(x, self.final_layer_norm.weight, self.final_layer_norm.bias)
