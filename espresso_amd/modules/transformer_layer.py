"""Transformer encoder layer with optional relative positions — storage and forward order of
fairseq/modules/transformer_layer.py:19-226 as specialised by
espresso/modules/transformer_with_relative_positional_embedding_layer.py:17-43
(self_attn, self_attn_layer_norm, fc1, fc2, final_layer_norm; pre-LN when normalize_before)."""
import torch.nn as nn

from .. import functional as F
from .conformer_layer import MultiheadAttentionParams, _pe_table
from .params import LayerNormParams, LinearParams


class TransformerWithRelativePositionalEmbeddingEncoderLayer(nn.Module):
    def __init__(self, cfg, positional_embedding=None):
        super().__init__()
        self.cfg = cfg
        d = cfg.encoder.embed_dim
        self.embed_dim = d
        self.num_heads = cfg.encoder.attention_heads
        self.normalize_before = cfg.encoder.normalize_before
        self.positional_embedding = [positional_embedding]
        self.self_attn = MultiheadAttentionParams(d, self.num_heads, relpos=positional_embedding is not None,
                                                  positional_embedding=positional_embedding)
        self.self_attn_layer_norm = LayerNormParams(d)
        self.fc1 = LinearParams(d, cfg.encoder.ffn_embed_dim)
        self.fc2 = LinearParams(cfg.encoder.ffn_embed_dim, d)
        self.final_layer_norm = LayerNormParams(d)
        self.activation_fn = cfg.activation_fn

    use_native_runtime = True  # whole layer per C-ABI call (csrc/engine.hip); False: per-kernel composition

    def forward(self, x, B, T, key_len=None, attn_mask=None):
        cfg = self.cfg
        tr = self.training
        p_drop = cfg.dropout if tr else 0.0
        p_act = cfg.activation_dropout if tr else 0.0
        p_att = cfg.attention_dropout if tr else 0.0
        a = self.self_attn
        pe = self.positional_embedding[0]
        if (self.use_native_runtime and pe is not None and self.normalize_before and x.is_cuda
                and self.activation_fn in ("relu", "silu", "swish")):
            return F.transformer_layer_native(x, _pe_table(pe, T, x.device, self.num_heads, self.embed_dim), self, key_len, attn_mask,
                                              B, T, p_drop, p_act, p_att, tr, "silu" if self.activation_fn == "swish" else self.activation_fn)
        wqkv, bqkv, wqkv16 = a.fused_qkv()
        x = F.relpos_mhsa(x, self.self_attn_layer_norm.weight, self.self_attn_layer_norm.bias, wqkv, bqkv,
                          a.out_proj.weight, a.out_proj.bias, a.pos_bias_u, a.pos_bias_v,
                          a.pos_proj.weight if a.pos_proj is not None else None,
                          _pe_table(pe, T, x.device, self.num_heads, self.embed_dim), key_len, attn_mask, B, T, self.num_heads,
                          p_attn=p_att, p_out=p_drop, wqkv16=wqkv16, pre_ln=self.normalize_before)
        if not self.normalize_before:  # post-LN (fairseq transformer_layer.py:176-226 with normalize_before False)
            x = F.layer_norm(x, self.self_attn_layer_norm.weight, self.self_attn_layer_norm.bias)
            x = F.ffn_module(x, None, None, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, act=self.activation_fn,
                             p_act=p_act, p_out=p_drop, out_scale=1.0)
            return F.layer_norm(x, self.final_layer_norm.weight, self.final_layer_norm.bias)
        return F.ffn_module(x, self.final_layer_norm.weight, self.final_layer_norm.bias, self.fc1.weight, self.fc1.bias,
                            self.fc2.weight, self.fc2.bias, act=self.activation_fn, p_act=p_act, p_out=p_drop, out_scale=1.0)
