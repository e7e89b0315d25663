"""Transformer decoder layer — storage and forward order of fairseq/modules/transformer_layer.py:241-529 as built by
espresso/modules/transformer_with_relative_positional_embedding_layer.py:66-116 (pre-LN: causal self-attention,
encoder-decoder attention with static K/V, ReLU FFN; parameter names self_attn, self_attn_layer_norm, encoder_attn,
encoder_attn_layer_norm, fc1, fc2, final_layer_norm)."""
import torch
import torch.nn as nn

from .. import functional as F
from .conformer_layer import MultiheadAttentionParams
from .params import LayerNormParams, LinearParams


class TransformerDecoderLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        d = cfg.decoder.embed_dim
        self.embed_dim, self.num_heads = d, cfg.decoder.attention_heads
        if not cfg.decoder.normalize_before:
            raise NotImplementedError("post-LN decoder layers (recipes use normalize_before: true)")
        if cfg.decoder.relative_positional_embeddings:
            raise NotImplementedError("decoder relative positions (recipes: false)")
        self.self_attn = MultiheadAttentionParams(d, self.num_heads, relpos=False)
        self.self_attn_layer_norm = LayerNormParams(d)
        self.encoder_attn = MultiheadAttentionParams(d, self.num_heads, relpos=False, kdim=cfg.encoder.embed_dim,
                                                     vdim=cfg.encoder.embed_dim)
        self.encoder_attn_layer_norm = LayerNormParams(d)
        self.fc1 = LinearParams(d, cfg.decoder.ffn_embed_dim)
        self.fc2 = LinearParams(cfg.decoder.ffn_embed_dim, d)
        self.final_layer_norm = LayerNormParams(d)
        self.activation_fn = cfg.activation_fn

    use_native_runtime = True  # whole layer per C-ABI call (csrc/engine.hip) when the parameters live in the flat buffers

    def forward(self, x, enc, enc_len, B, U, S):
        """x bf16 [B*U][C]; enc bf16 [B*S][C_enc]; enc_len int32 [B] (valid encoder frames)."""
        cfg, tr = self.cfg, self.training
        p_drop = cfg.dropout if tr else 0.0
        p_act = cfg.activation_dropout if tr else 0.0
        p_att = cfg.attention_dropout if tr else 0.0
        if (self.use_native_runtime and x.is_cuda and self.embed_dim // self.num_heads == 64 and enc.shape[1] == self.embed_dim
                and self.activation_fn in ("relu", "silu", "swish")):
            bind = F.decoder_layer_binding(self)  # needs the flat parameter layout of the trainer (else: per-kernel path)
            if bind is not None:
                return F.decoder_layer_native(x, enc, bind, self, enc_len, B, U, S, p_drop, p_act, p_att, tr,
                                              "silu" if self.activation_fn == "swish" else self.activation_fn)
        a = self.self_attn
        wqkv, bqkv, wqkv16 = a.fused_qkv()
        x = F.relpos_mhsa(x, self.self_attn_layer_norm.weight, self.self_attn_layer_norm.bias, wqkv, bqkv, a.out_proj.weight,
                          a.out_proj.bias, None, None, None, None, None, None, B, U, self.num_heads, p_attn=p_att, p_out=p_drop,
                          wqkv16=wqkv16, causal=True)
        e = self.encoder_attn
        wkv = torch.cat([e.k_proj.weight, e.v_proj.weight], 0)
        bkv = torch.cat([e.k_proj.bias, e.v_proj.bias], 0)
        x = F.cross_mha(x, enc, self.encoder_attn_layer_norm.weight, self.encoder_attn_layer_norm.bias, e.q_proj.weight,
                        e.q_proj.bias, wkv, bkv, e.out_proj.weight, e.out_proj.bias, enc_len, B, U, S, self.num_heads,
                        p_attn=p_att, p_out=p_drop)
        return F.ffn_module(x, self.final_layer_norm.weight, self.final_layer_norm.bias, self.fc1.weight, self.fc1.bias,
                            self.fc2.weight, self.fc2.bias, act=self.activation_fn, p_act=p_act, p_out=p_drop, out_scale=1.0)
