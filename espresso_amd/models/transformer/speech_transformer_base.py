"""`speech_transformer_base` — attention encoder-decoder ASR model
(espresso/models/transformer/speech_transformer_base.py:28-201) with the decoder of
espresso/models/transformer/speech_transformer_decoder.py:43-516 /
fairseq/models/transformer/transformer_decoder.py:254-398 (teacher-forced training path: token + sinusoidal
positional embedding, optional layernorm_embedding, N pre-LN decoder layers, final LayerNorm, output projection).
Scheduled sampling (speech_transformer_decoder.py:283-324) is a host-side loop over this forward and is not used
by the LibriSpeech recipe (scheduled_sampling_probs 1.0)."""
import math

import torch
import torch.nn as nn

from ... import functional as F
from ... import kernels as K
from ...modules.params import LayerNormParams, LinearParams
from ...modules.speech_convolutions import ConvBNReLU
from ...modules.transformer_decoder_layer import TransformerDecoderLayer
from ...registry import register_model
from ...tools import utils as speech_utils
from .speech_transformer_config import SpeechTransformerConfig
from .speech_transformer_encoder_model import SpeechTransformerEncoderBase


def sinusoidal_positional_table(num_embeddings, dim, padding_idx):
    """fairseq/modules/sinusoidal_positional_embedding.py:36-58 — constant table, built with the reference's fp32 formula."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(num_embeddings, -1)
    if dim % 2 == 1:
        e = torch.cat([e, torch.zeros(num_embeddings, 1)], dim=1)
    if padding_idx is not None:
        e[padding_idx, :] = 0
    return e


class EmbeddingParams(nn.Module):
    """nn.Embedding storage; init N(0, d^-0.5), pad row zero (fairseq Embedding helper)."""

    def __init__(self, num_embeddings, dim, padding_idx):
        super().__init__()
        self.padding_idx, self.embedding_dim = padding_idx, dim
        self.weight = nn.Parameter(torch.empty(num_embeddings, dim))
        nn.init.normal_(self.weight, mean=0, std=dim ** -0.5)
        nn.init.constant_(self.weight[padding_idx], 0)


class SpeechTransformerDecoderBase(nn.Module):
    def __init__(self, cfg, dictionary, embed_tokens):
        super().__init__()
        self.cfg = cfg
        self.dictionary = dictionary
        self.register_buffer("version", torch.Tensor([3]))
        d = cfg.decoder.embed_dim
        self.embed_dim = d
        self.padding_idx = embed_tokens.padding_idx
        self.max_target_positions = cfg.max_target_positions
        self.embed_tokens = embed_tokens
        self.embed_scale = 1.0 if cfg.no_scale_embedding else math.sqrt(d)
        if embed_tokens.embedding_dim != d:
            raise NotImplementedError("project_in_dim (decoder.input_dim != embed_dim)")
        if cfg.decoder.learned_pos or cfg.no_token_positional_embeddings:
            raise NotImplementedError("learned / disabled decoder positions (recipes: sinusoidal)")
        self._pos_cache = {}
        self.layernorm_embedding = LayerNormParams(d) if cfg.layernorm_embedding else None
        self.layers = nn.ModuleList([TransformerDecoderLayer(cfg) for _ in range(cfg.decoder.layers)])
        self.layer_norm = LayerNormParams(d) if cfg.decoder.normalize_before else None
        if cfg.share_decoder_input_output_embed:
            self.output_projection = None  # tied to embed_tokens.weight
        else:
            self.output_projection = LinearParams(cfg.decoder.output_dim, len(dictionary), bias=False, init="torch")
            nn.init.normal_(self.output_projection.weight, mean=0, std=cfg.decoder.output_dim ** -0.5)

    def _positions(self, prev_output_tokens):
        """(int32 positions [B*U], fp32 table) as utils.make_positions + SinusoidalPositionalEmbedding produce them."""
        B, U = prev_output_tokens.shape
        mask = prev_output_tokens.ne(self.padding_idx).int()
        pos = (torch.cumsum(mask, 1) * mask + self.padding_idx).to(torch.int32)
        n = self.padding_idx + 1 + U
        key = (n, str(prev_output_tokens.device))
        tab = self._pos_cache.get(key)
        if tab is None:
            tab = sinusoidal_positional_table(n, self.embed_dim, self.padding_idx).to(prev_output_tokens.device).contiguous()
            self._pos_cache[key] = tab
        return pos.reshape(-1).contiguous(), tab

    def forward(self, prev_output_tokens, encoder_out=None, features_only=False, **unused):
        """-> (logits bf16 (B, U, V) view, extra dict)."""
        cfg, tr = self.cfg, self.training
        B, U = prev_output_tokens.shape
        enc = encoder_out["_x_bt"][0] if "_x_bt" in encoder_out else None
        if enc is None:
            e = encoder_out["encoder_out"][0]  # T x B x C
            enc = e.transpose(0, 1).reshape(-1, e.shape[-1]).contiguous()
        S = encoder_out["encoder_padding_mask"][0].shape[1]
        enc_len = encoder_out["src_lengths"][0].to(torch.int32).contiguous()
        pos, tab = self._positions(prev_output_tokens)
        tok = prev_output_tokens.to(torch.int32).reshape(-1).contiguous()
        x = F.embedding(self.embed_tokens.weight, tok, pos, tab, self.embed_scale, self.padding_idx)
        p = cfg.dropout if tr else 0.0
        if self.layernorm_embedding is not None:
            x = F.layer_norm(x, self.layernorm_embedding.weight, self.layernorm_embedding.bias, drop_p=p)
        elif p > 0:
            x = F.dropout(x, p)
        for layer in self.layers:
            x = layer(x, enc, enc_len, B, U, S)
        if self.layer_norm is not None:
            x = F.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias)
        if features_only:
            return x.view(B, U, -1), {}
        w = self.embed_tokens.weight if self.output_projection is None else self.output_projection.weight
        logits = F.linear(x, w, None)  # [B*U][V] (row-padded view)
        V = logits.shape[1]
        return logits.view(B, U, V), {"_logits_bu": logits}

    def max_positions(self):
        return self.max_target_positions


@register_model("speech_transformer_base", dataclass=SpeechTransformerConfig)
class SpeechTransformerModelBase(nn.Module):
    def __init__(self, cfg, encoder, decoder):
        super().__init__()
        self.cfg, self.encoder, self.decoder = cfg, encoder, decoder
        self.num_updates = 0

    @classmethod
    def build_model(cls, cfg, task):
        ev = speech_utils.eval_str_nested_list_or_tuple
        tgt_dict = task.target_dictionary
        embed = EmbeddingParams(len(tgt_dict), cfg.decoder.input_dim, tgt_dict.pad())
        out_channels = ev(cfg.encoder.conv_channels, type=int)
        conv = ConvBNReLU(out_channels, ev(cfg.encoder.conv_kernel_sizes, type=int), ev(cfg.encoder.conv_strides, type=int),
                          in_channels=task.feat_in_channels)
        in_size = conv.output_feat_dim(task.feat_dim // task.feat_in_channels)
        encoder = SpeechTransformerEncoderBase(cfg, pre_encoder=conv, input_size=in_size)
        decoder = SpeechTransformerDecoderBase(cfg, tgt_dict, embed)
        return cls(cfg, encoder, decoder)

    def set_num_updates(self, n):
        self.num_updates = n
        self.encoder.set_num_updates(n)

    def forward(self, src_tokens, src_lengths, prev_output_tokens, **kwargs):
        encoder_out = self.encoder(src_tokens, src_lengths)
        return self.decoder(prev_output_tokens, encoder_out=encoder_out)

    def forward_encoder(self, src_tokens, src_lengths):
        return self.encoder(src_tokens, src_lengths)

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        logits = net_output[1].get("_logits_bu") if isinstance(net_output[1], dict) else None
        if logits is None:
            lg = net_output[0]
            logits = lg.reshape(-1, lg.shape[-1])
        M, V = logits.shape
        lp = K.log_softmax(logits.detach(), M, V, logits.stride(0))
        if not log_probs:
            lp = lp.exp_()
        return lp.view(net_output[0].shape[0], net_output[0].shape[1], V)

    def get_targets(self, sample, net_output):
        return sample["target"]

    def max_positions(self):
        return (self.encoder.max_positions(), self.decoder.max_positions())

    def upgrade_state_dict_named(self, state_dict, name):
        for k in list(state_dict.keys()):
            if "conv_layers_before" in k:
                state_dict[k.replace("conv_layers_before", "pre_encoder")] = state_dict.pop(k)
        for k in list(state_dict.keys()):
            if k.endswith("positional_embedding._float_tensor") or k.endswith("embed_positions._float_tensor"):
                state_dict.pop(k)
        return state_dict
