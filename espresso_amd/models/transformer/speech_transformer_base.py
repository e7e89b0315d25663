"""`speech_transformer_base` — attention encoder-decoder ASR model
(espresso/models/transformer/speech_transformer_base.py:28-201) with the decoder of
espresso/models/transformer/speech_transformer_decoder.py:43-516 /
fairseq/models/transformer/transformer_decoder.py:254-398 (teacher-forced training path: token + sinusoidal
positional embedding, optional layernorm_embedding, N pre-LN decoder layers, final LayerNorm, output projection).
Scheduled sampling (speech_transformer_decoder.py:254-322): the fed tokens come from a no-grad roll-out on the incremental
decoding kernels, the training pass is one teacher-forced pass over them (same logits / gradients as the reference's
step-by-step loop, see `_scheduled_sampling_tokens`)."""
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
        return pos.reshape(-1).contiguous(), self._pos_table(self.padding_idx + 1 + U, prev_output_tokens.device)

    def _pos_table(self, n, device):
        """First `n` rows of ONE device-resident sinusoidal table (built once for max_target_positions and grown by doubling):
        a per-length host build + pageable host-to-device copy would drain the stream at every new target length."""
        key = str(device)
        tab = self._pos_cache.get(key)
        if tab is None or tab.shape[0] < n:
            rows = max(n, self.max_target_positions + self.padding_idx + 2, 2 * (tab.shape[0] if tab is not None else 0))
            tab = sinusoidal_positional_table(rows, self.embed_dim, self.padding_idx).to(device).contiguous()
            self._pos_cache[key] = tab
        return tab[:n]

    def forward(self, prev_output_tokens, encoder_out=None, features_only=False, **kwargs):
        """-> (logits bf16 (B, U, V) view, extra dict)."""
        sched = getattr(self, "scheduled_sampling_rate_scheduler", None)
        if self.training and sched is not None and not features_only:
            p_true = sched.step(kwargs.get("epoch", 1))
            if p_true < 1.0:  # speech_transformer_decoder.py:254-271
                prev_output_tokens = self._scheduled_sampling_tokens(prev_output_tokens, encoder_out, p_true)
        return self._forward_teacher_forced(prev_output_tokens, encoder_out, features_only)

    @torch.no_grad()
    def _scheduled_sampling_tokens(self, prev_output_tokens, encoder_out, p_true):
        """The token sequence the reference's step-by-step loop feeds (speech_transformer_decoder.py:283-322): position 0 is
        the gold token; at every later step each sentence keeps the gold token with probability `p_true`, otherwise it is fed
        the arg-max of the previous step's output.  The reference runs that loop WITH gradients through the incremental
        state; because the decoder is causal, its logits and parameter gradients are exactly those of ONE teacher-forced
        pass over the fed sequence (the discrete choices carry no gradient), so the roll-out here runs without autograd on
        the incremental-decoding kernels (no dropout noise in the roll-out) and the training pass follows on the result."""
        B, U = prev_output_tokens.shape
        was_training = self.training
        self.eval()
        try:
            st = self.init_incremental(encoder_out, B, 1)
            fed = prev_output_tokens.clone()
            for step in range(U):
                if step > 0:
                    keep = torch.rand(B, 1, device=fed.device).lt(p_true)
                    fed[:, step:step + 1] = torch.where(keep, prev_output_tokens[:, step:step + 1], pred)
                lp = self.step(st, fed[:, : step + 1], step, None)
                pred = lp.argmax(-1, keepdim=True)
        finally:
            self.train(was_training)
        return fed

    def _forward_teacher_forced(self, prev_output_tokens, encoder_out, features_only=False):
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

    # ------------------------------------------------------------------ incremental decoding (beam search)
    @torch.no_grad()
    def init_incremental(self, encoder_out, bsz, beam):
        """Per-layer state: ping-pong self-attention K/V caches [N][Lmax][2C] and the encoder K/V of every SENTENCE
        ([B][S][2C], projected once — multihead_attention.py:661-671 static_kv + beam dedup)."""
        enc = encoder_out["_x_bt"][0] if "_x_bt" in encoder_out else None
        if enc is None:
            e = encoder_out["encoder_out"][0]
            enc = e.transpose(0, 1).reshape(-1, e.shape[-1]).contiguous()
        dev = enc.device
        S = encoder_out["encoder_padding_mask"][0].shape[1]
        C = self.embed_dim
        N = bsz * beam
        Lmax = self.max_target_positions + 2
        st = {"bsz0": bsz, "beam": beam, "S": S, "L": 0, "Lmax": Lmax,
              "enc_len": encoder_out["src_lengths"][0].to(torch.int32).contiguous(),
              "kv_row": torch.arange(bsz, device=dev, dtype=torch.int32).repeat_interleave(beam).contiguous(),
              "layers": []}
        for layer in self.layers:
            e = layer.encoder_attn
            wkv16 = torch.cat([F.bf16_weight(e.k_proj.weight), F.bf16_weight(e.v_proj.weight)], 0)
            bkv = torch.cat([e.k_proj.bias.detach(), e.v_proj.bias.detach()], 0)
            enc_kv = torch.empty(enc.shape[0], 2 * C, dtype=torch.bfloat16, device=dev)
            K.gemm(enc, wkv16, enc_kv, enc.shape[0], 2 * C, enc.shape[1], lda=enc.shape[1], ldb=enc.shape[1], ldc=2 * C, bias=bkv)
            st["layers"].append({
                "enc_kv": enc_kv,
                "cache": [torch.empty(N, Lmax, 2 * C, dtype=torch.bfloat16, device=dev) for _ in range(2)],
                "cur": 0,
            })
        return st

    @torch.no_grad()
    def step(self, st, tokens, step, parent):
        """One decoding step for the N live hypotheses.  tokens: [N][step+1] (only the last column is consumed);
        parent: int64 [N] previous-step hypothesis each row continues (None at step 0).  -> fp32 log-probs [N][V]."""
        C, H = self.embed_dim, self.layers[0].num_heads
        dh = C // H
        scaling = dh ** -0.5
        N = tokens.shape[0]
        dev = tokens.device
        tok = tokens[:, -1].to(torch.int32).contiguous()
        if parent is not None:
            par = parent.to(torch.int32).contiguous()
            st["kv_row"] = st["kv_row"].index_select(0, parent).contiguous()
        else:
            par = None
        L = st["L"]
        pos = torch.full((N,), self.padding_idx + step + 1, dtype=torch.int32, device=dev)
        tab = self._pos_table(self.padding_idx + 2 + step, dev)
        x = K.embedding_fwd(tok, pos, self.embed_tokens.weight, tab, self.embed_scale)
        if self.layernorm_embedding is not None:
            x, _, _ = K.layernorm_fwd(x, self.layernorm_embedding.weight, self.layernorm_embedding.bias, save_stats=False)
        for layer, ls in zip(self.layers, st["layers"]):
            a = layer.self_attn
            _, bqkv, wqkv16 = a.fused_qkv()
            xn, _, _ = K.layernorm_fwd(x, layer.self_attn_layer_norm.weight, layer.self_attn_layer_norm.bias, save_stats=False)
            q = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
            kvn = torch.empty(N, 2 * C, dtype=torch.bfloat16, device=dev)
            K.gemm(xn, wqkv16, q, N, C, C, lda=C, ldb=C, ldc=C, bias=bqkv[:C].contiguous())
            K.gemm(xn, wqkv16, kvn, N, 2 * C, C, lda=C, ldb=C, ldc=2 * C, b_off=C * C, bias=bqkv[C:].contiguous())
            qs, _ = K.relpos_q_prep(q, C, None, None, N, C, scaling, want_qv=False)
            old, new = ls["cache"][ls["cur"]], ls["cache"][1 - ls["cur"]]
            K.kv_append_reorder(old, new, kvn, par, N, L, st["Lmax"], 2 * C)
            ls["cur"] = 1 - ls["cur"]
            att = K.decode_attention(qs, new, None, None, None, N, H, dh, st["Lmax"] * 2 * C, 2 * C, 0, C, L + 1)
            x2 = torch.empty_like(x)
            K.gemm(att, F.bf16_weight(a.out_proj.weight), x2, N, C, C, lda=C, ldb=C, ldc=C, bias=a.out_proj.bias, resid=x, ldr=C)
            x = x2
            e = layer.encoder_attn
            xn, _, _ = K.layernorm_fwd(x, layer.encoder_attn_layer_norm.weight, layer.encoder_attn_layer_norm.bias, save_stats=False)
            q = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
            K.gemm(xn, F.bf16_weight(e.q_proj.weight), q, N, C, C, lda=C, ldb=C, ldc=C, bias=e.q_proj.bias)
            qs, _ = K.relpos_q_prep(q, C, None, None, N, C, scaling, want_qv=False)
            S = st["S"]
            att = K.decode_attention(qs, ls["enc_kv"], None, st["kv_row"], st["enc_len"], N, H, dh, S * 2 * C, 2 * C, 0, C, S)
            x2 = torch.empty_like(x)
            K.gemm(att, F.bf16_weight(e.out_proj.weight), x2, N, C, C, lda=C, ldb=C, ldc=C, bias=e.out_proj.bias, resid=x, ldr=C)
            x = x2
            Fd = layer.fc1.weight.shape[0]
            xn, _, _ = K.layernorm_fwd(x, layer.final_layer_norm.weight, layer.final_layer_norm.bias, save_stats=False)
            h = torch.empty(N, Fd, dtype=torch.bfloat16, device=dev)
            K.gemm(xn, F.bf16_weight(layer.fc1.weight), h, N, Fd, C, lda=C, ldb=C, ldc=Fd, bias=layer.fc1.bias, act=layer.activation_fn)
            x2 = torch.empty_like(x)
            K.gemm(h, F.bf16_weight(layer.fc2.weight), x2, N, C, Fd, lda=Fd, ldb=Fd, ldc=C, bias=layer.fc2.bias, resid=x, ldr=C)
            x = x2
        st["L"] = L + 1
        if self.layer_norm is not None:
            x, _, _ = K.layernorm_fwd(x, self.layer_norm.weight, self.layer_norm.bias, save_stats=False)
        w = self.embed_tokens.weight if self.output_projection is None else self.output_projection.weight
        V = w.shape[0]
        Vp = (V + 7) // 8 * 8
        logits = torch.empty(N, Vp, dtype=torch.bfloat16, device=dev)
        K.gemm(x, F.bf16_weight(w), logits, N, V, C, lda=C, ldb=C, ldc=Vp)
        return K.log_softmax(logits, N, V, Vp)


@register_model("speech_transformer_base", dataclass=SpeechTransformerConfig)
class SpeechTransformerModelBase(nn.Module):
    config_class = SpeechTransformerConfig

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
        # speech_transformer_base.py:129-143: probability of feeding the TRUE previous token, per epoch
        from ..speech_lstm import ScheduledSamplingRateScheduler

        probs = cfg.scheduled_sampling_probs
        if isinstance(probs, str):  # "[1.0, 0.9]" or "1.0,0.9"
            probs = [float(x) for x in probs.strip("[]() ").split(",") if x.strip()]
        decoder.scheduled_sampling_rate_scheduler = ScheduledSamplingRateScheduler(
            [float(x) for x in probs], getattr(cfg, "start_scheduled_sampling_epoch", 1))
        return cls(cfg, encoder, decoder)

    def set_num_updates(self, n):
        self.num_updates = n
        self.encoder.set_num_updates(n)

    def forward(self, src_tokens, src_lengths, prev_output_tokens, **kwargs):
        encoder_out = self.encoder(src_tokens, src_lengths)
        return self.decoder(prev_output_tokens, encoder_out=encoder_out, epoch=kwargs.get("epoch", 1))

    def forward_encoder(self, src_tokens, src_lengths):
        return self.encoder(src_tokens, src_lengths)

    def max_decoder_positions(self):
        return self.decoder.max_positions()

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
