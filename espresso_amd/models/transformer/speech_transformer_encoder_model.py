"""`speech_transformer_encoder_model` — encoder-only ASR model for CTC
(espresso/models/transformer/speech_transformer_encoder_model.py:35-210) on top of the HIP encoder
(espresso/models/transformer/speech_transformer_encoder.py:44-453 semantics).

State-dict keys match the reference (`encoder.pre_encoder.convolutions.N.*`, `encoder.fc0.*`,
`encoder.layernorm_embedding.*`, `encoder.layers.N.*`, `encoder.fc_out.*`, `encoder.version`), so
reference checkpoints load and vice versa."""
import os

import torch
import torch.nn as nn

from ... import functional as F
from ... import kernels as K
from ...modules.conformer_layer import ConformerWithRelativePositionalEmbeddingEncoderLayer
from ...modules.params import LayerNormParams, LinearParams
from ...modules.learned_relative_positional_embedding import LearnedRelativePositionalEmbedding
from ...modules.sinusoidal_relative_positional_embedding import SinusoidalRelativePositionalEmbedding
from ...modules.speech_convolutions import ConvBNReLU
from ...modules.transformer_layer import TransformerWithRelativePositionalEmbeddingEncoderLayer
from ...registry import register_model, register_model_architecture
from ...tools import utils as speech_utils
from .speech_transformer_config import DEFAULT_MAX_SOURCE_POSITIONS, SpeechTransformerConfig

_LOGITS_F32 = os.environ.get("EA_LOGITS_F32", "0") == "1"


class AbsolutePositionTable(nn.Module):
    """fairseq LearnedPositionalEmbedding storage (`embed_positions.weight`, padding row 0 zero, N(0, d^-0.5) init)."""

    def __init__(self, num_embeddings, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num_embeddings, dim))
        nn.init.normal_(self.weight, mean=0, std=dim ** -0.5)
        nn.init.constant_(self.weight[0], 0)


class SpeechTransformerEncoderBase(nn.Module):
    def __init__(self, cfg, pre_encoder=None, input_size=83):
        super().__init__()
        self.cfg = cfg
        self.register_buffer("version", torch.Tensor([3]))
        d = cfg.encoder.embed_dim
        self.embed_dim = d
        self.max_source_positions = cfg.max_source_positions
        self.pre_encoder = pre_encoder
        self.fc0 = LinearParams(input_size, d) if input_size != d else None
        self.embed_scale = 1.0 if (cfg.no_scale_embedding or self.fc0 is not None) else d ** 0.5
        # absolute positions (speech_transformer_encoder.py:95-103: the legacy `speech_transformer_{wsj,swbd,librispeech}` presets):
        # sinusoidal table (no parameters) or learned `embed_positions.weight`; padding_idx 0, positions 1..len
        self.embed_positions = None
        self.abs_positions = not cfg.encoder.relative_positional_embeddings and not cfg.no_token_positional_embeddings
        if self.abs_positions and cfg.encoder.learned_pos:
            n = int(self.output_lengths(cfg.max_source_positions)) if pre_encoder is not None else cfg.max_source_positions
            self.embed_positions = AbsolutePositionTable(n + 1, d)  # fairseq PositionalEmbedding: num_embeddings + padding_idx + 1
        self._sin_table = None
        self.layernorm_embedding = LayerNormParams(d) if cfg.layernorm_embedding else None
        nl = cfg.encoder.layers
        if not cfg.encoder.relative_positional_embeddings:
            rels = [None] * nl
        elif cfg.encoder.learned_pos:
            # speech_transformer_encoder.py:121-147: one learned table per layer unless shared across layers; dim = head
            # dim when shared across heads; max_size = sub-sampled max_source_positions
            dim = d // cfg.encoder.attention_heads if getattr(cfg.encoder, "share_learned_relative_positional_embeddings_across_heads", False) else d
            size = int(self.output_lengths(cfg.max_source_positions)) if pre_encoder is not None else cfg.max_source_positions
            if getattr(cfg.encoder, "share_learned_relative_positional_embeddings_across_layers", False):
                rels = [LearnedRelativePositionalEmbedding(dim, max_size=size)] * nl
            else:
                rels = [LearnedRelativePositionalEmbedding(dim, max_size=size) for _ in range(nl)]
        else:
            rels = [SinusoidalRelativePositionalEmbedding(d)] * nl
        self.rel_pos_embed = [rels[0]]
        if cfg.encoder.layer_type == "conformer":
            layer_cls = ConformerWithRelativePositionalEmbeddingEncoderLayer
        elif cfg.encoder.layer_type == "transformer":
            layer_cls = TransformerWithRelativePositionalEmbeddingEncoderLayer
        else:
            raise NotImplementedError(cfg.encoder.layer_type)
        self.layers = nn.ModuleList([layer_cls(cfg, positional_embedding=rels[i]) for i in range(nl)])
        self.num_layers = len(self.layers)
        if cfg.encoder.normalize_before and cfg.encoder.layer_type != "conformer":
            self.layer_norm = LayerNormParams(d)
        else:
            self.layer_norm = None
        self.transformer_context = speech_utils.eval_str_nested_list_or_tuple(cfg.encoder.transformer_context, type=int)
        self.num_updates = 0

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates

    def output_lengths(self, in_lengths):
        return in_lengths if self.pre_encoder is None else self.pre_encoder.output_lengths(in_lengths)

    def get_attn_mask(self, max_len, device, in_lengths=None):
        """Additive fp32 [T][T] mask from the chunk-streaming setting (speech_transformer_encoder.py:240-248: drawn under
        `numpy_seed(num_updates)`, last chunk partial at inference) or `transformer_context` (:250-263), already filled with
        -1e8 where masked (conformer layer :107-110), or None."""
        if self.cfg.encoder.chunk_size > 0:
            from ...data.data_utils import numpy_seed

            with numpy_seed(self.num_updates):
                vis = speech_utils.chunk_streaming_mask(in_lengths, self.cfg.encoder.chunk_size,
                                                        left_window=self.cfg.encoder.chunk_left_window,
                                                        right_window=self.cfg.encoder.chunk_right_window,
                                                        always_partial_in_last=not self.training)
            return torch.zeros(max_len, max_len, device=device).masked_fill(~vis.to(device), -1e8).contiguous()
        tc = self.transformer_context
        if tc is None or (tc[0] is None and tc[1] is None):
            return None
        ones = torch.ones(max_len, max_len, dtype=torch.bool, device=device)
        if tc[0] is None:
            m = ones.triu(tc[1] + 1)
        elif tc[1] is None:
            m = ones.tril(-tc[0] - 1)
        else:
            m = ones.triu(tc[1] + 1) | ones.tril(-tc[0] - 1)
        return torch.zeros(max_len, max_len, device=device).masked_fill(m, -1e8).contiguous()

    def _count_batchnorm_forward(self, device):
        """`num_batches_tracked += 1` of every BatchNorm of the encoder (4 in the sub-sampler + one per Conformer layer) as ONE
        launch: the counters are re-seated once as views of one int64 buffer (state_dict keys and values unchanged)."""
        flat = getattr(self, "_bn_counters", None)
        mods = getattr(self, "_bn_counter_mods", None)
        ok = (flat is not None and flat.device == device and len(mods) > 0
              and mods[0].num_batches_tracked.data_ptr() == flat.data_ptr())
        if not ok:
            owners = ([self.pre_encoder] if self.pre_encoder is not None else []) + [l for l in self.layers if hasattr(l, "conv_module")]
            mods = list(self.pre_encoder.batchnorms) if self.pre_encoder is not None else []
            mods += [l.conv_module.batch_norm for l in self.layers if hasattr(l, "conv_module")]
            if not mods:
                self._bn_counters, self._bn_counter_mods = None, []
                return
            flat = torch.stack([m.num_batches_tracked.detach().to(device).long().reshape(()) for m in mods])
            for i, m in enumerate(mods):
                m._buffers["num_batches_tracked"] = flat[i]
            for o in owners:
                o._counters_managed = True
            self._bn_counters, self._bn_counter_mods = flat, mods
        flat.add_(1)

    def _add_positions(self, x, padding_mask):
        """x = embed_scale * x + embed_positions(make_positions(~padding_mask))  (speech_transformer_encoder.py:343-347)."""
        B, Tp = padding_mask.shape
        valid = (~padding_mask).to(torch.int64)
        pos = (torch.cumsum(valid, dim=1) * valid).reshape(-1)  # fairseq utils.make_positions with padding_idx 0
        if not self.abs_positions:
            table = torch.zeros(1, self.embed_dim, device=x.device)
            pos = torch.zeros_like(pos)
        elif self.embed_positions is not None:
            table = self.embed_positions.weight
        else:
            if self._sin_table is None or self._sin_table.shape[0] < Tp + 1 or self._sin_table.device != x.device:
                from .speech_transformer_base import sinusoidal_positional_table

                self._sin_table = sinusoidal_positional_table(max(Tp + 1, 1024), self.embed_dim, 0).to(x.device)
            table = self._sin_table
        return F.add_positions(x, table, pos, self.embed_scale)

    def _fc0_weight(self):
        """fc0 consumes (c*F' + f)-ordered features in the reference; the channels-last sub-sampler
        emits f*C + c, so the weight's input axis is permuted once per call (autograd un-permutes)."""
        w = self.fc0.weight
        C = self.pre_encoder.out_channels[-1]
        Fp = w.shape[1] // C
        return w.view(w.shape[0], C, Fp).permute(0, 2, 1).reshape(w.shape[0], Fp * C)

    def encode(self, src_tokens, src_lengths, return_all_hiddens=False):
        """-> (x bf16 [B*T'][C] batch-major, lengths' [B], padding_mask [B][T'], B, T')"""
        cfg = self.cfg
        tr = self.training
        B = src_tokens.shape[0]
        p = cfg.dropout if tr else 0.0
        if tr and float(getattr(cfg.encoder, "layerdrop", 0.0) or 0.0) == 0.0 and src_tokens.is_cuda:
            self._count_batchnorm_forward(src_tokens.device)
        x, x_lengths, padding_mask, row_zero = self.pre_encoder(src_tokens, src_lengths, p_drop=p if self.fc0 is not None else 0.0)
        Tp = padding_mask.shape[1]
        if self.fc0 is not None:
            x = F.linear(x, self._fc0_weight(), self.fc0.bias)
        if self.abs_positions or self.embed_scale != 1.0:
            x = self._add_positions(x, padding_mask)
        if self.layernorm_embedding is not None:
            x = F.layer_norm(x, self.layernorm_embedding.weight, self.layernorm_embedding.bias, row_zero=row_zero, drop_p=p)
        else:  # legacy presets: dropout straight on the embedding, padded frames zeroed (speech_transformer_encoder.py:350-355)
            if p > 0:
                x = F.dropout(x, p)
            if row_zero is not None:
                x = F.zero_rows(x, row_zero)
        key_len = x_lengths.to(torch.int32).contiguous()
        attn_mask = self.get_attn_mask(Tp, x.device, in_lengths=x_lengths)
        states = [x] if return_all_hiddens else []
        # LayerDrop (fairseq/modules/layer_drop.py:38-44: one uniform draw per layer per forward, layer kept when the draw exceeds p)
        ld = float(getattr(cfg.encoder, "layerdrop", 0.0) or 0.0)
        keep = torch.empty(len(self.layers)).uniform_() > ld if (tr and ld > 0) else None
        wt_event = None
        if tr and torch.is_grad_enabled() and x.is_cuda:
            native = [l for i, l in enumerate(self.layers) if (keep is None or bool(keep[i])) and _native_conformer(l)]
            if native:  # transposed weight copies of the backward pass: refreshed off the compute stream, under this forward pass
                wt_event = F.refresh_layer_transposes(native, B, Tp)
        kept = [l for i, l in enumerate(self.layers) if keep is None or bool(keep[i])]
        if (not return_all_hiddens and x.is_cuda and all(_native_conformer(l) for l in kept) and F.conformer_stack_supported(kept, x)):
            # the whole layer loop as one C call per direction (same launches; host time at small batches: functional.py)
            l0 = kept[0]
            c0 = l0.cfg
            x = F.conformer_stack_native(x, kept, key_len, attn_mask, l0.positional_embedding[0].table(Tp, x.device), B, Tp,
                                         c0.dropout if tr else 0.0, c0.activation_dropout if tr else 0.0,
                                         c0.attention_dropout if tr else 0.0, tr)
            if tr:
                for l in kept:
                    if not getattr(l, "_counters_managed", False):
                        l.conv_module.batch_norm.num_batches_tracked += 1
            kept = []
        for i, layer in enumerate(kept):        kept = [l for i, l in enumerate(self.layers) if keep is None or bool(keep[i])]
        for i, layer in enumerate(kept):
            nxt = kept[i + 1] if i + 1 < len(kept) else None
            if nxt is not None and not return_all_hiddens and x.is_cuda and _native_conformer(layer) and _native_conformer(nxt):
                # consecutive native layers and nobody else reads the output: the two calls share the LayerNorm kernel at their boundary
                x = layer(x, B, Tp, key_len=key_len, attn_mask=attn_mask, chain_next=nxt)
            else:
                x = layer(x, B, Tp, key_len=key_len, attn_mask=attn_mask)
            if return_all_hiddens:
                states.append(x)
        if wt_event is not None:
            torch.cuda.current_stream(x.device).wait_event(wt_event)
        if self.layer_norm is not None:
            x = F.layer_norm(x, self.layer_norm.weight, self.layer_norm.bias)
        return x, x_lengths, padding_mask, B, Tp, states

    def forward(self, src_tokens, src_lengths, return_all_hiddens=False):
        x, x_lengths, padding_mask, B, Tp, states = self.encode(src_tokens, src_lengths, return_all_hiddens)
        C = x.shape[1]
        return {
            "encoder_out": [x.view(B, Tp, C).transpose(0, 1)],  # T x B x C (view of the batch-major buffer)
            "encoder_padding_mask": [padding_mask],
            "encoder_embedding": [],
            "encoder_states": [s.view(B, Tp, C).transpose(0, 1) for s in states],
            "fc_results": [],
            "src_tokens": [],
            "src_lengths": [x_lengths],
            "_x_bt": [x],
        }

    def max_positions(self):
        return self.max_source_positions

    def reorder_encoder_out(self, encoder_out, new_order):
        out = dict(encoder_out)
        out["encoder_out"] = [e.index_select(1, new_order) for e in encoder_out["encoder_out"]]
        out["encoder_padding_mask"] = [m.index_select(0, new_order) for m in encoder_out["encoder_padding_mask"]]
        out["src_lengths"] = [l.index_select(0, new_order) for l in encoder_out["src_lengths"]]
        out["encoder_states"] = [s.index_select(1, new_order) for s in encoder_out["encoder_states"]]
        out.pop("_x_bt", None)
        return out


def _native_conformer(layer) -> bool:
    """A Conformer layer that runs on the native layer runtime (csrc/engine.hip) with the sinusoidal relative-position table."""
    pe = getattr(layer, "positional_embedding", [None])[0]
    return (hasattr(layer, "conv_module") and getattr(layer, "use_native_runtime", False) and pe is not None
            and not getattr(pe, "learnable", False))


class SpeechTransformerEncoderForPrediction(SpeechTransformerEncoderBase):
    """Encoder + optional output layer `fc_out` (speech_transformer_encoder_model.py:153-210)."""

    def __init__(self, cfg, pre_encoder=None, input_size=83, vocab_size=None):
        super().__init__(cfg, pre_encoder=pre_encoder, input_size=input_size)
        self.fc_out = LinearParams(cfg.encoder.embed_dim, vocab_size) if vocab_size is not None else None

    def forward(self, src_tokens, src_lengths, return_all_hiddens=False):
        out = super().forward(src_tokens, src_lengths, return_all_hiddens=return_all_hiddens)
        if self.fc_out is not None:
            x = out.pop("_x_bt")[0]
            B, Tp = out["encoder_padding_mask"][0].shape
            # bf16 [B*T'][V] (row-padded view).  EA_LOGITS_F32=1 keeps the vocabulary logits in fp32 (the output GEMM's fp32
            # epilogue, loss gradient re-pitched by ea_cast_f32_to_bf16_rows): measured in round 5, it does NOT bring the logits
            # closer to the reference's fp32 run — max |error| 0.0313 either way on the dh64 fixture, the error is the bf16
            # storage of the 12 layers' activations, not of the logits — and costs 0.1 ms per update step (63 MB more per pass
            # at the recipe batch), so bf16 stays the default.
            logits = F.linear(x, self.fc_out.weight, self.fc_out.bias, out_f32=_LOGITS_F32)
            V = logits.shape[1]
            out["encoder_out"] = [logits.view(B, Tp, V).transpose(0, 1)]  # T x B x V
            out["_logits_bt"] = [logits]
        return out


@register_model("speech_transformer_encoder_model", dataclass=SpeechTransformerConfig)
class SpeechTransformerEncoderModel(nn.Module):
    config_class = SpeechTransformerConfig

    def __init__(self, cfg, encoder):
        super().__init__()
        self.cfg = cfg
        self.encoder = encoder
        self.num_updates = 0

    @classmethod
    def build_model(cls, cfg, task):
        if cfg.max_source_positions is None:
            cfg.max_source_positions = DEFAULT_MAX_SOURCE_POSITIONS
        ev = speech_utils.eval_str_nested_list_or_tuple
        out_channels = ev(cfg.encoder.conv_channels, type=int)
        kernel_sizes = ev(cfg.encoder.conv_kernel_sizes, type=int)
        strides = ev(cfg.encoder.conv_strides, type=int)
        assert task.feat_dim % task.feat_in_channels == 0
        conv_layers = ConvBNReLU(out_channels, kernel_sizes, strides, in_channels=task.feat_in_channels) if out_channels is not None else None
        in_size = task.feat_dim // task.feat_in_channels
        if conv_layers is not None:
            in_size = conv_layers.output_feat_dim(in_size)
        else:
            in_size = task.feat_dim
        vocab = len(task.target_dictionary) if task.target_dictionary is not None else None
        encoder = cls.build_encoder(cfg, pre_encoder=conv_layers, input_size=in_size, vocab_size=vocab)
        return cls(cfg, encoder)

    @classmethod
    def build_encoder(cls, cfg, pre_encoder=None, input_size=83, vocab_size=None):
        return SpeechTransformerEncoderForPrediction(cfg, pre_encoder=pre_encoder, input_size=input_size, vocab_size=vocab_size)

    def set_num_updates(self, num_updates):
        self.num_updates = num_updates
        self.encoder.set_num_updates(num_updates)

    def forward(self, src_tokens, src_lengths, **kwargs):
        return self.encoder(src_tokens, src_lengths, **kwargs)

    def output_lengths(self, in_lengths):
        return self.encoder.output_lengths(in_lengths)

    def max_positions(self):
        return self.encoder.max_positions()

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        """(T x B x V) fp32 (log-)probabilities — speech_transformer_encoder_model.py:141-150."""
        lg = net_output.get("_logits_bt")
        if lg:
            logits = lg[0]
            B, Tp = net_output["encoder_padding_mask"][0].shape
        else:
            e = net_output["encoder_out"][0]
            Tp, B, V = e.shape
            logits = e.transpose(0, 1).reshape(B * Tp, V)
        M, V = logits.shape
        lp = K.log_softmax(logits.detach(), M, V, logits.stride(0))
        if not log_probs:
            lp = lp.exp_()
        return lp.view(B, Tp, V).transpose(0, 1)

    def get_targets(self, sample, net_output):
        return sample["target"]

    def upgrade_state_dict_named(self, state_dict, name):
        """Accept reference checkpoints: rename conv_layers_before -> pre_encoder
        (speech_transformer_encoder.py:415-428) and drop the positional-embedding dummy buffers."""
        for k in list(state_dict.keys()):
            if "conv_layers_before" in k:
                state_dict[k.replace("conv_layers_before", "pre_encoder")] = state_dict.pop(k)
        for k in list(state_dict.keys()):
            if k.endswith("positional_embedding._float_tensor") or k.endswith("embed_positions._float_tensor"):
                state_dict.pop(k)
        return state_dict


@register_model_architecture("speech_transformer_encoder_model", "speech_transformer_encoder_model")
def base_architecture(cfg):
    return cfg


@register_model_architecture("speech_transformer_encoder_model", "speech_conformer_encoder_model_librispeech")
def conformer_ctc_librispeech(cfg=None):
    """examples/asr_librispeech/config/transformer_ctc_librispeech.yaml:66-85 + `model.encoder.layer_type=conformer`."""
    cfg = cfg or SpeechTransformerConfig()
    e = cfg.encoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = 512, 2048, 12, 8
    e.normalize_before, e.learned_pos, e.relative_positional_embeddings = True, False, True
    e.layer_type = "conformer"
    cfg.attention_dropout = cfg.activation_dropout = cfg.dropout = 0.1
    cfg.activation_fn = "relu"
    cfg.layernorm_embedding = True
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    return cfg
