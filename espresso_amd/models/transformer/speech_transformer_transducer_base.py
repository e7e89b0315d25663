"""`speech_transformer_transducer_base` — espresso/models/transformer/speech_transformer_transducer_base.py:43-330:
Conformer/Transformer encoder (no output projection) + LSTM predictor (`SpeechLSTMDecoder`, attention-free) + joint
network  logits[b,t,u] = fc_out(relu(LN(proj_encoder(h_t)) + LN(proj_decoder(g_u))))  with a weight-normalised `fc_out`
(unless the embeddings are shared).  State-dict names follow the reference (`proj_encoder`, `laynorm_proj_encoder`,
`proj_decoder`, `laynorm_proj_decoder`, `fc_out.weight_g|weight_v|bias`).

Compute: encoder and predictor run on the HIP kernels (native Conformer runtime, LSTM layer op); the joint is
functional._TransducerJoint — broadcast add + ReLU kernel, then one MFMA GEMM over all B*T'*(U+1) lattice nodes into bf16
logits, which csrc/rnnt.hip consumes directly (bf16 in, bf16 gradient out)."""
import math

import torch
import torch.nn as nn

from ... import functional as F
from ...modules.params import LayerNormParams
from ...modules.speech_convolutions import ConvBNReLU
from ...registry import register_model
from ...tools import utils as speech_utils
from ..speech_lstm import SpeechLSTMDecoder, lstm_linear
from .speech_transformer_base import EmbeddingParams
from .speech_transformer_config import SpeechTransformerTransducerConfig
from .speech_transformer_encoder_model import SpeechTransformerEncoderBase


class WeightNormLinearParams(nn.Module):
    """nn.utils.weight_norm(nn.Linear(in, out), name="weight"): weight = weight_v * (weight_g / ||weight_v||_row)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        v = torch.empty(out_features, in_features)
        nn.init.kaiming_uniform_(v, a=math.sqrt(5))
        self.weight_g = nn.Parameter(v.norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(v)
        bound = 1 / math.sqrt(in_features)
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))

    def effective_weight(self):
        if not torch.is_grad_enabled():  # decoding: the normalised weight is a constant, keep it (and its bf16 cast) per version
            ver = (self.weight_v._version, self.weight_g._version, self.weight_v.data_ptr())
            hit = getattr(self, "_infer_weight", None)
            if hit is None or hit[0] != ver:
                hit = (ver, self.weight_v * (self.weight_g / self.weight_v.norm(dim=1, keepdim=True)))
                self._infer_weight = hit
            return hit[1]
        return self.weight_v * (self.weight_g / self.weight_v.norm(dim=1, keepdim=True))


@register_model("speech_transformer_transducer_base")
class SpeechTransformerTransducerModelBase(nn.Module):
    config_class = SpeechTransformerTransducerConfig

    def __init__(self, cfg, encoder, decoder):
        super().__init__()
        self.cfg, self.encoder, self.decoder = cfg, encoder, decoder
        J = cfg.joint_dim
        self.proj_encoder = lstm_linear(cfg.encoder.embed_dim, J)
        self.laynorm_proj_encoder = LayerNormParams(J)
        self.proj_decoder = lstm_linear(cfg.decoder.hidden_size, J)
        self.laynorm_proj_decoder = LayerNormParams(J)
        V = self.decoder.embed_tokens.weight.shape[0]
        self.share_embed = bool(cfg.share_decoder_input_output_embed)
        if self.share_embed:
            assert J == cfg.decoder.embed_dim, "joint_dim and decoder.embed_dim must be the same if the two embeddings are to be shared"
            self.fc_out_bias = nn.Parameter(torch.empty(V).uniform_(-1 / math.sqrt(J), 1 / math.sqrt(J)))
        else:
            self.fc_out = WeightNormLinearParams(J, V)
        self.num_updates = 0

    @classmethod
    def build_model(cls, cfg, task):
        ev = speech_utils.eval_str_nested_list_or_tuple
        tgt_dict = task.target_dictionary
        embed = EmbeddingParams(len(tgt_dict), cfg.decoder.embed_dim, tgt_dict.pad())
        out_channels = ev(cfg.encoder.conv_channels, type=int)
        conv = ConvBNReLU(out_channels, ev(cfg.encoder.conv_kernel_sizes, type=int), ev(cfg.encoder.conv_strides, type=int),
                          in_channels=task.feat_in_channels)
        in_size = conv.output_feat_dim(task.feat_dim // task.feat_in_channels)
        encoder = SpeechTransformerEncoderBase(cfg, pre_encoder=conv, input_size=in_size)
        d = cfg.decoder
        decoder = SpeechLSTMDecoder(tgt_dict, embed_dim=d.embed_dim, hidden_size=d.hidden_size, out_embed_dim=d.hidden_size,
                                    num_layers=d.layers, dropout_in=d.dropout_in if d.dropout_in is not None else cfg.dropout,
                                    dropout_out=d.dropout_out if d.dropout_out is not None else cfg.dropout, residual=d.residual,
                                    pretrained_embed=embed, share_input_output_embed=True)
        return cls(cfg, encoder, decoder)

    def set_num_updates(self, n):
        self.num_updates = n
        self.encoder.set_num_updates(n)

    def fc_out_params(self):
        if self.share_embed:
            return self.decoder.embed_tokens.weight, self.fc_out_bias
        return self.fc_out.effective_weight(), self.fc_out.bias

    def _joint_decoder_branch(self, dec_bu):
        return F.layer_norm(F.linear(dec_bu, self.proj_decoder.weight, self.proj_decoder.bias), self.laynorm_proj_decoder.weight,
                            self.laynorm_proj_decoder.bias, out_f32=True)

    def joint(self, enc_bt, dec_bu, B, T, U1, apply_output_layer=True, _D=None, _out=None, lazy=False):
        """enc_bt bf16 [B*T][C], dec_bu bf16 [B*U1][H] -> bf16 logits [B][T][U1][V] (:276-299).  The two LayerNorm outputs E, D
        stay fp32 and relu(E + D) is evaluated in fp32 (:292-294 under autocast); only fc_out's operand is rounded to bf16.
        (_D: the predictor branch already projected + normalised; _out: (w, b, holder) from F.joint_weight_late — forward())"""
        E = F.layer_norm(F.linear(enc_bt, self.proj_encoder.weight, self.proj_encoder.bias), self.laynorm_proj_encoder.weight,
                         self.laynorm_proj_encoder.bias, out_f32=True)
        D = _D if _D is not None else self._joint_decoder_branch(dec_bu)
        if not apply_output_layer:
            raise NotImplementedError("joint features without the output layer are never materialised (B*T*U*J)")
        w, b, late = _out if _out is not None else (self.fc_out_params() + (None,))
        if lazy:  # the criterion fuses the output layer with the loss (F.joint_rnnt_loss): the logits are never written
            return F.LazyJointLogits(E, D, w, b, B, T, U1, late=late)
        return F.transducer_joint(E, D, w, b, B, T, U1, late=late)

    # ---- inference helpers: the encoder branch of the joint is computed once per utterance batch, the predictor branch
    # once per expansion (espresso/tools/transducer_greedy_decoder.py:163-176 evaluates joint() on one frame at a time)
    @torch.no_grad()
    def joint_encoder_branch(self, enc_bt):
        return F.layer_norm(F.linear(enc_bt, self.proj_encoder.weight, self.proj_encoder.bias), self.laynorm_proj_encoder.weight,
                            self.laynorm_proj_encoder.bias, out_f32=True)

    @torch.no_grad()
    def joint_step(self, E_rows, dec_rows):
        """E_rows fp32 [N][J] (already projected + normalised), dec_rows bf16 [N][H] -> fp32 logits [N][V]."""
        from ... import kernels as K

        N = E_rows.shape[0]
        D = F.layer_norm(F.linear(dec_rows, self.proj_decoder.weight, self.proj_decoder.bias), self.laynorm_proj_decoder.weight,
                         self.laynorm_proj_decoder.bias, out_f32=True)
        Z = K.joint_add_relu(E_rows.contiguous(), D.contiguous(), N, 1, 1)
        w, b = self.fc_out_params()
        # (without grad mode `w` is the cached constant itself: its bf16 cast is then cached on it too)
        return F.linear(Z, w.detach() if w.requires_grad else w, b, out_f32=True)

    supports_lazy_joint = True  # forward(..., lazy_joint=True) -> F.LazyJointLogits for the fused loss (criterions/transducer_loss.py)

    def forward(self, src_tokens, src_lengths, prev_output_tokens, lazy_joint=False, **kwargs):
        """-> (logits bf16 [B][T'][U+1][V], encoder_out_lengths [B])  (:221-243).  lazy_joint (set by the `transducer_loss`
        criterion): the first element is a `F.LazyJointLogits` — the joint's two branches and output layer, for the fused loss."""
        dev = src_tokens.device
        lazy_joint = bool(lazy_joint) and torch.is_grad_enabled()
        B, U1 = prev_output_tokens.shape
        if not (F.branch_overlap() and dev.type == "cuda" and torch.is_grad_enabled()):
            enc = self.encoder(src_tokens, src_lengths)
            x = enc["_x_bt"][0]
            dec, _ = self.decoder.extract_features(prev_output_tokens)
            return self.joint(x, dec.reshape(B * U1, -1), B, x.shape[0] // B, U1, lazy=lazy_joint), enc["src_lengths"][0]
        # The predictor network depends on the targets only: it runs on its own stream next to the encoder (5-6 utterances per
        # product-rule batch leave most of the device idle under either), and autograd runs its backward pass on that stream
        # too.  Host order: encoder first, predictor second -> the engine issues the predictor's backward first.  The output
        # layer's weight gradient overlaps the encoder's backward pass the same way (F.joint_weight_late).
        w, b = self.fc_out_params()
        out = F.joint_weight_late(w, b) if (w.requires_grad and b is not None) else (w, b, None)
        cur, side = torch.cuda.current_stream(dev), F.aux_stream(dev, 1)
        start = cur.record_event()  # the inputs, the updated weights and the cleared accumulator pool are all behind this point
        enc = self.encoder(src_tokens, src_lengths)
        x = enc["_x_bt"][0]
        with torch.cuda.stream(side):
            side.wait_event(start)
            dec, _ = self.decoder.extract_features(prev_output_tokens)
            D = self._joint_decoder_branch(dec.reshape(B * U1, -1))
        cur.wait_stream(side)
        D.record_stream(cur)
        return self.joint(x, None, B, x.shape[0] // B, U1, _D=D, _out=out, lazy=lazy_joint), enc["src_lengths"][0]

    def forward_encoder(self, src_tokens, src_lengths):
        return self.encoder(src_tokens, src_lengths)

    def max_positions(self):
        return (self.encoder.max_positions(), self.decoder.max_positions())

    def max_decoder_positions(self):
        return self.decoder.max_positions()

    def get_targets(self, sample, net_output):
        return sample["target"]

    def upgrade_state_dict_named(self, state_dict, name):
        for k in list(state_dict.keys()):
            if "conv_layers_before" in k:
                state_dict[k.replace("conv_layers_before", "pre_encoder")] = state_dict.pop(k)
        for k in list(state_dict.keys()):
            if k.endswith("positional_embedding._float_tensor") or k.endswith("embed_positions._float_tensor"):
                state_dict.pop(k)
        return state_dict
