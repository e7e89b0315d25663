"""`speech_lstm` — espresso/models/speech_lstm.py: `SpeechLSTMModel` (:169-357), `SpeechLSTMEncoder` (:358-598, conv front-end +
packed BiLSTM stack) and `SpeechLSTMDecoder` (:600-1048).

The decoder serves three roles: attention-free (attn_type None) as the predictor of the transducer
(espresso/models/transformer/speech_transformer_transducer_base.py:195-212) and as the decoder-only LSTM language model
(espresso/models/lstm_lm.py:88-198); with Bahdanau attention + input feeding as the decoder of the `speech_lstm`
encoder-decoder (BASELINE config 1).  Parameter names follow the reference's state_dict (`embed_tokens.weight`,
`layers.{i}.weight_ih|weight_hh|bias_ih|bias_hh`, optional `additional_fc`, `fc_out`).  Without attention / input feeding
the per-step loop of :846-893 factorises into stacked sequence LSTMs, so each layer runs over the whole teacher-forced
sequence (one input-projection GEMM + per-step recurrent GEMM + cell kernel; functional._LSTMLayer); incremental decoding
keeps (h, c) per layer and advances one step at a time (`step`), reordered by the surviving beams with a gather kernel."""
import torch
import torch.nn as nn

from .. import functional as F
from .. import kernels as K
from ..modules.params import LinearParams
from ..registry import register_model


class LSTMCellParams(nn.Module):
    """torch.nn.LSTMCell storage, init U(-0.1, 0.1) like fairseq/models/lstm.py:LSTMCell."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.weight_ih = nn.Parameter(torch.empty(4 * hidden_size, input_size).uniform_(-0.1, 0.1))
        self.weight_hh = nn.Parameter(torch.empty(4 * hidden_size, hidden_size).uniform_(-0.1, 0.1))
        self.bias_ih = nn.Parameter(torch.empty(4 * hidden_size).uniform_(-0.1, 0.1))
        self.bias_hh = nn.Parameter(torch.empty(4 * hidden_size).uniform_(-0.1, 0.1))


class LSTMEmbedding(nn.Module):
    """fairseq/models/lstm.py:Embedding — U(-0.1, 0.1), pad row zero."""

    def __init__(self, num_embeddings, dim, padding_idx):
        super().__init__()
        self.padding_idx, self.embedding_dim, self.num_embeddings = padding_idx, dim, num_embeddings
        self.weight = nn.Parameter(torch.empty(num_embeddings, dim).uniform_(-0.1, 0.1))
        nn.init.constant_(self.weight[padding_idx], 0)


def lstm_linear(in_features, out_features, bias=True):
    """fairseq/models/lstm.py:Linear — U(-0.1, 0.1) weights and bias."""
    m = LinearParams(in_features, out_features, bias=bias)
    m.weight.data.uniform_(-0.1, 0.1)
    if bias:
        m.bias.data.uniform_(-0.1, 0.1)
    return m


class SpeechLSTMDecoder(nn.Module):
    def __init__(self, dictionary, embed_dim=512, hidden_size=512, out_embed_dim=512, num_layers=1, dropout_in=0.1,
                 dropout_out=0.1, encoder_output_units=0, attn_type=None, attn_dim=0, need_attn=False, residual=False,
                 pretrained_embed=None, share_input_output_embed=False, max_target_positions=1024):
        super().__init__()
        self.dictionary = dictionary
        self.dropout_in, self.dropout_out = float(dropout_in), float(dropout_out)
        self.hidden_size, self.num_layers, self.residual = hidden_size, num_layers, residual
        self.share_input_output_embed = share_input_output_embed
        self.max_target_positions = max_target_positions
        no_attn = attn_type is None or str(attn_type).lower() == "none"
        if no_attn:
            encoder_output_units = 0
        self.encoder_output_units = encoder_output_units
        self.need_attn = need_attn and not no_attn
        pad = dictionary.pad()
        self.embed_tokens = pretrained_embed if pretrained_embed is not None else LSTMEmbedding(len(dictionary), embed_dim, pad)
        embed_dim = self.embed_tokens.embedding_dim
        self.layers = nn.ModuleList([LSTMCellParams(encoder_output_units + (embed_dim if i == 0 else hidden_size), hidden_size)
                                     for i in range(num_layers)])
        if no_attn:
            self.attention = None
        elif str(attn_type).lower() == "bahdanau":
            self.attention = BahdanauAttentionParams(hidden_size, encoder_output_units, attn_dim)
        else:
            raise NotImplementedError(f"attention type {attn_type} (the recipes use bahdanau)")
        if hidden_size + encoder_output_units != out_embed_dim:
            self.additional_fc = lstm_linear(hidden_size + encoder_output_units, out_embed_dim)
        if not share_input_output_embed:
            self.fc_out = lstm_linear(out_embed_dim, len(dictionary))

    def max_positions(self):
        return self.max_target_positions

    # ---------------------------------------------------------------- teacher-forced path
    def extract_features(self, prev_output_tokens, encoder_out=None, **unused):
        """prev_output_tokens [B][U] -> (features bf16 [B][U][H_out], attn or None).  speech_lstm.py:766-919."""
        if self.attention is not None:
            return self._extract_features_attention(prev_output_tokens, encoder_out)
        B, U = prev_output_tokens.shape
        tr = self.training
        tok = prev_output_tokens.t().contiguous().view(-1).to(torch.int32)  # time-major rows t*B + b
        x = F.embedding(self.embed_tokens.weight, tok, None, None, 1.0, self.embed_tokens.padding_idx)
        if tr and self.dropout_in > 0:
            x = F.dropout(x, self.dropout_in)
        for i, cell in enumerate(self.layers):
            hs, _, _ = F.lstm_layer(x, cell, B, U)
            out = F.dropout(hs, self.dropout_out) if (tr and self.dropout_out > 0) else hs
            if self.residual and i > 0:
                out = out + x
            x = out
        H = x.shape[1]
        x = x.view(U, B, H).transpose(0, 1).contiguous()  # B x U x H
        if hasattr(self, "additional_fc"):
            x = F.linear(x.view(B * U, H), self.additional_fc.weight, self.additional_fc.bias)
            if tr and self.dropout_out > 0:
                x = F.dropout(x, self.dropout_out)
            x = x.view(B, U, -1)
        return x, None

    def output_layer(self, features):
        """features bf16 [..., H] -> fp32 logits [..., V] (speech_lstm.py:921-930)."""
        shp = features.shape
        f2 = features.reshape(-1, shp[-1])
        if self.share_input_output_embed:
            y = F.linear(f2, self.embed_tokens.weight, None, out_f32=True)
        else:
            y = F.linear(f2, self.fc_out.weight, self.fc_out.bias, out_f32=True)
        return y[:, : len(self.dictionary)].reshape(*shp[:-1], -1)

    def forward(self, prev_output_tokens, encoder_out=None, incremental_state=None, epoch=1, **kwargs):
        sched = getattr(self, "scheduled_sampling_rate_scheduler", None)
        if self.training and sched is not None and self.attention is not None:
            p = sched.step(epoch)
            if p < 1.0:  # scheduled sampling (:735-764): feed the model's own previous prediction with probability 1 - p
                return self._extract_features_attention(prev_output_tokens, encoder_out, sampling_prob=p)
        x, attn = self.extract_features(prev_output_tokens, encoder_out=encoder_out)
        return self.output_layer(x), attn

    def _extract_features_attention(self, prev_output_tokens, encoder_out, sampling_prob=1.0):
        """Attention decoder with input feeding (:846-893): per step layer-0 cell -> Bahdanau attention on its hidden state ->
        the context is appended to every upper layer's input and fed to the next step's layer 0.  Each step is a handful of
        autograd nodes on the HIP kernels; weight / key / value gradients are accumulated in place (functional.GradSink)."""
        B, U = prev_output_tokens.shape
        tr = self.training
        enc = encoder_out["_x_tb"][0]                      # bf16 [T*B][Cv] time-major
        lens = encoder_out["src_lengths"][0].to(torch.int32).contiguous()
        T = enc.shape[0] // B
        H, Cv = self.hidden_size, self.encoder_output_units
        at = self.attention
        key = F.linear(enc, at.value_proj.weight, None)    # [T*B][A]
        kv = F.GradSink(key, enc)
        nv = at.g * at.v / torch.norm(at.v)
        sinks = [F.GradSink(c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh) for c in self.layers]
        bsum = [(c.bias_ih + c.bias_hh).detach().float().contiguous() for c in self.layers]
        sampling = sampling_prob < 1.0
        dev = enc.device
        if not sampling:
            tok = prev_output_tokens.t().contiguous().view(-1).to(torch.int32)
            x = F.embedding(self.embed_tokens.weight, tok, None, None, 1.0, self.embed_tokens.padding_idx)
            if tr and self.dropout_in > 0:
                x = F.dropout(x, self.dropout_in)
        h = [torch.zeros(B, H, dtype=torch.bfloat16, device=dev) for _ in self.layers]
        c = [torch.zeros(B, H, dtype=torch.float32, device=dev) for _ in self.layers]
        feed = torch.zeros(B, Cv, dtype=torch.bfloat16, device=dev)
        outs, step_logits, pred = [], [], None
        for j in range(U):
            if sampling:
                gold = prev_output_tokens[:, j]
                if j > 0:
                    keep_gold = torch.rand(B, device=dev).lt(sampling_prob)
                    gold = torch.where(keep_gold, gold, pred)
                xj = F.embedding(self.embed_tokens.weight, gold.to(torch.int32).contiguous(), None, None, 1.0, self.embed_tokens.padding_idx)
                if tr and self.dropout_in > 0:
                    xj = F.dropout(xj, self.dropout_in)
            else:
                xj = x[j * B:(j + 1) * B]
            inp = torch.cat((xj, feed), dim=1)
            ctx = None
            for i, cell in enumerate(self.layers):
                h[i], c[i] = F.lstm_cell_ag(inp, h[i], c[i], cell, sinks[i], bsum[i])
                prev_in = inp[:, :H] if (self.residual and i > 0) else None
                if i == 0:
                    qp = F.linear(h[0], at.query_proj.weight, None)
                    ctx, _ = F.bahdanau_step(qp, kv, nv, at.b, lens, T, B)
                inp = torch.cat((h[i], ctx), dim=1)
                if tr and self.dropout_out > 0:
                    inp = F.dropout(inp, self.dropout_out)
                if prev_in is not None:
                    inp = torch.cat((inp[:, :H] + prev_in, inp[:, H:]), dim=1)
            feed = ctx
            outs.append(inp)
            if sampling:
                yj = inp
                if hasattr(self, "additional_fc"):
                    yj = F.linear(yj.contiguous(), self.additional_fc.weight, self.additional_fc.bias)
                    if tr and self.dropout_out > 0:
                        yj = F.dropout(yj, self.dropout_out)
                lg = self.output_layer(yj)          # fp32 [B][V]
                step_logits.append(lg)
                pred = lg.detach().argmax(-1)
        if sampling:
            return torch.stack(step_logits, 1), None  # B x U x V
        y = torch.stack(outs, 0).transpose(0, 1).contiguous()  # B x U x (H + Cv)
        if hasattr(self, "additional_fc"):
            y = F.linear(y.view(B * U, -1), self.additional_fc.weight, self.additional_fc.bias)
            if tr and self.dropout_out > 0:
                y = F.dropout(y, self.dropout_out)
            y = y.view(B, U, -1)
        return y, None

    # ---------------------------------------------------------------- incremental path (inference)
    def init_state(self, n, device):
        """Zero (h, c) for n hypotheses — speech_lstm.py:932-952 initialize_cached_state."""
        z32 = lambda: torch.zeros(n, self.hidden_size, dtype=torch.float32, device=device)
        z16 = lambda: torch.zeros(n, self.hidden_size, dtype=torch.bfloat16, device=device)
        return {"h16": [z16() for _ in self.layers], "h32": [z32() for _ in self.layers], "c": [z32() for _ in self.layers]}

    @torch.no_grad()
    def advance(self, tokens, state, keep_row=None):
        """Attention-free single step (predictor / LM).
        tokens int [N] (last emitted token of each hypothesis) -> (features bf16 [N][H_out], new state).
        keep_row uint8 [N]: rows whose state must not advance (speech_lstm.py:1001-1040 masked_copy_cached_state)."""
        tok = tokens.view(-1).to(torch.int32).contiguous()
        x = F.embedding(self.embed_tokens.weight, tok, None, None, 1.0, self.embed_tokens.padding_idx)
        new = {"h16": [], "h32": [], "c": []}
        for i, cell in enumerate(self.layers):
            h16, h32, c = F.lstm_cell_step(x, cell, state["h16"][i], state["h32"][i], state["c"][i], keep_row=keep_row)
            if keep_row is not None:
                h16 = K.cast_f32_to_bf16(h32)  # frozen rows carry h_prev: rebuild the bf16 copy from the fp32 state
            new["h16"].append(h16)
            new["h32"].append(h32)
            new["c"].append(c)
            out = h16
            if self.residual and i > 0:
                out = out + x
            x = out
        if hasattr(self, "additional_fc"):
            x = F.linear(x, self.additional_fc.weight, self.additional_fc.bias)
        return x, new

    # ---------------------------------------------------------------- SequenceGenerator hooks (attention decoder)
    @torch.no_grad()
    def init_incremental(self, encoder_out, bsz, beam):
        """Encoder keys (value_proj applied once) and values stay ONE copy per sentence; hypotheses address them through
        `kv_col` (the reference re-orders / replicates encoder_out per beam, speech_lstm.py:531-566)."""
        assert self.attention is not None
        enc = encoder_out["_x_tb"][0]
        dev = enc.device
        T = enc.shape[0] // bsz
        N = bsz * beam
        H, Cv = self.hidden_size, self.encoder_output_units
        at = self.attention
        return {
            "T": T, "Bkv": bsz, "value": enc, "key": F.linear(enc, at.value_proj.weight, None).contiguous(),
            "len": encoder_out["src_lengths"][0].to(torch.int32).contiguous(),
            "nv": (at.g * at.v / torch.norm(at.v)).float().contiguous(), "bias": at.b.detach().float().contiguous(),
            "kv_col": torch.arange(bsz, device=dev, dtype=torch.int32).repeat_interleave(beam).contiguous(),
            "h16": [torch.zeros(N, H, dtype=torch.bfloat16, device=dev) for _ in self.layers],
            "h32": [torch.zeros(N, H, dtype=torch.float32, device=dev) for _ in self.layers],
            "c": [torch.zeros(N, H, dtype=torch.float32, device=dev) for _ in self.layers],
            "feed": torch.zeros(N, Cv, dtype=torch.bfloat16, device=dev),
        }

    @torch.no_grad()
    def step(self, st, tokens, step, parent):
        """One beam-search step of the attention decoder: tokens [N][step+1], parent int64 [N] (rows of the previous step each
        hypothesis continues; None at step 0) -> fp32 log-probs [N][V]."""
        if parent is not None:
            idx = parent.to(torch.int32).contiguous()
            for k in ("h16", "h32", "c"):
                st[k] = [K.gather_rows(t, idx) for t in st[k]]
            st["feed"] = K.gather_rows(st["feed"], idx)
            st["kv_col"] = K.gather_rows(st["kv_col"].view(-1, 1).view(torch.float32), idx).view(torch.int32).view(-1)
        N = tokens.shape[0]
        H = self.hidden_size
        x = F.embedding(self.embed_tokens.weight, tokens[:, -1].to(torch.int32).contiguous(), None, None, 1.0, self.embed_tokens.padding_idx)
        inp = torch.cat((x, st["feed"]), dim=1)
        ctx = None
        for i, cell in enumerate(self.layers):
            h16, h32, c = F.lstm_cell_step(inp, cell, st["h16"][i], st["h32"][i], st["c"][i])
            prev_in = inp[:, :H] if (self.residual and i > 0) else None
            if i == 0:
                qp = F.linear(h16, self.attention.query_proj.weight, None)
                _, ctx = K.bahdanau_fwd(qp.contiguous(), st["key"], st["value"], st["nv"], st["bias"], st["len"], st["T"], N,
                                        kv_col=st["kv_col"], Bkv=st["Bkv"])
            inp = torch.cat((h16, ctx), dim=1)
            if prev_in is not None:
                inp = torch.cat((inp[:, :H] + prev_in, inp[:, H:]), dim=1)
            st["h16"][i], st["h32"][i], st["c"][i] = h16, h32, c
        st["feed"] = ctx
        y = inp
        if hasattr(self, "additional_fc"):
            y = F.linear(y.contiguous(), self.additional_fc.weight, self.additional_fc.bias)
        logits = self.output_layer(y)
        return K.log_softmax(logits, N, logits.shape[1], logits.stride(0))

    @staticmethod
    def reorder_state(state, new_order):
        """index_select of every cached tensor by the surviving beams (speech_lstm.py:981-999)."""
        idx = new_order.to(torch.int32).contiguous()
        return {k: [K.gather_rows(t.contiguous(), idx) for t in v] for k, v in state.items()}


class ScheduledSamplingRateScheduler:
    """espresso/tools/scheduled_sampling_rate_scheduler.py:9-41: probability of feeding the TRUE previous token, per epoch."""

    def __init__(self, scheduled_sampling_probs=(1.0,), start_scheduled_sampling_epoch=1):
        self.scheduled_sampling_probs = list(scheduled_sampling_probs)
        self.start_scheduled_sampling_epoch = start_scheduled_sampling_epoch

    def step(self, epoch: int) -> float:
        ps = self.scheduled_sampling_probs
        if (len(ps) > 1 or ps[0] < 1.0) and epoch >= self.start_scheduled_sampling_epoch:
            return ps[min(epoch - self.start_scheduled_sampling_epoch, len(ps) - 1)]
        return 1.0


class BahdanauAttentionParams(nn.Module):
    """espresso/modules/speech_attention.py:38-64 storage (normalize=True): query_proj, value_proj (no bias), v, b, g."""

    def __init__(self, query_dim, value_dim, embed_dim):
        super().__init__()
        import math

        self.query_proj = LinearParams(query_dim, embed_dim, bias=False)
        self.value_proj = LinearParams(value_dim, embed_dim, bias=False)
        self.query_proj.weight.data.uniform_(-0.1, 0.1)
        self.value_proj.weight.data.uniform_(-0.1, 0.1)
        self.v = nn.Parameter(torch.empty(embed_dim).uniform_(-0.1, 0.1))
        self.b = nn.Parameter(torch.zeros(embed_dim))
        self.g = nn.Parameter(torch.full((1,), math.sqrt(1.0 / embed_dim)))


class LSTMParams(nn.Module):
    """Single-layer torch.nn.LSTM storage (weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0 and the *_reverse set),
    init U(-0.1, 0.1) like fairseq/models/lstm.py:LSTM."""

    def __init__(self, input_size, hidden_size, bidirectional=False):
        super().__init__()
        self.bidirectional = bidirectional
        for sfx in ([""] + (["_reverse"] if bidirectional else [])):
            setattr(self, "weight_ih_l0" + sfx, nn.Parameter(torch.empty(4 * hidden_size, input_size).uniform_(-0.1, 0.1)))
            setattr(self, "weight_hh_l0" + sfx, nn.Parameter(torch.empty(4 * hidden_size, hidden_size).uniform_(-0.1, 0.1)))
            setattr(self, "bias_ih_l0" + sfx, nn.Parameter(torch.empty(4 * hidden_size).uniform_(-0.1, 0.1)))
            setattr(self, "bias_hh_l0" + sfx, nn.Parameter(torch.empty(4 * hidden_size).uniform_(-0.1, 0.1)))

    def direction(self, sfx):
        return tuple(getattr(self, n + sfx) for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"))


class SpeechLSTMEncoder(nn.Module):
    """espresso/models/speech_lstm.py:358-598: ConvBNReLU front-end, then `num_layers` single-layer (Bi)LSTMs over packed
    sequences (state frozen and output zero beyond each utterance's length), dropout between layers, optional residuals."""

    def __init__(self, pre_encoder=None, input_size=83, hidden_size=512, num_layers=1, dropout_in=0.1, dropout_out=0.1,
                 bidirectional=False, residual=False, max_source_positions=3600):
        super().__init__()
        self.pre_encoder = pre_encoder
        self.num_layers, self.hidden_size, self.bidirectional, self.residual = num_layers, hidden_size, bidirectional, residual
        self.dropout_in, self.dropout_out = float(dropout_in), float(dropout_out)
        self.max_source_positions = max_source_positions
        d = 2 if bidirectional else 1
        self.lstm = nn.ModuleList([LSTMParams(input_size if i == 0 else d * hidden_size, hidden_size, bidirectional)
                                   for i in range(num_layers)])
        self.output_units = d * hidden_size

    def output_lengths(self, in_lengths):
        return in_lengths if self.pre_encoder is None else self.pre_encoder.output_lengths(in_lengths)

    def max_positions(self):
        return self.max_source_positions

    def _first_layer_weight(self, w):
        """The channels-last sub-sampler emits features ordered (f, c); the reference's LSTM expects (c, f): permute the input
        axis of layer 0's weight_ih once per call (autograd un-permutes), as the Transformer encoder does for fc0."""
        C = self.pre_encoder.out_channels[-1]
        Fp = w.shape[1] // C
        return w.view(w.shape[0], C, Fp).permute(0, 2, 1).reshape(w.shape[0], Fp * C)

    def forward(self, src_tokens, src_lengths, **unused):
        tr = self.training
        B = src_tokens.shape[0]
        x, x_lengths, padding_mask, _ = self.pre_encoder(src_tokens, src_lengths, p_drop=self.dropout_in if tr else 0.0)
        T = padding_mask.shape[1]
        D = x.shape[1]
        x = x.view(B, T, D).transpose(0, 1).contiguous().view(T * B, D)  # time-major rows t*B + b
        frozen = padding_mask.t().contiguous().to(torch.uint8)            # [T][B], 1 where t >= length
        for i, layer in enumerate(self.lstm):
            prev_x = x
            outs = []
            for sfx, rev in ((("", False),) + ((("_reverse", True),) if self.bidirectional else ())):
                w_ih, w_hh, b_ih, b_hh = layer.direction(sfx)
                if i == 0 and self.pre_encoder is not None:
                    w_ih = self._first_layer_weight(w_ih)
                outs.append(F.lstm_direction(x, w_ih, w_hh, b_ih, b_hh, B, T, reverse=rev, frozen=frozen))
            x = torch.cat(outs, dim=1) if len(outs) > 1 else outs[0]
            if i < len(self.lstm) - 1 and tr and self.dropout_out > 0:
                x = F.dropout(x, self.dropout_out)
            if self.residual and i > 0:
                x = x + prev_x
        C = x.shape[1]
        return {
            "encoder_out": [x.view(T, B, C)],                      # T x B x C
            "encoder_padding_mask": [padding_mask.t()] if bool(padding_mask.any()) else [],  # T x B
            "encoder_embedding": [], "encoder_states": [], "src_tokens": [],
            "src_lengths": [x_lengths],
            "_x_tb": [x],
        }


ARCHS = {
    "speech_lstm": dict(dropout=0.4, encoder_conv_channels="[64, 64, 128, 128]", encoder_conv_kernel_sizes="[(3, 3), (3, 3), (3, 3), (3, 3)]",
                        encoder_conv_strides="[(1, 1), (2, 2), (1, 1), (2, 2)]", encoder_rnn_hidden_size=320, encoder_rnn_layers=3,
                        encoder_rnn_bidirectional=True, encoder_rnn_residual=False, decoder_embed_dim=48, decoder_hidden_size=320,
                        decoder_layers=3, decoder_out_embed_dim=960, decoder_rnn_residual=True, attention_type="bahdanau",
                        attention_dim=320, need_attention=False, share_decoder_input_output_embed=False),
}
ARCHS["speech_conv_lstm_wsj"] = dict(ARCHS["speech_lstm"])
ARCHS["speech_conv_lstm_librispeech"] = dict(ARCHS["speech_lstm"], dropout=0.3, encoder_rnn_hidden_size=1024, encoder_rnn_layers=4,
                                             decoder_embed_dim=512, decoder_hidden_size=1024, decoder_layers=3,
                                             decoder_out_embed_dim=3072, attention_dim=512)
ARCHS["speech_conv_lstm_swbd"] = dict(ARCHS["speech_lstm"], dropout=0.5, encoder_rnn_hidden_size=640, encoder_rnn_layers=4,
                                      decoder_embed_dim=640, decoder_hidden_size=640, decoder_layers=3, decoder_out_embed_dim=1920,
                                      attention_dim=640)


@register_model("speech_lstm")
class SpeechLSTMModel(nn.Module):
    """`speech_lstm` (espresso/models/speech_lstm.py:169-357): encoder-decoder with attention; `forward` returns
    (fp32 logits [B][U][V], attention or None) like the reference's decoder output."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder
        self.num_updates = 0

    @classmethod
    def build_model(cls, args, task):
        from ..modules.speech_convolutions import ConvBNReLU
        from ..tools import utils as speech_utils

        a = dict(ARCHS[(args.get("arch") if isinstance(args, dict) else getattr(args, "arch", None)) or "speech_lstm"])
        a.update({k: v for k, v in (args if isinstance(args, dict) else vars(args)).items() if v is not None})
        ev = speech_utils.eval_str_nested_list_or_tuple
        ch = ev(a["encoder_conv_channels"], type=int)
        conv = ConvBNReLU(ch, ev(a["encoder_conv_kernel_sizes"], type=int), ev(a["encoder_conv_strides"], type=int),
                          in_channels=task.feat_in_channels)
        in_size = conv.output_feat_dim(task.feat_dim // task.feat_in_channels)
        drop = a["dropout"]
        enc = SpeechLSTMEncoder(pre_encoder=conv, input_size=in_size, hidden_size=a["encoder_rnn_hidden_size"],
                                num_layers=a["encoder_rnn_layers"], dropout_in=a.get("encoder_rnn_dropout_in", drop),
                                dropout_out=a.get("encoder_rnn_dropout_out", drop), bidirectional=a["encoder_rnn_bidirectional"],
                                residual=a["encoder_rnn_residual"], max_source_positions=a.get("max_source_positions", 3600))
        if a["share_decoder_input_output_embed"] and a["decoder_embed_dim"] != a["decoder_out_embed_dim"]:
            raise ValueError("--share-decoder-input-output-embed requires --decoder-embed-dim to match --decoder-out-embed-dim")
        dec = SpeechLSTMDecoder(task.target_dictionary, embed_dim=a["decoder_embed_dim"], hidden_size=a["decoder_hidden_size"],
                                out_embed_dim=a["decoder_out_embed_dim"], num_layers=a["decoder_layers"],
                                dropout_in=a.get("decoder_dropout_in", drop), dropout_out=a.get("decoder_dropout_out", drop),
                                encoder_output_units=enc.output_units, attn_type=a["attention_type"], attn_dim=a["attention_dim"],
                                need_attn=a["need_attention"], residual=a["decoder_rnn_residual"],
                                share_input_output_embed=a["share_decoder_input_output_embed"],
                                max_target_positions=a.get("max_target_positions", 1024))
        probs = a.get("scheduled_sampling_probs", [1.0])
        probs = [float(x) for x in (probs.split(",") if isinstance(probs, str) else probs)]
        dec.scheduled_sampling_rate_scheduler = ScheduledSamplingRateScheduler(probs, a.get("start_scheduled_sampling_epoch", 1))
        return cls(enc, dec)

    def set_num_updates(self, n):
        self.num_updates = n

    def forward(self, src_tokens, src_lengths, prev_output_tokens, epoch=1, **kwargs):
        enc = self.encoder(src_tokens, src_lengths)
        return self.decoder(prev_output_tokens, encoder_out=enc, epoch=epoch)

    def forward_encoder(self, src_tokens, src_lengths):
        return self.encoder(src_tokens, src_lengths)

    def max_positions(self):
        return (self.encoder.max_positions(), self.decoder.max_positions())

    def max_decoder_positions(self):
        return self.decoder.max_positions()

    def get_targets(self, sample, net_output):
        return sample["target"]

    def upgrade_state_dict_named(self, state_dict, name):
        """`conv_layers_before` -> `pre_encoder` (speech_lstm.py:583-597); checkpoints trained with
        `encoder_multilayer_rnn_as_single_module` hold one nn.LSTM (`lstm.weight_ih_l{k}`): map to the per-layer modules."""
        import re

        for k in list(state_dict.keys()):
            if "conv_layers_before" in k:
                state_dict[k.replace("conv_layers_before", "pre_encoder")] = state_dict.pop(k)
        for k in list(state_dict.keys()):
            m = re.match(r"^(.*encoder\.lstm\.)(weight_ih|weight_hh|bias_ih|bias_hh)_l(\d+)(_reverse)?$", k)
            if m:
                state_dict[f"{m.group(1)}{m.group(3)}.{m.group(2)}_l0{m.group(4) or ''}"] = state_dict.pop(k)
        return state_dict
