"""LSTM decoder / predictor — espresso/models/speech_lstm.py:600-1048 (`SpeechLSTMDecoder`).

This round implements the attention-free mode (attn_type None): the predictor of the transducer
(espresso/models/transformer/speech_transformer_transducer_base.py:195-212) and the decoder-only LSTM language model
(espresso/models/lstm_lm.py:88-198).  Parameter names follow the reference's state_dict (`embed_tokens.weight`,
`layers.{i}.weight_ih|weight_hh|bias_ih|bias_hh`, optional `additional_fc`, `fc_out`).  Without attention / input feeding
the per-step loop of :846-893 factorises into stacked sequence LSTMs, so each layer runs over the whole teacher-forced
sequence (one input-projection GEMM + per-step recurrent GEMM + cell kernel; functional._LSTMLayer); incremental decoding
keeps (h, c) per layer and advances one step at a time (`step`), reordered by the surviving beams with a gather kernel."""
import torch
import torch.nn as nn

from .. import functional as F
from .. import kernels as K
from ..modules.params import LinearParams


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
        if attn_type is not None and str(attn_type).lower() != "none":
            raise NotImplementedError("attention LSTM decoder (speech_lstm enc-dec, config 1) is scheduled after the transducer path")
        self.dictionary = dictionary
        self.dropout_in, self.dropout_out = float(dropout_in), float(dropout_out)
        self.hidden_size, self.num_layers, self.residual = hidden_size, num_layers, residual
        self.share_input_output_embed = share_input_output_embed
        self.max_target_positions = max_target_positions
        self.encoder_output_units = 0
        self.attention = None
        pad = dictionary.pad()
        self.embed_tokens = pretrained_embed if pretrained_embed is not None else LSTMEmbedding(len(dictionary), embed_dim, pad)
        embed_dim = self.embed_tokens.embedding_dim
        self.layers = nn.ModuleList([LSTMCellParams(embed_dim if i == 0 else hidden_size, hidden_size) for i in range(num_layers)])
        if hidden_size != out_embed_dim:
            self.additional_fc = lstm_linear(hidden_size, out_embed_dim)
        if not share_input_output_embed:
            self.fc_out = lstm_linear(out_embed_dim, len(dictionary))

    def max_positions(self):
        return self.max_target_positions

    # ---------------------------------------------------------------- teacher-forced path
    def extract_features(self, prev_output_tokens, **unused):
        """prev_output_tokens [B][U] -> (features bf16 [B][U][H_out], None).  speech_lstm.py:766-919 with encoder_out None."""
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

    def forward(self, prev_output_tokens, encoder_out=None, incremental_state=None, **kwargs):
        x, attn = self.extract_features(prev_output_tokens)
        return self.output_layer(x), attn

    # ---------------------------------------------------------------- incremental path (inference)
    def init_state(self, n, device):
        """Zero (h, c) for n hypotheses — speech_lstm.py:932-952 initialize_cached_state."""
        z32 = lambda: torch.zeros(n, self.hidden_size, dtype=torch.float32, device=device)
        z16 = lambda: torch.zeros(n, self.hidden_size, dtype=torch.bfloat16, device=device)
        return {"h16": [z16() for _ in self.layers], "h32": [z32() for _ in self.layers], "c": [z32() for _ in self.layers]}

    @torch.no_grad()
    def step(self, tokens, state, keep_row=None):
        """tokens int [N] (last emitted token of each hypothesis) -> (features bf16 [N][H_out], new state).
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

    @staticmethod
    def reorder_state(state, new_order):
        """index_select of every cached tensor by the surviving beams (speech_lstm.py:981-999)."""
        idx = new_order.to(torch.int32).contiguous()
        return {k: [K.gather_rows(t.contiguous(), idx) for t in v] for k, v in state.items()}
