"""`lstm_lm_espresso` — espresso/models/lstm_lm.py:88-252: a decoder-only LSTM language model (sub-word or word level) built
from `SpeechLSTMDecoder` without attention.  Architectures `lstm_lm_wsj`, `lstm_lm_librispeech`, `lstm_lm_swbd`,
`lstm_wordlm_wsj` carry the reference's hyper-parameters.

Training: `forward(src_tokens)` -> (fp32 logits [B][U][V], None) through the sequence LSTM op.
Decoding (shallow fusion inside SequenceGenerator, fairseq/sequence_generator.py:385-393): `init_incremental` / `step`
keep one (h, c) per layer and hypothesis on the device, reordered by the surviving beams with ea_gather_rows."""
import torch
import torch.nn as nn

from .. import kernels as K
from ..registry import register_model
from .speech_lstm import LSTMEmbedding, SpeechLSTMDecoder

ARCHS = {
    "lstm_lm_wsj": dict(dropout=0.1, decoder_embed_dim=48, decoder_hidden_size=650, decoder_layers=2, decoder_out_embed_dim=650,
                        share_embed=False, is_wordlm=False),
    "lstm_lm_librispeech": dict(dropout=0.0, decoder_embed_dim=800, decoder_hidden_size=800, decoder_layers=4,
                                decoder_out_embed_dim=800, share_embed=True, is_wordlm=False),
    "lstm_lm_swbd": dict(dropout=0.3, decoder_embed_dim=1800, decoder_hidden_size=1800, decoder_layers=3,
                         decoder_out_embed_dim=1800, share_embed=True, is_wordlm=False),
    "lstm_wordlm_wsj": dict(dropout=0.35, decoder_embed_dim=1200, decoder_hidden_size=1200, decoder_layers=3,
                            decoder_out_embed_dim=1200, share_embed=True, is_wordlm=True),
}


@register_model("lstm_lm_espresso")
class LSTMLanguageModelEspresso(nn.Module):
    def __init__(self, decoder, is_wordlm=False):
        super().__init__()
        self.decoder = decoder
        self.is_wordlm = is_wordlm

    @classmethod
    def build_model(cls, args, task):
        a = dict(ARCHS.get(getattr(args, "arch", "lstm_lm_wsj"), ARCHS["lstm_lm_wsj"]))
        a.update({k: v for k, v in (vars(args) if not isinstance(args, dict) else args).items() if v is not None})
        if a.get("is_wordlm") and hasattr(task, "word_dictionary"):
            dictionary = task.word_dictionary
        else:
            dictionary = getattr(task, "target_dictionary", None) or task.source_dictionary
        if a["share_embed"] and a["decoder_embed_dim"] != a["decoder_out_embed_dim"]:
            raise ValueError("--share-embed requires --decoder-embed-dim to match --decoder-out-embed-dim")
        drop = a.get("dropout", 0.1)
        decoder = SpeechLSTMDecoder(dictionary, embed_dim=a["decoder_embed_dim"], hidden_size=a["decoder_hidden_size"],
                                    out_embed_dim=a["decoder_out_embed_dim"], num_layers=a["decoder_layers"],
                                    dropout_in=a.get("decoder_dropout_in", drop), dropout_out=a.get("decoder_dropout_out", drop),
                                    residual=a.get("decoder_rnn_residual", False), share_input_output_embed=a["share_embed"],
                                    max_target_positions=a.get("max_target_positions") or a.get("tokens_per_sample", 1024),
                                    pretrained_embed=LSTMEmbedding(len(dictionary), a["decoder_embed_dim"], dictionary.pad()))
        return cls(decoder, is_wordlm=bool(a.get("is_wordlm", False)))

    def forward(self, src_tokens, **kwargs):
        return self.decoder(src_tokens)

    def max_positions(self):
        return self.decoder.max_positions()

    def get_normalized_probs(self, net_output, log_probs, sample=None):
        lg = net_output[0]
        V = lg.shape[-1]
        lp = K.log_softmax(lg.reshape(-1, V).contiguous(), lg.numel() // V, V, V).view(lg.shape)
        return lp if log_probs else lp.exp_()

    # ---------------------------------------------------------------- shallow fusion hooks of SequenceGenerator
    def init_incremental(self, bsz, beam):
        dev = self.decoder.embed_tokens.weight.device
        return {"lstm": self.decoder.init_state(bsz * beam, dev)}

    @torch.no_grad()
    def step(self, state, tokens, step, parent=None):
        """tokens [N][step+1]; parent int [N] = surviving beams (None at step 0) -> fp32 log-probs [N][V]."""
        if parent is not None:
            state["lstm"] = self.decoder.reorder_state(state["lstm"], parent)
        feat, state["lstm"] = self.decoder.advance(tokens[:, -1], state["lstm"])
        logits = self.decoder.output_layer(feat)
        N, V = logits.shape
        return K.log_softmax(logits, N, V, logits.stride(0))

    def shrink(self, state, keep_rows):
        """Drop the rows of finished sentences (the generator's batch compaction)."""
        state["lstm"] = self.decoder.reorder_state(state["lstm"], keep_rows)
