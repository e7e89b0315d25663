"""Legacy (argparse) Transformer encoder-decoder `speech_transformer` and its presets `speech_transformer_wsj`,
`speech_transformer_librispeech`, `speech_transformer_swbd` — espresso/models/transformer/speech_transformer_legacy.py:23-232
(the models behind examples/asr_{wsj,swbd}/run.sh --arch ...).  Flat `--encoder-embed-dim`-style arguments are folded into the
nested SpeechTransformerConfig (the reference's `SpeechTransformerConfig.from_namespace`); the presets differ from the recipe
YAMLs in using ABSOLUTE sinusoidal encoder positions and no embedding LayerNorm."""
from ...registry import register_model
from .speech_transformer_base import SpeechTransformerModelBase
from .speech_transformer_config import SpeechDecoderConfig, SpeechEncoderConfig, SpeechTransformerConfig

# base_architecture (:103-178)
BASE = dict(encoder_conv_channels="[64, 64, 128, 128]", encoder_conv_kernel_sizes="[(3, 3), (3, 3), (3, 3), (3, 3)]",
            encoder_conv_strides="[(1, 1), (2, 2), (1, 1), (2, 2)]", encoder_embed_dim=256, encoder_ffn_embed_dim=1024,
            encoder_layers=12, encoder_attention_heads=4, encoder_normalize_before=True, encoder_learned_pos=False,
            encoder_relative_positional_embeddings=False, encoder_transformer_context=None, decoder_layers=6,
            decoder_attention_heads=4, decoder_normalize_before=True, decoder_learned_pos=False,
            decoder_relative_positional_embeddings=False, attention_dropout=0.2, activation_dropout=0.2, activation_fn="relu",
            dropout=0.2, share_decoder_input_output_embed=False, no_token_positional_embeddings=False, no_scale_embedding=False,
            layernorm_embedding=False, encoder_layerdrop=0.0, decoder_layerdrop=0.0)
_BIG = dict(encoder_embed_dim=512, encoder_ffn_embed_dim=2048, encoder_layers=12, decoder_layers=6)
ARCHS = {
    "speech_transformer": {},
    "speech_transformer_wsj": {},  # :180-182
    "speech_transformer_librispeech": dict(_BIG, encoder_attention_heads=8, decoder_attention_heads=8, attention_dropout=0.1,
                                           activation_dropout=0.1, dropout=0.1),  # :185-207
    "speech_transformer_swbd": dict(_BIG, encoder_attention_heads=4, decoder_attention_heads=4, attention_dropout=0.25,
                                    activation_dropout=0.25, dropout=0.25),  # :210-232
}


def config_from_flat(args) -> SpeechTransformerConfig:
    """Preset defaults, then the user's flat arguments; `decoder_*` sizes default to the encoder's (:132-136,:159-162)."""
    a = dict(args if isinstance(args, dict) else vars(args))
    arch = a.pop("arch", None) or "speech_transformer"
    a.pop("_name", None)
    flat = dict(BASE)
    flat.update(ARCHS[arch])
    flat.update({k: v for k, v in a.items() if v is not None})
    flat.setdefault("decoder_embed_dim", flat["encoder_embed_dim"])
    flat.setdefault("decoder_ffn_embed_dim", flat["encoder_ffn_embed_dim"])
    flat.setdefault("decoder_output_dim", flat["decoder_embed_dim"])
    flat.setdefault("decoder_input_dim", flat["decoder_embed_dim"])
    enc_f = set(SpeechEncoderConfig.__dataclass_fields__)
    dec_f = set(SpeechDecoderConfig.__dataclass_fields__)
    top_f = set(SpeechTransformerConfig.__dataclass_fields__) - {"encoder", "decoder"}
    enc = {k[len("encoder_"):]: v for k, v in flat.items() if k.startswith("encoder_") and k[len("encoder_"):] in enc_f}
    dec = {k[len("decoder_"):]: v for k, v in flat.items() if k.startswith("decoder_") and k[len("decoder_"):] in dec_f}
    top = {k: v for k, v in flat.items() if k in top_f}
    unknown = sorted(k for k in a if k not in top_f and not (k.startswith("encoder_") and k[8:] in enc_f)
                     and not (k.startswith("decoder_") and k[8:] in dec_f))
    if unknown:
        raise NotImplementedError(f"speech_transformer: arguments without an implementation here: {unknown}")
    return SpeechTransformerConfig(encoder=SpeechEncoderConfig(**enc), decoder=SpeechDecoderConfig(**dec), **top)


@register_model("speech_transformer")
class SpeechTransformerModel(SpeechTransformerModelBase):
    config_class = None  # configured by flat argparse-style arguments (folded into SpeechTransformerConfig by build_model)

    @classmethod
    def build_model(cls, args, task):
        cfg = args if isinstance(args, SpeechTransformerConfig) else config_from_flat(args)
        return super().build_model(cfg, task)
