"""Model configuration with the reference's field names and defaults
(espresso/models/transformer/speech_transformer_config.py:28-365, fairseq/models/transformer/transformer_config.py).
Only fields that affect the ASR hot path are carried; YAML keys of the recipes map 1:1."""
from dataclasses import dataclass, field
from typing import List, Optional

DEFAULT_MAX_SOURCE_POSITIONS = 10240
DEFAULT_MAX_TARGET_POSITIONS = 1024


@dataclass
class SpeechEncoderConfig:
    embed_dim: int = 512
    ffn_embed_dim: int = 2048
    layers: int = 6
    attention_heads: int = 8
    normalize_before: bool = False
    learned_pos: bool = False
    layerdrop: float = 0.0
    relative_positional_embeddings: bool = False
    share_learned_relative_positional_embeddings_across_layers: bool = False
    share_learned_relative_positional_embeddings_across_heads: bool = False
    conv_channels: str = "[64, 64, 128, 128]"
    conv_kernel_sizes: str = "[(3, 3), (3, 3), (3, 3), (3, 3)]"
    conv_strides: str = "[(1, 1), (2, 2), (1, 1), (2, 2)]"
    conv_apply_batchnorm: bool = True
    transformer_context: Optional[str] = None
    layer_type: str = "transformer"
    chunk_size: int = 0
    chunk_left_window: int = 0
    chunk_right_window: int = 0
    depthwise_conv_kernel_size: int = 31


@dataclass
class SpeechDecoderConfig:
    embed_dim: int = 512
    ffn_embed_dim: int = 2048
    layers: int = 6
    attention_heads: int = 8
    normalize_before: bool = False
    learned_pos: bool = False
    layerdrop: float = 0.0
    relative_positional_embeddings: bool = False
    share_learned_relative_positional_embeddings_across_layers: bool = False  # (read only with relative positions in the decoder)
    share_learned_relative_positional_embeddings_across_heads: bool = False
    input_dim: Optional[int] = None
    output_dim: Optional[int] = None
    relaxed_attention_weight: float = 0.0

    def __post_init__(self):
        if self.input_dim is None:
            self.input_dim = self.embed_dim
        if self.output_dim is None:
            self.output_dim = self.embed_dim


@dataclass
class SpeechTransformerConfig:
    activation_fn: str = "relu"
    dropout: float = 0.1
    attention_dropout: float = 0.0
    activation_dropout: float = 0.0
    adaptive_input: bool = False
    encoder: SpeechEncoderConfig = field(default_factory=SpeechEncoderConfig)
    decoder: SpeechDecoderConfig = field(default_factory=SpeechDecoderConfig)
    max_source_positions: int = DEFAULT_MAX_SOURCE_POSITIONS
    max_target_positions: int = DEFAULT_MAX_TARGET_POSITIONS
    share_decoder_input_output_embed: bool = False
    no_token_positional_embeddings: bool = False
    layernorm_embedding: bool = False
    no_scale_embedding: bool = False
    scheduled_sampling_probs: List[float] = field(default_factory=lambda: [1.0])  # P(feed the true token) per epoch
    start_scheduled_sampling_epoch: int = 1

    @classmethod
    def from_dict(cls, d):
        """Build from a nested dict with the recipe YAML's `model:` layout."""
        d = dict(d or {})
        d.pop("_name", None)
        enc = SpeechEncoderConfig(**{k: v for k, v in dict(d.pop("encoder", {}) or {}).items()})
        dec = SpeechDecoderConfig(**{k: v for k, v in dict(d.pop("decoder", {}) or {}).items()})
        known = {f for f in cls.__dataclass_fields__}
        return cls(encoder=enc, decoder=dec, **{k: v for k, v in d.items() if k in known})


@dataclass
class SpeechLSTMPredictorConfig:
    """espresso/models/transformer/speech_transformer_transducer_config.py:30-54 (the `decoder:` block of the transducer)."""
    embed_dim: int = 48
    hidden_size: int = 320
    layers: int = 3
    residual: bool = False
    dropout_in: Optional[float] = None
    dropout_out: Optional[float] = None


@dataclass
class SpeechTransformerTransducerConfig:
    """espresso/models/transformer/speech_transformer_transducer_config.py:56-133."""
    activation_fn: str = "relu"
    dropout: float = 0.1
    attention_dropout: float = 0.0
    activation_dropout: float = 0.0
    encoder: SpeechEncoderConfig = field(default_factory=SpeechEncoderConfig)
    decoder: SpeechLSTMPredictorConfig = field(default_factory=SpeechLSTMPredictorConfig)
    joint_dim: int = 512
    share_decoder_input_output_embed: bool = False
    no_token_positional_embeddings: bool = False
    layernorm_embedding: bool = True
    no_scale_embedding: bool = False
    max_source_positions: int = DEFAULT_MAX_SOURCE_POSITIONS
    max_target_positions: int = DEFAULT_MAX_TARGET_POSITIONS

    @classmethod
    def from_dict(cls, d):
        d = dict(d or {})
        d.pop("_name", None)
        enc = SpeechEncoderConfig(**dict(d.pop("encoder", {}) or {}))
        dec = SpeechLSTMPredictorConfig(**dict(d.pop("decoder", {}) or {}))
        known = {f for f in cls.__dataclass_fields__}
        return cls(encoder=enc, decoder=dec, **{k: v for k, v in d.items() if k in known})
