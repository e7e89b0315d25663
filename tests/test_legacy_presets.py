"""The argparse presets `speech_transformer_{wsj,librispeech,swbd}` (espresso/models/transformer/speech_transformer_legacy.py)
against the reference itself: same nested configuration after `base_architecture`, same parameter names and shapes.
Needs the reference tree (build container only; skipped on the GPU box)."""
import argparse
import os
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")


@pytest.mark.parametrize("arch,overrides", [("speech_transformer_wsj", {}), ("speech_transformer_librispeech", {"encoder_layers": 2, "decoder_layers": 1}),
                                            ("speech_transformer_swbd", {"encoder_layers": 2, "decoder_layers": 1, "dropout": 0.3})])
def test_preset_config_and_parameter_names_equal_the_reference(arch, overrides):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "oracle", "ref_stubs"))
    sys.path.insert(1, REF)
    import torch
    import fairseq  # noqa: F401
    import espresso  # noqa: F401
    from fairseq.models import ARCH_CONFIG_REGISTRY
    from espresso.models.transformer.speech_transformer_config import SpeechTransformerConfig as RefCfg

    import espresso_amd  # noqa: F401
    from espresso_amd.models.transformer.speech_transformer_legacy import SpeechTransformerModel, config_from_flat

    ns = argparse.Namespace(**overrides)
    ARCH_CONFIG_REGISTRY[arch](ns)  # the reference's preset function fills in every default
    ref = RefCfg.from_namespace(ns)
    mine = config_from_flat(dict(overrides, arch=arch))
    for f in ("embed_dim", "ffn_embed_dim", "layers", "attention_heads", "normalize_before", "learned_pos", "relative_positional_embeddings",
              "conv_channels", "conv_kernel_sizes", "conv_strides"):
        assert getattr(mine.encoder, f) == getattr(ref.encoder, f), ("encoder." + f, getattr(mine.encoder, f), getattr(ref.encoder, f))
    for f in ("embed_dim", "ffn_embed_dim", "layers", "attention_heads", "normalize_before", "learned_pos", "input_dim", "output_dim"):
        assert getattr(mine.decoder, f) == getattr(ref.decoder, f), ("decoder." + f, getattr(mine.decoder, f), getattr(ref.decoder, f))
    for f in ("dropout", "attention_dropout", "activation_dropout", "activation_fn", "layernorm_embedding", "no_scale_embedding",
              "no_token_positional_embeddings", "share_decoder_input_output_embed"):
        assert getattr(mine, f) == getattr(ref, f), (f, getattr(mine, f), getattr(ref, f))

    if ref.encoder.layers > 2:
        return  # parameter-name comparison on the shrunken variants only (a 12-layer 512-wide model is slow to build twice on CPU)

    class _Task:
        feat_dim, feat_in_channels = 80, 1

        def __init__(self):
            from espresso_amd.data.asr_dictionary import AsrDictionary

            self.target_dictionary = AsrDictionary.from_symbols([f"u{i}" for i in range(20)], enable_bos=True)
            self.source_dictionary = None

    task = _Task()
    model = SpeechTransformerModel.build_model(dict(overrides, arch=arch), task)
    from espresso.models.transformer.speech_transformer_legacy import SpeechTransformerModel as RefModel

    ns2 = argparse.Namespace(**overrides, max_source_positions=3600, max_target_positions=200, scheduled_sampling_probs=[1.0],
                             start_scheduled_sampling_epoch=1)
    ARCH_CONFIG_REGISTRY[arch](ns2)

    class _RefTask:
        feat_dim, feat_in_channels = 80, 1
        target_dictionary = task.target_dictionary
        source_dictionary = None

    rmodel = RefModel.build_model(ns2, _RefTask())
    rsd = {k: v for k, v in rmodel.state_dict().items()}
    rsd = model.upgrade_state_dict_named(dict(rsd), "")
    msd = model.state_dict()
    assert set(rsd) == set(msd), (sorted(set(rsd) - set(msd))[:8], sorted(set(msd) - set(rsd))[:8])
    for k in msd:
        assert tuple(msd[k].shape) == tuple(rsd[k].shape), k
    model.load_state_dict(rsd, strict=True)
    assert torch.equal(model.state_dict()["decoder.embed_tokens.weight"], rsd["decoder.embed_tokens.weight"])
