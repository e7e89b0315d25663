"""Round 6, CPU only: would an fp32 residual stream bring the encoder's logits within 1e-2 of the
reference's fp32 run?  The bf16-emulating oracle (oracle/torch_ref.py) rounds where the HIP path stores bf16; with
`resid_f32=True` the residual adds and the layers' final LayerNorm outputs stay fp32 (the reference's autocast behaviour from the
first layer's final LayerNorm on).  Prints max |logit error| of both emulations against the reference fixtures (tiny, dh 64) and
against the fp32 restatement for the recipe-size 12-layer encoder — the three sizes tests/test_gpu_parity.py bounds.
    python tools/probes/resid_f32_emulation.py [--layers 12]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import torch_ref  # noqa: E402


def fixture(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=True)
    sd = {k[len("sd::"):]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith("sd::")}
    return g, sd


def run(feats, lengths, sd, H, logits_f32=False, weights_only=False, **kw):
    old = os.environ.get("EA_LOGITS_F32")
    os.environ["EA_LOGITS_F32"] = "1" if logits_f32 else "0"
    try:
        if weights_only:  # the floor: bf16 weight shadows, every activation fp32
            sd = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() and v.dim() >= 2 else v) for k, v in sd.items()}
            kw = dict(on=False)
        with torch.no_grad(), torch_ref.bf16_emulation(**kw):
            lo, _ = torch_ref.encoder(feats, lengths, sd, H=H, layer_type="conformer", training=False)
    finally:
        os.environ.pop("EA_LOGITS_F32") if old is None else os.environ.__setitem__("EA_LOGITS_F32", old)
    return lo


def modes(flash):
    return (("bf16_stream", dict(on=True, flash=flash)), ("fp32_stream", dict(on=True, flash=flash, resid_f32=True)),
            ("fp32_stream_fp32_logits", dict(on=True, flash=flash, resid_f32=True, logits_f32=True)),
            ("bf16_weights_only", dict(weights_only=True)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=12)
    a = ap.parse_args()
    out = {}
    for name, H in (("ref_conformer_ctc_tiny", 4), ("ref_conformer_ctc_dh64", 2)):  # (tests/gpu_checks.py:_fixture_shape)
        g, sd = fixture(name)
        d = sd["fc0.weight"].shape[0]
        ref = torch.from_numpy(g["out::eval_logits"])
        feats, lengths = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"])
        flash = d // H == 64
        r = {}
        for tag, kw in modes(flash):
            lo = run(feats, lengths, sd, H, **kw)
            r[tag] = float((lo - ref).abs().max())
            r[tag + "_logprob"] = float((torch.log_softmax(lo, -1) - torch.log_softmax(ref, -1)).abs().max())
        r["fp32_restatement"] = float((run(feats, lengths, sd, H, on=False) - ref).abs().max())
        r["logit_scale"] = float(ref.abs().max())
        out[name] = r
    # recipe-size encoder, random weights as tests/gpu_checks.py:check_fullsize_layer_vs_oracle builds them
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerConfig
    from espresso_amd.models.transformer.speech_transformer_encoder_model import SpeechTransformerEncoderModel
    from tests.gpu_checks import _Task

    torch.manual_seed(0)
    cfg = SpeechTransformerConfig()
    e = cfg.encoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = 512, 2048, a.layers, 8
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, "conformer"
    e.conv_channels = "[64, 64, 128, 128]"
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = 0.0
    cfg.layernorm_embedding = True
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    model = SpeechTransformerEncoderModel.build_model(cfg, _Task(5004))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd = {k[len("encoder."):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("encoder.")}
    g = torch.Generator().manual_seed(1)
    lens = [400, 333, 250, 120]
    feats = torch.zeros(len(lens), max(lens), 80)
    for b, n in enumerate(lens):
        feats[b, :n] = torch.randn(n, 80, generator=g)
    lengths = torch.tensor(lens)
    ref = run(feats, lengths, sd, 8, on=False)
    r = {"logit_scale": float(ref.abs().max())}
    for tag, kw in modes(True):
        lo = run(feats, lengths, sd, 8, **kw)
        r[tag] = float((lo - ref).abs().max())
        r[tag + "_logprob"] = float((torch.log_softmax(lo, -1) - torch.log_softmax(ref, -1)).abs().max())
    out[f"fullsize_{a.layers}_layers"] = r
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
