"""CPU tests that PIN the oracle: the restatements in oracle/ must reproduce what the reference's own
modules produced (golden fixtures written by oracle/gen_golden.py from /root/reference)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import fbank_ref, torch_ref


def _load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    return g, sd


@pytest.mark.parametrize("layer_type,fixture,H", [("conformer", "ref_conformer_ctc_tiny", 4), ("transformer", "ref_transformer_ctc_tiny", 4),
                                                  ("transformer", "ref_transformer_learnedpos_ctc_tiny", 4),
                                                  ("conformer", "ref_conformer_ctc_dh64", 2), ("transformer", "ref_transformer_ctc_dh64", 2)])
def test_encoder_restatement_matches_reference(golden_dir, layer_type, fixture, H):
    """`*_dh64`: embed 128 / 2 heads = head dim 64 (the recipes' 512 / 8 shape class, fused attention kernels), 75 encoder frames."""
    g, sd = _load(golden_dir, fixture)
    feats, lengths = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"])
    lo, ol = torch_ref.encoder(feats, lengths, sd, H=H, layer_type=layer_type, training=False)
    assert ol.tolist() == g["out::out_lengths"].tolist()
    assert float((lo - torch.from_numpy(g["out::eval_logits"])).abs().max()) < 1e-5
    upd = {}
    lo, ol = torch_ref.encoder(feats, lengths, sd, H=H, layer_type=layer_type, training=True, update=upd)
    assert float((lo - torch.from_numpy(g["out::train_logits"])).abs().max()) < 1e-5
    tgt = torch.from_numpy(g["targets"])
    tl = (tgt != 1).sum(-1)
    loss = torch_ref.ctc_loss_sum(lo, tgt, ol, tl)
    assert float(loss) == pytest.approx(float(g["out::train_loss"]), rel=1e-6)
    for k, v in upd.items():
        assert float((v - torch.from_numpy(g["bn_after::" + k])).abs().max()) < 1e-5
    # independent float64 alpha recursion agrees with the ATen kernel the reference calls
    lp = torch.log_softmax(lo.float(), -1).numpy()
    tot = sum(torch_ref.ctc_nll_numpy(lp[: int(ol[b]), b], tgt[b, : int(tl[b])].tolist()) for b in range(tgt.shape[0]))
    assert tot == pytest.approx(float(g["out::train_loss"]), rel=1e-5)


@pytest.mark.parametrize("layer_type,fixture,H", [("transformer", "ref_transformer_ctc_legacy", 2), ("transformer", "ref_transformer_ctc_postln_chunk", 4),
                                                  ("conformer", "ref_conformer_ctc_abspos", 4)])
def test_encoder_restatement_matches_reference_legacy_configurations(golden_dir, layer_type, fixture, H):
    """The encoder options of the argparse presets `speech_transformer_{wsj,swbd,librispeech}` (absolute sinusoidal positions, no
    embedding LayerNorm) and the non-default ones the reference accepts (learned absolute positions, post-LN, chunk-streaming
    masks incl. the training-time coin flip under numpy_seed(num_updates)): fixtures from the reference's own encoder."""
    import json

    g, sd = _load(golden_dir, fixture)
    meta = json.loads(str(g["meta"]))
    feats, lengths = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"])
    lo, ol = torch_ref.encoder(feats, lengths, sd, H=H, layer_type=layer_type, training=False,
                               **torch_ref.legacy_encoder_kwargs(meta, lengths, False))
    assert ol.tolist() == g["out::out_lengths"].tolist()
    assert float((lo - torch.from_numpy(g["out::eval_logits"])).abs().max()) < 2e-5
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k and k != "version" and not k.endswith("_float_tensor"):
            v.requires_grad_(True)
    lo, ol = torch_ref.encoder(feats, lengths, sd, H=H, layer_type=layer_type, training=True,
                               **torch_ref.legacy_encoder_kwargs(meta, lengths, True))
    assert float((lo - torch.from_numpy(g["out::train_logits"])).abs().max()) < 2e-5
    tgt = torch.from_numpy(g["targets"])
    loss = torch_ref.ctc_loss_sum(lo, tgt, ol, (tgt != 1).sum(-1))
    assert float(loss) == pytest.approx(float(g["out::train_loss"]), rel=1e-6)
    loss.backward()
    for name in [k for k in ("fc_out.weight", "embed_positions.weight", "layers.0.fc1.weight", "layers.1.self_attn.q_proj.weight", "fc0.weight") if k in sd]:
        ref = torch.from_numpy(g["grad::" + name])
        assert float((sd[name].grad - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6, name


def test_encoder_restatement_gradients(golden_dir):
    g, sd = _load(golden_dir, "ref_conformer_ctc_tiny")
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k and k != "version":
            v.requires_grad_(True)
    feats, lengths = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"])
    lo, ol = torch_ref.encoder(feats, lengths, sd, H=4, layer_type="conformer", training=True)
    tgt = torch.from_numpy(g["targets"])
    torch_ref.ctc_loss_sum(lo, tgt, ol, (tgt != 1).sum(-1)).backward()
    for name in ("fc_out.weight", "layers.0.self_attn.pos_bias_u", "layers.1.conv_module.depthwise_conv.weight",
                 "pre_encoder.convolutions.0.weight", "fc0.weight"):
        ref = torch.from_numpy(g["grad::" + name])
        assert float((sd[name].grad - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-6, name


def test_label_smoothing_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "label_smoothing.npz"))
    logits, target = torch.from_numpy(g["logits"]), torch.from_numpy(g["target"])
    for eps in (0.0, 0.1):
        loss, nll = torch_ref.label_smoothed_nll(logits, target, eps, pad_idx=1)
        assert float(loss) == pytest.approx(float(g[f"loss_{eps}"]), rel=1e-6)
        assert float(nll) == pytest.approx(float(g[f"nll_{eps}"]), rel=1e-6)


@pytest.mark.parametrize("kind", ["unigram", "temporal"])
def test_label_smoothing_variants_match_reference(golden_dir, kind):
    g = np.load(os.path.join(golden_dir, "label_smoothing.npz"))
    logits = torch.from_numpy(g["logits"]).requires_grad_(True)
    target = torch.from_numpy(g["target2"])
    loss, nll = torch_ref.label_smoothed_nll(logits, target, 0.1, 1, smoothing=kind, prior=torch.from_numpy(g["prior"]),
                                             tgt_len=target.numel())
    assert float(loss) == pytest.approx(float(g[f"{kind}_loss"]), rel=1e-6)
    assert float(nll) == pytest.approx(float(g[f"{kind}_nll"]), rel=1e-6)
    loss.backward()
    assert float((logits.grad - torch.from_numpy(g[f"{kind}_dlogits"])).abs().max()) < 1e-6


def test_fbank_frame_count_matches_reference_formula():
    """espresso/tools/utils.py:457-486 pins only the frame count of the torchaudio fbank."""
    from espresso_amd.tools.utils import num_samples_to_num_frames

    rng = np.random.default_rng(0)
    for n in (0, 399, 400, 401, 559, 560, 16000, 16000 * 3 + 77):
        w = (rng.standard_normal(n) * 1000).astype(np.float32)
        assert fbank_ref.fbank(w).shape == (num_samples_to_num_frames([n], 16000)[0], 80)


def test_fbank_restatement_matches_an_independent_kaldi_compatible_frontend():
    """torchaudio is absent, but `transformers.audio_utils` ships a numpy Kaldi-compatible front-end (povey window, per-frame DC
    removal, pre-emphasis, power spectrum, Kaldi mel scale with triangles in mel space, natural log floored at float32 eps) that
    its own feature extractors use in place of `torchaudio.compliance.kaldi.fbank` and test against it.  Same waveform through
    both restatements: an independent third-party pin of the oracle at the fbank boundary."""
    au = pytest.importorskip("transformers.audio_utils")
    from espresso_amd.data import synthetic

    win = au.window_function(400, "povey", periodic=False)
    mel = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000, sampling_rate=16000,
                             norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    rng = np.random.default_rng(0)
    for n in (400, 5000, 16000 * 3 + 123):
        w = synthetic.waveform(n, rng).astype(np.float32)  # int16-range band-limited noise, like the bench's audio
        theirs = au.spectrogram(w.astype(np.float64), win, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=False,
                                preemphasis=0.97, mel_filters=mel, log_mel="log", mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
        ours = fbank_ref.fbank(w)
        assert ours.shape == theirs.shape
        assert np.abs(ours - theirs).max() < 5e-4, (n, np.abs(ours - theirs).max())  # log-mel values reach ~22


def test_fbank_closed_form_properties():
    """Analytic checks of the restatement: a DC signal has zero energy after DC removal (log floor),
    and a pure tone peaks in the mel bin that contains it."""
    x = np.full(16000, 1234.0, dtype=np.float32)
    f = fbank_ref.fbank(x)
    assert np.allclose(f, np.log(np.finfo(np.float32).eps), atol=1e-3)
    t = np.arange(16000) / 16000.0
    tone = (8000 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
    f = fbank_ref.fbank(tone)
    bank = fbank_ref.mel_banks()
    expect = int(np.argmax(bank[:, 32]))  # 1000 Hz = FFT bin 32
    assert abs(int(np.argmax(f.mean(0))) - expect) <= 1


def test_product_mel_tables_equal_oracle_bank():
    from espresso_amd.data.fbank_tables import build_mel_bank

    np.testing.assert_array_equal(build_mel_bank(), fbank_ref.mel_banks())


def test_decoder_restatement_matches_reference(golden_dir):
    """speech_transformer_base: the restated decoder reproduces the reference's logits at every non-pad position and its
    label-smoothed CE values exactly (pad rows differ: the reference additionally masks padded keys, which no loss sees)."""
    g = np.load(os.path.join(golden_dir, "ref_transformer_encdec_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    feats, lengths, prev, target = (torch.from_numpy(g[k]) for k in ("feats", "lengths", "prev", "target"))
    valid = target.ne(0)
    for training, key in ((False, "out::eval_logits"), (True, "out::train_logits")):
        lo = torch_ref.encdec(feats, lengths, prev, sd, 4, pad_idx=0, training=training)
        assert float((lo - torch.from_numpy(g[key]))[valid].abs().max()) < 1e-5
    loss, nll = torch_ref.label_smoothed_nll(lo.reshape(-1, lo.shape[-1]), target.reshape(-1), 0.1, 0)
    assert float(loss) == pytest.approx(float(g["out::loss"]), rel=1e-6)
    assert float(nll) == pytest.approx(float(g["out::nll"]), rel=1e-6)


def test_dropout_hash_restatement_matches_the_library():
    """oracle/dropout_ref.py `ea_hash` == csrc/common.h `ea_hash` (the library's host evaluation, no GPU): seeds with high words,
    element indices beyond 2^32, the threshold rule, the seed table of the native layer runtime"""
    import ctypes

    from espresso_amd import _lib
    from oracle import dropout_ref as D

    lib = _lib.lib()
    for seed in (0, 1, 0x5EED00000001, ((0x5EED << 32) | 7) * 64 % (1 << 63), 2 ** 64 - 1):
        for idx0 in (0, 12345, 2 ** 32 - 5, 2 ** 40 + 3):
            out = (ctypes.c_uint32 * 4096)()
            assert lib.ea_dropout_hash_host(seed, idx0, 4096, out) == 0
            assert (np.frombuffer(out, dtype=np.uint32) == D.ea_hash(seed, np.arange(idx0, idx0 + 4096, dtype=np.uint64))).all()
    assert D.drop_threshold(0.0) == 0 and D.drop_threshold(0.1) == int(0.1 * 2 ** 32) and D.drop_threshold(1.0) == 2 ** 32 - 1
    assert abs(float(D.keep_mask(99, 1 << 20, 0.1).mean()) - 0.9) < 1e-3
    assert len({int(lib.ea_layer_dropout_seed(640, s)) for s in range(7)}) == 7  # the 7 conformer sites draw 7 different streams
    assert all(640 < int(lib.ea_layer_dropout_seed(640, s)) < 704 for s in range(9))


@pytest.mark.parametrize("name,layer_type", [("ref_dropout_conformer_ctc_tiny", "conformer"), ("ref_dropout_transformer_ctc_tiny", "transformer")])
def test_encoder_restatement_with_dropout_matches_reference(golden_dir, name, layer_type):
    """TRAINING mode with dropout 0.1: the reference's own encoder ran with every dropout call fed a given mask
    (oracle/gen_golden.py dropout_fixtures); the restatement applies the same masks at its dropout sites and must reproduce
    logits, loss and every gradient — pins WHERE the oracle drops (fc0 input, after the embedding LayerNorm, FFN activation /
    output, attention probabilities / output, convolution-module output) and in which tensor layout, to the reference's code."""
    import json

    from espresso_amd import _lib
    from oracle import dropout_ref as D

    gd = np.load(os.path.join(golden_dir, name + ".npz"))
    g, sd = _load(golden_dir, str(gd["source"]))
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k and k != "version":
            v.requires_grad_(True)
    plan = D.MaskPlan(json.loads(str(gd["trace"])), _lib.lib().ea_layer_dropout_seed)
    with torch_ref.dropout_masks(plan):
        lo, ol = torch_ref.encoder(torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), sd, H=4, layer_type=layer_type, training=True)
    plan.done()
    assert len(plan.queue) == (16 if layer_type == "conformer" else 10) and not plan.skipped
    assert float((lo - torch.from_numpy(gd["out::train_logits"])).abs().max()) < 2e-5
    assert float((lo - torch.from_numpy(g["out::train_logits"])).abs().max()) > 0.5  # (the masks do something)
    tgt = torch.from_numpy(g["targets"])
    loss = torch_ref.ctc_loss_sum(lo, tgt, ol, (tgt != 1).sum(-1))
    assert float(loss) == pytest.approx(float(gd["out::train_loss"]), rel=1e-6)
    loss.backward()
    for k in gd.files:
        if k.startswith("grad::"):
            ref = torch.from_numpy(gd[k])
            assert float((sd[k[6:]].grad - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6, k


def test_decoder_restatement_with_dropout_matches_reference(golden_dir):
    """the same for speech_transformer_base: encoder sites + decoder embedding dropout + self-attention / encoder-attention
    probabilities and outputs + FFN of both decoder layers (fairseq transformer_decoder.py:324-327, transformer_layer.py:384-529)"""
    import json

    from espresso_amd import _lib
    from oracle import dropout_ref as D

    gd = np.load(os.path.join(golden_dir, "ref_dropout_transformer_encdec_tiny.npz"))
    g, sd = _load(golden_dir, str(gd["source"]))
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k and not k.endswith("version") and "_float_tensor" not in k:
            v.requires_grad_(True)
    feats, lengths, prev, target = (torch.from_numpy(g[k]) for k in ("feats", "lengths", "prev", "target"))
    plan = D.MaskPlan(json.loads(str(gd["trace"])), _lib.lib().ea_layer_dropout_seed)
    with torch_ref.dropout_masks(plan):
        lo = torch_ref.encdec(feats, lengths, prev, sd, 4, pad_idx=0, training=True)
    plan.done()
    assert len(plan.queue) == 2 + 2 * 4 + 1 + 2 * 6 and not plan.skipped
    valid = target.ne(0)
    assert float((lo - torch.from_numpy(gd["out::train_logits"]))[valid].abs().max()) < 2e-5
    loss, nll = torch_ref.label_smoothed_nll(lo.reshape(-1, lo.shape[-1]), target.reshape(-1), 0.1, 0)
    assert float(loss) == pytest.approx(float(gd["out::loss"]), rel=1e-6)
    loss.backward()
    worst = 0.0
    for k in gd.files:
        if k.startswith("grad::") and sd[k[6:]].grad is not None:
            ref = torch.from_numpy(gd[k])
            worst = max(worst, float((sd[k[6:]].grad - ref).abs().max() / (float(ref.abs().max()) + 1e-9)))
    assert worst < 5e-4, worst


def test_transducer_restatement_with_dropout_matches_reference(golden_dir):
    """speech_transformer_transducer_base in training mode with dropout 0.1: encoder sites + the LSTM predictor's dropout_in on
    the embeddings and dropout_out after every layer at every step (espresso/models/speech_lstm.py:811,866), masks fed to the
    reference's own modules (oracle/gen_golden.py dropout_fixtures) — logits and every gradient of sum(logits * R)."""
    import json

    from espresso_amd import _lib
    from oracle import dropout_ref as D

    gd = np.load(os.path.join(golden_dir, "ref_dropout_conformer_transducer_tiny.npz"))
    g, sd = _load(golden_dir, str(gd["source"]))
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k and not k.endswith("version"):
            v.requires_grad_(True)
    feats, lengths, prev = (torch.from_numpy(g[k]) for k in ("feats", "lengths", "prev"))
    plan = D.MaskPlan(json.loads(str(gd["trace"])), _lib.lib().ea_layer_dropout_seed)
    with torch_ref.dropout_masks(plan):
        lo, ol = torch_ref.transducer(feats, lengths, prev, sd, H=4, residual=True, training=True)
    plan.done()
    assert len(plan.queue) == 2 + 2 * 7 + 3 and not plan.skipped
    valid = torch.zeros(lo.shape[:3], dtype=torch.bool)
    for b in range(lo.shape[0]):
        valid[b, : int(ol[b])] = True
    ref = torch.from_numpy(gd["out::train_logits"])
    assert float((lo.detach() - ref)[valid].abs().max()) < 5e-5
    (lo * torch.from_numpy(g["R"])).sum().backward()
    worst = 0.0
    for k in gd.files:
        # (conv biases in front of BatchNorm and the key bias have an exactly zero true gradient: both sides hold round-off)
        if k.startswith("grad::") and sd[k[6:]].grad is not None and not (k.endswith("k_proj.bias") or (".convolutions." in k and k.endswith(".bias"))):
            r = torch.from_numpy(gd[k])
            worst = max(worst, float((sd[k[6:]].grad - r).abs().max() / (float(r.abs().max()) + 1e-9)))
    assert worst < 1e-4, worst


def test_rnnt_torch_loss_equals_the_numpy_restatement():
    """oracle/rnnt_ref.py rnnt_loss_torch (differentiable, used by the model-level gradient checks) == rnnt_loss_one: value and
    gradient w.r.t. the logits"""
    from oracle import rnnt_ref

    rng = np.random.default_rng(3)
    for T, U, V in ((9, 4, 11), (5, 1, 7), (3, 0, 5)):
        lg = rng.standard_normal((T, U + 1, V))
        tgt = [int(t) for t in rng.integers(1, V, U)]
        want, gw = rnnt_ref.rnnt_loss_one(lg, tgt, blank=0, want_grad=True)
        x = torch.tensor(lg, requires_grad=True)
        got = rnnt_ref.rnnt_loss_torch(x, tgt, 0)
        got.backward()
        assert float(got.detach()) == pytest.approx(want, rel=1e-12)
        assert float(np.abs(gw - x.grad.numpy()).max()) < 1e-12


def test_rnnt_restatement_vs_bruteforce_and_finite_differences():
    """Pins oracle/rnnt_ref.py: the alpha recursion equals an explicit sum over ALL alignments on tiny lattices, and the
    analytic gradient equals central finite differences."""
    from oracle import rnnt_ref

    rng = np.random.default_rng(0)
    for T, U, V in ((1, 0, 3), (2, 1, 4), (3, 2, 5), (4, 3, 4), (5, 2, 6)):
        z = rng.standard_normal((T, U + 1, V)) * 2
        y = rng.integers(1, V, size=U).tolist()
        nll = rnnt_ref.rnnt_loss_one(z, y)
        assert nll == pytest.approx(rnnt_ref.rnnt_loss_bruteforce(z, y), rel=1e-10)
    z = rng.standard_normal((4, 3, 5))
    y = [2, 4]
    nll, g = rnnt_ref.rnnt_loss_one(z, y, want_grad=True)
    eps = 1e-6
    for idx in [(0, 0, 0), (1, 1, 2), (3, 2, 0), (2, 0, 4), (3, 1, 4)]:
        zp, zm = z.copy(), z.copy()
        zp[idx] += eps
        zm[idx] -= eps
        fd = (rnnt_ref.rnnt_loss_one(zp, y) - rnnt_ref.rnnt_loss_one(zm, y)) / (2 * eps)
        assert g[idx] == pytest.approx(fd, abs=1e-6)


def test_transducer_restatement_matches_reference(golden_dir):
    """encoder + LSTM predictor + joint (weight-normed fc_out) restated in oracle/torch_ref.py vs the reference model's
    own logits (eval and train mode) and parameter gradients of sum(logits * R)."""
    g, sd = _load(golden_dir, "ref_conformer_transducer_tiny")
    feats, lengths, prev = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"])
    lo, ol = torch_ref.transducer(feats, lengths, prev, sd, H=4, pad_idx=1, residual=True, training=False)
    assert ol.tolist() == g["out::out_lengths"].tolist()
    assert float((lo - torch.from_numpy(g["out::eval_logits"])).abs().max()) < 2e-5
    leaf = {k: v.clone().requires_grad_(True) if (v.is_floating_point() and "running_" not in k) else v.clone() for k, v in sd.items()}
    lo, _ = torch_ref.transducer(feats, lengths, prev, leaf, H=4, pad_idx=1, residual=True, training=True, update={})
    assert float((lo - torch.from_numpy(g["out::train_logits"])).abs().max()) < 2e-5
    (lo * torch.from_numpy(g["R"])).sum().backward()
    for k in ("decoder.layers.0.weight_hh", "decoder.layers.1.weight_ih", "decoder.embed_tokens.weight", "fc_out.weight_g",
              "fc_out.weight_v", "proj_decoder.weight", "laynorm_proj_encoder.weight", "encoder.fc0.weight"):
        ref = torch.from_numpy(g["grad::" + k])
        assert float((leaf[k].grad - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), k


def test_speech_lstm_restatement_matches_reference(golden_dir):
    """conv front-end + packed BiLSTM encoder + Bahdanau-attention LSTM decoder (input feeding, residuals, additional_fc)
    restated in oracle/torch_ref.py vs the reference SpeechLSTMModel's own logits, label-smoothed loss and gradients."""
    g, sd = _load(golden_dir, "ref_speech_lstm_tiny")
    feats, lengths, prev = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"])
    lo, _, _ = torch_ref.speech_lstm(feats, lengths, prev, sd, enc_residual=True, dec_residual=True, pad_idx=0, training=False)
    assert float((lo - torch.from_numpy(g["out::eval_logits"])).abs().max()) < 5e-5
    leaf = {k: v.clone().requires_grad_(True) if (v.is_floating_point() and "running_" not in k) else v.clone() for k, v in sd.items()}
    lo, _, _ = torch_ref.speech_lstm(feats, lengths, prev, leaf, enc_residual=True, dec_residual=True, pad_idx=0, training=True, update={})
    assert float((lo - torch.from_numpy(g["out::train_logits"])).abs().max()) < 5e-5
    target = torch.from_numpy(g["target"])
    loss, nll = torch_ref.label_smoothed_nll(lo.reshape(-1, lo.shape[-1]), target.reshape(-1), 0.1, 0)  # pad = 0 (no <s> in this dictionary)
    assert float(loss) == pytest.approx(float(g["out::loss"]), rel=1e-5)
    loss.backward()
    for k in ("encoder.lstm.0.weight_hh_l0_reverse", "encoder.lstm.1.weight_ih_l0", "decoder.attention.v", "decoder.attention.g",
              "decoder.attention.value_proj.weight", "decoder.layers.1.weight_ih", "decoder.embed_tokens.weight"):
        ref = torch.from_numpy(g["grad::" + k])
        assert float((leaf[k].grad - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max())), k


def test_lstm_lm_restatement_matches_reference(golden_dir):
    """oracle/torch_ref.py lstm_lm (embedding -> LSTMCell stack over the padded token matrix -> tied / untied output projection)
    vs the reference's lstm_lm_espresso training step: logits, summed NLL and every parameter gradient (fixture from
    oracle/gen_golden.py lmtrain)."""
    g = np.load(os.path.join(golden_dir, "ref_lstm_lm_train_tiny.npz"))
    pad = int(g["pad"])
    for tag in ("tied", "untied"):
        sd = {k[len(tag) + 6:]: torch.from_numpy(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith(tag + "::sd::")}
        src, target = torch.from_numpy(g[tag + "::src"]), torch.from_numpy(g[tag + "::target"])
        lo = torch_ref.lstm_lm(src, sd, pad_idx=pad)
        valid = target.ne(pad)
        assert float((lo - torch.from_numpy(g[tag + "::logits"]))[valid].abs().max()) < 1e-5, tag
        lp = torch.log_softmax(lo.float(), -1)
        loss = -(lp.gather(-1, target.unsqueeze(-1)).squeeze(-1) * valid).sum()
        assert float(loss.detach()) == pytest.approx(float(g[tag + "::loss"]), rel=1e-5), tag
        loss.backward()
        for k, v in sd.items():
            ref = torch.from_numpy(g[f"{tag}::grad::{k}"])
            assert float((v.grad - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (tag, k)


def test_incremental_decode_restatement_matches_full_forward(golden_dir):
    """oracle/decode_ref.py (K/V-cached one-token-per-step decoder used by bench.py's CPU decode baseline) reproduces the
    teacher-forced log-probs of the full-forward decoder restatement, which is itself pinned to the reference's outputs."""
    from oracle import decode_ref

    g, sd = _load(golden_dir, "ref_transformer_encdec_tiny")
    sd = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    feats, lengths, prev = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"])
    full = torch_ref.encdec(feats, lengths, prev, sd, H=4, pad_idx=0)
    valid = torch.from_numpy(g["target"]).ne(0)
    assert float((full - torch.from_numpy(g["out::eval_logits"]))[valid].abs().max()) < 1e-5  # the pin (non-pad positions)
    want = torch.log_softmax(full.float(), -1)
    enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    x, out_len = torch_ref.encoder(feats, lengths, enc_sd, 4, layer_type="transformer", training=False)
    pad = torch.arange(x.shape[0]).unsqueeze(0) >= out_len.unsqueeze(1)
    dec = decode_ref.IncrementalDecoder(sd, 4, pad_idx=0)
    dec.init(x, pad, beam=1)
    for step in range(int((prev != 0).sum(1).min())):
        lp = dec.step(prev[:, step], step)
        assert float((lp - want[:, step]).abs().max()) < 1e-4, step
    toks, scores = decode_ref.beam_search(feats, lengths, sd, 4, pad=0, eos=1, unk=2, beam=3, min_steps=5)
    assert toks.shape == (3, 5) and bool((toks[:, -1] == 1).all()) and bool(torch.isfinite(scores).all())


def test_trajectory_task_is_learnable_by_the_oracle():
    """tests/trajectory.py: the synthetic task the +n2 GPU test trains on is learned by the fp32 oracle (held-out greedy token
    error rate under 5 % after 80 updates, from a loss of ~ln(V) per token), and the fp32 and bf16-emulating oracle runs — same
    arithmetic up to rounding — agree to 1.5 % per update over the first 24 updates (the plateau) and to 2 tokens per 100 at the end: the bounds
    the GPU test holds the HIP path to are the ones two correct implementations meet"""
    from tests import trajectory as TR
    from tests.gpu_checks import load_fixture

    _, sd, _, _ = load_fixture(TR.FIXTURE)
    train, held = TR.make_batches(TR.TRAIN_BATCHES, seed=0), TR.make_batches(TR.HELDOUT_BATCHES, seed=1)
    l32, e32, tot = TR.train_oracle(sd, train, held, TR.STEPS, False)
    lem, eem, _ = TR.train_oracle(sd, train, held, TR.STEPS, True)
    assert l32[0] > 30 and sum(l32[-10:]) / 10 < 1.5, l32[::8]
    assert e32 / tot < 0.05, (e32, tot)
    assert max(abs(a - b) / b for a, b in zip(l32[:24], lem[:24])) < 1.5e-2
    assert abs(e32 - eem) / tot <= 0.02, (e32, eem, tot)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("layer_type", ["conformer", "transformer"])
def test_encoder_restatement_matches_reference_at_8_heads(layer_type):
    """The recipes' attention geometry — embed 512, EIGHT heads of 64 — against the reference's own encoder, live (the state
    dict would make a 20 MB fixture, so nothing is stored: the reference is imported through oracle/ref_stubs exactly as
    oracle/gen_golden.py does and run on seeded inputs here).  Closes the chain for H = 8: reference == oracle (this test, 1e-5)
    and oracle == HIP path (the full-size -m gpu tests, same oracle functions).  The committed dh-64 fixtures have 2 heads."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import gen_golden as GG

    torch.manual_seed(7)
    V, H = 40, 8
    cfg = GG.ref_config(layer_type, d=512, heads=H, ffn=256, layers=2)
    enc = GG.build_ref_encoder(cfg, V)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    feats = torch.randn(2, 150, 80)
    lengths = torch.tensor([150, 97])
    feats[1, 97:] = 0
    tgt = torch.tensor([[5, 9, 11, 4, 30], [7, 8, 1, 1, 1]])
    tl = (tgt != 1).sum(-1)
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    enc.eval()
    with torch.no_grad():
        ref_eval = enc(feats, lengths)["encoder_out"][0]
    enc.train()
    o = enc(feats, lengths)
    lp = torch.log_softmax(o["encoder_out"][0].float(), -1)
    flat = torch.cat([tgt[b, : int(tl[b])] for b in range(2)])
    ref_loss = torch.nn.functional.ctc_loss(lp, flat, o["src_lengths"][0], tl, blank=0, reduction="sum", zero_infinity=True)
    ref_loss.backward()
    ref_grads = {n: p.grad.clone() for n, p in enc.named_parameters()}

    lo, ol = torch_ref.encoder(feats, lengths, sd, H=H, layer_type=layer_type, training=False)
    assert float((lo - ref_eval).abs().max()) < 1e-5
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "version" else v.clone()) for k, v in sd.items()}
    lt, ol = torch_ref.encoder(feats, lengths, sdo, H=H, layer_type=layer_type, training=True)
    loss = torch_ref.ctc_loss_sum(lt, tgt, ol, tl)
    assert float(loss) == pytest.approx(float(ref_loss), rel=1e-6)
    loss.backward()
    worst = max(float((sdo[n].grad - g).abs().max() / (g.abs().max() + 1e-12)) for n, g in ref_grads.items() if sdo[n].grad is not None)
    assert worst < 2e-4, worst
    assert sum(sdo[n].grad is not None for n in ref_grads) == len(ref_grads)


def test_bf16_floor_of_the_logit_tolerance(golden_dir):
    """Round 6, the experiment behind the GPU tests' logit bounds — CPU only, on the emulating oracle, whose
    eval-logit error tracks the HIP path's (0.0231 / 0.0306 / 0.0210 here vs 0.0271 / 0.0312 / 0.0225 measured on the MI355X for
    the tiny / dh 64 / 12-layer encoders, profiles/r05_logits_f32_ab.txt).  north_star asks for log-probs within 1e-2 of the
    reference's fp32 run in bf16.  Three facts, pinned on the dh-64 fixture (tools/probes/resid_f32_emulation.py prints all
    sizes, profiles/r06_resid_f32_emulation.json):
      * bf16 WEIGHTS ALONE (every activation, statistic and sum in fp32) already move the logits by ~1.0e-2: no bf16 run — the
        reference's own autocast run included — can sit inside 1e-2 with margin;
      * keeping the residual stream in fp32, as the reference's autocast run does from the first layer's final LayerNorm on
        (`bf16_emulation(resid_f32=True)`), brings 0.031 -> 0.024 (12 layers: 0.021 -> 0.0135): closer, still above 1e-2;
      * fp32 logits on top change nothing (the error is accumulated upstream).
    So the fp32 stream was priced and NOT built (+ ~19 MB of HBM traffic per sub-layer, ~0.5 ms per update step, for a bound
    that stays unmet), and the GPU tests state absolute bounds from these measurements instead of a rescaled 1e-2."""
    g = np.load(os.path.join(golden_dir, "ref_conformer_ctc_dh64.npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    feats, lengths = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"])
    ref = torch.from_numpy(g["out::eval_logits"])

    def err(sdx, **kw):
        with torch.no_grad(), torch_ref.bf16_emulation(**kw):
            lo, _ = torch_ref.encoder(feats, lengths, sdx, H=2, layer_type="conformer", training=False)
        return float((lo - ref).abs().max()), float((torch.log_softmax(lo, -1) - torch.log_softmax(ref, -1)).abs().max())

    w16 = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() and v.dim() >= 2 else v) for k, v in sd.items()}
    floor, floor_lp = err(w16, on=False)
    bf16_stream, _ = err(sd, on=True, flash=True)
    f32_stream, f32_stream_lp = err(sd, on=True, flash=True, resid_f32=True)
    assert err(sd, on=False)[0] == 0.0                       # (the restatement itself is pinned: exactly the reference's logits)
    assert 7e-3 < floor < 1.4e-2 and 7e-3 < floor_lp < 1.4e-2, (floor, floor_lp)
    assert 2.5e-2 < bf16_stream < 3.6e-2, bf16_stream        # the HIP path's rounding points (GPU test bound: 3.5e-2)
    assert 1.2e-2 < f32_stream < bf16_stream and f32_stream_lp > 1e-2, (f32_stream, f32_stream_lp)
