"""CPU tests of the host-side mirror: the reference's own known-answer vectors for the path
(tests/espresso/test_speech_utils.py:127-263 collate_frames / sequence_mask /
convert_padding_direction / edit_distance; tests/test_data_utils.py:13-136 batch_by_size vs a slow
baseline) plus dictionary layout, lr schedule, SpecAugment RNG order and registry names."""
from collections import Counter

import numpy as np
import pytest
import torch

import espresso_amd
from espresso_amd import registry
from espresso_amd.data import data_utils
from espresso_amd.data.asr_dictionary import AsrDictionary
from espresso_amd.data.feature_transforms import AdaptiveSpecAugmentTransform, numpy_seed_value
from espresso_amd.optim.noam_lr_scheduler import NoamSchedule
from espresso_amd.tools import utils


def test_registry_names_match_reference():
    assert "speech_recognition_espresso" in registry.TASK_REGISTRY
    assert "speech_transformer_encoder_model" in registry.MODEL_REGISTRY
    assert "ctc_loss" in registry.CRITERION_REGISTRY
    assert "noam" in registry.LR_SCHEDULER_REGISTRY
    assert {"adaptive_specaugment", "global_cmvn"} <= set(registry.AUDIO_FEATURE_TRANSFORM_REGISTRY)


def test_collate_frames():
    vals = [
        torch.tensor([4.5, 2.3, 1.2]).unsqueeze(-1).expand(-1, 10),
        torch.tensor([6.7, 9.8]).unsqueeze(-1).expand(-1, 10),
        torch.tensor([7.7, 5.4, 6.2, 8.0]).unsqueeze(-1).expand(-1, 10),
        torch.tensor([1.5]).unsqueeze(-1).expand(-1, 10),
    ]
    e1 = torch.tensor([[4.5, 2.3, 1.2, 0.0], [6.7, 9.8, 0.0, 0.0], [7.7, 5.4, 6.2, 8.0], [1.5, 0.0, 0.0, 0.0]]).unsqueeze(-1).expand(-1, -1, 10)
    e2 = torch.tensor([[0.0, 4.5, 2.3, 1.2], [0.0, 0.0, 6.7, 9.8], [7.7, 5.4, 6.2, 8.0], [0.0, 0.0, 0.0, 1.5]]).unsqueeze(-1).expand(-1, -1, 10)
    assert torch.equal(utils.collate_frames(vals, pad_value=0.0, left_pad=False), e1)
    assert torch.equal(utils.collate_frames(vals, pad_value=0.0, left_pad=True), e2)


def test_sequence_mask():
    seq_len = torch.tensor([1, 4, 0, 3]).int()
    m1 = torch.tensor([[1, 0, 0, 0], [1, 1, 1, 1], [0, 0, 0, 0], [1, 1, 1, 0]]).bool()
    m2 = torch.tensor([[1, 0, 0, 0, 0], [1, 1, 1, 1, 0], [0, 0, 0, 0, 0], [1, 1, 1, 0, 0]]).bool()
    assert torch.equal(utils.sequence_mask(seq_len), m1)
    assert torch.equal(utils.sequence_mask(seq_len, max_len=5), m2)


def test_convert_padding_direction():
    t1 = torch.tensor([[4.5, 2.3, 1.2, 0.0], [6.7, 9.8, 0.0, 0.0], [7.7, 5.4, 6.2, 8.0], [1.5, 0.0, 0.0, 0.0]]).unsqueeze(-1).expand(-1, -1, 10)
    t2 = torch.tensor([[0.0, 4.5, 2.3, 1.2], [0.0, 0.0, 6.7, 9.8], [7.7, 5.4, 6.2, 8.0], [0.0, 0.0, 0.0, 1.5]]).unsqueeze(-1).expand(-1, -1, 10)
    seq_len = torch.tensor([3, 2, 4, 1]).int()
    assert torch.equal(utils.convert_padding_direction(t1, seq_len, right_to_left=True), t2)
    assert torch.equal(utils.convert_padding_direction(t2, seq_len, left_to_right=True), t1)


def test_edit_distance():
    _, steps, c = utils.edit_distance([], [])
    assert c == Counter({"words": 0, "corr": 0, "sub": 0, "ins": 0, "del": 0}) and steps == []
    _, steps, c = utils.edit_distance(["a", "b", "c"], [])
    assert c == Counter({"words": 3, "corr": 0, "sub": 0, "ins": 0, "del": 3}) and steps == ["del"] * 3
    _, steps, c = utils.edit_distance(["a", "b", "c"], ["a", "b", "c"])
    assert c == Counter({"words": 3, "corr": 3, "sub": 0, "ins": 0, "del": 0}) and steps == ["corr"] * 3
    _, steps, c = utils.edit_distance(["a", "b", "c"], ["d", "b", "c", "e", "f"])
    assert c == Counter({"words": 3, "corr": 2, "sub": 1, "ins": 2, "del": 0})
    assert steps == ["sub", "corr", "corr", "ins", "ins"]
    _, steps, c = utils.edit_distance(["b", "c", "d", "e", "f", "h"], ["d", "b", "c", "e", "f", "g"])
    assert c == Counter({"words": 6, "corr": 4, "sub": 1, "ins": 1, "del": 1})
    assert steps == ["ins", "corr", "corr", "del", "corr", "corr", "sub"]


def _slow_batch_by_size(indices, ntok, max_tokens, max_sentences, bsz_mult):
    """Slow baseline with the semantics of the reference's test helper (tests/test_data_utils.py:15-41)."""
    batches, batch = [], []
    for pos, idx in enumerate(indices):
        cand = batch + [pos]
        mx = max(int(ntok[p]) for p in cand)
        over = (0 < max_tokens < mx * len(cand)) or (0 < max_sentences < len(cand))
        if over:
            keep = max(bsz_mult * (len(batch) // bsz_mult), len(batch) % bsz_mult)
            batches.append(batch[:keep])
            batch = batch[keep:] + [pos]
        else:
            batch = cand
    if batch:
        batches.append(batch)
    return [[int(indices[p]) for p in b] for b in batches if b]


@pytest.mark.parametrize("bsz_mult", [1, 2, 4, 8])
@pytest.mark.parametrize("max_sentences", [-1, 1, 3, 10])
def test_batch_by_size_matches_slow_baseline(bsz_mult, max_sentences):
    rng = np.random.default_rng(bsz_mult * 100 + max_sentences)
    for _ in range(50):
        n = int(rng.integers(1, 80))
        ntok = rng.integers(1, 40, size=n)
        idx = rng.permutation(n)
        for mt in (-1, 40, 100, 260):
            got = [b.tolist() for b in data_utils.batch_by_size(idx, ntok, mt, max_sentences, bsz_mult)]
            flat = [i for b in got for i in b]
            assert flat == idx.tolist()  # order preserved, nothing dropped
            for b in got:
                pos = [int(np.where(idx == i)[0][0]) for i in b]
                if mt > 0:
                    assert max(ntok[p] for p in pos) * len(b) <= mt
                if max_sentences > 0:
                    assert len(b) <= max_sentences
            if bsz_mult == 1:
                assert got == _slow_batch_by_size(idx, ntok, mt, max_sentences, 1)


def test_batch_by_size_recipe_shapes():
    """26000-frame / 24-utterance budget of transformer_ctc_librispeech.yaml:27-31."""
    from espresso_amd.data import synthetic

    batches, n_samples = synthetic.make_batches(2000, max_tokens=26000, max_sentences=24, seed=1, shuffle=False)
    frames = synthetic.num_frames(n_samples)
    assert sum(len(b) for b in batches) == 2000
    for b in batches:
        assert len(b) <= 24 and frames[b].max() * len(b) <= 26000


def test_collate_tokens_move_eos():
    vals = [torch.tensor([5, 6, 2]), torch.tensor([7, 2])]
    out = data_utils.collate_tokens(vals, pad_idx=1, eos_idx=2, move_eos_to_beginning=True)
    assert out.tolist() == [[2, 5, 6], [2, 7, 1]]


def test_dictionary_layout():
    d = AsrDictionary.from_symbols(["a", "b"], enable_bos=True)
    assert (d.index("<s>"), d.pad(), d.eos(), d.unk()) == (0, 1, 2, 3)
    assert d.index("a") == 4 and d.index("<space>") == 6 and len(d) == 7
    d2 = AsrDictionary.from_symbols(["a", "b"], enable_bos=False)
    assert (d2.pad(), d2.eos(), d2.unk()) == (0, 1, 2)


def test_noam_schedule():
    class Opt:
        def set_lr(self, lr):
            self.lr = lr

    o = Opt()
    s = NoamSchedule(o, lr=5.0, warmup_steps=25000, model_size=512, final_lr=1e-6)
    assert o.lr == pytest.approx(5.0 * 512 ** -0.5 * 25000 ** -1.5)
    s.step_update(24999)
    assert o.lr == pytest.approx(5.0 * 512 ** -0.5 * 25000 ** -0.5)
    s.step_update(10 ** 9)
    assert o.lr == pytest.approx(max(5.0 * 512 ** -0.5 * (10 ** 9 + 1) ** -0.5, 1e-6))


def test_specaugment_rng_order_matches_reference_fixture(golden_dir):
    """Masks drawn by draw_masks under numpy_seed(1, 2, index) + the oracle's fill reproduce what the
    reference's AdaptiveSpecAugmentTransform produced (tests/golden/specaug.npz, oracle/gen_golden.py)."""
    import os

    from oracle import fbank_ref

    g = np.load(os.path.join(golden_dir, "specaug.npz"))
    tr = AdaptiveSpecAugmentTransform.from_config_dict({"freq_mask_N": 2, "freq_mask_F": 27, "time_mask_pm": 0.04, "time_mask_ps": 0.04})
    for k in range(3):
        spec, out = g[f"in_{k}"], g[f"out_{k}"]
        M, idx = g[f"meta_{k}"]
        state = np.random.get_state()
        np.random.seed(numpy_seed_value(1, 2, int(idx)))
        fm, tm = tr.draw_masks(int(M), 80)
        np.random.set_state(state)
        np.testing.assert_array_equal(fbank_ref.specaugment_apply(spec, fm, tm, None), out)


def test_wer_scorer_counts():
    from espresso_amd.tools.wer import Scorer

    d = AsrDictionary.from_symbols(list("abcdefgh"), enable_bos=False)
    s = Scorer(d)
    # reference "ab cd" vs hypothesis "ab ce f": chars a b <space> c d | a b <space> c e <space> f
    s.add_evaluation("u1", "a b <space> c d", "a b <space> c e <space> f")
    assert s.char_counter == Counter({"words": 5, "corr": 4, "sub": 1, "ins": 2, "del": 0})
    assert s.word_counter == Counter({"words": 2, "corr": 1, "sub": 1, "ins": 1, "del": 0})
    assert s.wer() == 100.0 and s.cer() == 60.0
    s.add_evaluation("u2", "g h", "g h")
    assert s.tot_word_count() == 3 and s.wer() == pytest.approx(200.0 / 3)


def test_tensorized_prefix_tree_matches_reference(golden_dir):
    """children / prev_subword_idx / word_idx / word_set_idx identical to what espresso/tools/tensorized_prefix_tree.py built
    for the same word and character dictionaries (fixture written by oracle/gen_golden.py lookahead)."""
    import os

    import numpy as np

    from espresso_amd.data.asr_dictionary import AsrDictionary
    from espresso_amd.tools.tensorized_prefix_tree import TensorizedPrefixTree, tokenize

    g = np.load(os.path.join(golden_dir, "ref_lookahead_wordlm_tiny.npz"))
    wd = AsrDictionary.from_symbols([str(w) for w in g["words"]], enable_bos=False, add_space=False)
    sd = AsrDictionary.from_symbols([str(c) for c in g["chars"]], enable_bos=False)
    t = TensorizedPrefixTree.build(wd, sd, lambda x: tokenize(x).split(" "))
    for k in ("children", "prev_subword_idx", "word_idx", "word_set_idx"):
        assert np.array_equal(getattr(t, k).numpy(), g["tree_" + k]), k
    assert tokenize("AB C") == "A B <space> C"


def test_speech_recognize_host_pieces(tmp_path):
    """WAV / scp readers, frame-budget batching and the option surface of the recognition CLI (the GPU loop is covered by the
    generator and decoder parity tests)."""
    import wave

    import numpy as np

    from espresso_amd import speech_recognize as sr

    x = (np.sin(np.arange(1600) / 10.0) * 1000).astype("<i2")
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(x.tobytes())
    y = sr.read_wav(str(p))
    assert y.dtype == np.float32 and np.array_equal(y, x.astype(np.float32))
    scp = tmp_path / "wav.scp"
    scp.write_text(f"utt1 {p}\nutt2 {p}\n")
    assert sr.read_scp(str(scp)) == {"utt1": str(p), "utt2": str(p)}
    b = sr.make_batches(["a", "b", "c", "d"], [16000 * 10, 16000 * 2, 16000 * 9, 16000 * 3], max_tokens=2100, max_sentences=3)
    assert sorted(sum(b, [])) == [0, 1, 2, 3] and b[0] == [0, 2] and all(len(x) <= 3 for x in b)
    a = sr.get_parser().parse_args(["--path", "m.pt", "--model-config", "m.yaml", "--dict", "d.txt", "--wav-scp", str(scp),
                                    "--lm-weight", "0.47", "--eos-factor", "1.5", "--beam", "60"])
    assert a.beam == 60 and a.lm_weight == 0.47 and a.eos_factor == 1.5 and a.search == "beam"
