"""CPU tests of the host-side mirror: the reference's own known-answer vectors for the path
(tests/espresso/test_speech_utils.py:127-263 collate_frames / sequence_mask /
convert_padding_direction / edit_distance; tests/test_data_utils.py:13-136 batch_by_size vs a slow
baseline) plus dictionary layout, lr schedule, SpecAugment RNG order and registry names."""
from collections import Counter

import os
import math
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


def test_specaugment_time_warp():
    """time_warp_W > 0 (adaptive_specaugment.py:94-109; no recipe uses it, cv2 is absent here: parity unpinned).  (1) RNG order:
    w0 and w are drawn before the masks and only when 2 W < num_frames — checked against a literal transcription of the
    reference's draws; (2) the source-row table + the torch gather / lerp of the front-end == the oracle's restatement of
    cv2.resize(INTER_LINEAR) on both segments, bit for bit in float32; (3) length preserved, w = 0 is the identity, a linear ramp
    stays monotonic."""
    import torch

    from espresso_amd.data.gpu_frontend import apply_time_warp
    from oracle import fbank_ref

    tr = AdaptiveSpecAugmentTransform.from_config_dict({"time_warp_W": 5, "freq_mask_N": 2, "freq_mask_F": 27, "time_mask_pm": 0.04,
                                                        "time_mask_ps": 0.04})
    rng = np.random.default_rng(0)
    for m in (9, 10, 11, 40, 163):
        np.random.seed(100 + m)
        warp, fm, tm = tr.draw_masks(m, 80, with_warp=True)
        np.random.seed(100 + m)  # literal transcription of the reference's draw order
        want = None
        if 2 * 5 < m:
            want = (int(np.random.randint(5, m - 5)), int(np.random.randint(-5 + 1, 5)))
        fm2 = []
        for _ in range(2):
            f = np.random.randint(0, 27)
            fm2.append((int(np.random.randint(0, 80 - f)), int(f)))
        assert warp == want and fm == fm2, (m, warp, want)
        if warp is None:
            assert m <= 10
            continue
        spec = rng.standard_normal((m, 80)).astype(np.float32)
        ref = fbank_ref.time_warp(spec, *warp)
        assert ref.shape == spec.shape
        i0, i1, fr = tr.warp_indices(m, *warp)
        got = apply_time_warp(torch.from_numpy(spec)[None], torch.from_numpy(i0)[None], torch.from_numpy(i1)[None],
                              torch.from_numpy(fr)[None])[0].numpy()
        np.testing.assert_array_equal(got, ref)
        ramp = np.repeat(np.arange(m, dtype=np.float32)[:, None], 3, 1)
        assert (np.diff(fbank_ref.time_warp(ramp, *warp)[:, 0]) >= 0).all()
    spec = rng.standard_normal((30, 4)).astype(np.float32)
    np.testing.assert_array_equal(fbank_ref.time_warp(spec, 12, 0), spec)


def test_transducer_beam_state_pools_are_bounded_by_grouping():
    """The batched transducer beam search keeps predictor / LM states in append-only row pools (released when a group of
    searches ends): a batch whose estimated pool size (frames x expansions x beam rows, (layers x H x 10 + H x 2) bytes each)
    exceeds the budget is searched in several groups that share the encoder pass; ordinary batches stay one group."""
    import torch

    from espresso_amd.tools.transducer_beam_search_decoder import TransducerBeamSearchDecoder

    class Dec:
        hidden_size = 8
        layers = [0, 1]

        def init_state(self, n, device):
            z = lambda dt: torch.zeros(n, self.hidden_size, dtype=dt)
            return {"h16": [z(torch.bfloat16) for _ in self.layers], "h32": [z(torch.float32) for _ in self.layers],
                    "c": [z(torch.float32) for _ in self.layers]}

    class Model:
        decoder = Dec()

        def eval(self):
            return self

    d = AsrDictionary.from_symbols(list("abcdefgh"), enable_bos=True)
    dec = TransducerBeamSearchDecoder([Model()], d, beam_size=5, max_num_expansions_per_step=2)
    assert dec._pool_row_bytes() == 2 * 8 * (2 + 4 + 4) + 2 * 8
    lens = [100, 250, 40, 300, 7]
    assert dec._pool_groups(list(range(5)), lens) == [[0, 1, 2, 3, 4]]          # 4 GiB default: one group
    dec.state_pool_budget_bytes = dec._pool_row_bytes() * 10 * 300               # room for 300 frame-rounds of 10 rows
    groups = dec._pool_groups(list(range(5)), lens)
    assert groups == [[0], [1, 2], [3], [4]], groups                             # batch order kept, every utterance exactly once
    assert dec._pool_groups([1, 3], lens) == [[1], [3]]                          # `only`-restricted searches are grouped too


def test_wer_scorer_counts():
    from espresso_amd.tools.wer import Scorer

    d = AsrDictionary.from_symbols(list("abcdefgh"), enable_bos=False)
    s = Scorer(d)
    # reference "ab cd" vs hypothesis "ab ce f": chars a b <space> c d | a b <space> c e <space> f
    s.add_evaluation("u1", "a b <space> c d", "a b <space> c e <space> f")
    assert s.char_counter == Counter({"words": 5, "corr": 4, "sub": 1, "ins": 2, "del": 0})
    assert s.word_counter == Counter({"words": 2, "corr": 1, "sub": 1, "ins": 1, "del": 0})
    assert s.wer() == (100.0, 50.0, 50.0, 0.0) and s.cer() == (60.0, 20.0, 40.0, 0.0)  # (rate, sub, ins, del) like the reference
    s.add_evaluation("u2", "g h", "g h")
    assert s.tot_word_count() == 3 and s.wer()[0] == pytest.approx(200.0 / 3)
    assert s.summary_lines()[0].startswith("WER=66.67%, Sub=33.33%, Ins=33.33%, Del=0.00%, #words=3")


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
    # sharded decoding (espresso/speech_recognize.py:188-189 -> ShardedIterator): replica i takes batches i, i+n, ...; together
    # the replicas cover every batch exactly once; defaults come from the launcher's RANK / WORLD_SIZE
    assert a.num_shards == 1 and a.shard_id == 0
    bl = [[i] for i in range(7)]
    parts = [sr.shard_batches(bl, 3, i) for i in range(3)]
    assert parts[0] == [[0], [3], [6]] and parts[1] == [[1], [4]] and sorted(sum(sum(parts, []), [])) == list(range(7))
    with pytest.raises(ValueError):
        sr.shard_batches(bl, 2, 2)
    a = sr.get_parser().parse_args(["--path", "m.pt", "--dict", "d.txt", "--wav-scp", str(scp), "--num-shards", "8", "--shard-id", "5"])
    assert (a.num_shards, a.shard_id) == (8, 5)


# ---- datasets / collater / task hooks (reference: tests/espresso/test_asr_dataset.py:102-189) -----------------------------
def _toy_corpus(tmp_path, n=9, seed=0, kind="wave"):
    import json

    from espresso_amd.data import audio_utils, kaldi_io
    from espresso_amd.data.asr_dictionary import AsrDictionary

    rng = np.random.default_rng(seed)
    letters = list("ABCDEFGHIJKLMNOPQRSTUVWXYZ")
    d = AsrDictionary.from_symbols(letters, enable_bos=False)
    d.build_bpe("characters_asr")
    utts, arrays, feats = {}, {}, {}
    for i in range(n):
        u = f"utt{i:03d}"
        ns = int(rng.integers(4000, 16000))
        x = np.round(rng.standard_normal(ns) * 3000).astype(np.float32)
        arrays[u] = x
        text = " ".join("".join(rng.choice(letters, size=int(rng.integers(1, 5)))) for _ in range(int(rng.integers(1, 4))))
        if kind == "wave":
            path = str(tmp_path / f"{u}.wav")
            audio_utils.write_wav(path, x)
            utts[u] = {"wave": path, "text": text}
        else:
            feats[u] = rng.standard_normal((int(rng.integers(20, 60)), 8)).astype(np.float32)
            utts[u] = {"text": text}
    if kind == "feat":
        scp = kaldi_io.write_ark(str(tmp_path / "feats.ark"), feats)
        for u in utts:
            utts[u]["feat"] = scp[u]
            utts[u]["utt2num_frames"] = str(feats[u].shape[0])
    with open(tmp_path / "train.json", "w") as f:
        json.dump(utts, f)
    return d, utts, arrays, feats


@pytest.mark.parametrize("kind", ["wave", "feat"])
def test_asr_dataset_round_trip(tmp_path, kind):
    from espresso_amd.data.asr_dataset import get_asr_dataset_from_json, samples_to_frames

    d, utts, arrays, feats = _toy_corpus(tmp_path, kind=kind)
    ds = get_asr_dataset_from_json(str(tmp_path), "train", d, autoregressive=True)
    assert len(ds) == len(utts)
    idx = list(range(len(ds)))
    for s in range(0, len(idx), 4):
        b = ds.collater([ds[i] for i in idx[s:s + 4]])
        bsz = b["nsentences"]
        assert bsz == len(b["utt_id"]) == b["target"].shape[0] == b["net_input"]["src_lengths"].numel()
        lens = b["net_input"]["src_lengths"].tolist()
        assert lens == sorted(lens, reverse=True)
        hyp = d.string(b["target"], extra_symbols_to_ignore={d.pad()}).split("\n")
        for j, u in enumerate(b["utt_id"]):
            assert d.wordpiece_decode(hyp[j]) == utts[u]["text"] == b["text"][j]
            # shifted decoder input: </s> first, then the target without its final </s>
            t = b["target"][j][b["target"][j] != d.pad()]
            p = b["net_input"]["prev_output_tokens"][j][: len(t)]
            assert p[0] == d.eos() and torch.equal(p[1:], t[:-1])
            if kind == "wave":
                o0, o1 = int(b["wav_offsets"][j]), int(b["wav_offsets"][j + 1])
                assert o1 - o0 == b["num_samples"][j] == len(arrays[u])
                assert np.array_equal(b["wav"][o0:o1].numpy(), np.clip(arrays[u], -32768, 32767))
                assert lens[j] == samples_to_frames(len(arrays[u]))
            else:
                assert torch.equal(b["net_input"]["src_tokens"][j, : lens[j]], torch.from_numpy(feats[u]))
                assert float(b["net_input"]["src_tokens"][j, lens[j]:].abs().sum()) == 0.0
        assert b["ntokens"] == int((b["target"] != d.pad()).sum())
        assert b["id"].tolist() == [ds.src.utt_ids.index(u) for u in b["utt_id"]]


def test_asr_dataset_matches_src_and_tgt():
    from espresso_amd.data.asr_dataset import AsrDataset, AsrTextDataset, AudioFeatDataset
    from espresso_amd.data.asr_dictionary import AsrDictionary

    d = AsrDictionary.from_symbols(list("ABC"))
    feats = [np.full((3 + i, 2), float(i + 1), dtype=np.float32) for i in range(4)]
    src = AudioFeatDataset(["a", "b", "c", "d"], feats)
    tgt = AsrTextDataset(["d", "b", "x"], ["A B", "C", "A"], d)
    ds = AsrDataset(src, src.sizes, tgt, tgt.sizes, d, shuffle=False)
    assert ds.src.utt_ids == ds.tgt.utt_ids == ["b", "d"]
    assert ds.src_sizes.tolist() == [4, 6] and ds.tgt_sizes.tolist() == [2, 3]
    assert ds.ordered_indices().tolist() == [0, 1]
    kept, ignored = ds.filter_indices_by_size(np.array([0, 1]), (5, 10))
    assert kept.tolist() == [0] and ignored == [1]
    assert ds.num_tokens(1) == 6 and ds.size(1) == (6, 3)


def test_kaldi_compressed_matrix_formats():
    import io
    import struct

    from espresso_amd.data import kaldi_io

    rng = np.random.default_rng(0)
    rows, cols = 7, 3
    mn, rg = -2.0, 5.0
    hdr = np.sort(rng.integers(0, 65536, size=(cols, 4)), axis=1).astype("<u2")
    data = rng.integers(0, 256, size=(cols, rows)).astype(np.uint8)
    buf = b"\0BCM " + struct.pack("<ffii", mn, rg, rows, cols) + hdr.tobytes() + data.tobytes()
    got = kaldi_io.read_mat_fd(io.BytesIO(buf))
    p = mn + rg * hdr.astype(np.float64) / 65535.0
    for c in range(cols):
        for r in range(rows):
            v = float(data[c, r])
            if v <= 64:
                e = p[c, 0] + (p[c, 1] - p[c, 0]) * v / 64.0
            elif v <= 192:
                e = p[c, 1] + (p[c, 2] - p[c, 1]) * (v - 64) / 128.0
            else:
                e = p[c, 2] + (p[c, 3] - p[c, 2]) * (v - 192) / 63.0
            assert abs(got[r, c] - e) < 1e-4
    v16 = rng.integers(0, 65536, size=(rows, cols)).astype("<u2")
    got = kaldi_io.read_mat_fd(io.BytesIO(b"\0BCM2 " + struct.pack("<ffii", mn, rg, rows, cols) + v16.tobytes()))
    assert np.allclose(got, mn + rg * v16 / 65535.0, atol=1e-5)
    v8 = rng.integers(0, 256, size=(rows, cols)).astype(np.uint8)
    got = kaldi_io.read_mat_fd(io.BytesIO(b"\0BCM3 " + struct.pack("<ffii", mn, rg, rows, cols) + v8.tobytes()))
    assert np.allclose(got, mn + rg * v8 / 255.0, atol=1e-5)
    dm = rng.standard_normal((rows, cols))
    got = kaldi_io.read_mat_fd(io.BytesIO(b"\0BDM \4" + struct.pack("<i", rows) + b"\4" + struct.pack("<i", cols) + dm.astype("<f8").tobytes()))
    assert np.allclose(got, dm.astype(np.float32))


def test_task_batches_shard_like_the_epoch_iterator(tmp_path):
    from espresso_amd.data.asr_dataset import get_asr_dataset_from_json
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask

    d, utts, _, _ = _toy_corpus(tmp_path, n=23, kind="wave")
    cfg = SpeechRecognitionEspressoConfig(data=str(tmp_path), autoregressive=True, criterion_name="label_smoothed_cross_entropy_v2")
    task = SpeechRecognitionEspressoTask(cfg, d)
    ds = task.load_dataset("train")
    assert task.dataset("train") is ds and task.feat_dim == 80
    full = task.get_batches(ds, max_tokens=250, max_sentences=4, seed=3, epoch=2)
    again = task.get_batches(ds, max_tokens=250, max_sentences=4, seed=3, epoch=2)
    other = task.get_batches(ds, max_tokens=250, max_sentences=4, seed=3, epoch=3)
    assert [b.tolist() for b in full] == [b.tolist() for b in again]
    assert sorted(i for b in full for i in b.tolist()) == list(range(len(ds)))
    assert [b.tolist() for b in full] != [b.tolist() for b in other]
    for b in full:
        assert len(b) <= 4 and len(b) * int(ds.src_sizes[b].max()) <= 250
    shards = [task.get_batches(ds, max_tokens=250, max_sentences=4, seed=3, epoch=2, num_shards=2, shard_id=r) for r in range(2)]
    assert len(shards[0]) == len(shards[1]) == (len(full) + 1) // 2
    merged = [b.tolist() for pair in zip(*shards) for b in pair if len(b)]
    assert merged == [b.tolist() for b in full]


def test_characters_asr_and_tokenize_follow_reference_order():
    from espresso_amd.data.asr_dictionary import AsrDictionary
    from espresso_amd.data.encoders import CharactersAsr, tokenize

    assert tokenize("  HI  <noise> YOU ", non_lang_syms=["<noise>"]) == "H I <space> <noise> <space> Y O U"
    # alternation order, not longest match (espresso/tools/utils.py:44 joins the symbols with "|")
    assert tokenize("<a><ab>", non_lang_syms=["<a>", "<a><ab>"]) == "<a> < a b >"
    enc = CharactersAsr()
    assert enc.encode("AB C") == "A B <space> C <space>"
    assert enc.decode("A B <space> C <space>") == "AB C"
    d = AsrDictionary.from_symbols(list("ABC"))
    assert d.wordpiece_encode("AB C") == "AB C"
    d.build_bpe("characters_asr")
    ids = d.encode_line(d.wordpiece_encode("AB C"))
    assert d.wordpiece_decode(d.string(ids)) == "AB C"


def test_layer_runtime_rejects_short_arenas_before_launching():
    """ea_conformer_layer_fwd/bwd size their arenas with the same dry pass as the workspace query and refuse (-5) a short
    arena before any kernel is enqueued (no GPU needed to see the refusal)."""
    import ctypes

    from espresso_amd import _lib
    from espresso_amd._lib import EaConformerLayer, EaLayerShape

    lib = _lib.lib()
    sh = EaLayerShape()
    sh.B, sh.T, sh.C, sh.H, sh.F, sh.KW = 3, 50, 64, 4, 128, 31
    sh.training, sh.p_drop, sh.p_act, sh.p_attn, sh.seed, sh.has_attn_mask = 1, 0.1, 0.1, 0.1, 7, 0
    a, b = ctypes.c_long(0), ctypes.c_long(0)
    assert lib.ea_conformer_layer_workspace(ctypes.byref(sh), ctypes.byref(a), ctypes.byref(b)) == 0
    assert a.value > 0 and b.value > 0
    L = EaConformerLayer()
    null = ctypes.c_void_p(0)
    assert lib.ea_conformer_layer_fwd(ctypes.byref(L), ctypes.byref(sh), null, null, null, null, null, null, a.value // 2, null,
                                      b.value, null) == -5
    assert lib.ea_conformer_layer_bwd(ctypes.byref(L), ctypes.byref(sh), null, null, null, null, null, null, a.value, null, 1024,
                                      null) == -5
    sh.T = 5000  # beyond the runtime's limit: shape error, not a crash
    assert lib.ea_conformer_layer_workspace(ctypes.byref(sh), ctypes.byref(a), ctypes.byref(b)) == -2


# ---- learning-rate schedules named by the recipes (formulas of the reference classes, torch's plateau logic as the oracle) ----
class _Opt:
    def __init__(self):
        self.lr = None

    def set_lr(self, lr):
        self.lr = lr

    def get_lr(self):
        return self.lr


def test_tri_stage_schedule_matches_reference_formulas():
    from espresso_amd.optim.lr_schedulers import TriStageLRSchedule

    o = _Opt()
    s = TriStageLRSchedule(o, lr=1e-3, warmup_steps=2200, hold_steps=80000, decay_steps=100000)  # transformer_librispeech.yaml
    assert o.lr == pytest.approx(1e-5)
    assert s.step_update(1100) == pytest.approx(1e-5 + (1e-3 - 1e-5) * 1100 / 2200)
    assert s.step_update(2200) == pytest.approx(1e-3) and s.step_update(50000) == pytest.approx(1e-3)
    assert s.step_update(82200 + 50000) == pytest.approx(1e-3 * math.exp(math.log(0.01) * 0.5))
    assert s.step_update(182200) == pytest.approx(1e-5) and s.step_update(10 ** 7) == pytest.approx(1e-5)
    s2 = TriStageLRSchedule(_Opt(), lr=2.0, phase_ratio=(0.1, 0.4, 0.5), max_update=1000)
    assert (s2.warmup_steps, s2.hold_steps, s2.decay_steps) == (100, 400, 500)


def test_polynomial_decay_v2_schedule():
    from espresso_amd.optim.lr_schedulers import PolynomialDecayV2LRSchedule

    o = _Opt()
    s = PolynomialDecayV2LRSchedule(o, lr=0.002, warmup_updates=1000, end_learning_rate=1e-5, power=2.0, total_num_update=10000)
    assert o.lr == pytest.approx(0.002 / 1000)
    assert s.step_update(500) == pytest.approx(0.001)
    assert s.step_update(5500) == pytest.approx((0.002 - 1e-5) * 0.5 ** 2 + 1e-5)
    assert s.step_update(10000) == pytest.approx(1e-5) and s.step_update(99999) == pytest.approx(1e-5)


def test_reduce_lr_on_plateau_v2_follows_torch_plateau_logic():
    from espresso_amd.optim.lr_schedulers import ReduceLROnPlateauLRScheduleV2

    losses = [5.0, 4.0, 4.0, 3.99999, 4.2, 3.0, 3.1, 3.2, 3.3, 2.0, 2.0, 2.0, 2.0, 2.0]
    o = _Opt()
    s = ReduceLROnPlateauLRScheduleV2(o, lr=0.1, lr_shrink=0.5, lr_patience=1, start_reduce_lr_epoch=3, final_lr_scale=0.1)
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.SGD([p], lr=0.1)
    tsch = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, patience=1, factor=0.5, mode="min", threshold=1e-4, min_lr=0.01)
    for epoch, l in enumerate(losses, start=1):
        got = s.step(epoch, l)
        if epoch < 3:  # the reference only records the epoch and keeps lr[0] before start_reduce_lr_epoch
            tsch.last_epoch = epoch
            assert got == pytest.approx(0.1)
        else:
            tsch.step(l)
            assert got == pytest.approx(topt.param_groups[0]["lr"]), (epoch, got)
    assert o.lr == pytest.approx(0.01)  # shrunk 0.1 -> 0.05 -> 0.025 -> 0.0125 -> floor final_lr_scale * lr
    w = ReduceLROnPlateauLRScheduleV2(_Opt(), lr=0.1, warmup_updates=10)
    assert w.optimizer.lr == 0 and w.step_update(5) == pytest.approx(0.05) and w.step_update(10) == pytest.approx(0.1)
    w.step_update(11)
    assert w.warmup_end


# ---- training entry point: recipe configuration, checkpoint rules, trainer state (fairseq_cli/train.py, checkpoint_utils.py) ----

_RECIPE = """
common: {seed: 3, log_interval: 10}
checkpoint: {save_dir: ckpt, save_interval_updates: 2, keep_interval_updates: 2, keep_last_epochs: 2, best_checkpoint_metric: wer}
task:
  _name: speech_recognition_espresso
  data: ???
  dict: ???
  autoregressive: false
dataset: {max_tokens: 26000, batch_size: 24, train_subset: train, valid_subset: valid}
criterion: {_name: ctc_loss, zero_infinity: true}
optimization: {max_epoch: 100, clip_norm: 2.0, sentence_avg: true, update_freq: [2, 1], lr: [5.0]}
optimizer: {_name: adam, adam_betas: "(0.9,0.98)", adam_eps: 1e-08, weight_decay: 0.0}
lr_scheduler: {_name: noam, warmup_steps: 25000, model_size: "${model.encoder.embed_dim}", final_lr: 1e-6}
model:
  _name: speech_transformer_encoder_model
  encoder: {embed_dim: 512, layers: 12, conv_channels: "[64, 64, 128, 128]"}
"""


def test_recipe_config_loader(tmp_path):
    from espresso_amd import config as C

    p = tmp_path / "recipe.yaml"
    p.write_text(_RECIPE)
    with pytest.raises(ValueError, match="task.data"):
        C.load_config(str(p))
    cfg = C.load_config(str(p), ["task.data=/d", "task.dict=/d/dict.txt", "model.encoder.embed_dim=256", "+checkpoint.patience=3",
                                 "optimization.lr=[2.5]", "dataset.valid_subset=valid,dev"])
    assert cfg["task"]["data"] == "/d" and cfg["checkpoint"]["patience"] == 3
    assert cfg["lr_scheduler"]["model_size"] == 256 and isinstance(cfg["lr_scheduler"]["model_size"], int)  # interpolation keeps the type
    assert cfg["checkpoint"]["restore_file"] == "checkpoint_last.pt" and cfg["dataset"]["required_batch_size_multiple"] == 8  # defaults
    assert C.literal(cfg["optimizer"]["adam_betas"]) == (0.9, 0.98)
    # YAML 1.1 reads `1e-08` / `1e-6` as strings; omegaconf (the reference's parser) and this loader read floats
    assert cfg["optimizer"]["adam_eps"] == 1e-8 and cfg["lr_scheduler"]["final_lr"] == 1e-6
    assert C.load_config(str(p), ["task.data=/d", "task.dict=x", "lr_scheduler.final_lr=2e-5"])["lr_scheduler"]["final_lr"] == 2e-5
    assert C.literal(cfg["model"]["encoder"]["conv_channels"]) == [64, 64, 128, 128]
    assert C.as_list(cfg["optimization"]["lr"]) == [2.5]
    assert [C.per_epoch(cfg["optimization"]["update_freq"], e) for e in (1, 2, 3, 9)] == [2, 1, 1, 1]
    assert cfg["dataset"]["valid_subset"] == "valid,dev"
    with pytest.raises(ValueError):
        C.load_config(str(p), ["task.data"])


class _FakeTrainer:
    def __init__(self):
        self.num_updates, self.saved = 0, []

    def save_checkpoint(self, filename, extra_state=None):
        self.saved.append((os.path.basename(filename), dict(extra_state)))
        with open(filename, "w") as f:
            f.write("x")


def test_checkpoint_naming_and_retention(tmp_path):
    """The file names and pruning of fairseq/checkpoint_utils.py:34-172 for a minimising and a maximising metric."""
    from espresso_amd.checkpoint_utils import CheckpointSaver, checkpoint_paths
    from espresso_amd.config import DEFAULTS

    cfg = dict(DEFAULTS["checkpoint"], save_dir=str(tmp_path / "c"), save_interval_updates=2, keep_interval_updates=2, keep_last_epochs=2,
               keep_best_checkpoints=2, best_checkpoint_metric="wer")
    saver, tr = CheckpointSaver(cfg), _FakeTrainer()
    ls = lambda: sorted(os.listdir(cfg["save_dir"]))
    tr.num_updates = 2
    files = saver.save(tr, 1, False, {"epoch": 1, "iterations_in_epoch": 2}, 30.0)
    assert [os.path.basename(f) for f in files][:2] == ["checkpoint_1_2.pt", "checkpoint_best.pt"] and "checkpoint_last.pt" in ls()
    assert tr.saved[-1][1]["train_iterator"]["iterations_in_epoch"] == 2 and tr.saved[-1][1]["best"] == 30.0
    tr.num_updates = 4
    saver.save(tr, 1, False, {}, 35.0)  # worse: no new checkpoint_best
    assert len(tr.saved) == 2 and tr.saved[-1][0] == "checkpoint_1_4.pt" and tr.saved[-1][1]["best"] == 30.0
    tr.num_updates = 6
    saver.save(tr, 1, False, {}, 20.0)
    names = ls()
    assert "checkpoint_1_2.pt" not in names and {"checkpoint_1_4.pt", "checkpoint_1_6.pt"} <= set(names)  # keep_interval_updates = 2
    kept_best = [n for n in names if n.startswith("checkpoint.best_wer_")]
    assert len(kept_best) == 2 and all(n.startswith(("checkpoint.best_wer_20.000", "checkpoint.best_wer_30.000")) for n in kept_best)
    for ep in (1, 2, 3):
        tr.num_updates += 1
        saver.save(tr, ep, True, {"epoch": ep}, None)  # no validation: never touches checkpoint_best
    names = ls()
    assert "checkpoint1.pt" not in names and {"checkpoint2.pt", "checkpoint3.pt"} <= set(names)  # keep_last_epochs = 2
    assert [os.path.basename(p) for p in checkpoint_paths(cfg["save_dir"])] == ["checkpoint3.pt", "checkpoint2.pt"]
    assert saver.best == 20.0
    cfg2 = dict(cfg, save_dir=str(tmp_path / "m"), maximize_best_checkpoint_metric=True, keep_best_checkpoints=-1)
    s2, t2 = CheckpointSaver(cfg2), _FakeTrainer()
    for n, score in ((2, 0.5), (4, 0.4), (6, 0.7)):
        t2.num_updates = n
        s2.save(t2, 1, False, {}, score)
    assert [("checkpoint_best.pt" in os.listdir(cfg2["save_dir"])), s2.best] == [True, 0.7]
    assert sum(1 for name, _ in t2.saved if name.startswith("checkpoint_1_")) == 3
    cfg3 = dict(cfg, save_dir=str(tmp_path / "n"), no_save=True)
    assert CheckpointSaver(cfg3).save(_FakeTrainer(), 1, True, {}, 1.0) == [] and not os.path.exists(cfg3["save_dir"])


def test_trainer_checkpoint_round_trip(tmp_path):
    """Trainer.state_dict has the reference's top-level keys and a torch.optim.Adam-shaped optimizer state
    (fairseq/trainer.py:387-431); loading restores masters, moments, update count and the schedule on a fresh trainer."""
    import torch.nn as nn

    from espresso_amd.trainer import Trainer

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = nn.Linear(8, 16), nn.Linear(16, 4)
            self.register_buffer("running", torch.zeros(3))
            self.n = 0

        def set_num_updates(self, n):
            self.n = n

    class Crit:
        pass

    def make(seed):
        torch.manual_seed(seed)
        return Trainer(None, M(), Crit(), torch.device("cpu"), lr=1e-3, lr_scheduler=("tri_stage", dict(warmup_steps=10, hold_steps=10, decay_steps=10)))

    t = make(0)
    t.optimizer.exp_avg.normal_()
    t.optimizer.exp_avg_sq.uniform_()
    t.optimizer.step_count = t.num_updates = 7
    t.lr_scheduler.step_update(7)
    t.model.running.fill_(2.5)
    sd = t.state_dict({"train_iterator": {"epoch": 2, "iterations_in_epoch": 5}})
    assert {"args", "cfg", "model", "criterion", "optimizer_history", "task_state", "extra_state", "last_optimizer_state"} <= set(sd)
    assert sd["optimizer_history"][-1] == {"criterion_name": "Crit", "optimizer_name": "FlatAdam", "lr_scheduler_state": {}, "num_updates": 7}
    los = sd["last_optimizer_state"]
    assert list(los["state"]) == [0, 1, 2, 3] and los["state"][0]["exp_avg"].shape == (16, 8) and los["state"][3]["step"] == 7
    assert los["param_groups"][0]["params"] == [0, 1, 2, 3] and los["param_groups"][0]["betas"] == (0.9, 0.98)
    # the same layout torch's own Adam produces for this module
    ref = torch.optim.Adam(M().parameters())
    assert set(ref.state_dict()["param_groups"][0]) >= {"lr", "betas", "eps", "weight_decay", "amsgrad", "params"}
    path = str(tmp_path / "ck.pt")
    t.save_checkpoint(path, {"train_iterator": {"epoch": 2, "iterations_in_epoch": 5}})
    t2 = make(1)
    assert not torch.equal(t2.model.a.weight, t.model.a.weight)
    extra = t2.load_checkpoint(path)
    assert extra["train_iterator"] == {"epoch": 2, "iterations_in_epoch": 5}
    a, b = t._optimizer_state()["state"], t2._optimizer_state()["state"]  # (alignment padding between parameters is not state)
    assert torch.equal(t2.flat.p32, t.flat.p32) and torch.equal(t2.model.running, t.model.running)
    assert all(torch.equal(a[i][k], b[i][k]) for i in a for k in ("exp_avg", "exp_avg_sq")) and float(b[2]["exp_avg"].abs().sum()) > 0
    assert t2.model.a.weight.data_ptr() == t2.flat.p32.data_ptr() + 4 * t2.flat.offsets[id(t2.model.a.weight)]  # still views
    assert (t2.num_updates, t2.optimizer.step_count, t2.model.n) == (7, 7, 7) and t2.get_lr() == pytest.approx(t.get_lr())
    t3 = make(2)
    t3.load_checkpoint(path, reset_optimizer=True)
    assert t3.num_updates == 0 and float(t3.optimizer.exp_avg.abs().sum()) == 0.0 and torch.equal(t3.flat.p32, t.flat.p32)
    assert make(3).load_checkpoint(str(tmp_path / "absent.pt")) is None
    bad = dict(sd)
    bad["model"] = {k: v for k, v in sd["model"].items() if k != "b.bias"}
    torch.save(bad, str(tmp_path / "bad.pt"))
    with pytest.raises(RuntimeError, match="does not match the model"):
        make(4).load_checkpoint(str(tmp_path / "bad.pt"))


_REF_RECIPES = "/root/reference/examples/asr_librispeech/config"


@pytest.mark.skipif(not os.path.isdir(_REF_RECIPES), reason="the reference's recipe YAMLs are only present in the build container")
def test_every_asr_recipe_of_the_reference_builds(tmp_path):
    """Each ASR recipe YAML of the reference, read as it is, yields task + model + criterion + optimizer + schedule through the
    training entry point's builders (what `fairseq-hydra-train --config-name <recipe>` resolves through hydra)."""
    import glob

    from espresso_amd import speech_train as st
    from espresso_amd.config import load_config
    from espresso_amd.trainer import Trainer

    (tmp_path / "dict.txt").write_text("".join(f"u{i} 1\n" for i in range(5000)))
    want = {  # recipe -> (model, criterion, dictionary size incl. specials [+ <s> as blank], schedule)
        "conformer_librispeech": ("speech_transformer_base", "label_smoothed_cross_entropy_v2", 5003, "TriStageLRSchedule"),
        "conformer_transducer_librispeech": ("speech_transformer_transducer_base", "transducer_loss", 5004, "NoamSchedule"),
        "lstm_librispeech": ("speech_lstm", "label_smoothed_cross_entropy_v2", 5003, "ReduceLROnPlateauLRScheduleV2"),
        "lstm_librispeech_specaug": ("speech_lstm", "label_smoothed_cross_entropy_v2", 5003, "TriStageLRSchedule"),
        "transformer_ctc_librispeech": ("speech_transformer_encoder_model", "ctc_loss", 5004, "NoamSchedule"),
        "transformer_librispeech": ("speech_transformer_base", "label_smoothed_cross_entropy_v2", 5003, "TriStageLRSchedule"),
        "transformer_librispeech_specaug": ("speech_transformer_base", "label_smoothed_cross_entropy_v2", 5003, "TriStageLRSchedule"),
        "transformer_transducer_librispeech": ("speech_transformer_transducer_base", "transducer_loss", 5004, "PolynomialDecayV2LRSchedule"),
        "lstm_lm_librispeech": ("lstm_lm_espresso", "cross_entropy", 5003, "ReduceLROnPlateauLRSchedule"),
    }
    seen = set()
    for y in sorted(glob.glob(os.path.join(_REF_RECIPES, "*.yaml"))):
        name = os.path.basename(y)[:-5]
        if name not in want:
            continue
        cfg = load_config(y, [f"task.data={tmp_path}", f"task.dict={tmp_path}/dict.txt", "bpe.sentencepiece_model=none"])
        cfg["bpe"] = {}
        task = st.build_task(cfg)
        model, crit = st.build_model(cfg, task), st.build_criterion(cfg, task)
        tr = Trainer.from_cfg(cfg, task, model, crit, torch.device("cpu"))
        lrs = [tr.lr_scheduler.step_update(n) for n in (1, 1000, 100000)]
        got = (cfg["model"]["_name"], cfg["criterion"]["_name"], len(task.target_dictionary), type(tr.lr_scheduler).__name__)
        assert got == want[name], (name, got)
        assert all(isinstance(v, float) and 0.0 < v < 1.0 for v in lrs), (name, lrs)
        assert type(task).__name__ == ("LanguageModelingForASRTask" if name == "lstm_lm_librispeech" else "SpeechRecognitionEspressoTask")
        assert (task.blank_symbol == "<s>") == (got[1] in ("ctc_loss", "transducer_loss")), name
        if name != "lstm_lm_librispeech":  # input feeding everywhere but CTC (the LSTM recipes leave it to the task's default: true)
            assert task.cfg.autoregressive == (got[1] != "ctc_loss"), name
        assert tr.clip_norm == float(cfg["optimization"]["clip_norm"]) and tr.optimizer.eps == 1e-8, name
        seen.add(name)
        del tr, model
    assert seen == set(want)
    # the headline recipe: 12 x 512 encoder; `model.encoder.layer_type=conformer` is the command-line override of run_torchaudio.sh
    cfg = load_config(os.path.join(_REF_RECIPES, "transformer_ctc_librispeech.yaml"),
                      [f"task.data={tmp_path}", f"task.dict={tmp_path}/dict.txt", "bpe.sentencepiece_model=none", "model.encoder.layer_type=conformer"])
    cfg["bpe"] = {}
    task = st.build_task(cfg)
    n = sum(p.numel() for p in st.build_model(cfg, task).parameters())
    assert 79e6 < n < 81e6, n


def test_transducer_beam_history_pool_matches_dense_histories():
    """The slot-pool history of the transducer beam search (`_Pool` / `_Hist`) against the dense `[n][L][H]` tensor semantics it
    replaces (index_select / pad / cat / write-at-lens-1 / read-at-lens-1 of espresso/tools/transducer_utils.py:62-102, 386-417,
    492-637), over a random sequence of those operations."""
    import torch.nn.functional as TF

    from espresso_amd.tools.transducer_beam_search_decoder import _Hist, _Pool

    g = torch.Generator().manual_seed(0)
    H = 8
    pool = _Pool(H, torch.device("cpu"), cap=4)  # tiny capacity: growth is exercised
    hist = _Hist(pool, torch.zeros(1, 1, dtype=torch.long))
    dense = torch.zeros(1, 1, H, dtype=torch.bfloat16)
    lens = torch.ones(1, dtype=torch.long)

    def mat(h, lens):  # dense view of the positions that have been written (0 .. lens-1)
        out = torch.zeros(h.slots.shape[0], h.slots.shape[1], H, dtype=torch.bfloat16)
        for i in range(h.slots.shape[0]):
            for k in range(int(lens[i])):
                out[i, k] = h.at(i, k)
        return out

    def mask(d, lens):
        d = d.clone()
        for i in range(d.shape[0]):
            d[i, int(lens[i]):] = 0
        return d

    rows = torch.randn(1, H, generator=g).to(torch.bfloat16)
    hist.put(lens, rows)
    dense[torch.arange(1), lens - 1] = rows
    saved = []
    for step in range(60):
        op = int(torch.randint(0, 4, (1,), generator=g))
        n = hist.slots.shape[0]
        if op == 0:  # select (with repetition, like the k-expansion)
            idx = torch.randint(0, n, (int(torch.randint(1, 7, (1,), generator=g)),), generator=g)
            hist, dense, lens = hist.select(idx), dense.index_select(0, idx), lens[idx]
        elif op == 1:  # some hypotheses grow by one token, then the predictor output is written at the new last position
            grow = torch.rand(n, generator=g) < 0.6
            if bool((lens[grow] == hist.slots.shape[1]).any()):
                hist = hist.padded(hist.slots.shape[1] + 1)
                dense = TF.pad(dense, (0, 0, 0, 1))
            lens = lens + grow.long()
            keep = grow.nonzero().view(-1)
            if keep.numel():
                sub_h, sub_d, sub_l = hist.select(keep), dense.index_select(0, keep), lens[keep]
                rows = torch.randn(keep.numel(), H, generator=g).to(torch.bfloat16)
                sub_h.put(sub_l, rows)
                sub_d[torch.arange(keep.numel()), sub_l - 1] = rows
                saved.append((sub_h, sub_d, sub_l))
                hist, dense, lens = sub_h, sub_d, sub_l
        elif op == 2 and saved:  # combine with an earlier batch (pad the position axis, concatenate)
            oh, od, ol = saved[int(torch.randint(0, len(saved), (1,), generator=g))]
            L = max(hist.slots.shape[1], oh.slots.shape[1])
            hist = _Hist(pool, torch.cat((hist.padded(L).slots, oh.padded(L).slots)))
            dense = torch.cat((TF.pad(dense, (0, 0, 0, L - dense.shape[1])), TF.pad(od, (0, 0, 0, L - od.shape[1]))))
            lens = torch.cat((lens, ol))
        else:  # read the newest outputs
            assert torch.equal(hist.last(lens), dense[torch.arange(dense.shape[0]), lens - 1])
        assert hist.slots.shape[:2] == dense.shape[:2]
        assert torch.equal(mat(hist, lens), mask(dense, lens)), step
    ri, ci = [0, hist.slots.shape[0] - 1], [0, int(lens[-1]) - 1]
    assert torch.equal(hist.rows(ri, ci), dense[ri, ci])
    assert pool.buf.shape[0] > 4 and pool.n <= pool.buf.shape[0]


def test_recognize_cli_rebuilds_model_from_checkpoint_cfg(tmp_path):
    """`speech_recognize --path checkpoint_best.pt` needs no --model-config: name and `model:` block come from the checkpoint's
    cfg (what speech_train / the reference's trainer store); explicit arguments win; a bare state_dict still needs the file."""
    import json

    from espresso_amd.speech_recognize import resolve_model_config

    block = {"_name": "speech_transformer_encoder_model", "encoder": {"embed_dim": 128, "layers": 2}, "dropout": 0.1}
    ck = {"cfg": {"model": dict(block), "task": {"_name": "speech_recognition_espresso"}}, "model": {}}
    name, got = resolve_model_config(None, None, ck)
    assert name == "speech_transformer_encoder_model" and got == block and got is not ck["cfg"]["model"]
    assert resolve_model_config("speech_transformer_base", None, ck)[0] == "speech_transformer_base"
    y = tmp_path / "m.yaml"
    y.write_text("encoder:\n  embed_dim: 256\nlayernorm_embedding: true\n")
    name, got = resolve_model_config(None, str(y), ck)
    assert name == "speech_transformer_encoder_model" and got == {"encoder": {"embed_dim": 256}, "layernorm_embedding": True}
    j = tmp_path / "recipe.json"
    j.write_text(json.dumps({"model": {"_name": "speech_transformer_transducer_base", "joint_dim": 64}, "task": {}}))
    assert resolve_model_config(None, str(j), {})[0] == "speech_transformer_transducer_base"
    assert resolve_model_config(None, str(y), {"w": torch.zeros(1)})[0] == "speech_transformer_base"
    with pytest.raises(ValueError, match="model-config"):
        resolve_model_config(None, None, {"w": torch.zeros(1)})


def test_language_model_data_follows_fairseq_monolingual_dataset(tmp_path):
    """Token files (fairseq mmap format written and read back; raw text), `eos` / `none` sample-break modes, the shifted source /
    future target of fairseq's MonolingualDataset, right-padded collation and the task's batch plan."""
    from espresso_amd.data.lm_dataset import MMapTokenFile, MonolingualDataset, load_token_file
    from espresso_amd.tasks.language_modeling_for_asr import LanguageModelingForASRConfig, LanguageModelingForASRTask

    d = AsrDictionary.from_symbols([f"w{i}" for i in range(20)], enable_bos=False, add_space=False)
    (tmp_path / "dict.txt").write_text("".join(f"w{i} 1\n" for i in range(20)))
    lines = ["w1 w2 w3", "w4", "w5 w6 w7 w8 w9", "w10 w11"]
    (tmp_path / "valid").write_text("\n".join(lines) + "\n")
    raw = load_token_file(str(tmp_path / "valid"), d)
    items = [raw[i] for i in range(len(raw))]
    assert [len(x) for x in items] == [4, 2, 6, 3] and all(int(x[-1]) == d.eos() for x in items)
    MMapTokenFile.write(str(tmp_path / "train"), items, dtype=np.int32)
    mm = load_token_file(str(tmp_path / "train"), d)
    assert isinstance(mm, MMapTokenFile) and len(mm) == 4 and mm.sizes.tolist() == [4, 2, 6, 3]
    assert all(np.array_equal(mm[i], items[i]) and mm[i].dtype == np.int64 for i in range(4))
    idx = open(tmp_path / "train.idx", "rb").read()
    assert idx[:9] == b"MMIDIDX\x00\x00" and idx[17] == 4 and len(idx) == 9 + 8 + 1 + 8 + 4 * 4 + 8 * 4  # dtype code 4 = int32
    assert np.frombuffer(idx, dtype=np.int64, count=4, offset=9 + 8 + 1 + 8 + 16).tolist() == [0, 16, 24, 48]  # byte pointers

    ds = MonolingualDataset(mm, d, "eos", shuffle=False)
    assert len(ds) == 4 and ds.sizes.tolist() == [4, 2, 6, 3]
    it = ds[2]
    assert it["target"].tolist() == items[2].tolist() and it["source"].tolist() == [d.eos()] + items[2][:-1].tolist()
    assert ds[0]["source"][0] == d.eos()
    assert ds.ordered_indices().tolist() == [1, 3, 0, 2]  # stable by length
    batch = ds.collater([ds[0], ds[2]])
    assert batch["ntokens"] == 10 and batch["nsentences"] == 2 and batch["net_input"]["src_lengths"].tolist() == [4, 6]
    assert batch["target"][0].tolist() == items[0].tolist() + [d.pad()] * 2 and batch["net_input"]["src_tokens"][0, 4:].tolist() == [d.pad()] * 2
    blocks = MonolingualDataset(mm, d, "none", tokens_per_sample=4, shuffle=False)
    flat = np.concatenate(items)
    assert blocks.sizes.tolist() == [4, 4, 4, 3] and blocks[1]["target"].tolist() == flat[4:8].tolist()
    assert blocks[1]["source"].tolist() == flat[3:7].tolist() and blocks[0]["source"][0] == d.eos()
    keep, dropped = ds.filter_indices_by_size(np.arange(4), 4)
    assert keep.tolist() == [0, 1, 3] and dropped == [2]

    task = LanguageModelingForASRTask.setup_task(LanguageModelingForASRConfig(data=str(tmp_path), dict=str(tmp_path / "dict.txt"),
                                                                              sample_break_mode="eos", tokens_per_sample=5))
    tds = task.load_dataset("train")
    plan = task.get_batches(tds, max_tokens=8, max_sentences=None, max_positions=task.max_positions(), seed=1, epoch=1, shuffle=False)
    assert sorted(int(i) for b in plan for i in b) == [0, 1, 3]  # the 6-token sentence exceeds tokens_per_sample
    assert all(len(b) * int(tds.sizes[b].max()) <= 8 for b in plan)
    two = task.get_batches(tds, max_tokens=8, max_positions=task.max_positions(), seed=1, epoch=1, num_shards=2, shard_id=1, shuffle=False)
    assert len(two) == (len(plan) + 1) // 2
    assert registry.CRITERION_REGISTRY["cross_entropy"](task).sentence_avg is False


def test_training_loop_update_groups_and_early_stop():
    """Micro-batch grouping of the epoch loop (whole chunks of update_freq, a shorter last one, aligned after a mid-epoch restart)
    and the `patience` rule of fairseq_cli/train.py:205-233."""
    from espresso_amd.speech_train import EarlyStop, update_group_sizes

    assert update_group_sizes(5, 0, 2) == [2, 2, 1] and update_group_sizes(5, 4, 2) == [1] and update_group_sizes(5, 5, 2) == []
    assert update_group_sizes(6, 0, 3) == [3, 3] and update_group_sizes(6, 3, 3) == [3] and update_group_sizes(4, 0, 1) == [1, 1, 1, 1]
    for n, uf in ((7, 3), (8, 2), (5, 4)):  # a restart at any update boundary continues with the uninterrupted run's chunks
        full = update_group_sizes(n, 0, uf)
        done = 0
        for k, size in enumerate(full):
            assert update_group_sizes(n, done, uf) == full[k:]
            done += size
    cfg = {"checkpoint": {"patience": 2, "maximize_best_checkpoint_metric": False}}
    es = EarlyStop(cfg)
    assert [es(v) for v in (5.0, 4.0, 4.5, 3.9, 4.0, 4.1)] == [False, False, False, False, False, True]
    assert es(None) is False
    up = EarlyStop({"checkpoint": {"patience": 1, "maximize_best_checkpoint_metric": True}})
    assert [up(v) for v in (0.5, 0.6, 0.6)] == [False, False, True]
    off = EarlyStop({"checkpoint": {"patience": -1, "maximize_best_checkpoint_metric": False}})
    assert [off(v) for v in (1.0, 2.0, 3.0)] == [False, False, False]


def test_legacy_command_lines_of_the_wsj_and_swbd_recipes(tmp_path):
    """`fairseq_cli/train.py DATA --task … --arch … --flag value` (examples/asr_wsj/run.sh:292-303, asr_swbd/run.sh:293-303,
    asr_wsj/run.sh:197-207) through the legacy front end: same grouped configuration, task / model / criterion / schedule build."""
    from espresso_amd import speech_train as st
    from espresso_amd.config import from_legacy_argv, is_legacy_argv, literal
    from espresso_amd.trainer import Trainer

    (tmp_path / "dict.txt").write_text("".join(f"u{i} 1\n" for i in range(50)) + "<space> 1\n")
    (tmp_path / "nlsyms.txt").write_text("")
    wsj = ["data", "--task", "speech_recognition_espresso", "--seed", "1", "--log-interval", "400", "--log-format", "simple",
           "--print-training-sample-interval", "1000", "--num-workers", "6", "--data-buffer-size", "0", "--max-tokens", "24000",
           "--batch-size", "32", "--curriculum", "2", "--empty-cache-freq", "2", "--valid-subset", "valid", "--batch-size-valid", "64",
           "--ddp-backend", "legacy_ddp", "--update-freq", "2", "--distributed-world-size", "1", "--required-batch-size-multiple", "8",
           "--optimizer", "adam", "--lr", "0.001", "--weight-decay", "0.0", "--save-dir", "exp/lstm", "--restore-file", "checkpoint_last.pt",
           "--save-interval-updates", "400", "--keep-interval-updates", "5", "--keep-last-epochs", "5", "--validate-interval", "1",
           "--best-checkpoint-metric", "wer", "--criterion", "label_smoothed_cross_entropy_v2", "--label-smoothing", "0.05",
           "--smoothing-type", "temporal", "--dict", str(tmp_path / "dict.txt"), "--bpe", "characters_asr", "--non-lang-syms",
           str(tmp_path / "nlsyms.txt"), "--max-source-positions", "3000", "--max-target-positions", "300",
           "--arch", "speech_conv_lstm_wsj", "--max-epoch", "35", "--lr-scheduler", "reduce_lr_on_plateau_v2", "--lr-shrink", "0.5",
           "--start-reduce-lr-epoch", "11", "--scheduled-sampling-probs", "0.5", "--start-scheduled-sampling-epoch", "6"]
    assert is_legacy_argv(wsj) and not is_legacy_argv(["--config", "r.yaml", "task.data=x"])
    cfg = from_legacy_argv(wsj)
    assert cfg["task"]["data"] == "data" and cfg["task"]["_name"] == "speech_recognition_espresso" and cfg["task"]["max_source_positions"] == 3000
    assert cfg["dataset"]["max_tokens"] == 24000 and cfg["dataset"]["curriculum"] == 2 and cfg["dataset"]["required_batch_size_multiple"] == 8
    assert cfg["optimization"]["update_freq"] == [2] and cfg["optimization"]["lr"] == [0.001] and cfg["optimization"]["clip_norm"] == 0.0
    assert cfg["checkpoint"]["best_checkpoint_metric"] == "wer" and cfg["checkpoint"]["keep_interval_updates"] == 5
    assert cfg["criterion"] == {"_name": "label_smoothed_cross_entropy_v2", "print_training_sample_interval": 1000,
                                "label_smoothing": 0.05, "smoothing_type": "temporal"}
    assert cfg["lr_scheduler"] == {"_name": "reduce_lr_on_plateau_v2", "lr_shrink": 0.5, "start_reduce_lr_epoch": 11}
    assert cfg["model"] == {"_name": "speech_lstm", "arch": "speech_conv_lstm_wsj", "scheduled_sampling_probs": [0.5],
                            "start_scheduled_sampling_epoch": 6}
    assert cfg["bpe"]["_name"] == "characters_asr" and literal(cfg["optimizer"]["adam_betas"]) == (0.9, 0.999)
    task = st.build_task(cfg)
    assert task.cfg.autoregressive and task.blank_symbol is None and task.cfg.non_lang_syms.endswith("nlsyms.txt")
    model, crit = st.build_model(cfg, task), st.build_criterion(cfg, task)
    tr = Trainer.from_cfg(cfg, task, model, crit, torch.device("cpu"))
    assert type(tr.lr_scheduler).__name__ == "ReduceLROnPlateauLRScheduleV2" and tr.clip_norm == 0.0 and tr.get_lr() == 0.001
    assert crit.smoothing_type == "temporal" and 5e6 < sum(p.numel() for p in model.parameters()) < 4e7

    swbd = ["data", "--task", "speech_recognition_espresso", "--max-tokens", "26000", "--batch-size", "48", "--optimizer", "adam", "--lr", "0.001",
            "--clip-norm", "2.0", "--criterion", "label_smoothed_cross_entropy_v2", "--label-smoothing", "0.1", "--smoothing-type", "uniform",
            "--dict", str(tmp_path / "dict.txt"), "--arch", "speech_conv_lstm_swbd", "--max-epoch", "100", "--lr-scheduler", "tri_stage",
            "--warmup-steps", "1000", "--hold-steps", "180000", "--decay-steps", "360000", "--encoder-rnn-hidden-size", "1024",
            "--encoder-rnn-layers", "5", "--decoder-embed-dim", "512", "--decoder-hidden-size", "1024", "--decoder-out-embed-dim", "3072",
            "--attention-dim", "512", "--dropout", "0.4", "--specaugment-config",
            "{'time_warp_W': 0, 'freq_mask_F': 18, 'time_mask_T': 70, 'freq_mask_N': 2, 'time_mask_N': 2, 'time_mask_p': 0.2}"]
    c2 = from_legacy_argv(swbd)
    assert c2["model"]["encoder_rnn_layers"] == 5 and c2["model"]["dropout"] == 0.4 and c2["optimization"]["clip_norm"] == 2.0
    assert c2["task"]["specaugment_config"]["time_mask_T"] == 70 and c2["lr_scheduler"]["hold_steps"] == 180000
    assert type(Trainer.from_cfg(c2, st.build_task(c2), torch.nn.Linear(2, 2), object(), torch.device("cpu")).lr_scheduler).__name__ == "TriStageLRSchedule"

    lm = ["lmdata", "--task", "language_modeling_for_asr", "--dict", str(tmp_path / "dict.txt"), "--max-tokens", "25600", "--batch-size", "128",
          "--max-epoch", "25", "--optimizer", "adam", "--lr", "0.001", "--weight-decay", "5e-06", "--lr-scheduler", "reduce_lr_on_plateau",
          "--lr-shrink", "0.5", "--arch", "lstm_lm_wsj", "--criterion", "cross_entropy", "--sample-break-mode", "eos"]
    c3 = from_legacy_argv(lm)
    assert c3["task"] == {"_name": "language_modeling_for_asr", "dict": str(tmp_path / "dict.txt"), "sample_break_mode": "eos", "data": "lmdata"}
    assert c3["optimizer"]["weight_decay"] == 5e-6 and c3["model"] == {"_name": "lstm_lm_espresso", "arch": "lstm_lm_wsj"}
    t3 = st.build_task(c3)
    m3 = st.build_model(c3, t3)
    assert type(t3).__name__ == "LanguageModelingForASRTask" and m3.decoder.hidden_size == 650
    # the argparse Transformer presets resolve to the legacy model (absolute encoder positions, no embedding LayerNorm) ...
    c4 = from_legacy_argv(["data", "--arch", "speech_transformer_wsj", "--dict", str(tmp_path / "dict.txt"), "--encoder-layers", "2",
                           "--decoder-layers", "1", "--criterion", "label_smoothed_cross_entropy_v2"])
    assert c4["model"] == {"_name": "speech_transformer", "arch": "speech_transformer_wsj", "encoder_layers": 2, "decoder_layers": 1}
    m4 = st.build_model(c4, st.build_task(c4))
    assert type(m4).__name__ == "SpeechTransformerModel" and m4.encoder.abs_positions and m4.encoder.layernorm_embedding is None
    assert m4.cfg.encoder.embed_dim == 256 and m4.cfg.dropout == 0.2 and len(m4.encoder.layers) == 2
    # ... and an architecture nobody registered is refused by name
    with pytest.raises(NotImplementedError, match="recipe YAMLs"):
        from_legacy_argv(["data", "--arch", "speech_tdnn_wsj"])


def test_language_model_data_path_matches_the_reference_fixture(tmp_path, golden_dir):
    """tests/golden/ref_lm_data_tiny.npz (oracle/gen_golden.py lmdata): fairseq's MMapIndexedDataset read token files written by
    THIS repo's writer, its TokenBlockDataset + MonolingualDataset built the samples (`eos` / `none`), `ordered_indices` ran
    under numpy_seed(seed) and the reference's own Cython `batch_by_size` planned the batches.  Same files through this repo's
    reader / dataset / task: identical samples, order, batch plan and collation."""
    from espresso_amd.data.lm_dataset import MMapTokenFile
    from espresso_amd.tasks.language_modeling_for_asr import LanguageModelingForASRConfig, LanguageModelingForASRTask

    g = np.load(os.path.join(golden_dir, "ref_lm_data_tiny.npz"))
    d = AsrDictionary.from_symbols([f"w{i}" for i in range(30)], enable_bos=False, add_space=False)
    assert (d.pad(), d.eos(), len(d)) == (int(g["pad"]), int(g["eos"]), int(g["V"]))
    bounds = np.concatenate(([0], np.cumsum(g["sizes"])))
    sents = [g["flat"][bounds[i]:bounds[i + 1]] for i in range(int(g["n_sent"]))]
    MMapTokenFile.write(str(tmp_path / "train"), sents, dtype=np.int32)
    for mode, tps in (("eos", 8), ("none", 7)):
        task = LanguageModelingForASRTask.setup_task(
            LanguageModelingForASRConfig(data=str(tmp_path), sample_break_mode=mode, tokens_per_sample=tps), dictionary=d)
        ds = task.load_dataset("train")
        assert len(ds) == int(g[f"{mode}::n"]) and ds.sizes.tolist() == g[f"{mode}::sizes"].tolist()
        items = [ds[i] for i in range(len(ds))]
        assert [len(it["source"]) for it in items] == g[f"{mode}::len"].tolist()
        assert np.concatenate([it["source"].numpy() for it in items]).tolist() == g[f"{mode}::src"].tolist()
        assert np.concatenate([it["target"].numpy() for it in items]).tolist() == g[f"{mode}::tgt"].tolist()
        for seed in (1, 5):
            with data_utils.numpy_seed(seed):
                assert ds.ordered_indices().tolist() == g[f"{mode}::order::{seed}"].tolist()
            plan = task.get_batches(ds, max_tokens=40, max_sentences=6, max_positions=task.max_positions(), seed=seed, epoch=1,
                                    shuffle=False, bsz_mult=4)
            assert [len(b) for b in plan] == g[f"{mode}::batch_sizes::{seed}"].tolist()
            assert np.concatenate(plan).tolist() == g[f"{mode}::batches::{seed}"].tolist()
        b = ds.collater([ds[i] for i in (0, 3, 2)])
        assert b["net_input"]["src_tokens"].tolist() == g[f"{mode}::collate::src"].tolist()
        assert b["target"].tolist() == g[f"{mode}::collate::tgt"].tolist()
        assert b["net_input"]["src_lengths"].tolist() == g[f"{mode}::collate::lens"].tolist() and b["ntokens"] == int(g[f"{mode}::collate::ntokens"])


def test_batch_by_size_matches_the_reference_cython_planner(golden_dir):
    """tests/golden/ref_batch_by_size.npz: 24 cases planned by the reference's own Cython `batch_by_size_vec` / `batch_by_size_fn`
    (fairseq/data/data_utils_fast.pyx, compiled from where it lies by oracle/build_ref_cython.py) — recipe settings, batch-size
    multiples, no sentence cap, no token cap, tiny budgets.  The host planner of this repo gives the same batches."""
    g = np.load(os.path.join(golden_dir, "ref_batch_by_size.npz"))
    n = int(g["n_cases"])
    assert n >= 20
    for c in range(n):
        sizes, order = g[f"{c}::sizes"], g[f"{c}::order"]
        mt, ms, mult = (int(v) for v in g[f"{c}::args"])
        got = data_utils.batch_by_size(order, sizes[order], max_tokens=None if mt < 0 else mt, max_sentences=None if ms < 0 else ms,
                                       bsz_mult=mult)
        assert [len(b) for b in got] == g[f"{c}::lens"].tolist(), (c, mt, ms, mult)
        assert np.concatenate(got).tolist() == g[f"{c}::flat"].tolist(), (c, mt, ms, mult)


def test_per_rank_epoch_batches_match_the_reference_epoch_iterator(golden_dir):
    """tests/golden/ref_epoch_batches.npz: what the reference's EpochBatchIterator hands each of 1 / 4 data-parallel ranks in
    epochs 1 and 2 (batches shuffled with `seed + epoch`, every num_shards-th batch per rank, empty batches as fill).
    `task.get_batches` — the plan `speech_train` iterates — gives every rank the same batches in the same order."""
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoTask

    g = np.load(os.path.join(golden_dir, "ref_epoch_batches.npz"))
    sizes = g["sizes"]
    max_tokens, max_sentences, mult, seed = (int(v) for v in g["args"])

    class DS:
        def ordered_indices(self):
            return np.argsort(sizes, kind="mergesort")

        def num_tokens_vec(self, indices):
            return sizes[indices]

    for shards in (1, 4):
        for shard in range(shards):
            for epoch in (1, 2):
                plan = SpeechRecognitionEspressoTask.get_batches(None, DS(), max_tokens=max_tokens, max_sentences=max_sentences, seed=seed,
                                                                 epoch=epoch, num_shards=shards, shard_id=shard, shuffle=True, bsz_mult=mult)
                assert [len(b) for b in plan] == g[f"{shards}::{shard}::{epoch}::lens"].tolist(), (shards, shard, epoch)
                flat = [int(i) for b in plan for i in b]
                assert flat == g[f"{shards}::{shard}::{epoch}::flat"].tolist(), (shards, shard, epoch)
    assert 0 in g["4::3::1::lens"].tolist()  # 26 batches over 4 ranks: the short ranks end with an empty fill batch


def test_checkpoint_rules_match_the_reference_listing_by_listing(tmp_path, golden_dir):
    """tests/golden/ref_checkpoint_rules.json: directory listings after each call of the reference's own
    `checkpoint_utils.save_checkpoint` over a scripted run (mid-epoch saves, epoch ends, improving / worsening / tied scores,
    keep_interval_updates / keep_last_epochs / keep_best_checkpoints, minimising and maximising).  CheckpointSaver leaves the
    same files after every call, including the tie-breaking digit of the `checkpoint.best_*` names."""
    import json

    from espresso_amd.checkpoint_utils import CheckpointSaver
    from espresso_amd.config import DEFAULTS

    g = json.load(open(os.path.join(golden_dir, "ref_checkpoint_rules.json")))
    for tag, maximize in (("min", False), ("max", True)):
        cfg = dict(DEFAULTS["checkpoint"], save_dir=str(tmp_path / tag), save_interval_updates=2, keep_interval_updates=2, keep_last_epochs=2,
                   keep_best_checkpoints=2, best_checkpoint_metric="wer", maximize_best_checkpoint_metric=maximize)
        saver, tr = CheckpointSaver(cfg), _FakeTrainer()
        for (epoch, end, n, val), want in zip(g["script"], g[tag]["listings"]):
            tr.num_updates = n
            saver.save(tr, epoch, end, {"epoch": epoch}, val)
            assert sorted(os.listdir(cfg["save_dir"])) == want, (tag, epoch, end, n, val)
        assert saver.best == g[tag]["best"]


def test_lr_schedules_match_the_reference_trajectories(golden_dir):
    """tests/golden/ref_lr_schedules.json: learning rates the reference's own scheduler classes produced (noam, tri_stage with
    steps and with phase ratios, polynomial_decay_v2, reduce_lr_on_plateau_v2 with warm-up / start epoch / floor, fairseq's
    reduce_lr_on_plateau, a maximising plateau schedule) over scripted update counts and validation scores."""
    import json

    from espresso_amd.optim import lr_schedulers  # noqa: F401

    class Opt:
        lr = 1.0

        def set_lr(self, lr):
            self.lr = lr

        def get_lr(self):
            return self.lr

    g = json.load(open(os.path.join(golden_dir, "ref_lr_schedules.json")))
    for tag in ("noam", "noam_small", "tri_stage", "tri_stage_ratio", "polynomial_decay_v2"):
        cfg = dict(g[tag]["cfg"])
        lr0 = cfg.pop("lr")[0]
        cfg.pop("force_anneal", None)
        if cfg.get("phase_ratio") is None:
            cfg.pop("phase_ratio", None)
        name = {"noam_small": "noam", "tri_stage_ratio": "tri_stage"}.get(tag, tag)
        opt = Opt()
        sch = registry.LR_SCHEDULER_REGISTRY[name](opt, lr=lr0, **cfg)
        got = []
        for n in g["updates"]:
            sch.step_update(n)
            got.append(opt.get_lr())
        assert got == pytest.approx(g[tag]["lr"], rel=1e-9, abs=1e-15), tag
    for tag in ("reduce_lr_on_plateau_v2", "reduce_lr_on_plateau", "reduce_lr_on_plateau_v2_max"):
        cfg = dict(g[tag]["cfg"])
        lr0 = cfg.pop("lr")[0]
        opt = Opt()
        sch = registry.LR_SCHEDULER_REGISTRY["reduce_lr_on_plateau" if tag == "reduce_lr_on_plateau" else "reduce_lr_on_plateau_v2"](opt, lr=lr0, **cfg)
        warm = []
        for n in g[tag]["warm_updates"]:
            sch.step_update(n)
            warm.append(opt.get_lr())
        assert warm == pytest.approx(g[tag]["warm_lr"], rel=1e-9), tag
        epochs = []
        for e, v in enumerate(g[tag]["scores"], start=1):
            sch.step(e, -v if cfg["maximize_best_checkpoint_metric"] else v)
            sch.step_update(2000 + e)
            epochs.append(opt.get_lr())
        assert epochs == pytest.approx(g[tag]["epoch_lr"], rel=1e-9), (tag, epochs, g[tag]["epoch_lr"])


def test_wer_scorer_matches_the_reference_scorer(tmp_path, golden_dir):
    """tests/golden/ref_wer_scorer.json: the reference's Scorer (espresso/tools/wer.py) on scripted character-unit utterances with
    <space> word boundaries, non-language symbols (dropped before scoring) and a sed-style WER output filter: the 4-tuples of
    wer() / cer() and the running totals after every utterance."""
    import json

    from espresso_amd.tools.wer import Scorer

    g = json.load(open(os.path.join(golden_dir, "ref_wer_scorer.json")))
    (tmp_path / "dict.txt").write_text("".join(f"{c} 1\n" for c in "abcdefghijklmnopqrstuvwxyz'") + "<space> 1\n<noise> 1\n<laugh> 1\n")
    (tmp_path / "nlsyms.txt").write_text("<noise>\n<laugh>\n")
    (tmp_path / "filter").write_text("#!/bin/sed -f\ns/uh //g\ns: um::g\n")
    for tag, filt in (("plain", None), ("filtered", str(tmp_path / "filter"))):
        d = AsrDictionary.load(str(tmp_path / "dict.txt"), f_non_lang_syms=str(tmp_path / "nlsyms.txt"))
        d.build_bpe("characters_asr")
        sc = Scorer(d, wer_output_filter=filt)
        for (utt, ref, hyp), want in zip(g["pairs"], g[tag]):
            sc.add_evaluation(utt, ref, hyp)
            assert list(sc.wer()) == pytest.approx(want["wer"]) and list(sc.cer()) == pytest.approx(want["cer"]), (tag, utt)
            assert (sc.tot_word_error(), sc.tot_word_count(), sc.tot_char_error(), sc.tot_char_count()) == (
                want["word_error"], want["word_count"], want["char_error"], want["char_count"]), (tag, utt)
    assert g["plain"][-1]["word_count"] != g["filtered"][-1]["word_count"]  # the filter removed words


def test_asr_dictionary_matches_the_reference_dictionary(tmp_path, golden_dir):
    """tests/golden/ref_asr_dictionary.json: the reference's AsrDictionary + `characters_asr` encoder + `tokenize` on scripted
    sentences (apostrophes, non-language symbols, repeated / leading spaces, out-of-vocabulary characters, empty text): symbol
    layout with and without `<s>`, counts, text -> pieces -> ids -> string -> text."""
    import json

    from espresso_amd.tools.utils import tokenize

    g = json.load(open(os.path.join(golden_dir, "ref_asr_dictionary.json")))
    (tmp_path / "dict.txt").write_text("".join(f"{c} {i + 3}\n" for i, c in enumerate("abcdefghijklmnopqrstuvwxyz'")) + "<space> 9\n<noise> 2\n<laugh> 1\n")
    (tmp_path / "nlsyms.txt").write_text("<noise>\n<laugh>\n")
    for tag, bos in (("bos", True), ("nobos", False)):
        want = g[tag]
        d = AsrDictionary.load(str(tmp_path / "dict.txt"), enable_bos=bos, f_non_lang_syms=str(tmp_path / "nlsyms.txt"))
        d.build_bpe("characters_asr")
        assert (len(d), d.pad(), d.eos(), d.unk(), d.space()) == (want["len"], want["pad"], want["eos"], want["unk"], want["space"])
        assert list(d.symbols) == want["symbols"] and [int(c) for c in d.count] == want["count"]
        if bos:
            assert d.bos() == want["bos"]
        for text, row in zip(g["texts"], want["rows"]):
            pieces = d.wordpiece_encode(text)
            assert pieces == row["pieces"], (tag, text)
            ids = d.encode_line(pieces, append_eos=True).tolist()
            assert ids == row["ids"], (tag, text)
            s = d.string(torch.tensor(ids))
            assert s == row["string"] and d.wordpiece_decode(s) == row["decoded"], (tag, text)
            assert tokenize(text, space=d.space_word, non_lang_syms=d.non_lang_syms) == row["tokenize"], (tag, text)


def test_asr_collate_matches_the_reference_collate(golden_dir):
    """tests/golden/ref_asr_collate.npz: the reference's `collate` on scripted feature samples — descending-length sort (ties in
    input order), zero frame padding, token padding, input feeding with </s> moved to the front or <s> prepended,
    pad_to_multiple, and samples without targets."""
    from espresso_amd.data.asr_dataset import collate

    g = np.load(os.path.join(golden_dir, "ref_asr_collate.npz"))
    pad, eos, bos = int(g["pad"]), int(g["eos"]), int(g["bos"])
    samples = [{"id": i, "utt_id": f"utt{i}", "source": torch.from_numpy(g[f"in::{i}::source"]), "target": torch.from_numpy(g[f"in::{i}::target"]),
                "text": f"text {i}"} for i in range(int(g["n"]))]
    cases = {"feed_eos": dict(input_feeding=True), "feed_bos": dict(input_feeding=True, maybe_bos_idx=bos),
             "no_feed": dict(input_feeding=False), "mult4": dict(input_feeding=True, pad_to_multiple=4)}
    for tag, kw in cases.items():
        b = collate(samples, pad_idx=pad, eos_idx=eos, left_pad_source=False, left_pad_target=False, **kw)
        assert b["id"].tolist() == g[f"{tag}::id"].tolist() and list(b["utt_id"]) == g[f"{tag}::utt_id"].tolist(), tag
        assert torch.equal(b["net_input"]["src_tokens"], torch.from_numpy(g[f"{tag}::src"])), tag
        assert b["net_input"]["src_lengths"].tolist() == g[f"{tag}::src_lengths"].tolist(), tag
        assert b["target"].tolist() == g[f"{tag}::target"].tolist() and b["ntokens"] == int(g[f"{tag}::ntokens"]), tag
        if f"{tag}::prev" in g.files:
            assert b["net_input"]["prev_output_tokens"].tolist() == g[f"{tag}::prev"].tolist(), tag
        else:
            assert "prev_output_tokens" not in b["net_input"], tag
    nt = [{k: v for k, v in s.items() if k not in ("target", "text")} for s in samples]
    b = collate(nt, pad_idx=pad, eos_idx=eos)
    assert b["id"].tolist() == g["notgt::id"].tolist() and b["net_input"]["src_lengths"].tolist() == g["notgt::src_lengths"].tolist()
    assert sorted(b.keys()) == g["notgt::keys"].tolist() and b["target"] is None



def test_post_accumulate_grad_hook_fires_for_a_parameter_whose_function_returned_none():
    """espresso_amd.functional._grad_sink: kernels accumulate some parameter gradients straight into `p.grad` and the autograd
    Function returns None for them; the data-parallel wrapper still learns that the gradient is complete from the
    post-accumulate-grad hook — autograd runs the parameter's AccumulateGrad node (and its hooks) even for an undefined
    gradient.  Pinned here because the bucket accounting of OverlappedDistributedDataParallel depends on it."""
    import torch

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.params = (w, b)
            return x * 2

        @staticmethod
        def backward(ctx, dy):
            ctx.params[0].grad += 1.0  # "the kernel wrote it"
            return dy * 2, None, None

    w, b = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(3))
    w.grad, b.grad = torch.zeros(3), torch.zeros(3)
    fired = []
    w.register_post_accumulate_grad_hook(lambda p: fired.append("w"))
    b.register_post_accumulate_grad_hook(lambda p: fired.append("b"))
    Fn.apply(torch.ones(3, requires_grad=True), w, b).sum().backward()
    assert sorted(fired) == ["b", "w"] and float(w.grad.sum()) == 3.0


def test_direct_gradient_route_is_gated_on_the_scope_and_the_measured_torch_capability():
    """`functional._grad_sink` hands a kernel the live `p.grad` only inside `accumulating_backward()` (what a trainer
    that owns the gradient buffers opens around `loss.backward()`), and only when this torch was MEASURED to fire post-accumulate-grad
    hooks for a parameter whose Function returned None (`_none_grad_hooks_fire`, the pin above as a runtime check).  On CPU tensors
    the sink is never used (`g.is_cuda`), so what is checked here is the gating and the scope's nesting."""
    import torch

    from espresso_amd import functional as F

    assert F._sink_scope == 0
    assert F._none_grad_hooks_fire() is True and F._none_grad_hooks_ok is True
    p = torch.nn.Parameter(torch.ones(4))
    p.grad = torch.zeros(4)
    assert F._grad_sink(p, 4) is None  # outside the scope
    with F.accumulating_backward():
        assert F._sink_scope == 1
        with F.accumulating_backward():
            assert F._sink_scope == 2
        assert F._sink_scope == 1
        assert F._grad_sink(p, 4) is None  # CPU gradient: autograd route
    assert F._sink_scope == 0
    try:
        with F.accumulating_backward():
            raise RuntimeError("backward failed")
    except RuntimeError:
        pass
    assert F._sink_scope == 0  # the scope closes on errors too


def test_trainer_skips_the_update_after_an_out_of_memory_error(capsys):
    """fairseq/trainer.py:842-857 with one worker: an out-of-memory RuntimeError raised in the forward / backward pass of a
    micro-batch is logged, accumulated gradients are dropped, `train_step` returns None (no optimizer update, no statistics) and
    the next call trains normally; any other RuntimeError propagates."""
    import torch.nn as nn

    from espresso_amd.trainer import Trainer

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Linear(8, 4)

        def forward(self, x):
            return self.a(x)

    class Task:
        def prepare_sample(self, sample, train=True):
            return sample

    class Crit:
        def __init__(self):
            self.fail = None

        def __call__(self, model, sample):
            loss = model(sample["x"]).pow(2).sum()
            if self.fail == "oom_after_backward_of_part":
                (loss * 0.5).backward()  # gradients of an earlier micro-batch are sitting in the buffer when the error arrives
                raise RuntimeError("HIP out of memory. Tried to allocate 1.50 GiB")
            if self.fail == "other":
                raise RuntimeError("shape mismatch")
            return loss, 2, {"ntokens": 5, "nsentences": 2}

    torch.manual_seed(0)
    crit = Crit()
    t = Trainer(Task(), M(), crit, torch.device("cpu"), lr=1e-3, lr_scheduler=("tri_stage", dict(warmup_steps=10, hold_steps=10, decay_steps=10)))
    sample = {"x": torch.randn(2, 8)}
    stepped = []
    t.optimizer.clip_and_step = lambda **kw: stepped.append(kw) or torch.zeros(2)  # (the fused optimizer kernel needs a GPU; not under test)
    crit.fail = "oom_after_backward_of_part"
    assert t.train_step([sample]) is None
    assert t.ooms == 1 and t.num_updates == 0 and not stepped
    assert float(t.flat.g32.abs().sum()) == 0.0 and all(float(p.grad.abs().sum()) == 0.0 for p in t.model.parameters())
    assert "ran out of memory" in capsys.readouterr().err
    crit.fail = None
    out = t.train_step([sample])
    assert out is not None and t.num_updates == 1 and len(stepped) == 1 and float(out[0]) == 2.0 and float(t.flat.g32.abs().sum()) > 0
    crit.fail = "other"
    with pytest.raises(RuntimeError, match="shape mismatch"):
        t.train_step([sample])
    # more than one rank: buckets of the update may already be in flight — the error is re-raised, with the reason
    t.world_size, crit.fail = 2, "oom_after_backward_of_part"
    with pytest.raises(RuntimeError, match="cannot be skipped consistently"):
        t.train_step([sample])
    assert t.ooms == 2
