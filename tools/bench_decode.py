"""Decode real-time factor at beam 10 (BASELINE.json metric, second half): the attention encoder-decoder of config 2/5
(conv sub-sampling + 12-layer rel-pos Transformer encoder + 6-layer decoder, V = 5004) with optional LSTM-LM shallow fusion,
batched beam search on the HIP incremental-decoding kernels.  Synthetic 16 kHz audio, random-init weights: no hypothesis
ends early, every sentence decodes to max_len = 0.08 * frames (the recipe's --max-len-a), i.e. the worst case.

Prints one JSON line: RTF = decode wall time / audio duration (front-end and encoder included)."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

VOCAB = 5004


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--beam", type=int, default=10)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--max-tokens", type=int, default=15000)
    ap.add_argument("--batch-size", type=int, default=24)
    ap.add_argument("--lm", action="store_true", help="shallow fusion with an LSTM LM (lstm_lm_librispeech: 4 x 800)")
    ap.add_argument("--wordlm", action="store_true",
                    help="look-ahead word-LM fusion (config 5): character units, 65 000-word lexicon, lstm_wordlm_wsj (3 x 1200) through "
                         "TensorizedLookaheadLanguageModel")
    ap.add_argument("--lm-weight", type=float, default=0.47)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    import espresso_amd  # noqa: F401
    from espresso_amd.data import synthetic
    from espresso_amd.data.asr_dictionary import AsrDictionary
    from espresso_amd.models.lstm_lm import LSTMLanguageModelEspresso
    from espresso_amd.models.transformer.speech_transformer_base import SpeechTransformerModelBase
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerConfig
    from espresso_amd.sequence_generator import SequenceGenerator
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask

    torch.manual_seed(1)
    vocab = VOCAB
    if args.wordlm:  # the look-ahead LM walks a character prefix tree: character units + <space>, like the reference's WSJ recipe
        chars = [chr(ord("a") + i) for i in range(26)] + ["'", ".", "-"] + [f"<n{i}>" for i in range(18)]
        d = AsrDictionary.from_symbols(chars, enable_bos=False)
        vocab = len(d)
    else:
        d = AsrDictionary.from_symbols([f"u{i}" for i in range(VOCAB - 4)], enable_bos=False)
    task = SpeechRecognitionEspressoTask.setup_task(SpeechRecognitionEspressoConfig(seed=1), tgt_dict=d)
    cfg = SpeechTransformerConfig()
    e, dc = cfg.encoder, cfg.decoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = 512, 2048, 12, 8
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, "transformer"
    e.conv_channels = "[64, 64, 128, 128]"
    dc.embed_dim, dc.ffn_embed_dim, dc.layers, dc.attention_heads, dc.normalize_before = 512, 2048, 6, 8, True
    dc.input_dim = dc.output_dim = 512
    cfg.layernorm_embedding = True
    cfg.max_source_positions, cfg.max_target_positions = 3600, 1024
    model = SpeechTransformerModelBase.build_model(cfg, task).to(dev).eval()
    lm = None
    if args.lm:
        lm = LSTMLanguageModelEspresso.build_model(dict(arch="lstm_lm_librispeech"), task).to(dev).eval()
    if args.wordlm:
        from espresso_amd.models.tensorized_lookahead_language_model import TensorizedLookaheadLanguageModel

        rng = np.random.default_rng(7)
        words = set()
        while len(words) < 65000:  # random lexicon over the 26 letters, lengths 2..10
            words.add("".join(chr(ord("a") + int(c)) for c in rng.integers(0, 26, size=int(rng.integers(2, 11)))))
        wd = AsrDictionary.from_symbols(sorted(words), enable_bos=False, add_space=False)

        class _LMTask:
            word_dictionary = target_dictionary = source_dictionary = wd
        wlm = LSTMLanguageModelEspresso.build_model(dict(arch="lstm_wordlm_wsj", dropout=0.0), _LMTask).to(dev).eval()
        lm = TensorizedLookaheadLanguageModel(wlm, d, oov_penalty=1e-4, open_vocab=True)
    batches, n_samples = synthetic.make_batches(2000, max_tokens=args.max_tokens, max_sentences=args.batch_size, seed=3)
    samples = [synthetic.make_sample(b, n_samples, vocab, d.pad(), dev, seed=3) for b in batches[: args.batches + 1]]
    task.build_frontend(dev)
    gen = SequenceGenerator([model], d, beam_size=args.beam, max_len_a=0.08, max_len_b=0, lm_model=lm, lm_weight=args.lm_weight,
                            eos_factor=1.5 if (args.lm or args.wordlm) else None)

    def run(s):
        s = task.prepare_sample(s, train=False)
        return gen.generate([model], s)

    run(samples[0])  # warm-up (allocator, positional tables)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ntok = 0
    for s in samples[1:]:
        hyps = run(s)
        ntok += sum(len(h[0]["tokens"]) for h in hyps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    audio = sum(s["audio_seconds"] for s in samples[1:])
    nsent = sum(s["nsentences"] for s in samples[1:])
    print(json.dumps({"metric": "decode RTF", "value": el / audio, "beam": args.beam, "lm_fusion": "lookahead_wordlm_65k_3x1200" if args.wordlm else bool(args.lm), "sentences": nsent,
                      "audio_seconds": audio, "wall_seconds": el, "sentences_per_s": nsent / el, "best_hyp_tokens_per_s": ntok / el,
                      "model": f"conv4 + 12-layer rel-pos Transformer encoder + 6-layer decoder, V={vocab}, bf16, random init (max-length hypotheses)",
                      "data": "synthetic 16 kHz"}))


if __name__ == "__main__":
    main()
