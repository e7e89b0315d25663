"""Transducer decoding real-time factor (config 4's model: conv4 + Conformer-16 encoder + 2-layer LSTM predictor + joint,
V = 5004): the batched greedy decoder and the per-utterance beam search (modified adaptive expansion search, the reference's
defaults: beam 5, 2 expansions per frame).  Synthetic 16 kHz audio, random-init weights — a random joint rarely prefers blank, so
every frame spends all its expansions and hypotheses grow to 2 tokens per frame: the worst case.  `--emit-rate R` (tokens per
second of audio, LibriSpeech BPE ≈ 4.5) instead biases the joint's blank logit until the greedy decoder emits at that rate, i.e.
the lattice shape a trained model produces.  One JSON line per decoder: RTF = wall time / audio duration
(front-end and encoder included)."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

VOCAB = 5004


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--batches", type=int, default=2)
    ap.add_argument("--beam-utts", type=int, default=6, help="utterances of the batch decoded by the beam search (searches batched across utterances)")
    ap.add_argument("--max-tokens", type=int, default=15000)
    ap.add_argument("--batch-size", type=int, default=24)
    ap.add_argument("--emit-rate", type=float, default=0.0, help="calibrate the blank bias to this many emitted tokens per audio second")
    ap.add_argument("--beam-only", action="store_true", help="time the beam search only (bench.py's f3 block)")
    ap.add_argument("--all-utts", action="store_true", help="beam search over EVERY utterance of the timed batches (f3 block: >= 200 utterances)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    import espresso_amd  # noqa: F401
    from espresso_amd.data import synthetic
    from espresso_amd.data.asr_dictionary import AsrDictionary
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerTransducerConfig
    from espresso_amd.models.transformer.speech_transformer_transducer_base import SpeechTransformerTransducerModelBase
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask
    from espresso_amd.tools.transducer_beam_search_decoder import TransducerBeamSearchDecoder
    from espresso_amd.tools.transducer_greedy_decoder import TransducerGreedyDecoder

    torch.manual_seed(1)
    d = AsrDictionary.from_symbols([f"u{i}" for i in range(VOCAB - 5)], enable_bos=True)
    task = SpeechRecognitionEspressoTask.setup_task(SpeechRecognitionEspressoConfig(criterion_name="transducer_loss", seed=1), tgt_dict=d)
    cfg = SpeechTransformerTransducerConfig()
    e, dc = cfg.encoder, cfg.decoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = 512, 2048, 16, 8
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, "conformer"
    e.conv_channels = "[64, 64, 128, 128]"
    dc.embed_dim, dc.hidden_size, dc.layers = 512, 512, 2
    cfg.joint_dim = 512
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    model = SpeechTransformerTransducerModelBase.build_model(cfg, task).to(dev).eval()
    batches, n_samples = synthetic.make_batches(2000, max_tokens=args.max_tokens, max_sentences=args.batch_size, seed=3)
    samples = [synthetic.make_sample(b, n_samples, VOCAB, d.pad(), dev, seed=3) for b in batches[: args.batches + 1]]
    task.build_frontend(dev)
    blank = d.index(task.blank_symbol)
    common = dict(max_num_expansions_per_step=2, bos=d.eos(), blank=blank)
    rate = None

    def n_emitted(hyps):  # the greedy decoder returns its (frame, expansion) grid with blanks in it
        return sum(int((h[0]["tokens"] != blank).sum()) for h in hyps)

    if args.emit_rate > 0:  # bisection on the blank bias with the (fast, batched) greedy decoder on the warm-up batch
        g = TransducerGreedyDecoder([model], d, **common)
        s0 = task.prepare_sample(samples[0], train=False)
        lo, hi = 0.0, 40.0
        base = float(model.fc_out.bias[blank].detach())
        for _ in range(9):
            mid = 0.5 * (lo + hi)
            with torch.no_grad():
                model.fc_out.bias[blank] = base + mid
            rate = n_emitted(g.generate([model], s0)) / samples[0]["audio_seconds"]
            lo, hi = (mid, hi) if rate > args.emit_rate else (lo, mid)
    decoders = [("greedy", TransducerGreedyDecoder([model], d, **common), samples),
                ("beam", TransducerBeamSearchDecoder([model], d, beam_size=args.beam, **common), samples if args.all_utts else None)]
    if args.beam_only:
        decoders = decoders[1:]
    for name, dec, use in decoders:
        if use is None:  # a few utterances of the first timed batch
            s = samples[1]
            k = min(args.beam_utts, s["nsentences"])
            off = s["wav_offsets"]
            s = dict(s, wav=s["wav"][: int(off[k])], wav_offsets=off[: k + 1], num_samples=s["num_samples"][:k], nsentences=k,
                     audio_seconds=float(sum(s["num_samples"][:k])) / 16000.0, id_list=s["id_list"][:k] if "id_list" in s else None)
            use = [s, s]

        def run(s):
            return dec.generate([model], task.prepare_sample(s, train=False))

        run(use[0])  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ntok = 0
        for s in use[1:]:
            ntok += n_emitted(run(s))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        audio = sum(s["audio_seconds"] for s in use[1:])
        nsent = sum(s["nsentences"] for s in use[1:])
        print(json.dumps({"metric": "transducer decode RTF", "decoder": name, "value": el / audio, "beam": 1 if name == "greedy" else args.beam,
                          "max_num_expansions_per_step": 2, "sentences": nsent, "audio_seconds": audio, "wall_seconds": el,
                          "best_hyp_tokens_per_s": ntok / el, "emitted_tokens_per_audio_second": ntok / audio,
                          "blank_bias_calibrated_to_tokens_per_s": rate,
                          "model": "conv4 + Conformer-16 + 2x512 LSTM predictor + joint 512, V=5004, bf16, random init", "data": "synthetic 16 kHz"}),
              flush=True)


if __name__ == "__main__":
    main()
