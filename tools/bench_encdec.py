"""Training throughput of BASELINE config 2 (LibriSpeech Transformer encoder-decoder, label-smoothed CE) on one GPU:
conv4 sub-sampling + 12-layer rel-pos Transformer encoder + 6-layer decoder, V = 5003, <= 26000 frames & <= 24 utterances per
batch (examples/asr_librispeech/config/transformer_librispeech.yaml), on-GPU fbank + SpecAugment, dropout 0.1, Adam.  Prints
one JSON line (audio-hours/s); diagnostic tool, the contract metric is bench.py (config 3)."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

VOCAB = 5003


def run(steps=10, warmup=3):
    """-> the result dict (also what bench.py embeds as its `config2_encdec` block)"""
    args = argparse.Namespace(steps=steps, warmup=warmup)
    dev = torch.device("cuda:0")
    import espresso_amd  # noqa: F401
    from espresso_amd.data import synthetic
    from espresso_amd.data.asr_dictionary import AsrDictionary
    from espresso_amd.models.transformer.speech_transformer_base import SpeechTransformerModelBase
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerConfig
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask
    from espresso_amd.trainer import Trainer

    torch.manual_seed(1)
    d = AsrDictionary.from_symbols([f"u{i}" for i in range(VOCAB - 4)], enable_bos=False)
    tcfg = SpeechRecognitionEspressoConfig(
        autoregressive=True, specaugment_config="{'freq_mask_N': 2, 'freq_mask_F': 27, 'time_mask_pm': 0.04, 'time_mask_ps': 0.04}", seed=1)
    task = SpeechRecognitionEspressoTask.setup_task(tcfg, tgt_dict=d)
    cfg = SpeechTransformerConfig()
    e, dc = cfg.encoder, cfg.decoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = 512, 2048, 12, 8
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, "transformer"
    e.learned_pos = True  # transformer_librispeech.yaml: learned relative tables, one per layer
    e.conv_channels = "[64, 64, 128, 128]"
    dc.embed_dim, dc.ffn_embed_dim, dc.layers, dc.attention_heads, dc.normalize_before = 512, 2048, 6, 8, True
    dc.input_dim = dc.output_dim = 512
    cfg.layernorm_embedding = True
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = 0.1
    cfg.max_source_positions, cfg.max_target_positions = 3600, 1024
    model = SpeechTransformerModelBase.build_model(cfg, task)
    crit = task.build_criterion("label_smoothed_cross_entropy_v2", sentence_avg=False, label_smoothing=0.1)
    trainer = Trainer(task, model, crit, dev, clip_norm=2.0, lr=5.0, warmup_steps=25000, adam_betas=(0.9, 0.98), adam_eps=1e-8, seed=1)
    batches, n_samples = synthetic.make_batches(20000, max_tokens=26000, max_sentences=24, seed=1)
    need = args.steps + args.warmup
    pad, eos = d.pad(), d.eos()
    samples = []
    for b in batches[:need]:
        s = synthetic.make_sample(b, n_samples, VOCAB, pad, dev, seed=1)
        tgt = s["target"]
        lens = (tgt != pad).sum(1)
        tgt = torch.cat([tgt, torch.full((tgt.shape[0], 1), pad, dtype=tgt.dtype, device=dev)], 1)
        tgt[torch.arange(tgt.shape[0]), lens] = eos  # AsrTextDataset appends </s>
        prev = torch.full_like(tgt, pad)
        prev[:, 1:] = tgt[:, :-1]
        prev[prev == eos] = pad
        prev[:, 0] = eos  # input feeding: </s> moved to the front
        s["target"], s["net_input"] = tgt, {"prev_output_tokens": prev}
        s["ntokens"] = int(lens.sum()) + tgt.shape[0]
        samples.append(s)
    task.build_frontend(dev)
    task.begin_epoch(1)
    # start-up only (as bench.py): size the arenas / allocator pools for the longest utterance and the largest batch
    by_T = max(samples, key=lambda s: max(s["num_samples"]))
    by_M = max(samples, key=lambda s: s["audio_seconds"])
    trainer.reserve([by_M] if by_M is by_T else [by_M, by_T])
    for i in range(args.warmup):
        trainer.train_step([samples[i]])
    torch.cuda.synchronize()
    if os.environ.get("EA_NO_GC"):
        import gc
        gc.collect()
        gc.freeze()
        gc.disable()
    if os.environ.get("EA_PER_STEP"):
        for i in range(args.warmup, need):
            t1 = time.perf_counter()
            trainer.train_step([samples[i]])
            torch.cuda.synchronize()
            print(f"step {i}: {1e3 * (time.perf_counter() - t1):7.2f} ms  B={samples[i]['nsentences']} audio={samples[i]['audio_seconds']:.0f}s "
                  f"Tmax={max(samples[i]['num_samples']) // 160}", file=sys.stderr)
    t0 = time.perf_counter()
    for i in range(args.warmup, need):
        trainer.train_step([samples[i]])
    torch.cuda.synchronize()
    first = time.perf_counter() - t0  # every batch shape seen for the first time (allocator growth, per-shape tables)
    t0 = time.perf_counter()
    for i in range(args.warmup, need):
        trainer.train_step([samples[i]])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    audio = sum(s["audio_seconds"] for s in samples[args.warmup:])
    return ({"metric": "audio-hours/sec training (LibriSpeech Transformer enc-dec, label-smoothed CE)", "value": audio / 3600 / el,
                      "ms_per_step": el * 1e3 / args.steps, "first_visit_ms_per_step": first * 1e3 / args.steps,
                      "note": "value = second pass over the same batches (steady state); first_visit_ms_per_step = the pass in which every batch shape is new",
                      "steps": args.steps, "audio_seconds_per_step": audio / args.steps,
                      "loss_per_token": float(trainer._stats[1] / max(1.0, float(trainer._stats[2]))),
                      "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "dtype": "bf16", "data": "synthetic 16 kHz",
                      "command": "python tools/bench_encdec.py"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.warmup)))


if __name__ == "__main__":
    main()
