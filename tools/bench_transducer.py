"""Training throughput of BASELINE config 4 (LibriSpeech Conformer-16 transducer, RNN-T loss) on one GPU: batches by the
reference's product rule (sum over the batch of padded src_frames x tgt_len <= 590000, <= 16 utterances:
examples/asr_librispeech/config/conformer_transducer_librispeech.yaml:28-50, espresso/data/asr_dataset.py:369-382),
V = 5004, on-GPU fbank + SpecAugment, dropout 0.1, Adam, two batches per update (the recipe's `update_freq: [2]`, :50).
Prints one JSON line (audio-hours/s); diagnostic tool, the contract metric is bench.py (config 3)."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

VOCAB = 5004


def run(steps=8, warmup=3, max_product=590000, max_sentences=16, update_freq=2):
    """-> the result dict (also what bench.py embeds as its `config4_transducer` block); a step = one update = `update_freq` batches"""
    args = argparse.Namespace(steps=steps, warmup=warmup, max_product=max_product, max_sentences=max_sentences)
    uf = max(1, int(update_freq))
    dev = torch.device("cuda:0")
    import espresso_amd  # noqa: F401
    from espresso_amd.data import synthetic
    from espresso_amd.data.asr_dictionary import AsrDictionary
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerTransducerConfig
    from espresso_amd.models.transformer.speech_transformer_transducer_base import SpeechTransformerTransducerModelBase
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask
    from espresso_amd.trainer import Trainer

    torch.manual_seed(1)
    d = AsrDictionary.from_symbols([f"u{i}" for i in range(VOCAB - 5)], enable_bos=True)
    tcfg = SpeechRecognitionEspressoConfig(
        specaugment_config="{'freq_mask_N': 2, 'freq_mask_F': 27, 'time_mask_pm': 0.04, 'time_mask_ps': 0.04}", seed=1,
        criterion_name="transducer_loss")
    task = SpeechRecognitionEspressoTask.setup_task(tcfg, tgt_dict=d)
    cfg = SpeechTransformerTransducerConfig()
    e, dc = cfg.encoder, cfg.decoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = 512, 2048, 16, 8
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, "conformer"
    e.conv_channels = "[64, 64, 128, 128]"
    dc.embed_dim, dc.hidden_size, dc.layers, dc.dropout_in, dc.dropout_out = 512, 512, 2, 0.1, 0.1
    cfg.joint_dim = 512
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = 0.1
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    model = SpeechTransformerTransducerModelBase.build_model(cfg, task)
    crit = task.build_criterion("transducer_loss", sentence_avg=True)
    trainer = Trainer(task, model, crit, dev, clip_norm=2.0, lr=5.0, warmup_steps=12000, adam_betas=(0.9, 0.98), adam_eps=1e-8, seed=1)
    # product-rule batching on synthetic utterances (targets ~4.5 tokens/s as in synthetic.make_sample)
    dur = synthetic.durations(4000, 5)
    n_samples = (dur * synthetic.SAMPLE_RATE).astype(np.int64)
    frames = synthetic.num_frames(n_samples)
    tlen = np.maximum(1, np.round(4.5 * n_samples / synthetic.SAMPLE_RATE)).astype(np.int64) + 1
    order = np.argsort(frames, kind="mergesort")
    batches, cur, mf, mt = [], [], 0, 0
    for i in order:
        nf, nt = max(mf, frames[i]), max(mt, tlen[i])
        if cur and ((len(cur) + 1) * nf * nt > args.max_product or len(cur) + 1 > args.max_sentences):
            batches.append(np.array(cur))
            cur, nf, nt = [], frames[i], tlen[i]
        cur.append(i)
        mf, mt = nf, nt
    if cur:
        batches.append(np.array(cur))
    need = (args.steps + args.warmup) * uf
    # stratified over the length-sorted batch list (short many-utterance batches ... long single-utterance batches), then shuffled
    idx = np.linspace(0, len(batches) - 1, need).round().astype(int)
    batches = [batches[i] for i in idx]
    np.random.default_rng(7).shuffle(batches)
    pad, eos = d.pad(), d.eos()
    samples = []
    for b in batches[:need]:
        s = synthetic.make_sample(b, n_samples, VOCAB, pad, dev, seed=5)
        tgt = s["target"]
        lens = (tgt != pad).sum(1)
        tgt = torch.cat([tgt, torch.full((tgt.shape[0], 1), pad, dtype=tgt.dtype, device=dev)], 1)
        tgt[torch.arange(tgt.shape[0]), lens] = eos  # AsrDataset appends EOS
        prev = torch.full_like(tgt, pad)
        prev[:, 0] = eos
        prev[:, 1:] = tgt[:, :-1]
        prev[prev == eos] = pad
        prev[:, 0] = eos
        s["target"] = tgt
        s["net_input"] = {"prev_output_tokens": prev}
        s["ntokens"] = int(lens.sum()) + tgt.shape[0]
        samples.append(s)
    task.build_frontend(dev)
    task.begin_epoch(1)
    # allocator pools sized once for the largest lattices (what speech_train does at start-up): the (B, T', U+1, V) logits and
    # their gradient are 1.5 GB each and differ in size from batch to batch — without this every new maximum costs a 10 ms
    # device allocation in the middle of the step (tools/bench_transducer.py EA_TD_HOST_PROFILE=1: torch.empty 10 ms per batch)
    lattice = lambda s: int(s["target"].shape[0]) * int(s["target"].shape[1]) * max(int(x) for x in s["num_samples"])
    trainer.reserve(sorted(samples, key=lattice, reverse=True)[:2])
    for i in range(args.warmup):
        trainer.train_step(samples[i * uf:(i + 1) * uf])
    torch.cuda.synchronize()
    prof = None
    if os.environ.get("EA_TD_HOST_PROFILE") == "1":  # diagnostic: cProfile of the host side -> gpurun_out/host_profile_td.txt
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        trainer.train_step(samples[i * uf:(i + 1) * uf])
    host = time.perf_counter() - t0  # nothing in the loop waits for the device: this is what the host needs to enqueue the steps
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if prof is not None:
        import io, pstats
        prof.disable()
        out = io.StringIO()
        out.write(f"host enqueue {host * 1e3 / args.steps:.2f} ms per update (under cProfile), {args.steps} updates of {uf} batches\n")
        ps = pstats.Stats(prof, stream=out).sort_stats("cumulative")
        ps.print_stats(60)
        ps.sort_stats("tottime").print_stats(40)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        open(os.path.join(root, "gpurun_out", "host_profile_td.txt"), "w").write(out.getvalue())
    audio = sum(s["audio_seconds"] for s in samples[args.warmup * uf:])
    nodes = [int(s["target"].shape[0]) for s in samples[args.warmup * uf:]]
    return ({"metric": "audio-hours/sec training (LibriSpeech Conformer-16 transducer, RNN-T)", "value": audio / 3600 / el,
                      "ms_per_step": el * 1e3 / args.steps, "steps": args.steps, "update_freq": uf,
                      "ms_per_batch": el * 1e3 / args.steps / uf,
                      "host_enqueue_ms_per_step": host * 1e3 / args.steps, "utts_per_batch": float(np.mean(nodes)),
                      "audio_seconds_per_step": audio / args.steps, "loss_per_sentence": float(trainer._stats[1] / max(1.0, float(trainer._stats[0]))),
                      "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "dtype": "bf16", "data": "synthetic 16 kHz"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--max-product", type=int, default=590000)
    ap.add_argument("--max-sentences", type=int, default=16)
    ap.add_argument("--update-freq", type=int, default=2)
    ap.add_argument("--no-branch-overlap", action="store_true", help="A/B: predictor and joint weight gradient on the main stream")
    a = ap.parse_args()
    if a.no_branch_overlap:
        from espresso_amd import functional as F
        F.set_branch_overlap(False)
    print(json.dumps(run(a.steps, a.warmup, a.max_product, a.max_sentences, a.update_freq)))


if __name__ == "__main__":
    main()
