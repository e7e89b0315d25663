#!/usr/bin/env python
"""Hot-path benchmark: LibriSpeech Conformer-12 + CTC training steps on synthetic 16 kHz audio.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one optimizer update on one per-GPU batch of the recipe
(examples/asr_librispeech/config/transformer_ctc_librispeech.yaml + model.encoder.layer_type=conformer:
<= 26000 input frames and <= 24 utterances per GPU): raw waveforms already resident in HBM ->
fbank + CMVN + SpecAugment (HIP) -> Conformer-12 -> CTC -> backward -> overlapped RCCL gradient
all-reduce -> clip -> Adam.  Weak scaling: every rank processes its own batches.
Prints ONE JSON line on rank 0 (metric: audio-hours/sec, whole job).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

VOCAB = 5004
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0  # HBM3E peak, same guide


def build(device, seed=1):
    import espresso_amd  # noqa: F401
    from espresso_amd.data.asr_dictionary import AsrDictionary
    from espresso_amd.models.transformer.speech_transformer_encoder_model import conformer_ctc_librispeech
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask
    from espresso_amd.trainer import Trainer

    torch.manual_seed(seed)
    d = AsrDictionary.from_symbols([f"u{i}" for i in range(VOCAB - 5)], enable_bos=True)
    assert len(d) == VOCAB
    tcfg = SpeechRecognitionEspressoConfig(
        specaugment_config="{'freq_mask_N': 2, 'freq_mask_F': 27, 'time_mask_pm': 0.04, 'time_mask_ps': 0.04}", seed=seed,
        criterion_name="ctc_loss")
    task = SpeechRecognitionEspressoTask.setup_task(tcfg, tgt_dict=d)
    mcfg = conformer_ctc_librispeech()
    model = task.build_model(mcfg)
    criterion = task.build_criterion("ctc_loss", sentence_avg=True, zero_infinity=True)
    trainer = Trainer(task, model, criterion, device, clip_norm=2.0, lr=5.0, warmup_steps=25000, adam_betas=(0.9, 0.98),
                      adam_eps=1e-8, seed=seed)
    return task, model, criterion, trainer


def estimate_cmvn(task, sample, device):
    """Global CMVN statistics from synthetic audio (setup only, outside the timed region)."""
    from espresso_amd.data.feature_transforms import GlobalCMVN
    from espresso_amd.data.gpu_frontend import GpuFbankFrontend

    fe = GpuFbankFrontend(device)
    feat, lens, _ = fe(sample["wav"], sample["wav_offsets"], sample["num_samples"], train=False)
    rows = torch.cat([feat[b, : int(l)] for b, l in enumerate(lens.tolist())], 0).double()
    return GlobalCMVN(mean=rows.mean(0).cpu().numpy(), std=rows.std(0).cpu().numpy())


def cpu_baseline(n_threads=None, seconds_per_utt=12.0, n_utts=4, steps=3):
    """Oracle ("port") leg: the CPU restatement of the same update step (fbank -> Conformer-12 -> CTC ->
    backward -> clip -> Adam) on a BOUNDED sample, timed on this box's host cores."""
    from oracle import fbank_ref, torch_ref
    from espresso_amd.data import synthetic

    if n_threads:
        torch.set_num_threads(n_threads)
    cores = torch.get_num_threads()
    torch.manual_seed(0)
    d, H, ffn, L, V = 512, 8, 2048, 12, VOCAB
    sd = {}

    def lin(name, o, i, bias=True):
        sd[name + ".weight"] = torch.randn(o, i) * (i ** -0.5)
        if bias:
            sd[name + ".bias"] = torch.zeros(o)

    def ln(name, c):
        sd[name + ".weight"], sd[name + ".bias"] = torch.ones(c), torch.zeros(c)

    chans = [1, 64, 64, 128, 128]
    for i in range(4):
        sd[f"pre_encoder.convolutions.{i}.weight"] = torch.randn(chans[i + 1], chans[i], 3, 3) * 0.1
        sd[f"pre_encoder.convolutions.{i}.bias"] = torch.zeros(chans[i + 1])
        sd[f"pre_encoder.batchnorms.{i}.weight"], sd[f"pre_encoder.batchnorms.{i}.bias"] = torch.ones(chans[i + 1]), torch.zeros(chans[i + 1])
        sd[f"pre_encoder.batchnorms.{i}.running_mean"], sd[f"pre_encoder.batchnorms.{i}.running_var"] = torch.zeros(chans[i + 1]), torch.ones(chans[i + 1])
    lin("fc0", d, 2560)
    ln("layernorm_embedding", d)
    for l in range(L):
        p = f"layers.{l}."
        for f in ("ffn1.", "ffn2."):
            ln(p + f + "layer_norm", d); lin(p + f + "w_1", ffn, d); lin(p + f + "w_2", d, ffn)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(p + "self_attn." + n, d, d)
        lin(p + "self_attn.pos_proj", d, d, bias=False)
        sd[p + "self_attn.pos_bias_u"], sd[p + "self_attn.pos_bias_v"] = torch.zeros(d), torch.zeros(d)
        ln(p + "self_attn_layer_norm", d); ln(p + "final_layer_norm", d)
        c = p + "conv_module."
        ln(c + "layer_norm", d)
        sd[c + "pointwise_conv1.weight"] = torch.randn(2 * d, d, 1) * d ** -0.5
        sd[c + "depthwise_conv.weight"] = torch.randn(d, 1, 31) * 0.1
        sd[c + "batch_norm.weight"], sd[c + "batch_norm.bias"] = torch.ones(d), torch.zeros(d)
        sd[c + "batch_norm.running_mean"], sd[c + "batch_norm.running_var"] = torch.zeros(d), torch.ones(d)
        sd[c + "pointwise_conv2.weight"] = torch.randn(d, d, 1) * d ** -0.5
    lin("fc_out", V, d)
    params = [k for k in sd if "running" not in k]
    for k in params:
        sd[k].requires_grad_(True)
    opt = torch.optim.Adam([sd[k] for k in params], lr=1e-4, betas=(0.9, 0.98), eps=1e-8)
    rng = np.random.default_rng(0)
    n = int(seconds_per_utt * 16000)
    wavs = [synthetic.waveform(n, rng) for _ in range(n_utts)]
    tgt = torch.randint(4, V, (n_utts, int(4.5 * seconds_per_utt)))
    tl = torch.full((n_utts,), tgt.shape[1])
    times = []
    for s in range(steps + 1):
        t0 = time.perf_counter()
        feats = torch.from_numpy(np.stack([fbank_ref.fbank(w) for w in wavs]))
        lengths = torch.full((n_utts,), feats.shape[1])
        logits, ol = torch_ref.encoder(feats, lengths, sd, H=H, layer_type="conformer", training=True)
        loss = torch_ref.ctc_loss_sum(logits, tgt, ol, tl)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([sd[k] for k in params], 2.0)
        opt.step()
        times.append(time.perf_counter() - t0)
    t = float(np.mean(times[1:]))
    audio_s = n_utts * seconds_per_utt
    return {"value": audio_s / 3600.0 / t, "unit": "audio-hours/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} timed update steps (1 warm-up) of {n_utts} x {seconds_per_utt:.0f} s utterances, fp32, dropout 0, "
                      f"oracle/torch_ref.py + oracle/fbank_ref.py + torch Adam; {t:.2f} s/step"}


def decode_bench(device, n_utts=200, beam=10, seed=3):
    """BASELINE config 5 (SURVEY §8d decode workload): espresso/speech_recognize.py's batched beam search at beam 10 with
    look-ahead word-LM fusion — conv4 + 12-layer rel-pos Transformer encoder + 6-layer attention decoder, character units,
    65 000-word lexicon prefix tree + 3 x 1200 word LSTM LM (random init: every hypothesis runs to max_len = 0.08 * frames, the
    worst case), dev-other-like synthetic utterances (mean 6.4 s), batches of <= 15 000 frames / 24 utterances, front-end and
    encoder inside the timed region.  Returns the `decode` block of the bench line."""
    from espresso_amd.data import synthetic
    from espresso_amd.data.asr_dictionary import AsrDictionary
    from espresso_amd.models.lstm_lm import LSTMLanguageModelEspresso
    from espresso_amd.models.tensorized_lookahead_language_model import TensorizedLookaheadLanguageModel
    from espresso_amd.models.transformer.speech_transformer_base import SpeechTransformerModelBase
    from espresso_amd.models.transformer.speech_transformer_config import SpeechTransformerConfig
    from espresso_amd.sequence_generator import SequenceGenerator
    from espresso_amd.tasks.speech_recognition import SpeechRecognitionEspressoConfig, SpeechRecognitionEspressoTask

    torch.manual_seed(1)
    chars = [chr(ord("a") + i) for i in range(26)] + ["'", ".", "-"] + [f"<n{i}>" for i in range(18)]
    d = AsrDictionary.from_symbols(chars, enable_bos=False)
    task = SpeechRecognitionEspressoTask.setup_task(
        SpeechRecognitionEspressoConfig(seed=1, autoregressive=True, criterion_name="label_smoothed_cross_entropy_v2"), tgt_dict=d)
    cfg = SpeechTransformerConfig()
    e, dc = cfg.encoder, cfg.decoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = 512, 2048, 12, 8
    e.normalize_before, e.relative_positional_embeddings, e.layer_type = True, True, "transformer"
    e.conv_channels = "[64, 64, 128, 128]"
    dc.embed_dim, dc.ffn_embed_dim, dc.layers, dc.attention_heads, dc.normalize_before = 512, 2048, 6, 8, True
    dc.input_dim = dc.output_dim = 512
    cfg.layernorm_embedding = True
    cfg.max_source_positions, cfg.max_target_positions = 3600, 1024
    model = SpeechTransformerModelBase.build_model(cfg, task).to(device).eval()
    rng = np.random.default_rng(7)
    words = set()
    while len(words) < 65000:  # random lexicon over the 26 letters, lengths 2..10
        words.add("".join(chr(ord("a") + int(c)) for c in rng.integers(0, 26, size=int(rng.integers(2, 11)))))
    wd = AsrDictionary.from_symbols(sorted(words), enable_bos=False, add_space=False)

    class _LMTask:
        word_dictionary = target_dictionary = source_dictionary = wd
    wlm = LSTMLanguageModelEspresso.build_model(dict(arch="lstm_wordlm_wsj", dropout=0.0), _LMTask).to(device).eval()
    lm = TensorizedLookaheadLanguageModel(wlm, d, oov_penalty=1e-4, open_vocab=True)
    batches, n_samples = synthetic.make_batches(2864, max_tokens=15000, max_sentences=24, seed=seed, median_s=5.35, sigma=0.6)
    picked, n = [], 0
    for b in batches:  # the first batches of the shuffled plan until n_utts utterances
        picked.append(b)
        n += len(b)
        if n >= n_utts:
            break
    samples = [synthetic.make_sample(b, n_samples, len(d), d.pad(), device, seed=seed) for b in [batches[-1]] + picked]
    task.build_frontend(device)
    gen = SequenceGenerator([model], d, beam_size=beam, max_len_a=0.08, max_len_b=0, lm_model=lm, lm_weight=0.47, eos_factor=1.5)

    def run(smp):
        return gen.generate([model], task.prepare_sample(smp, train=False))

    run(samples[0])  # warm-up (allocator, positional tables)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ntok = nhyp_tok = 0
    for smp in samples[1:]:
        hyps = run(smp)
        ntok += sum(len(h[0]["tokens"]) for h in hyps)
        nhyp_tok += sum(len(h[0]["tokens"]) for h in hyps) * beam
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    audio = sum(smp["audio_seconds"] for smp in samples[1:])
    nsent = sum(smp["nsentences"] for smp in samples[1:])
    block = {"metric": "decode RTF @ beam 10 (lower is better)", "rtf": el / audio, "beam": beam,
             "lm_fusion": "look-ahead word LM (65k-word prefix tree, 3x1200 LSTM), lm_weight 0.47, eos_factor 1.5",
             "sentences": nsent, "audio_seconds": audio, "wall_seconds": el, "sentences_per_s": nsent / el,
             "best_hyp_tokens_per_s": ntok / el, "hypothesis_tokens_per_s": nhyp_tok / el,
             "workload": f"{nsent} of the 2864 synthetic dev-other-like utterances of SURVEY 8(d) (mean {audio / nsent:.1f} s), "
                         "<=15000 frames & <=24 utts per batch, max_len 0.08*frames, conv4 + Transformer-12 encoder + 6-layer decoder, "
                         "bf16, random init (every hypothesis runs to max_len: worst case)",
             "target_rtf": 0.05}
    block["roofline"] = decode_attention_roofline(device, beam)
    return block, model, d


def decode_attention_roofline(device, beam, n_sent=24, S=156, H=8, dh=64, reps=50):
    """HBM roofline of the decode path's dominant kernel (profiles/r05_decode_kernel_trace.txt): the cross-attention launch of one
    decoding step of a full batch — `n_sent * beam` single-row queries against the beam-deduplicated encoder K/V of `n_sent`
    utterances (S = 15000 frames / 24 utterances / 4 positions).  Algorithmic bytes = every K/V row once + q + out (the `beam`
    hypotheses of an utterance share its rows through L2); timed with events on the launch stream."""
    from espresso_amd import kernels as K

    C, N = H * dh, n_sent * beam
    g = torch.Generator(device="cpu").manual_seed(11)
    kv = torch.randn(n_sent, S, 2 * C, generator=g).to(device=device, dtype=torch.bfloat16)
    q = (torch.randn(N, C, generator=g) * dh ** -0.5).to(device=device, dtype=torch.bfloat16)
    kv_row = (torch.arange(N, device=device, dtype=torch.int32) // beam).contiguous()
    lens = torch.full((n_sent,), S, device=device, dtype=torch.int32)
    for _ in range(5):
        K.decode_attention(q, kv, None, kv_row, lens, N, H, dh, S * 2 * C, 2 * C, 0, C, S)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        K.decode_attention(q, kv, None, kv_row, lens, N, H, dh, S * 2 * C, 2 * C, 0, C, S)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    nbytes = kv.numel() * 2 + 2 * q.numel() * 2
    ach = nbytes / us * 1e-3
    return {"bound": "hbm", "kernel": "decode_attention_vec_kernel<64> (cross-attention of one decoding step)", "achieved": ach,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None, "us_per_launch": us,
            "algorithmic_bytes_per_launch": nbytes, "lane_bytes_per_launch": kv.numel() * 2 * beam + 2 * q.numel() * 2,
            "shape": f"{N} hypotheses ({n_sent} utterances x beam {beam}), {H} heads x {dh}, {S} encoder positions",
            "note": "back-to-back launches on one stream (includes the ~2 us launch boundary); 1920 single-wave blocks: latency- "
                    "and occupancy-bound, not bandwidth-bound"}


def decode_cpu_baseline(model, d, n_utts=6, seconds=6.4, beam=10):
    """CPU leg of the decode block: oracle/decode_ref.py (fp32 restatement of incremental beam-search decoding) with the SAME
    encoder-decoder weights on a bounded sample, host cores of this box; no LM fusion (the CPU side has no look-ahead-LM
    restatement — the baseline is therefore the cheaper task)."""
    from oracle import decode_ref, fbank_ref
    from espresso_amd.data import synthetic

    sd = {k: v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(5)
    n = int(seconds * 16000)
    feats = torch.from_numpy(np.stack([fbank_ref.fbank(synthetic.waveform(n, rng)) for _ in range(n_utts)]))
    lengths = torch.full((n_utts,), feats.shape[1])
    t0 = time.perf_counter()
    decode_ref.beam_search(feats, lengths, sd, H=8, pad=d.pad(), eos=d.eos(), unk=d.unk(), beam=beam, max_len_a=0.08)
    el = time.perf_counter() - t0
    return {"rtf": el / (n_utts * seconds), "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_utts} x {seconds:.0f} s utterances, beam {beam}, fp32, oracle/decode_ref.py (fbank excluded), no LM fusion; {el:.1f} s"}


def spawn_ranks(n, fn, args=()):
    """`python bench.py --gpus N` outside torchrun: start N ranks of this node (one per GPU) with the rendezvous environment the
    driver's torch.distributed.run launch would provide, wait for them, propagate failures."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_entry, args=(n, port, fn, args), nprocs=n, join=True)


def _rank_entry(local_rank, world, port, fn, args):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    fn(*args)


def main(argv=None):
    # stdout carries exactly ONE JSON line: native libraries that print to fd 1 (RCCL's version banner under the image's
    # NCCL_DEBUG=VERSION, rocm tools) are redirected to stderr; the JSON goes to a private duplicate of the original stdout
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ.pop("NCCL_DEBUG")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the config-5 decode block (beam-10 RTF)")
    ap.add_argument("--decode-utts", type=int, default=2864, help="utterances of the 2864-utterance decode workload (SURVEY 8d) to time; default: all of it")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the BASELINE config 2 (Transformer enc-dec) and config 4 (Conformer-16 transducer) training blocks")
    ap.add_argument("--gemm-nt-bytes", type=int, default=None, help="diagnostic: non-temporal store threshold of the GEMM epilogue (0 = off)")
    ap.add_argument("--gemm-dump", default=None, help="write the per-launch GEMM records of the roofline replay to this file")
    ap.add_argument("--n-utts", type=int, default=20000)
    ap.add_argument("--max-tokens", type=int, default=26000, help="diagnostic only: shrink the batches (host-overhead probes)")
    ap.add_argument("--no-bwd-overlap", action="store_true", help="A/B switch: keep weight-gradient GEMMs on the main stream")
    ap.add_argument("--no-fused-predrop", action="store_true", help="A/B switch: separate scale+dropout pass per residual block")
    ap.add_argument("--gemm-xcd-mask", type=int, default=None, help="A/B switch: XCD-aware tile order (bit 0 glds kernel, bit 1 register-staged)")
    ap.add_argument("--deferred-inline", action="store_true", help="A/B switch: deferred side work on the main stream (no overlap)")
    ap.add_argument("--no-conv-igemm", action="store_true", help="A/B switch: im2col + GEMM + col2im sub-sampler (round 1) instead of the implicit GEMM")
    ap.add_argument("--no-deferred", action="store_true", help="A/B switch: layer backward joins its side work inside every call")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        # not under torch.distributed.run: spawn one rank per GPU ourselves (rank 0 prints the line on the inherited stdout)
        os.dup2(json_fd, 1)
        spawn_ranks(args.gpus, main, (sys.argv[1:] if argv is None else argv,))
        return
    if args.no_bwd_overlap:
        from espresso_amd._lib import lib as _ealib
        _ealib().ea_set_backward_overlap(0)

    if args.no_fused_predrop:
        from espresso_amd._lib import lib as _ealib3
        _ealib3().ea_set_fused_predrop(0)
    if args.deferred_inline:
        from espresso_amd._lib import lib as _ealib5
        _ealib5().ea_set_backward_deferred_inline(1)
    if args.no_conv_igemm:
        from espresso_amd import functional as _F2
        _F2.set_conv_implicit_gemm(False)
    if args.no_deferred:
        from espresso_amd import functional as _F
        _F.set_backward_deferred(False)
    if args.gemm_nt_bytes is not None:
        from espresso_amd._lib import lib as _ealib3
        _ealib3().ea_set_gemm_nt_store_min_bytes(args.gemm_nt_bytes)
    if args.gemm_xcd_mask is not None:
        from espresso_amd._lib import lib as _ealib2
        _ealib2().ea_set_gemm_xcd_swizzle(args.gemm_xcd_mask)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("EA_DDP_FORCE") == "1":  # EA_DDP_FORCE: RCCL path with a single rank (diagnostic)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if world != args.gpus or (dist.is_initialized() and dist.get_world_size() != args.gpus and world > 1):
        raise RuntimeError(f"--gpus {args.gpus} but the communicator has {world} rank(s): launch with "
                           f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}` or plain `python bench.py --gpus {args.gpus}`")
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")

    from espresso_amd import kernels as K
    from espresso_amd.data import synthetic

    task, model, criterion, trainer = build(device)
    batches, n_samples = synthetic.make_batches(args.n_utts, max_tokens=args.max_tokens, max_sentences=24, seed=1)
    need = args.steps + args.warmup
    mine = [batches[(i * world + rank) % len(batches)] for i in range(need)]
    pad = task.target_dictionary.pad()
    samples = [synthetic.make_sample(b, n_samples, VOCAB, pad, device, seed=1) for b in mine]
    task.build_frontend(device, cmvn=estimate_cmvn(task, samples[0], device))
    task.begin_epoch(1)
    # start-up only: size the arenas for the longest utterance and the largest batch of this run (no parameter update)
    by_T = max(samples, key=lambda s: max(s["num_samples"]))
    by_M = max(samples, key=lambda s: s["audio_seconds"])
    trainer.reserve([by_M] if by_M is by_T else [by_M, by_T])
    torch.cuda.synchronize()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        trainer.train_step([samples[i]])
    sync()
    t0 = time.perf_counter()
    host_marks = [t0]
    for i in range(args.warmup, need):
        trainer.train_step([samples[i]])
        host_marks.append(time.perf_counter())  # (host clock only: nothing here waits for the device)
    host_enqueue = host_marks[-1] - t0  # time for the host to enqueue all steps (diagnostic only)
    # the host runs ahead of the device until the HIP queue is full and then blocks inside launches: the fastest steps show what
    # enqueueing a step costs, the mean above includes that blocking
    host_step_min = min(b - a for a, b in zip(host_marks, host_marks[1:])) if len(host_marks) > 1 else 0.0
    sync()
    elapsed = time.perf_counter() - t0
    audio = sum(s["audio_seconds"] for s in samples[args.warmup:])
    tens = torch.tensor([elapsed, audio], dtype=torch.float64, device=device)
    per_rank_ms = [elapsed * 1e3 / args.steps]
    if world > 1:
        mx = tens.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tens.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        each = [torch.zeros_like(tens) for _ in range(world)]
        dist.all_gather(each, tens)
        per_rank_ms = [float(t[0]) * 1e3 / args.steps for t in each]
        elapsed, audio = float(mx[0]), float(sm[1])
    # hardware-queue probe verdicts of every rank (with a process group in the process HIP may deal
    # a side stream onto the compute stream's queue; the probe rejects such candidates) — so that the first real multi-rank run
    # shows per rank what was decided: [rejected candidates of the layer runtime's stream, unprobed flag, rejected per Python stream ...]
    from espresso_amd import functional as F_
    rep = F_.side_stream_probe_report()
    probe = [[rep["layer_runtime"]["rejected"], int(rep["layer_runtime"]["unprobed"])] + [p["rejected"] for p in rep["python_streams"]]]
    if world > 1:
        row = torch.full((8,), -9.0, dtype=torch.float64, device=device)
        row[: min(8, len(probe[0]))] = torch.tensor(probe[0][:8], dtype=torch.float64)
        rows = [torch.zeros_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        probe = [[int(v) for v in r.tolist() if v > -9.0] for r in rows]
    loss_stats = trainer._stats.clone()

    roofline = None
    if rank == 0 and not args.no_roofline:
        # Instrumented replay of timed-region steps: the library records one HIP-event pair around every
        # launch of the dominant kernel (the bf16 MFMA GEMM) on its launch stream; flops = 2*M*N*K*batch.
        import ctypes
        from espresso_amd import _lib
        lib = _lib.lib()
        nrep = min(2, args.steps)
        lib.ea_gemm_profile_enable(1)
        for i in range(args.warmup, args.warmup + nrep):
            trainer.train_step([samples[i]])
        torch.cuda.synchronize()
        ms, fl = ctypes.c_double(0), ctypes.c_double(0)
        n = lib.ea_gemm_profile_read(ctypes.byref(ms), ctypes.byref(fl))
        alg = ctypes.c_double(0)
        lib.ea_gemm_profile_bytes(ctypes.byref(alg))
        if args.gemm_dump:
            lib.ea_gemm_profile_dump(args.gemm_dump.encode())
        lib.ea_gemm_profile_enable(0)
        tot_ms, tot_fl = ms.value, fl.value
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        # HBM bytes per launch of the same kernels from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this
        # command, tools/pmc_bench_traffic.sh; counters cannot be read from inside the process)
        traffic = None
        tname = next((n for n in ("r06_gemm_traffic.json", "r05_gemm_traffic.json", "r04_gemm_traffic.json", "r03_gemm_traffic.json", "r02_gemm_traffic.json", "r01_gemm_traffic.json")
                      if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", n))), None)
        if tname:
            traffic = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", tname))).get("hbm_bytes_per_launch")
        roofline = {"bound": "mfma", "kernel": "gemm_bf16_kernel+gemm_glds_kernel+gemm_w8_kernel+wgrad_group_kernel", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_unit": f"HBM bytes per launch (PMC, profiles/{tname})",
                    "traffic_source": "constant read from the committed rocprofv3 --pmc run named in traffic_unit (counters cannot be "
                                      "read inside the process); achieved / frac / launches ARE measured live by this run",
                    "algorithmic_bytes_per_launch": alg.value / max(n, 1),
                    "traffic_over_algorithmic": (traffic / (alg.value / max(n, 1))) if traffic else None,
                    "launches_per_step": n / nrep, "avg_launch_us": tot_ms * 1e3 / max(n, 1),
                    "gemm_ms_per_step": tot_ms / nrep, "gemm_flop_per_step": tot_fl / nrep}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
        # the same CPU port at the recipe's batch size (24 utterances: the host cores are better fed); ~45 s of CPU work
        cpu["at_recipe_batch"] = cpu_baseline(n_utts=24, steps=1)
        # the REFERENCE's own modules (imported from /root/reference, dropout 0.1) cannot run on the GPU box: timed once in the
        # build container (8 cores) by tools/cpu_reference_step.py and committed; quoted here next to the port
        ref_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_cpu_reference_container.json")
        if os.path.exists(ref_file):
            cpu["reference_in_build_container"] = json.load(open(ref_file))
    decode = None
    if rank == 0 and world == 1 and not args.no_decode:
        del trainer, model, criterion, samples  # the decode model gets the GPU to itself
        torch.cuda.empty_cache()
        decode, dmodel, ddict = decode_bench(device, n_utts=args.decode_utts)
        if not args.no_cpu_baseline:
            decode["cpu_baseline"] = decode_cpu_baseline(dmodel, ddict)
        del dmodel
    others = {}
    if rank == 0 and world == 1 and not args.no_other_configs:
        # BASELINE configs 2 and 4 (same machinery, other model families): reported next to the headline, never part of `value`
        # Each block runs in its own interpreter (tools/bench_encdec.py / bench_transducer.py print one JSON line): the transducer
        # step is within 20 % of being host-bound, and inside this process — a heap of ~10^6 objects left by the decode block, the
        # CPU baselines' thread pools — it measured 20.5 ms per batch against 17.9 ms on its own.
        import subprocess

        if args.no_decode:
            del trainer, model, criterion, samples
        torch.cuda.empty_cache()
        here = os.path.dirname(os.path.abspath(__file__))
        # f3: the transducer beam search (beam 5, 2 expansions per frame, the reference decoder's defaults) batched
        # across utterances, over 18 batches = 210 utterances of the decode workload; random-init weights = the WORST case (a
        # near-uniform joint's length-normalised scores prefer the longest hypothesis: every frame spends both expansions, 50
        # emitted tokens per audio second against ~4.5 for a trained model — tools/bench_transducer_decode.py)
        # ingest: the same update step fed by speech_train.py's data path from 2 048 WAV files on local disk
        # (tools/bench_ingest.py) next to the same batches resident in HBM
        # ingest_flac (round 6): the same from FLAC files — LibriSpeech's container — decoded by csrc/ingest.hip (512 files encoded by
        # the test-only encoder tests/flac_encode.py on the host's cores first; 24 + 6 steps)
        for key, script in (("ingest", ["bench_ingest.py"]),
                            ("ingest_flac", ["bench_ingest.py", "--format", "flac", "--files", "512", "--steps", "24", "--warmup", "6"]),
                            ("config2_encdec", ["bench_encdec.py"]), ("config4_transducer", ["bench_transducer.py"]),
                            ("f3_transducer_beam_search", ["bench_transducer_decode.py", "--beam-only", "--all-utts", "--batches", "18"])):
            try:
                out = subprocess.run([sys.executable, os.path.join(here, "tools", script[0])] + script[1:], capture_output=True, text=True,
                                     timeout=600)
                last = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
                others[key] = json.loads(last[-1]) if last else {"error": (out.stderr or out.stdout)[-300:]}
            except Exception as e:  # a secondary block must not take the headline line down with it
                others[key] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        value = audio / 3600.0 / elapsed
        line = {
            "metric": "audio-hours/sec training (LibriSpeech Conformer-12 + CTC)",
            "value": value,
            "unit": "audio-hours/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic 16 kHz waveforms (log-normal durations around 12.3 s), random-init weights",
            "config": {"workload": "LibriSpeech 960h Conformer-12 + CTC update step, <=26000 frames & <=24 utts per GPU, "
                                   "V=5004, on-GPU fbank+CMVN+SpecAugment, dropout 0.1, clip 2.0, Adam",
                       "parallelism": f"dp{world}", "audio_seconds_per_step_per_gpu": audio / args.steps / world,
                       "last_loss_per_sentence": float(loss_stats[1] / max(1.0, float(loss_stats[0]))),
                       "host_enqueue_ms_per_step": host_enqueue * 1e3 / args.steps,
                       "host_enqueue_ms_fastest_step": host_step_min * 1e3,
                       "per_rank_ms_per_step": per_rank_ms,
                       "side_stream_queue_probe_per_rank": probe},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "decode": decode,
            **others,
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
