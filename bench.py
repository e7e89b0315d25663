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


def main():
    # stdout carries exactly ONE JSON line: native libraries that print to fd 1 (RCCL's version banner under the image's
    # NCCL_DEBUG=VERSION, rocm tools) are redirected to stderr; the JSON goes to a private duplicate of the original stdout
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        os.environ.pop("NCCL_DEBUG")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gemm-dump", default=None, help="write the per-launch GEMM records of the roofline replay to this file")
    ap.add_argument("--n-utts", type=int, default=20000)
    ap.add_argument("--max-tokens", type=int, default=26000, help="diagnostic only: shrink the batches (host-overhead probes)")
    ap.add_argument("--no-bwd-overlap", action="store_true", help="A/B switch: keep weight-gradient GEMMs on the main stream")
    ap.add_argument("--no-fused-predrop", action="store_true", help="A/B switch: separate scale+dropout pass per residual block")
    ap.add_argument("--gemm-xcd-mask", type=int, default=None, help="A/B switch: XCD-aware tile order (bit 0 glds kernel, bit 1 register-staged)")
    ap.add_argument("--gemm-pk", type=int, default=None, help="A/B switch: persistent GEMM kernel (0 off, 1 automatic, 2+c configuration c)")
    ap.add_argument("--deferred-inline", action="store_true", help="A/B switch: deferred side work on the main stream (no overlap)")
    ap.add_argument("--no-deferred", action="store_true", help="A/B switch: layer backward joins its side work inside every call")
    args = ap.parse_args()
    if args.no_bwd_overlap:
        from espresso_amd._lib import lib as _ealib
        _ealib().ea_set_backward_overlap(0)

    if args.no_fused_predrop:
        from espresso_amd._lib import lib as _ealib3
        _ealib3().ea_set_fused_predrop(0)
    if args.gemm_pk is not None:
        from espresso_amd._lib import lib as _ealib4
        _ealib4().ea_set_gemm_persistent(args.gemm_pk)
    if args.deferred_inline:
        from espresso_amd._lib import lib as _ealib5
        _ealib5().ea_set_backward_deferred_inline(1)
    if args.no_deferred:
        from espresso_amd import functional as _F
        _F.set_backward_deferred(False)
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
    for i in range(args.warmup, need):
        trainer.train_step([samples[i]])
    host_enqueue = time.perf_counter() - t0  # time for the host to enqueue all steps (diagnostic only)
    sync()
    elapsed = time.perf_counter() - t0
    audio = sum(s["audio_seconds"] for s in samples[args.warmup:])
    tens = torch.tensor([elapsed, audio], dtype=torch.float64, device=device)
    if world > 1:
        mx = tens.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tens.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, audio = float(mx[0]), float(sm[1])
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
        if args.gemm_dump:
            lib.ea_gemm_profile_dump(args.gemm_dump.encode())
        lib.ea_gemm_profile_enable(0)
        tot_ms, tot_fl = ms.value, fl.value
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        # HBM bytes per launch of the same kernels from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this
        # command, tools/pmc_bench_traffic.sh; counters cannot be read from inside the process)
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_gemm_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        roofline = {"bound": "mfma", "kernel": "gemm_bf16_kernel+gemm_glds_kernel+wgrad_group_kernel", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch (PMC, profiles/r01_gemm_traffic.json)",
                    "launches_per_step": n / nrep, "avg_launch_us": tot_ms * 1e3 / max(n, 1),
                    "gemm_ms_per_step": tot_ms / nrep, "gemm_flop_per_step": tot_fl / nrep}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

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
                       "host_enqueue_ms_per_step": host_enqueue * 1e3 / args.steps},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
