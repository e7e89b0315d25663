"""End-to-end ingestion block of the bench line: the training loop's real data path — WAV files
on local disk -> the library's readers (csrc/ingest.hip, `dataset.num_workers` decoder threads, int16 into a pinned buffer) -> the
`Prefetcher` of espresso_amd/speech_train.py -> asynchronous int16 host-to-device copy -> on-GPU fbank / CMVN / SpecAugment ->
the Conformer-12 + CTC update step of bench.py — timed over whole updates, next to the SAME batches with the waveforms already in
HBM (what bench.py's `value` measures).  Prints ONE JSON line.

    python tools/bench_ingest.py [--files 2048] [--steps 40] [--warmup 8] [--workers 6]

The loop is speech_train.py's (`task.get_batches` plan -> Prefetcher -> task.to_device -> trainer.train_step), with a device
synchronisation only around the timed region.  Files: the synthetic LibriSpeech-like durations of espresso_amd/data/synthetic.py,
low-pass noise at int16 scale, one json manifest as espresso/tools/asr_prep_json.py packs it (`wave` + `text` + `utt2num_frames`)."""
import argparse
import json
import os
import shutil
import struct
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _encode_flac(job):
    """(worker process) int16 samples -> a FLAC file by the test-only encoder tests/flac_encode.py (fixed + LPC subframes, Rice
    partitions, MD5 signature: what the library's decoder is tested against)"""
    path, raw = job
    from tests import flac_encode as FE

    data = FE.encode(np.frombuffer(raw, dtype="<i2").astype(np.int64))
    with open(path, "wb") as f:
        f.write(data)
    return len(data)


def write_corpus(root, n_files, vocab, seed=1, fmt="wav"):
    from espresso_amd.data import synthetic

    dur = synthetic.durations(n_files, seed)
    rng = np.random.default_rng(seed)
    pool = np.clip(np.round(synthetic.waveform(16000 * 75, rng)), -32768, 32767).astype("<i2")  # 75 s of noise; files are slices of it
    os.makedirs(os.path.join(root, "wav"), exist_ok=True)
    manifest = {}
    nbytes = 0
    jobs = []
    for i, d in enumerate(dur):
        n = int(d * 16000)
        off = int(rng.integers(0, len(pool) - n))
        raw = pool[off:off + n].tobytes()
        path = os.path.join(root, "wav", f"utt{i:05d}.{fmt}")
        if fmt == "flac":
            jobs.append((path, raw))
        else:
            with open(path, "wb") as f:
                f.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16)
                        + b"data" + struct.pack("<I", len(raw)) + raw)
            nbytes += 44 + len(raw)
        L = max(1, int(round(4.5 * d)))
        manifest[f"utt{i:05d}"] = {"wave": path, "text": " ".join(f"u{int(t)}" for t in rng.integers(0, vocab - 5, size=L)),
                                   "utt2num_frames": str(int(1 + (n - 400) // 160))}
    if jobs:  # the pure-Python encoder takes ~1 s per 12 s file: spread over the host's cores
        import multiprocessing as mp

        with mp.get_context("fork").Pool(min(96, os.cpu_count() or 8)) as pl:
            nbytes = sum(pl.imap_unordered(_encode_flac, jobs, chunksize=4))
    with open(os.path.join(root, "train.json"), "w") as f:
        json.dump(manifest, f)
    return float(dur.sum()), nbytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workers", type=int, default=6, help="dataset.num_workers of the recipe (transformer_ctc_librispeech.yaml:28)")
    ap.add_argument("--format", choices=("wav", "flac"), default="wav", help="flac: LibriSpeech's container (decoded by csrc/ingest.hip)")
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    import bench
    from espresso_amd.speech_train import Prefetcher

    device = torch.device("cuda:0")
    torch.cuda.set_device(0)
    root = tempfile.mkdtemp(prefix="ea_ingest_", dir=os.environ.get("EA_INGEST_DIR", "/tmp"))
    try:
        t0 = time.perf_counter()
        total_s, nbytes = write_corpus(root, args.files, bench.VOCAB, fmt=args.format)
        t_write = time.perf_counter() - t0
        task, model, criterion, trainer = bench.build(device)
        task.cfg.data = root
        task.cfg.train_subset = "train"
        ds = task.load_dataset("train")
        batches = task.get_batches(ds, max_tokens=26000, max_sentences=24, max_positions=task.max_positions(), seed=1, epoch=1, shuffle=True)
        batches = [b for b in batches if len(b) > 0]
        need = args.steps + args.warmup
        plan = [batches[i % len(batches)] for i in range(need)]
        first = task.to_device(ds.collater([ds[int(i)] for i in plan[0]]), device)
        task.build_frontend(device, cmvn=bench.estimate_cmvn(task, {k: (v.float() if k == "wav" else v) for k, v in first.items()}, device))
        task.begin_epoch(1)
        sizes = ds.num_tokens_vec(np.arange(len(ds)))
        by_m = max(plan, key=lambda b: int(sizes[b].sum()))
        by_t = max(plan, key=lambda b: int(sizes[b].max()))
        trainer.reserve([task.to_device(ds.collater([ds[int(i)] for i in b]), device) for b in ([by_m] if by_m is by_t else [by_m, by_t])])

        def run(stream_of_samples):
            it = iter(stream_of_samples)
            for _ in range(args.warmup):
                trainer.train_step([task.to_device(next(it), device)])
            torch.cuda.synchronize()
            t = time.perf_counter()
            audio = 0.0
            for _ in range(args.steps):
                s = next(it)
                audio += s["audio_seconds"]
                trainer.train_step([task.to_device(s, device)])
            torch.cuda.synchronize()
            return audio, time.perf_counter() - t

        # (a) end to end: files -> decoder threads -> pinned int16 -> H2D -> step.  The page cache holds the corpus after the first
        # pass of the writer; what is timed is open + read + decode + pack + copy, not the disk
        audio_e, t_e = run(Prefetcher(ds, plan, depth=4, num_workers=args.workers))
        # (b) the same batches with the waveforms resident in HBM before the timed region (bench.py's convention)
        resident = [task.to_device(ds.collater([ds[int(i)] for i in b]), device) for b in plan]
        torch.cuda.synchronize()
        audio_r, t_r = run(resident)
        # (c) the reader alone: batches per second the prefetch thread sustains with nothing consuming GPU time
        t = time.perf_counter()
        n_b, a_b = 0, 0.0
        for s in Prefetcher(ds, plan, depth=4, num_workers=args.workers):
            n_b += 1
            a_b += s["audio_seconds"]
        t_reader = time.perf_counter() - t
        out = {
            "metric": f"audio-hours/sec training, END TO END from {args.format.upper()} files on local disk (Conformer-12 + CTC update step)",
            "format": args.format,
            "value": audio_e / 3600.0 / t_e, "ms_per_step": t_e * 1e3 / args.steps,
            "resident_value": audio_r / 3600.0 / t_r, "resident_ms_per_step": t_r * 1e3 / args.steps,
            "end_to_end_over_resident": (audio_e / t_e) / (audio_r / t_r),
            "reader_only_audio_hours_per_sec": a_b / 3600.0 / t_reader, "reader_only_ms_per_batch": t_reader * 1e3 / max(n_b, 1),
            "steps": args.steps, "warmup": args.warmup, "num_workers": args.workers, "files": args.files,
            "corpus_audio_hours": total_s / 3600.0, "corpus_bytes": nbytes, "corpus_write_s": round(t_write, 2),
            "staging": "int16 samples, pinned host buffer, one asynchronous copy per batch; fbank kernel reads int16",
            "data": f"synthetic 16 kHz {'FLAC (16-bit, tests/flac_encode.py)' if args.format == 'flac' else 'PCM WAV'} files (log-normal durations around 12.3 s) under " + os.path.dirname(root),
        }
        print(json.dumps(out), flush=True)
    finally:
        if not args.keep:
            shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
