#!/usr/bin/env python3
"""Global CMVN statistics on the GPU — replaces espresso/tools/compute_global_cmvn_stats.py:55-127.

Same CLI (`<wav.scp> <output_dir> [--feat-dim 80] [--max-num-utts N]`) and output (`gcmvn.npz` with float64 `mean`, `std`
over all frames, population variance), but the features never exist on the host: waveforms are decoded into a pinned
staging buffer, copied asynchronously, turned into log-mel features by the batched HIP fbank kernel and reduced by
`ea_feature_stats` into fp64 (sum, sum of squares, count) accumulators that stay on the device until the end."""
import argparse
import logging
import os
import sys
from concurrent.futures.thread import ThreadPoolExecutor

import numpy as np
import torch

logger = logging.getLogger("espresso_amd.tools.compute_global_cmvn_stats")


def get_parser():
    p = argparse.ArgumentParser(description="Compute global CMVN stats (GPU)")
    p.add_argument("file", type=str, nargs="?", help="lines of '<utt-id> <wav-path>' or '<utt-id> <command> |'")
    p.add_argument("output_dir", type=str)
    p.add_argument("--feature-type", type=str, default="fbank", choices=["fbank"])
    p.add_argument("--feat-dim", type=int, default=80)
    p.add_argument("--max-num-utts", type=int, default=None)
    p.add_argument("--num-workers", type=int, default=20, help="threads decoding audio")
    p.add_argument("--batch-seconds", type=float, default=600.0, help="audio per GPU launch")
    p.add_argument("--device", default="cuda:0")
    return p


def accumulate(waves, frontend, acc, device):
    """One launch pair over a list of host waveforms: fbank (no CMVN) + statistics."""
    from .. import kernels as K

    lens = [len(w) for w in waves]
    offs = np.zeros(len(waves) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    pinned = torch.empty(int(offs[-1]), dtype=torch.float32, pin_memory=torch.device(device).type == "cuda")
    buf = pinned.numpy()
    for k, w in enumerate(waves):
        buf[offs[k]:offs[k + 1]] = w
    feat, out_len, _ = frontend(pinned.to(device, non_blocking=True), torch.from_numpy(offs).to(device, non_blocking=True), lens,
                                train=False)
    K.feature_stats(feat, out_len, acc)


def finalize(acc: torch.Tensor, nmel: int):
    a = acc.cpu().numpy()
    n = a[2 * nmel]
    mean = a[:nmel] / n
    var = np.maximum(a[nmel:2 * nmel] / n - mean * mean, 0.0)
    return {"mean": mean, "std": np.sqrt(var), "frames": int(n)}


def compute(rxfiles, feat_dim=80, device="cuda:0", num_workers=20, batch_seconds=600.0):
    from ..data.audio_utils import get_waveform
    from ..data.gpu_frontend import GpuFbankFrontend

    frontend = GpuFbankFrontend(device, num_mel_bins=feat_dim)
    acc = torch.zeros(2 * feat_dim + 1, dtype=torch.float64, device=device)
    budget = int(batch_seconds * 16000)
    with ThreadPoolExecutor(max_workers=num_workers) as ex:
        pending, held = [], 0
        for wav, sr in ex.map(get_waveform, rxfiles):
            assert sr == 16000, "the front-end tables are built for 16 kHz audio"
            pending.append(wav)
            held += len(wav)
            if held >= budget:
                accumulate(pending, frontend, acc, device)
                pending, held = [], 0
        if pending:
            accumulate(pending, frontend, acc, device)
    return finalize(acc, feat_dim)


def main(args):
    logging.basicConfig(format="%(asctime)s | %(levelname)s | %(name)s | %(message)s", level=logging.INFO, stream=sys.stdout)
    rx = []
    with (open(args.file, "r", encoding="utf-8") if args.file else sys.stdin) as f:
        for i, line in enumerate(f):
            if args.max_num_utts is not None and i == args.max_num_utts:
                break
            rx.append(line.rstrip().split(None, 1)[1])
    logger.info(f"Computing {args.feature_type} global CMVN stats from {len(rx)} utterances in {args.file}")
    stats = compute(rx, args.feat_dim, args.device, args.num_workers, args.batch_seconds)
    path = os.path.join(args.output_dir, "gcmvn.npz")
    with open(path, "wb") as f:
        np.savez(f, mean=stats["mean"], std=stats["std"])
    logger.info(f"Saved CMVN stats file as {path} ({stats['frames']} frames)")


if __name__ == "__main__":
    main(get_parser().parse_args())
