#!/usr/bin/env python3
"""`<utt-id> <num_frames>` lines from a wav.scp, read from the WAV headers only (espresso/tools/wav2num_frames.py:41-63;
frame count = 1 + (N - 400) // 160 as in espresso/tools/utils.py:478-486)."""
import argparse
import re
import sys
from concurrent.futures.thread import ThreadPoolExecutor


def get_parser():
    p = argparse.ArgumentParser(description="Compute num_frames from raw waveform files and write them to stdout")
    p.add_argument("file", type=str, nargs="?", help="lines of '<utt-id> <wav-path>' or '<utt-id> <command> |'")
    p.add_argument("--num-workers", type=int, default=20)
    return p


def process(line: str) -> str:
    from ..data.asr_dataset import samples_to_frames
    from ..data.audio_utils import num_samples

    utt_id, rxfile = line.rstrip().split(None, 1)
    assert re.search(r"\.ark:\d+$", rxfile.strip()) is None, "Please provide raw waveform files"
    return utt_id + " " + str(samples_to_frames(num_samples(rxfile)))


def main(args, out=sys.stdout):
    with (open(args.file, "r", encoding="utf-8") if args.file else sys.stdin) as f:
        lines = [l for l in f if l.strip()]
    with ThreadPoolExecutor(max_workers=args.num_workers) as ex:
        for r in ex.map(process, lines):
            print(r, file=out)


if __name__ == "__main__":
    main(get_parser().parse_args())
