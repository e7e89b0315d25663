"""The REFERENCE's own CPU path timed in the build container: espresso's
SpeechTransformerEncoderForPrediction (Conformer-12, d 512, 8 heads, FFN 2048, conv4 [64, 64, 128, 128], V = 5004) imported
from /root/reference with the recipe's dropout 0.1, one update step = forward + F.ctc_loss (the reference criterion's call,
espresso/criterions/ctc_loss.py:85-94) + backward + clip_grad_norm 2.0 + Adam, on 4 x 12 s of 80-dim features, torch CPU threads
as given.  The fbank front-end is NOT in the timed region (the reference computes it with torchaudio, absent from this image);
the number is therefore an upper bound on the reference's CPU throughput.  Writes one JSON line.

    python tools/cpu_reference_step.py [--threads 8] [--utts 4] [--seconds 12] [--steps 3] > profiles/r05_cpu_reference_container.json

Runs only where /root/reference exists (the build container)."""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_stubs"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, os.path.join(ROOT, "oracle"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--utts", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    import gen_golden as GG

    cfg = GG.ref_config("conformer", d=512, heads=8, ffn=2048, layers=12, conv_channels="[64, 64, 128, 128]")
    cfg.dropout, cfg.attention_dropout, cfg.activation_dropout = 0.1, 0.1, 0.1
    cfg.activation_fn = "swish"
    torch.manual_seed(0)
    V = 5004
    model = GG.build_ref_encoder(cfg, V)
    model.train()
    nparam = sum(p.numel() for p in model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.9, 0.98), eps=1e-8)
    T = 1 + (int(args.seconds * 16000) - 400) // 160
    feats = torch.randn(args.utts, T, 80)
    lengths = torch.full((args.utts,), T)
    U = int(4.5 * args.seconds)
    tgt = torch.randint(4, V, (args.utts, U))
    tl = torch.full((args.utts,), U)
    times = []
    for s in range(args.steps + 1):
        t0 = time.perf_counter()
        out = model(feats, lengths)
        logits = out["encoder_out"][0]  # T' x B x V
        in_len = (~out["encoder_padding_mask"][0]).long().sum(-1) if out["encoder_padding_mask"] else torch.full((args.utts,), logits.shape[0])
        lp = torch.log_softmax(logits.float(), -1)
        loss = torch.nn.functional.ctc_loss(lp, tgt, in_len, tl, blank=0, reduction="sum", zero_infinity=True)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 2.0)
        opt.step()
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / len(times[1:])
    audio = args.utts * args.seconds
    print(json.dumps({
        "value": audio / 3600.0 / t, "unit": "audio-hours/sec", "cores": args.threads, "kind": "reference",
        "seconds_per_step": t, "steps_timed": args.steps, "parameters": nparam,
        "sample": f"{args.steps} timed update steps (1 warm-up) of {args.utts} x {args.seconds:.0f} s utterances: the reference's own "
                  "SpeechTransformerEncoderForPrediction (Conformer-12, dropout 0.1) + F.ctc_loss + clip 2.0 + torch Adam, fp32, "
                  f"{args.threads} torch threads in the BUILD container (no GPU box run: /root/reference is absent there); features given "
                  "(torchaudio fbank not timed)",
        "where": "build container", "torch": torch.__version__,
    }))


if __name__ == "__main__":
    main()
