"""Locate run-to-run variation in the speech_lstm forward: eval vs train, encoder vs decoder."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests import gpu_checks as G  # noqa: E402

g = np.load(os.path.join(G.GOLD, "ref_speech_lstm_tiny.npz"))
sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
model = G.build_tiny_speech_lstm().to(G.DEV)
model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
feats, lengths = torch.from_numpy(g["feats"]).to(G.DEV), torch.from_numpy(g["lengths"]).to(G.DEV)
prev = torch.from_numpy(g["prev"]).to(G.DEV)


def h(t):
    return float(t.float().double().sum()), float(t.float().abs().double().sum())


for mode in ("eval", "train"):
    model.train(mode == "train")
    for i in range(6):
        with torch.no_grad():
            enc = model.encoder(feats, lengths)
            eo = enc["encoder_out"][0]
            lo, _ = model(feats, lengths, prev)
        torch.cuda.synchronize()
        print(mode, i, "enc", h(eo), "logits", h(lo), flush=True)
# sub-sampler alone
pre = model.encoder.pre_encoder
for i in range(4):
    with torch.no_grad():
        model.train()
        out = pre(feats, lengths) if callable(pre) else None
    torch.cuda.synchronize()
    o = out[0] if isinstance(out, (tuple, list)) else out
    print("pre_encoder train", i, h(o), flush=True)
