"""Debug probe: candidate lists per step of the HIP beam search on sentence 1 of the dh64 enc-dec fixture (beam 3 / 5)."""
import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import gpu_checks as G
from espresso_amd import sequence_generator as SG
fixture = "ref_transformer_encdec_dh64"
g = np.load(os.path.join(G.GOLD, fixture + ".npz"))
sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
model = G._encdec_for(fixture).to(G.DEV)
model.load_state_dict(model.upgrade_state_dict_named(dict(sd), ""), strict=False)
model.eval()
d = G._TaskAR(40).target_dictionary
sample = {"net_input": {"src_tokens": torch.from_numpy(g["feats"]).to(G.DEV), "src_lengths": torch.from_numpy(g["lengths"]).to(G.DEV)}}
orig = SG.HipBeamSearch.step
def step(self, step, lprobs, prev, bsz, beam):
    cs, ct, cb = orig(self, step, lprobs, prev, bsz, beam)
    if step < 3 and bsz > 1:
        V = lprobs.shape[1]
        rows = lprobs.view(bsz, beam, V)[1]
        print(f" step {step}: cand scores {[round(float(x), 4) for x in cs[1]]} tok {ct[1].tolist()} beam {cb[1].tolist()}")
        for b in range(beam if step else 1):
            top = torch.topk(rows[b] + (prev.view(bsz, beam)[1, b] if prev is not None else 0.0), 5)
            print(f"    beam row {b}: top5 {[round(float(x), 4) for x in top.values]} tok {top.indices.tolist()}")
    return cs, ct, cb
SG.HipBeamSearch.step = step
for beam in (3, 5):
    print("beam", beam)
    gen = SG.SequenceGenerator([model], d, beam_size=beam, max_len_a=0.0, max_len_b=12)
    hyps = gen.generate([model], sample)
    for h in hyps[1][:3]:
        print("  HIP", h["tokens"].tolist(), round(float(h["score"]), 4))
