"""Round 6: the measured logit errors behind the absolute bounds of tests/test_gpu_parity.py (VERDICT r5 item 2b asked for the
"x logit scale" reading of north_star's 1e-2 to go): HIP vs the fp32 oracle / the emulating oracle at every size the tests bound.
    python tools/probes/r06_logit_errors.py  -> one JSON line per check"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_checks as G  # noqa: E402

keys = ("eval_logits_abs", "train_logits_abs", "eval_logits_vs_emulation", "train_logits_vs_emulation", "eval_logits_vs_fp32",
        "eval_logits_vs_emu", "logit_scale", "eval_logits_abs_valid", "ref_logit_scale")
pick = lambda r: {k: r[k] for k in keys if k in r}
for fx, lt in (("ref_conformer_ctc_tiny", "conformer"), ("ref_conformer_ctc_dh64", "conformer"), ("ref_transformer_ctc_dh64", "transformer")):
    print(fx, json.dumps(pick(G.check_encoder_vs_reference(lt, fixture=fx))), flush=True)
for lt in ("conformer", "transformer"):
    print("fullsize_1_layer", lt, json.dumps(pick(G.check_fullsize_layer_vs_oracle(lt))), flush=True)
print("fullsize_12_layers", json.dumps(pick(G.check_fullsize_layer_vs_oracle("conformer", layers=12, lens=(330, 211), tl=(8, 5)))), flush=True)
print("fullsize_encdec", json.dumps(pick(G.check_fullsize_encdec_vs_oracle())), flush=True)
