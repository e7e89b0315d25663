"""Round 5: does keeping the vocabulary logits in fp32 (EA_LOGITS_F32=1) bring them closer to the reference's
fp32 run?  Prints max |logit error| against the reference fixtures / the fp32 oracle for the tiny, head-dim-64 and full-size
12-layer encoders.  Run once per mode (the switch is read at import):  EA_LOGITS_F32=0|1 python tools/probes/logits_f32_ab.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_checks as G  # noqa: E402

out = {"EA_LOGITS_F32": os.environ.get("EA_LOGITS_F32", "0")}
for fx in ("ref_conformer_ctc_tiny", "ref_conformer_ctc_dh64"):
    r = G.check_encoder_vs_reference("conformer", fixture=fx)
    out[fx] = {k: r[k] for k in ("eval_logits_abs", "train_logits_abs", "eval_logits_vs_emulation") if k in r}
r = G.check_fullsize_layer_vs_oracle("conformer", layers=12)
out["fullsize_12_layers"] = {k: r[k] for k in ("eval_logits_vs_fp32", "logit_scale", "eval_logits_vs_emulation") if k in r}
print(json.dumps(out))
