import sys, json
sys.path.insert(0, '/root/repo')
from tests import gpu_checks as G
for lt, fx in (("conformer", None), ("transformer", None), ("transformer_learnedpos", None), ("conformer", "ref_conformer_ctc_dh64"), ("transformer", "ref_transformer_ctc_dh64")):
    r = G.check_encoder_vs_reference(lt, fixture=fx)
    print(lt, fx, json.dumps({k: (v if not isinstance(v, float) else round(v, 5)) for k, v in r.items()}))

for fx in ("ref_transformer_encdec_tiny", "ref_transformer_encdec_dh64"):
    r = G.check_encdec_vs_reference(fx)
    print(fx, json.dumps({k: (v if not isinstance(v, float) else round(v, 5)) for k, v in r.items()}))
    r = G.check_beam_search_vs_reference(fx)
    print(fx, "beam", json.dumps(r, default=str))
