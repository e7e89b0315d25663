"""Repeat the speech_lstm reference check: run-to-run variation of the loss and of the noisiest gradient."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as G  # noqa: E402

for i in range(10):
    r = G.check_speech_lstm_vs_reference()
    print("run", i, "loss %.6f" % r["loss"], "worst_scale", r["worst_scale"], "worst_l2", r["worst_l2"], flush=True)
