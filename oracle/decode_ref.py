"""TEST / BASELINE INFRASTRUCTURE ONLY — CPU fp32 restatement of batched beam-search decoding of the attention encoder-decoder
(espresso/speech_recognize.py:188-330 -> fairseq/sequence_generator.py:212-620 with incremental decoding:
fairseq/modules/multihead_attention.py:560-640 K/V caches, :964-989 cache reorder; fairseq/search.py:103-144 BeamSearch.step),
driven by a reference-format state_dict.  Used by bench.py's `decode.cpu_baseline` leg (timed on the GPU box's host cores on a
bounded sample; a reported baseline, never the product path) and checked against the full-forward restatement
(oracle/torch_ref.py decoder) in tests/test_oracle.py."""
import math

import torch
import torch.nn.functional as F

from . import torch_ref
from .search_ref import TorchRefSearch


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


class IncrementalDecoder:
    """Pre-LN Transformer decoder, one token per call, K/V caches per layer (transformer_layer.py:384-529)."""

    def __init__(self, sd, H, pad_idx, p="decoder."):
        self.sd, self.H, self.pad, self.p = sd, H, pad_idx, p
        self.W = sd[p + "embed_tokens.weight"]
        self.C = self.W.shape[1]
        self.L = 0
        while (p + f"layers.{self.L}.fc1.weight") in sd:
            self.L += 1
        self.pe = torch_ref.sinusoidal_abs_pe(pad_idx + 2 + 2048, self.C, pad_idx)

    def init(self, enc_out, enc_pad, beam):
        """enc_out (S, B, C), enc_pad (B, S) bool or None -> cross-attention K/V per layer for B*beam rows."""
        S, B, C = enc_out.shape
        H, dh = self.H, C // self.H
        self.xk, self.xv, self.sk, self.sv = [], [], [], []
        for i in range(self.L):
            lp = self.p + f"layers.{i}.encoder_attn."
            k = F.linear(enc_out, self.sd[lp + "k_proj.weight"], self.sd[lp + "k_proj.bias"])
            v = F.linear(enc_out, self.sd[lp + "v_proj.weight"], self.sd[lp + "v_proj.bias"])
            k = k.view(S, B, H, dh).permute(1, 2, 0, 3).repeat_interleave(beam, 0)  # (B*beam, H, S, dh)
            v = v.view(S, B, H, dh).permute(1, 2, 0, 3).repeat_interleave(beam, 0)
            self.xk.append(k)
            self.xv.append(v)
            self.sk.append(None)
            self.sv.append(None)
        self.xmask = None if enc_pad is None else enc_pad.repeat_interleave(beam, 0)[:, None, None, :]

    def reorder(self, idx):
        """fairseq reorders every cached tensor by the surviving beams each step (sequence_generator.py:560-566)."""
        for i in range(self.L):
            self.sk[i] = self.sk[i].index_select(0, idx)
            self.sv[i] = self.sv[i].index_select(0, idx)

    def step(self, tok, pos):
        """tok (N,) last tokens, pos: 0-based step -> log-probs (N, V)."""
        sd, H, C = self.sd, self.H, self.C
        dh = C // H
        N = tok.shape[0]
        x = math.sqrt(C) * F.embedding(tok, self.W) + self.pe[self.pad + 1 + pos]
        if (self.p + "layernorm_embedding.weight") in sd:
            x = _ln(x, sd, self.p + "layernorm_embedding.")
        for i in range(self.L):
            lp = self.p + f"layers.{i}."
            y = _ln(x, sd, lp + "self_attn_layer_norm.")
            a = lp + "self_attn."
            q = F.linear(y, sd[a + "q_proj.weight"], sd[a + "q_proj.bias"]).view(N, H, 1, dh) * dh ** -0.5
            k = F.linear(y, sd[a + "k_proj.weight"], sd[a + "k_proj.bias"]).view(N, H, 1, dh)
            v = F.linear(y, sd[a + "v_proj.weight"], sd[a + "v_proj.bias"]).view(N, H, 1, dh)
            self.sk[i] = k if self.sk[i] is None else torch.cat([self.sk[i], k], 2)
            self.sv[i] = v if self.sv[i] is None else torch.cat([self.sv[i], v], 2)
            w = torch.softmax(q @ self.sk[i].transpose(2, 3), -1)
            o = (w @ self.sv[i]).reshape(N, C)
            x = F.linear(o, sd[a + "out_proj.weight"], sd[a + "out_proj.bias"]) + x
            y = _ln(x, sd, lp + "encoder_attn_layer_norm.")
            a = lp + "encoder_attn."
            q = F.linear(y, sd[a + "q_proj.weight"], sd[a + "q_proj.bias"]).view(N, H, 1, dh) * dh ** -0.5
            w = q @ self.xk[i].transpose(2, 3)
            if self.xmask is not None:
                w = w.masked_fill(self.xmask, float("-inf"))
            o = (torch.softmax(w, -1) @ self.xv[i]).reshape(N, C)
            x = F.linear(o, sd[a + "out_proj.weight"], sd[a + "out_proj.bias"]) + x
            y = _ln(x, sd, lp + "final_layer_norm.")
            y = F.relu(F.linear(y, sd[lp + "fc1.weight"], sd[lp + "fc1.bias"]))
            x = F.linear(y, sd[lp + "fc2.weight"], sd[lp + "fc2.bias"]) + x
        if (self.p + "layer_norm.weight") in sd:
            x = _ln(x, sd, self.p + "layer_norm.")
        return torch.log_softmax(F.linear(x, sd[self.p + "output_projection.weight"]).float(), -1)


def beam_search(feats, lengths, sd, H, pad, eos, unk, beam=10, max_len_a=0.08, max_len_b=0, min_steps=None):
    """Length-synchronous beam search to max_len = max_len_a * frames + max_len_b with EOS only allowed at the last step — the
    regime bench.py's GPU decode runs in with random weights (no hypothesis ends early).  Returns (tokens (B, L), scores (B,))."""
    enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    with torch.no_grad():
        x, out_len = torch_ref.encoder(feats, lengths, enc_sd, H, layer_type="transformer", training=False)
        S, B, C = x.shape
        enc_pad = torch.arange(S).unsqueeze(0) >= out_len.unsqueeze(1)
        dec = IncrementalDecoder(sd, H, pad)
        dec.init(x, enc_pad if bool(enc_pad.any()) else None, beam)
        max_len = int(max_len_a * feats.shape[1] + max_len_b) if min_steps is None else min_steps
        search = TorchRefSearch()
        tokens = torch.full((B * beam, max_len + 1), pad, dtype=torch.long)
        tokens[:, 0] = eos
        scores = torch.zeros(B * beam)
        base = (torch.arange(B) * beam).unsqueeze(1)
        for step in range(max_len):
            lp = dec.step(tokens[:, step], step)
            last = step == max_len - 1
            lp = search.mask(lp, pad, unk, eos, 0.0, only_eos=last, forbid_eos=not last, eos_factor=None)
            sc, idx, beams = search.step(step, lp, scores, B, beam)
            sc, idx, beams = sc[:, :beam], idx[:, :beam], beams[:, :beam]
            src = (base + beams).reshape(-1)
            tokens = tokens.index_select(0, src)
            tokens[:, step + 1] = idx.reshape(-1)
            scores = sc.reshape(-1)
            dec.reorder(src)
        best = tokens.view(B, beam, -1)[:, 0, 1:]
        return best, scores.view(B, beam)[:, 0] / max_len
