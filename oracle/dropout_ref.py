"""TEST INFRASTRUCTURE ONLY — numpy restatement of the HIP path's dropout mask streams, so that the oracle
(oracle/torch_ref.py) can run the reference's training-mode arithmetic (FairseqDropout at every site of
fairseq/modules/conformer_layer.py:100,144,146, fairseq/modules/multihead_attention.py:874,
espresso/modules/conformer_with_relative_positional_embedding_encoder_layer.py:125,
espresso/models/transformer/speech_transformer_encoder.py:342,350, fairseq/modules/transformer_layer.py:196,212,216,456,481,
fairseq/models/transformer/transformer_decoder.py:327, espresso/models/speech_lstm.py:800-915) with exactly the keep decisions
the HIP kernels make.

The reference draws its masks from torch's Philox stream (F.dropout); the HIP path draws them from a counter-based hash of
(site seed, element index) (espresso_amd/csrc/common.h ea_hash / ea_keep) that its backward kernels re-evaluate instead of storing
a mask.  The two streams cannot be made equal, so parity of the training mode is established the other way round: the masks are
INPUTS — this file restates the hash (pinned bit for bit against the library's host evaluation `ea_dropout_hash_host` in
tests/test_oracle.py, and against the device kernels in the -m gpu tests), the test records which seed each site used
(`espresso_amd.functional.trace_dropout_seeds`), and the oracle applies the reference's `x * mask / (1 - p)` at the reference's
sites.  A missing 1/(1-p), a forward/backward mask mismatch, a mask indexed in the wrong layout or dropout on the wrong side of a
residual all show up as loss / gradient differences.

Element index of every site = position in the HIP path's own dense layout (include/espresso_amd.h, "Dropout sites"): activation
rows m = b*T + t; attention probabilities ((h*B + b)*T + i)*S + j.  The converters below map those onto the oracle's tensors
((T,B,C), (B,T,C), (B*H,T,S) ...)."""
import numpy as np
import torch

M32 = np.uint64(0xFFFFFFFF)


def ea_hash(seed: int, idx: np.ndarray) -> np.ndarray:
    """csrc/common.h ea_hash: murmur3-style 32-bit finaliser of (seed, idx); idx uint64 array -> uint32 array."""
    idx = np.asarray(idx, dtype=np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    s_lo, s_hi = seed & 0xFFFFFFFF, seed >> 32
    i_lo = (idx & M32).astype(np.uint64)
    i_hi = (idx >> np.uint64(32)).astype(np.uint64)
    hi = ((i_hi * np.uint64(0x85EBCA77)) & M32) ^ np.uint64(s_hi) ^ np.uint64((s_lo * 0xC2B2AE3D) & 0xFFFFFFFF)
    x = (i_lo * np.uint64(0x9E3779B1) + np.uint64(s_lo)) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & M32
    x ^= hi
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & M32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def drop_threshold(p: float) -> int:
    """espresso_amd/kernels.py drop_params / csrc/engine.hip drop_thr: thr = min(floor(p * 2^32), 2^32 - 1), 0 when p <= 0."""
    if p <= 0:
        return 0
    return min(int(float(p) * 4294967296.0), 4294967295)


def keep_mask(seed: int, n: int, p: float, idx0: int = 0) -> np.ndarray:
    """Boolean keep decisions of elements idx0 .. idx0+n-1 (ea_keep: kept iff hash >= thr)."""
    return ea_hash(seed, np.arange(idx0, idx0 + n, dtype=np.uint64)) >= np.uint32(drop_threshold(p))


def scale_mask(seed: int, shape, p: float) -> torch.Tensor:
    """fp32 multiplier (0 or 1/(1-p), the fp32 value the kernels use) over a dense row-major tensor of `shape`."""
    n = int(np.prod(shape))
    inv = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(keep_mask(seed, n, p).astype(np.float32) * inv).view(*shape)


class MaskPlan:
    """Hands the oracle one (seed, p) per dropout site, from the seeds the HIP path reported.

    `trace`: list of (site, seed, p) in the order the HIP forward drew them (espresso_amd.functional.trace_dropout_seeds).
    Sites: 'subsample.out', 'ln.out', 'dropout', 'ffn.act', 'ffn.out', 'attn.probs', 'attn.out', 'conv.out', or a whole-layer
    entry 'layer:conformer' / 'layer:transformer' / 'layer:decoder' of the native layer runtime, whose `p` is the dict
    {p_drop, p_act, p_attn} and whose per-site seeds come from the library's table `layer_seed_fn` = ea_layer_dropout_seed
    (include/espresso_amd.h EA_SITE_*).  The oracle asks by site name in its own forward order."""

    LAYER_SITES = {  # forward order of the sites of one native layer call: (generic name, EA_SITE_* index)
        "conformer": [("ffn.act", 0), ("ffn.out", 1), ("attn.probs", 2), ("attn.out", 3), ("conv.out", 4), ("ffn.act", 5), ("ffn.out", 6)],
        "transformer": [("attn.probs", 2), ("attn.out", 3), ("ffn.act", 0), ("ffn.out", 1)],
        "decoder": [("attn.probs", 2), ("attn.out", 3), ("attn.probs", 7), ("attn.out", 8), ("ffn.act", 0), ("ffn.out", 1)],
    }

    def __init__(self, trace, layer_seed_fn=None):
        self.queue = []
        for site, seed, p in trace:
            if site.startswith("layer:"):
                for name, idx in self.LAYER_SITES[site.split(":")[1]]:
                    ps = p["p_act"] if name.endswith(".act") else p["p_attn"] if name.endswith(".probs") else p["p_drop"]
                    if ps > 0:
                        self.queue.append((name, int(layer_seed_fn(seed, idx)), float(ps)))
            else:
                self.queue.append((site, int(seed), float(p)))
        self.pos = 0
        self.skipped = []

    def seed_for(self, site: str):
        """(seed, p) when the next seed the HIP path drew belongs to `site`, else None: that site ran with p = 0 there (the
        HIP path draws no seed for a probability of zero).  A site the HIP path has but the oracle never reaches, or the
        other way round, leaves the queue unfinished or out of step: `done()` after the oracle's forward raises."""
        if self.pos >= len(self.queue) or self.queue[self.pos][0] != site:
            self.skipped.append((site, self.pos))
            return None
        _, seed, p = self.queue[self.pos]
        self.pos += 1
        return seed, p

    def done(self):
        assert self.pos == len(self.queue), \
            f"the HIP path drew {len(self.queue)} site seeds, the oracle consumed {self.pos}; next: {self.queue[self.pos:][:3]}"


# ---- layout converters: HIP dense layout -> the oracle's tensor ------------------------------------------------------
def rows_btc(seed, p, B, T, C):
    """[B*T][C] rows m = b*T + t  ->  (B, T, C)."""
    return scale_mask(seed, (B, T, C), p)


def rows_tbc(seed, p, T, B, C):
    """-> (T, B, C) (the reference's time-major activations)."""
    return rows_btc(seed, p, B, T, C).transpose(0, 1)


def probs_zts(seed, p, B, H, T, S):
    """[H][B][T][S]  ->  (B*H, T, S) with z = b*H + h (fairseq's `view(T, B*H, dh).transpose(0, 1)` order)."""
    return scale_mask(seed, (H, B, T, S), p).transpose(0, 1).reshape(B * H, T, S)


def subsample_out(seed, p, B, T, Fp, C):
    """sub-sampler output [B*T'][F'*C] with feature index f*C + c  ->  (B, T', C*F') with the reference's c*F' + f."""
    return scale_mask(seed, (B, T, Fp, C), p).permute(0, 1, 3, 2).reshape(B, T, C * Fp)
