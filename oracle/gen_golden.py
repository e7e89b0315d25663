"""Generate golden fixtures under tests/golden/ by running the REFERENCE's own code
(/root/reference, imported through the stub packages in oracle/ref_stubs) on seeded inputs.

Runs only in the build container (the GPU box has no /root/reference); the fixtures it writes are
committed.  Usage:  python oracle/gen_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "ref_stubs"))
sys.path.insert(1, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def ref_config(layer_type="conformer", d=64, heads=4, ffn=128, layers=2, conv_channels="[64, 64, 16, 16]"):
    import fairseq  # noqa: F401
    import espresso  # noqa: F401
    from espresso.models.transformer.speech_transformer_config import SpeechTransformerConfig

    cfg = SpeechTransformerConfig()
    e = cfg.encoder
    e.embed_dim, e.ffn_embed_dim, e.layers, e.attention_heads = d, ffn, layers, heads
    e.normalize_before, e.learned_pos, e.relative_positional_embeddings = True, False, True
    e.layer_type = layer_type
    e.depthwise_conv_kernel_size = 31
    e.conv_channels = conv_channels
    e.xformers_att_config = None
    e.layerdrop = 0.0
    e.transformer_context = None
    e.chunk_size = 0
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    cfg.tpu = False
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = 0.0
    cfg.activation_fn = "relu"
    cfg.layernorm_embedding = True
    cfg.no_scale_embedding = False
    cfg.no_token_positional_embeddings = False
    cfg.adaptive_input = False
    cfg.quant_noise.pq = 0.0
    cfg.quant_noise.pq_block_size = 8
    cfg.export = False
    cfg.checkpoint_activations = False
    cfg.offload_activations = False
    cfg.min_params_to_wrap = int(1e8)
    return cfg


def build_ref_encoder(cfg, vocab):
    from espresso.models.transformer.speech_transformer_encoder_model import SpeechTransformerEncoderForPrediction
    from espresso.modules.speech_convolutions import ConvBNReLU
    import ast

    ch = ast.literal_eval(cfg.encoder.conv_channels)
    pre = ConvBNReLU(ch, [(3, 3)] * 4, [(1, 1), (2, 2), (1, 1), (2, 2)], in_channels=1)
    return SpeechTransformerEncoderForPrediction(cfg, pre_encoder=pre, input_size=20 * ch[-1], vocab_size=vocab)


def encoder_fixture(layer_type, name, learned_pos=False, d=64, heads=4, ffn=128, frames=70, legacy=None):
    """`d=128, heads=2` gives head dim 64 — the shape class of the recipes (512 / 8) that runs on the fused attention kernels;
    `frames=300` makes 75 encoder frames, i.e. more than one 64-key tile of those kernels."""
    torch.manual_seed(1234)
    V = 40
    cfg = ref_config(layer_type, d=d, heads=heads, ffn=ffn)
    if learned_pos:  # LibriSpeech enc-dec recipes: learned relative positions, one table per layer, full embedding dim
        cfg.encoder.learned_pos = True
        cfg.encoder.share_learned_relative_positional_embeddings_across_layers = False
        cfg.encoder.share_learned_relative_positional_embeddings_across_heads = False
    meta = {}
    if legacy is not None:
        # what the argparse presets `speech_transformer_{wsj,swbd,librispeech}` configure (speech_transformer_legacy.py:103-178):
        # absolute positions (sinusoidal or learned), no embedding LayerNorm; optionally post-LN and chunk-streaming masks
        cfg.encoder.relative_positional_embeddings = False
        cfg.encoder.learned_pos = bool(legacy.get("learned_pos", False))
        cfg.layernorm_embedding = bool(legacy.get("layernorm_embedding", False))
        cfg.encoder.normalize_before = bool(legacy.get("normalize_before", True))
        cfg.encoder.chunk_size = int(legacy.get("chunk_size", 0))
        cfg.encoder.chunk_left_window = int(legacy.get("chunk_left_window", 0))
        cfg.encoder.chunk_right_window = int(legacy.get("chunk_right_window", 0))
        meta = dict(legacy)
    enc = build_ref_encoder(cfg, V)
    # make BN affine / running stats and biases non-trivial so the check exercises them
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
        for n, b in enc.named_buffers():
            if "running_mean" in n:
                b.copy_(0.1 * torch.randn_like(b))
            if "running_var" in n:
                b.copy_(1.0 + 0.2 * torch.rand_like(b))
    B, T = 3, frames
    lengths = torch.tensor([frames, frames * 61 // 70, frames * 37 // 70])
    feats = torch.randn(B, T, 80)
    for b in range(B):
        feats[b, lengths[b]:] = 0.0
    targets = [torch.randint(4, V, (int(L),)) for L in (7, 5, 3)]
    tgt = torch.full((B, 7), 1, dtype=torch.long)
    for b, t in enumerate(targets):
        tgt[b, : len(t)] = t
    out = {}
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    # ---- eval mode ----
    enc.eval()
    with torch.no_grad():
        o = enc(feats, lengths)
    out["eval_logits"] = o["encoder_out"][0].numpy()  # T' x B x V
    out["out_lengths"] = o["src_lengths"][0].numpy()
    # ---- train mode (dropout 0): BN uses batch statistics ; loss + grads ----
    enc.train()
    o = enc(feats, lengths)
    logits = o["encoder_out"][0]
    lprobs = torch.log_softmax(logits.float(), dim=-1)
    in_len = o["src_lengths"][0]
    tl = torch.tensor([len(t) for t in targets])
    flat = torch.cat(targets)
    with torch.backends.cudnn.flags(enabled=False):
        loss = torch.nn.functional.ctc_loss(lprobs, flat, in_len, tl, blank=0, reduction="sum", zero_infinity=True)
    loss.backward()
    out["train_logits"] = logits.detach().numpy()
    out["train_loss"] = np.array(loss.item(), dtype=np.float64)
    grads = {}
    for n, p in enc.named_parameters():
        grads[n] = p.grad.detach().numpy()
    sd_after = {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        feats=feats.numpy(), lengths=lengths.numpy(), targets=tgt.numpy(),
        meta=np.array(json.dumps(meta)),
        **{"sd::" + k: v.numpy() for k, v in sd.items()},
        **{"out::" + k: v for k, v in out.items()},
        **{"grad::" + k: v for k, v in grads.items()},
        **{"bn_after::" + k: v.numpy() for k, v in sd_after.items()},
    )
    print(name, "loss", loss.item(), "params", sum(p.numel() for p in enc.parameters()))


def _build_ref_encdec(dm, heads, ffn, V=40):
    """The reference's speech_transformer_base at test size; returns (model, dictionary)."""
    from espresso.data.asr_dictionary import AsrDictionary
    from espresso.models.transformer.speech_transformer_base import SpeechTransformerModelBase

    cfg = ref_config("transformer", d=dm, heads=heads, ffn=ffn)
    d = cfg.decoder
    d.embed_dim, d.ffn_embed_dim, d.layers, d.attention_heads = dm, ffn, 2, heads
    d.input_dim = d.output_dim = dm
    d.normalize_before, d.learned_pos, d.relative_positional_embeddings = True, False, False
    d.layerdrop = 0.0
    d.xformers_att_config = None
    d.embed_path = None
    d.layers_to_keep = None
    cfg.encoder.layers_to_keep = None
    cfg.share_decoder_input_output_embed = False
    cfg.no_cross_attention = False
    cfg.cross_self_attention = False
    cfg.adaptive_softmax_cutoff = None
    cfg.tie_adaptive_weights = False
    cfg.scheduled_sampling_probs = [1.0]
    cfg.start_scheduled_sampling_epoch = 1
    cfg.layernorm_embedding = True
    cfg.no_decoder_final_norm = False
    cfg.scale_attn = cfg.scale_heads = cfg.scale_fc = cfg.scale_resids = False

    class T:
        feat_dim, feat_in_channels = 80, 1
    dic = AsrDictionary()
    for i in range(V - len(dic) - 1):
        dic.add_symbol(f"t{i}")
    dic.add_symbol("<space>")
    T.target_dictionary = dic
    assert len(dic) == V, len(dic)
    model = SpeechTransformerModelBase.build_model(cfg, T)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return model, dic


def ensemble_fixture(name="ref_transformer_encdec_ensemble", dm=64, heads=4, ffn=128, frames=70):
    """Beam search over an ENSEMBLE of two independently initialised models with the reference's own SequenceGenerator
    (fairseq/sequence_generator.py:837-939 EnsembleModel: log of the mean probability)."""
    from fairseq.sequence_generator import SequenceGenerator

    torch.manual_seed(777)
    m1, dic = _build_ref_encdec(dm, heads, ffn)
    m2, _ = _build_ref_encdec(dm, heads, ffn)
    m1.eval()
    m2.eval()
    B = 3
    lengths = torch.tensor([frames, frames * 61 // 70, frames * 37 // 70])
    feats = torch.randn(B, frames, 80)
    for b in range(B):
        feats[b, lengths[b]:] = 0.0
    beams = {}
    for tag, kw in (("b3", dict(beam_size=3, max_len_a=0.0, max_len_b=12)), ("b1", dict(beam_size=1, max_len_a=0.0, max_len_b=12))):
        gen = SequenceGenerator([m1, m2], dic, **kw)
        hyps = gen.generate([m1, m2], {"net_input": {"src_tokens": feats, "src_lengths": lengths}})
        for bi, hl in enumerate(hyps):
            for hi, hyp in enumerate(hl):
                beams[f"beam::{tag}::{bi}::{hi}::tokens"] = hyp["tokens"].numpy()
                beams[f"beam::{tag}::{bi}::{hi}::score"] = np.array(float(hyp["score"]))
                beams[f"beam::{tag}::{bi}::{hi}::pos"] = hyp["positional_scores"].numpy()
        print(tag, [[h["tokens"].tolist() for h in hl] for hl in hyps], [[round(float(h["score"]), 3) for h in hl] for hl in hyps])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), feats=feats.numpy(), lengths=lengths.numpy(), **beams,
                        **{"sd::" + k: v.numpy() for k, v in m1.state_dict().items()},
                        **{"sd2::" + k: v.numpy() for k, v in m2.state_dict().items()})


def encdec_fixture(name="ref_transformer_encdec_tiny", dm=64, heads=4, ffn=128, frames=70):
    """speech_transformer_base (conv front-end + rel-pos Transformer encoder + 2-layer decoder) with
    label_smoothed_cross_entropy_v2 (uniform, eps 0.1): logits, loss, gradients from the reference's own code."""
    from espresso.criterions.label_smoothed_cross_entropy_v2 import label_smoothed_nll_loss

    torch.manual_seed(4321)
    V = 40
    model, dic = _build_ref_encdec(dm, heads, ffn, V)
    B, Tn = 3, frames
    lengths = torch.tensor([frames, frames * 61 // 70, frames * 37 // 70])
    feats = torch.randn(B, Tn, 80)
    for b in range(B):
        feats[b, lengths[b]:] = 0.0
    pad, eos = dic.pad(), dic.eos()
    tl = [7, 5, 3]
    target = torch.full((B, 8), pad, dtype=torch.long)
    prev = torch.full((B, 8), pad, dtype=torch.long)
    for b, L in enumerate(tl):
        toks = torch.randint(dic.nspecial, V, (L,))
        target[b, :L] = toks
        target[b, L] = eos
        prev[b, 0] = eos
        prev[b, 1:L + 1] = toks
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    out = {}
    model.eval()
    with torch.no_grad():
        lo, _ = model(feats, lengths, prev)
    out["eval_logits"] = lo.numpy()
    # ---- beam search with the reference's own SequenceGenerator (fairseq/sequence_generator.py) ----
    from fairseq.sequence_generator import SequenceGenerator
    beams = {}
    for tag, kw in (("b3", dict(beam_size=3, max_len_a=0.0, max_len_b=12)),
                    ("b3_eosf", dict(beam_size=3, max_len_a=0.0, max_len_b=12, eos_factor=1.5)),
                    ("b1", dict(beam_size=1, max_len_a=0.0, max_len_b=12))):
        gen = SequenceGenerator([model], dic, **kw)
        hyps = gen.generate([model], {"net_input": {"src_tokens": feats, "src_lengths": lengths}})
        for bi, hl in enumerate(hyps):
            for hi, hyp in enumerate(hl):
                beams[f"beam::{tag}::{bi}::{hi}::tokens"] = hyp["tokens"].numpy()
                beams[f"beam::{tag}::{bi}::{hi}::score"] = np.array(float(hyp["score"]))
                beams[f"beam::{tag}::{bi}::{hi}::pos"] = hyp["positional_scores"].numpy()
        print(tag, [[h["tokens"].tolist() for h in hl] for hl in hyps], [[round(float(h["score"]), 3) for h in hl] for hl in hyps])
    # (beam search above ran BEFORE the train-mode forward: that forward updates the BatchNorm running statistics)
    model.train()
    lo, _ = model(feats, lengths, prev)
    lprobs = torch.log_softmax(lo.float(), -1).view(-1, V)
    loss, nll = label_smoothed_nll_loss(lprobs, target.view(-1, 1), 0.1, ignore_index=pad, reduce=True)
    loss.backward()
    out["train_logits"] = lo.detach().numpy()
    out["loss"] = np.array(loss.item())
    out["nll"] = np.array(nll.item())
    grads = {n: p.grad.detach().numpy() for n, p in model.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), feats=feats.numpy(), lengths=lengths.numpy(), prev=prev.numpy(),
                        target=target.numpy(), **beams, **{"sd::" + k: v.numpy() for k, v in sd.items()},
                        **{"out::" + k: v for k, v in out.items()}, **{"grad::" + k: v for k, v in grads.items()})
    print(name, "loss", loss.item(), "nll", nll.item(), "params", sum(p.numel() for p in model.parameters()))
    print([k for k in sd if k.startswith("decoder")][:40])


def encdec_trained_fixture(name="ref_transformer_encdec_trained", dm=64, heads=4, ffn=128, updates=4000, noise=1.0):
    """beam-search fixtures from a model that has LEARNED something, so that the margins at the beam edge are
    real (random-weight fixtures tie there).  The reference's speech_transformer_base is trained here, with the reference's own
    criterion arithmetic (label_smoothed_nll_loss, eps 0.1) and torch Adam, on the learnable synthetic task of tests/trajectory.py
    (every token owns an 80-dim template held for 8 frames + noise); then the reference's SequenceGenerator decodes 9 held-out
    utterances (3 batches of 3) with beam 3, beam 3 + eos_factor 1.5 and beam 1.  The fixture holds the TRAINED weights."""
    from espresso.criterions.label_smoothed_cross_entropy_v2 import label_smoothed_nll_loss
    from fairseq.sequence_generator import SequenceGenerator

    sys.path.insert(0, ROOT)
    from tests import trajectory as TJ

    torch.manual_seed(2024)
    V = 40
    model, dic = _build_ref_encdec(dm, heads, ffn, V)
    pad, eos = dic.pad(), dic.eos()
    train = TJ.make_batches(500, seed=11, noise=noise)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3, betas=(0.9, 0.98), eps=1e-8)
    model.train()
    for it in range(updates):
        feats, lens, tg = train[it % len(train)]
        B, U = tg.shape
        target = torch.full((B, U + 1), pad, dtype=torch.long)
        prev = torch.full((B, U + 1), pad, dtype=torch.long)
        for b in range(B):
            L = int((tg[b] != 1).sum())
            target[b, :L] = tg[b, :L]
            target[b, L] = eos
            prev[b, 0] = eos
            prev[b, 1:L + 1] = tg[b, :L]
        lo, _ = model(feats, lens, prev)
        lprobs = torch.log_softmax(lo.float(), -1).view(-1, V)
        loss, nll = label_smoothed_nll_loss(lprobs, target.view(-1, 1), 0.1, ignore_index=pad, reduce=True)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 2.0)
        opt.step()
        if it % 50 == 0 or it == updates - 1:
            print(f"update {it}: loss/token {loss.item() / (target != pad).sum().item():.3f} nll/token {nll.item() / (target != pad).sum().item():.3f}")
    model.eval()
    held = TJ.make_batches(3, seed=99, B=3, noise=noise)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    out = {}
    n_right = n_tot = 0
    for gi, (feats, lens, tg) in enumerate(held):
        out[f"g{gi}::feats"], out[f"g{gi}::lengths"], out[f"g{gi}::target"] = feats.numpy(), lens.numpy(), tg.numpy()
        for tag, kw in (("b3", dict(beam_size=3, max_len_a=0.0, max_len_b=12)),
                        ("b3_eosf", dict(beam_size=3, max_len_a=0.0, max_len_b=12, eos_factor=1.5)),
                        ("b1", dict(beam_size=1, max_len_a=0.0, max_len_b=12))):
            gen = SequenceGenerator([model], dic, **kw)
            with torch.no_grad():
                hyps = gen.generate([model], {"net_input": {"src_tokens": feats, "src_lengths": lens}})
            for bi, hl in enumerate(hyps):
                for hi, hyp in enumerate(hl):
                    out[f"g{gi}::beam::{tag}::{bi}::{hi}::tokens"] = hyp["tokens"].numpy()
                    out[f"g{gi}::beam::{tag}::{bi}::{hi}::score"] = np.array(float(hyp["score"]))
                    out[f"g{gi}::beam::{tag}::{bi}::{hi}::pos"] = hyp["positional_scores"].numpy()
                if tag == "b3":
                    ref = [int(t) for t in tg[bi] if t != 1]
                    n_right += int(hl[0]["tokens"].tolist()[:-1] == ref)
                    n_tot += 1
            print(gi, tag, [[h["tokens"].tolist() for h in hl][:2] for hl in hyps], [[round(float(h["score"]), 3) for h in hl] for hl in hyps])
    print(f"{n_right} of {n_tot} held-out utterances decoded exactly by the reference's own beam search")
    np.savez_compressed(os.path.join(OUT, name + ".npz"), groups=np.array(len(held)), **out, **{"sd::" + k: v.numpy() for k, v in sd.items()})


# ------------------------------------------------------------------------------------------------
# Training mode with dropout: the reference's own modules run with FairseqDropout fed from given masks
class _FeedDropout:
    """Replaces `torch.nn.functional.dropout` while active — the function behind FairseqDropout (fairseq_dropout.py:31-45),
    torch.nn.Dropout (the Conformer FFN / convolution modules, conformer_layer.py:77,130-131) and the attention-probability
    dropout inside `F.multi_head_attention_forward` (fairseq's plain MultiheadAttention, multihead_attention.py:590-640): the
    k-th dropout call of the reference's forward multiplies by the k-th mask of `sites` — the mask oracle/dropout_ref.py
    derives for (site seed, p) in the layout that call's tensor has — instead of drawing from torch's RNG.  `sites`:
    [(site, layout, kwargs)] in the reference's call order; every entry must be consumed, a tensor whose shape does not fit
    its layout raises.  `F.multi_head_attention_forward` is forced onto its explicit softmax -> dropout -> bmm branch
    (need_weights=True; with False it calls the fused SDPA kernel whose dropout cannot be fed) — same arithmetic."""

    def __init__(self, plan, sites):
        self.plan, self.sites, self.k = plan, sites, 0

    def __enter__(self):
        import torch.nn.functional as TF

        sys.path.insert(0, ROOT)
        from oracle import torch_ref

        feeder = self
        self.orig, self.orig_mha = TF.dropout, TF.multi_head_attention_forward

        def dropout(x, p=0.5, training=True, inplace=False):
            if not (p > 0 and training):
                return x
            site, layout, kw = feeder.sites[feeder.k]
            feeder.k += 1
            if callable(layout):  # a mask the caller cut out of a larger stream (per-step calls of the LSTM decoder)
                m = layout(x)
                assert tuple(m.shape) == tuple(x.shape), (site, m.shape, x.shape)
                return x * m
            with torch_ref.dropout_masks(feeder.plan):
                y = torch_ref._drop(x, site, layout, **kw)
            assert y is not x, (site, "the plan had no mask for this call")
            return y

        def mha_forward(*a, **kw):
            a = list(a)
            if len(a) > 15:
                a[15] = True  # positional `need_weights` (multihead_attention.py:631)
            else:
                kw["need_weights"] = True
            return feeder.orig_mha(*a, **kw)

        TF.dropout, TF.multi_head_attention_forward = dropout, mha_forward
        return self

    def __exit__(self, *a):
        import torch.nn.functional as TF

        TF.dropout, TF.multi_head_attention_forward = self.orig, self.orig_mha
        if a[0] is None:
            assert self.k == len(self.sites), (self.k, len(self.sites))
            self.plan.done()


def _layer_site_list(kind, H, B):
    at = [("attn.probs", "ZTS", dict(B=B, H=H)), ("attn.out", "TBC", {})]
    ffn = [("ffn.act", "TBC", {}), ("ffn.out", "TBC", {})]
    if kind == "conformer":
        return ffn + at + [("conv.out", "BCT", {})] + ffn
    if kind == "transformer":
        return at + ffn
    return at + at + ffn  # decoder: self-attention, encoder attention, FFN


def _synthetic_trace(kinds, p, base=0x5EED00000000):
    """(site, seed, p) entries as espresso_amd.functional.trace_dropout_seeds would report them for a model whose native layer
    calls are `kinds`; seeds are arbitrary distinct numbers (layer seeds multiples of 64)."""
    tr, c = [], 0
    for k in kinds:
        c += 1
        if k.startswith("layer:"):
            tr.append([k, (base + c) * 64 % (1 << 63), {"p_drop": p, "p_act": p, "p_attn": p}])
        else:
            tr.append([k, base + c, p])
    return tr


def dropout_fixtures(p=0.1):
    """`ref_dropout_*`: the reference's encoder / encoder-decoder in TRAINING mode with dropout = attention_dropout =
    activation_dropout = 0.1, every FairseqDropout call fed the mask of a counter-based stream (oracle/dropout_ref.py; torch.nn.Dropout calls too) —
    weights and inputs are those of the existing dropout-off fixtures, so only the trace and the outputs are stored."""
    sys.path.insert(0, ROOT)
    from espresso_amd import _lib
    from oracle import dropout_ref as D

    seed_fn = _lib.lib().ea_layer_dropout_seed
    for layer_type, src in (("conformer", "ref_conformer_ctc_tiny"), ("transformer", "ref_transformer_ctc_tiny")):
        g = np.load(os.path.join(OUT, src + ".npz"))
        sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
        cfg = ref_config(layer_type)
        cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = p
        enc = build_ref_encoder(cfg, 40)
        enc.load_state_dict(sd)
        feats, lengths, tgt = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["targets"])
        B, H = feats.shape[0], 4
        trace = _synthetic_trace(["subsample.out", "ln.out"] + ["layer:" + layer_type] * 2, p)
        sites = [("subsample.out", "SUB", dict(C=16)), ("ln.out", "BTC", {})] + _layer_site_list(layer_type, H, B) * 2
        enc.train()
        with _FeedDropout(D.MaskPlan(trace, seed_fn), sites):
            o = enc(feats, lengths)
        logits = o["encoder_out"][0]
        tl = (tgt != 1).sum(-1)
        flat = torch.cat([tgt[b, : int(tl[b])] for b in range(B)])
        loss = torch.nn.functional.ctc_loss(torch.log_softmax(logits.float(), -1), flat, o["src_lengths"][0], tl, blank=0,
                                            reduction="sum", zero_infinity=True)
        loss.backward()
        name = src.replace("ref_", "ref_dropout_")
        np.savez_compressed(os.path.join(OUT, name + ".npz"), source=np.array(src), trace=np.array(json.dumps(trace)),
                            **{"out::train_logits": logits.detach().numpy(), "out::train_loss": np.array(loss.item())},
                            **{"grad::" + n: q.grad.numpy() for n, q in enc.named_parameters()})
        print(name, "loss", loss.item(), "dropout-off loss", float(g["out::train_loss"]))
    # encoder-decoder (speech_transformer_base + label-smoothed CE)
    from espresso.criterions.label_smoothed_cross_entropy_v2 import label_smoothed_nll_loss

    src = "ref_transformer_encdec_tiny"
    g = np.load(os.path.join(OUT, src + ".npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    torch.manual_seed(4321)
    model, dic = _build_ref_encdec(64, 4, 128, 40)
    model.load_state_dict(sd)
    for m in model.modules():
        if m.__class__.__name__ in ("FairseqDropout", "Dropout"):
            m.p = p
    feats, lengths = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"])
    prev, target = torch.from_numpy(g["prev"]), torch.from_numpy(g["target"])
    B, H = feats.shape[0], 4
    trace = _synthetic_trace(["subsample.out", "ln.out"] + ["layer:transformer"] * 2 + ["ln.out"] + ["layer:decoder"] * 2, p)
    sites = ([("subsample.out", "SUB", dict(C=16)), ("ln.out", "BTC", {})] + _layer_site_list("transformer", H, B) * 2
             + [("ln.out", "BTC", {})] + _layer_site_list("decoder", H, B) * 2)
    model.train()
    with _FeedDropout(D.MaskPlan(trace, seed_fn), sites):
        lo, _ = model(feats, lengths, prev)
    loss, nll = label_smoothed_nll_loss(torch.log_softmax(lo.float(), -1).view(-1, 40), target.view(-1, 1), 0.1,
                                        ignore_index=dic.pad(), reduce=True)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "ref_dropout_transformer_encdec_tiny.npz"), source=np.array(src),
                        trace=np.array(json.dumps(trace)),
                        **{"out::train_logits": lo.detach().numpy(), "out::loss": np.array(loss.item()), "out::nll": np.array(nll.item())},
                        **{"grad::" + n: q.grad.numpy() for n, q in model.named_parameters() if q.grad is not None})
    print("ref_dropout_transformer_encdec_tiny loss", loss.item(), "dropout-off", float(g["out::loss"]))
    # transducer (Conformer encoder + LSTM predictor): predictor dropout_in on the embeddings, dropout_out after every layer at
    # every step (espresso/models/speech_lstm.py:811,866) — the reference calls it per (step, layer) on (B, H); the mask stream
    # is one [U*B][H] tensor per layer (time-major), cut into its step rows here
    src = "ref_conformer_transducer_tiny"
    g = np.load(os.path.join(OUT, src + ".npz"))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    torch.manual_seed(2468)
    model, dic = _ref_transducer_model(40)
    torch.nn.Module.load_state_dict(model, sd)  # (fairseq's override walks upgrade_state_dict_named, which the Conformer layer lacks)
    for m in model.modules():
        if m.__class__.__name__ in ("FairseqDropout", "Dropout"):
            m.p = p
    feats, lengths, prev = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"]), torch.from_numpy(g["prev"])
    B, U1 = prev.shape
    E, Hd, nl = 48, 64, 2
    trace = _synthetic_trace(["subsample.out", "ln.out"] + ["layer:conformer"] * 2 + ["dropout"] * (1 + nl), p)
    plan = D.MaskPlan(trace, seed_fn)
    enc_sites = [("subsample.out", "SUB", dict(C=16)), ("ln.out", "BTC", {})] + _layer_site_list("conformer", 4, B) * 2
    pred_q = plan.queue[len(enc_sites):]
    m_in = D.scale_mask(pred_q[0][1], (U1, B, E), p)
    m_out = [D.scale_mask(pred_q[1 + i][1], (U1, B, Hd), p) for i in range(nl)]
    sites = list(enc_sites) + [("dropout", (lambda x: m_in.transpose(0, 1)), {})]
    for j in range(U1):
        for i in range(nl):
            sites.append(("dropout", (lambda x, i=i, j=j: m_out[i][j]), {}))
    model.train()
    with _FeedDropout(plan, sites) as fd:
        lo, olen = model(feats, lengths, prev)
        plan.pos = len(plan.queue)  # (the predictor's masks were taken from the queue by hand above)
    R = torch.from_numpy(g["R"])
    (lo * R).sum().backward()
    np.savez_compressed(os.path.join(OUT, "ref_dropout_conformer_transducer_tiny.npz"), source=np.array(src),
                        trace=np.array(json.dumps(trace)), **{"out::train_logits": lo.detach().numpy()},
                        **{"grad::" + n: q.grad.numpy() for n, q in model.named_parameters() if q.grad is not None})
    print("ref_dropout_conformer_transducer_tiny logits", float(lo.abs().max()), "vs dropout-off diff",
          float((lo.detach() - torch.from_numpy(g["out::train_logits"])).abs().max()))


def _ref_transducer_model(V=40):
    """The reference's speech_transformer_transducer_base at test size (the configuration of `transducer_fixture`); -> (model, dictionary)."""
    from espresso.data.asr_dictionary import AsrDictionary
    from espresso.models.transformer.speech_transformer_transducer_base import SpeechTransformerTransducerModelBase
    from espresso.models.transformer.speech_transformer_transducer_config import SpeechTransformerTransducerConfig

    base = ref_config("conformer")
    cfg = SpeechTransformerTransducerConfig()
    cfg.encoder = base.encoder
    d = cfg.decoder
    d.embed_dim, d.hidden_size, d.layers, d.residual, d.dropout_in, d.dropout_out = 48, 64, 2, True, 0.0, 0.0
    d.embed_path = None
    cfg.encoder.layers_to_keep = None
    cfg.joint_dim = 64
    cfg.share_decoder_input_output_embed = False
    cfg.max_source_positions, cfg.max_target_positions = 3600, 200
    cfg.tpu = False
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = 0.0
    cfg.activation_fn = "relu"
    cfg.layernorm_embedding = True
    cfg.no_scale_embedding = False
    cfg.no_token_positional_embeddings = False
    cfg.adaptive_input = False
    cfg.quant_noise.pq = 0.0
    cfg.quant_noise.pq_block_size = 8
    cfg.export = False
    cfg.checkpoint_activations = False
    cfg.offload_activations = False
    cfg.min_params_to_wrap = int(1e8)

    class T:
        feat_dim, feat_in_channels = 80, 1
    dic = AsrDictionary(enable_bos=True)
    for i in range(V - len(dic) - 1):
        dic.add_symbol(f"t{i}")
    dic.add_symbol("<space>")
    T.target_dictionary = dic
    assert len(dic) == V, len(dic)
    model = SpeechTransformerTransducerModelBase.build_model(cfg, T)
    return model, dic


def transducer_fixture(name="ref_conformer_transducer_tiny"):
    """speech_transformer_transducer_base (conv front-end + rel-pos Conformer encoder + 2-layer LSTM predictor + joint with a
    weight-normed fc_out): logits (B, T', U+1, V) in eval and train mode and every parameter gradient of
    sum(logits * R) for a fixed random R, all from the reference's own modules.  (torchaudio is not installable here, so the
    RNN-T loss itself is pinned separately: oracle/rnnt_ref.py against brute-force alignment sums.)"""
    from espresso.data.asr_dictionary import AsrDictionary
    from espresso.models.transformer.speech_transformer_transducer_base import SpeechTransformerTransducerModelBase
    from espresso.models.transformer.speech_transformer_transducer_config import SpeechTransformerTransducerConfig

    torch.manual_seed(2468)
    V = 40
    model, dic = _ref_transducer_model(V)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and "weight_g" not in n:
                p.add_(0.1 * torch.randn_like(p))
        # make the search non-degenerate: input-dependent logits and a competitive blank
        # (relu output has a large common component: remove it from the output rows, sharpen, and bias towards blank)
        model.fc_out.weight_v.sub_(model.fc_out.weight_v.mean(dim=1, keepdim=True))
        model.fc_out.weight_g.copy_(3.0 * model.fc_out.weight_v.norm(dim=1, keepdim=True))
        model.fc_out.bias.zero_()
        model.fc_out.bias[dic.bos()] = 1.0
    B, Tn = 3, 70
    lengths = torch.tensor([70, 61, 37])
    feats = torch.randn(B, Tn, 80)
    for b in range(B):
        feats[b, lengths[b]:] = 0.0
    pad, eos = dic.pad(), dic.eos()
    tl = [7, 5, 3]
    U1 = 8
    prev = torch.full((B, U1), pad, dtype=torch.long)
    for b, L in enumerate(tl):
        prev[b, 0] = eos
        prev[b, 1:L + 1] = torch.randint(dic.nspecial, V, (L,))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    out = {}
    model.eval()
    with torch.no_grad():
        lo, olen = model(feats, lengths, prev)
    out["eval_logits"] = lo.numpy()
    out["out_lengths"] = olen.numpy()
    # greedy transducer search with the reference's own decoder (espresso/tools/transducer_greedy_decoder.py)
    from espresso.tools.transducer_greedy_decoder import TransducerGreedyDecoder
    for tag, kw in (("e2", dict(max_num_expansions_per_step=2)), ("e1_eos", dict(max_num_expansions_per_step=1, model_predicts_eos=True))):
        dec = TransducerGreedyDecoder([model], dic, print_alignment=True, **kw)
        toks, scores, ali = dec._generate({"net_input": {"src_tokens": feats, "src_lengths": lengths}})
        out[f"greedy_{tag}_tokens"] = toks.numpy()
        out[f"greedy_{tag}_scores"] = scores.numpy()
        print(tag, [[int(t) for t in row if int(t) != dec.blank] for row in toks], scores.tolist())
    # modified adaptive expansion search with the reference's own decoder (espresso/tools/transducer_beam_search_decoder.py)
    from espresso.tools.transducer_beam_search_decoder import TransducerBeamSearchDecoder
    for tag, kw in (("b3", dict(beam_size=3, max_num_expansions_per_step=2, prefix_alpha=1)),
                    ("b4_beta1_g2", dict(beam_size=4, max_num_expansions_per_step=2, expansion_beta=1, expansion_gamma=2.0, prefix_alpha=2)),
                    ("b2_nonorm", dict(beam_size=2, max_num_expansions_per_step=1, normalize_scores=False))):
        dec = TransducerBeamSearchDecoder([model], dic, **kw)
        toks_l, scores_l, _ = dec._generate({"net_input": {"src_tokens": feats, "src_lengths": lengths}})
        for bi, (tk, sc) in enumerate(zip(toks_l, scores_l)):
            out[f"beam_{tag}_{bi}_tokens"] = tk.numpy()
            out[f"beam_{tag}_{bi}_scores"] = sc.numpy()
        print(tag, [[int(t) for t in row if int(t) != dic.pad()] for row in toks_l[0]], scores_l[0].tolist())
    model.train()
    lo, _ = model(feats, lengths, prev)
    R = torch.randn_like(lo) * 0.1
    for b in range(B):  # frames beyond the encoder output length carry implementation-defined values: keep them out of the objective
        R[b, int(olen[b]):] = 0.0
    (lo * R).sum().backward()
    out["train_logits"] = lo.detach().numpy()
    grads = {n: p.grad.detach().numpy() for n, p in model.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), feats=feats.numpy(), lengths=lengths.numpy(), prev=prev.numpy(), R=R.numpy(),
                        **{"sd::" + k: v.numpy() for k, v in sd.items()}, **{"out::" + k: v for k, v in out.items()},
                        **{"grad::" + k: v for k, v in grads.items()})
    print(name, "logits", tuple(lo.shape), "params", sum(p.numel() for p in model.parameters()))
    print([k for k in sd if not k.startswith("encoder")])


def speech_lstm_fixture(name="ref_speech_lstm_tiny"):
    """speech_lstm (BASELINE config 1): conv front-end + 2-layer packed BiLSTM encoder + 2-layer LSTM decoder with Bahdanau
    attention, input feeding, residuals and additional_fc; label-smoothed CE (uniform 0.1).  Logits / loss / gradients and
    a beam search, all from the reference's own modules."""
    import argparse
    from espresso.data.asr_dictionary import AsrDictionary
    from espresso.models.speech_lstm import SpeechLSTMModel, base_architecture
    from espresso.criterions.label_smoothed_cross_entropy_v2 import label_smoothed_nll_loss
    from fairseq.sequence_generator import SequenceGenerator

    torch.manual_seed(8642)
    V = 40
    args = argparse.Namespace(dropout=0.0, encoder_conv_channels="[64, 64, 16, 16]", encoder_rnn_hidden_size=32, encoder_rnn_layers=2,
                              encoder_rnn_residual=True, decoder_embed_dim=24, decoder_hidden_size=32, decoder_layers=2,
                              decoder_out_embed_dim=48, attention_dim=40, criterion_name="label_smoothed_cross_entropy_v2",
                              scheduled_sampling_probs=[1.0], start_scheduled_sampling_epoch=1, max_source_positions=3600,
                              max_target_positions=200)
    base_architecture(args)

    class T:
        feat_dim, feat_in_channels = 80, 1
        cfg = argparse.Namespace(num_batch_buckets=0)
    dic = AsrDictionary()
    for i in range(V - len(dic) - 1):
        dic.add_symbol(f"t{i}")
    dic.add_symbol("<space>")
    T.target_dictionary = dic
    model = SpeechLSTMModel.build_model(args, T)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.mul_(2.5)  # U(-0.25, 0.25): LSTM gates and attention away from the linear regime
    B, Tn = 3, 70
    lengths = torch.tensor([70, 61, 37])
    feats = torch.randn(B, Tn, 80)
    for b in range(B):
        feats[b, lengths[b]:] = 0.0
    pad, eos = dic.pad(), dic.eos()
    tl = [7, 5, 3]
    target = torch.full((B, 8), pad, dtype=torch.long)
    prev = torch.full((B, 8), pad, dtype=torch.long)
    for b, L in enumerate(tl):
        toks = torch.randint(dic.nspecial, V, (L,))
        target[b, :L] = toks
        target[b, L] = eos
        prev[b, 0] = eos
        prev[b, 1:L + 1] = toks
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    out, beams = {}, {}
    model.eval()
    with torch.no_grad():
        lo, _ = model(feats, lengths, prev)
    out["eval_logits"] = lo.numpy()
    for tag, kw in (("b3", dict(beam_size=3, max_len_a=0.0, max_len_b=10)),):
        gen = SequenceGenerator([model], dic, **kw)
        hyps = gen.generate([model], {"net_input": {"src_tokens": feats, "src_lengths": lengths}})
        for bi, hl in enumerate(hyps):
            for hi, hyp in enumerate(hl):
                beams[f"beam::{tag}::{bi}::{hi}::tokens"] = hyp["tokens"].numpy()
                beams[f"beam::{tag}::{bi}::{hi}::score"] = np.array(float(hyp["score"]))
                beams[f"beam::{tag}::{bi}::{hi}::pos"] = hyp["positional_scores"].numpy()
        print(tag, [[h["tokens"].tolist() for h in hl] for hl in hyps], [[round(float(h["score"]), 3) for h in hl] for hl in hyps])
    model.train()
    lo, _ = model(feats, lengths, prev)
    lprobs = torch.log_softmax(lo.float(), -1).view(-1, V)
    loss, nll = label_smoothed_nll_loss(lprobs, target.view(-1, 1), 0.1, ignore_index=pad, reduce=True)
    loss.backward()
    out["train_logits"] = lo.detach().numpy()
    out["loss"], out["nll"] = np.array(loss.item()), np.array(nll.item())
    grads = {n: p.grad.detach().numpy() for n, p in model.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), feats=feats.numpy(), lengths=lengths.numpy(), prev=prev.numpy(),
                        target=target.numpy(), **beams, **{"sd::" + k: v.numpy() for k, v in sd.items()},
                        **{"out::" + k: v for k, v in out.items()}, **{"grad::" + k: v for k, v in grads.items()})
    print(name, "loss", loss.item(), "params", sum(p.numel() for p in model.parameters()))
    print(list(sd.keys()))


def lm_fusion_fixture(name="ref_lm_fusion_tiny"):
    """Shallow fusion inside the reference's SequenceGenerator (fairseq/sequence_generator.py:385-393): the enc-dec model of
    ref_transformer_encdec_tiny.npz (weights reloaded from that fixture) + a tiny sub-word LSTM LM, beam 3, lm_weight 0.5."""
    import argparse
    from espresso.data.asr_dictionary import AsrDictionary
    from espresso.models.lstm_lm import LSTMLanguageModelEspresso, base_lm_architecture
    from espresso.models.transformer.speech_transformer_base import SpeechTransformerModelBase
    from fairseq.sequence_generator import SequenceGenerator

    g = np.load(os.path.join(OUT, "ref_transformer_encdec_tiny.npz"))
    torch.manual_seed(1357)
    V = 40
    cfg = ref_config("transformer")
    d = cfg.decoder
    d.embed_dim, d.ffn_embed_dim, d.layers, d.attention_heads = 64, 128, 2, 4
    d.input_dim = d.output_dim = 64
    d.normalize_before, d.learned_pos, d.relative_positional_embeddings = True, False, False
    d.layerdrop = 0.0
    d.xformers_att_config = None
    d.embed_path = None
    d.layers_to_keep = None
    cfg.encoder.layers_to_keep = None
    cfg.share_decoder_input_output_embed = False
    cfg.no_cross_attention = False
    cfg.cross_self_attention = False
    cfg.adaptive_softmax_cutoff = None
    cfg.tie_adaptive_weights = False
    cfg.scheduled_sampling_probs = [1.0]
    cfg.start_scheduled_sampling_epoch = 1
    cfg.layernorm_embedding = True
    cfg.no_decoder_final_norm = False
    cfg.scale_attn = cfg.scale_heads = cfg.scale_fc = cfg.scale_resids = False

    class T:
        feat_dim, feat_in_channels = 80, 1
    dic = AsrDictionary()
    for i in range(V - len(dic) - 1):
        dic.add_symbol(f"t{i}")
    dic.add_symbol("<space>")
    T.target_dictionary = T.source_dictionary = dic
    model = SpeechTransformerModelBase.build_model(cfg, T)
    model.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")})
    model.eval()
    args = argparse.Namespace(decoder_embed_dim=24, decoder_hidden_size=32, decoder_layers=2, decoder_out_embed_dim=32, dropout=0.0,
                              share_embed=False, is_wordlm=False, criterion_name="cross_entropy", tokens_per_sample=64)
    base_lm_architecture(args)
    from espresso.tasks.speech_recognition import SpeechRecognitionEspressoTask  # noqa: F401 (isinstance check in build_model)
    lm = LSTMLanguageModelEspresso.build_model(args, T)
    with torch.no_grad():
        for p_ in lm.parameters():
            p_.mul_(3.0)
    lm.eval()
    feats, lengths = torch.from_numpy(g["feats"]), torch.from_numpy(g["lengths"])
    beams = {}
    for tag, kw in (("lm05", dict(beam_size=3, max_len_a=0.0, max_len_b=12, lm_weight=0.5)),
                    ("lm10_eosf", dict(beam_size=3, max_len_a=0.0, max_len_b=12, lm_weight=1.0, eos_factor=1.5))):
        gen = SequenceGenerator([model], dic, lm_model=lm, **kw)
        hyps = gen.generate([model], {"net_input": {"src_tokens": feats, "src_lengths": lengths}})
        for bi, hl in enumerate(hyps):
            for hi, hyp in enumerate(hl):
                beams[f"beam::{tag}::{bi}::{hi}::tokens"] = hyp["tokens"].numpy()
                beams[f"beam::{tag}::{bi}::{hi}::score"] = np.array(float(hyp["score"]))
                beams[f"beam::{tag}::{bi}::{hi}::pos"] = hyp["positional_scores"].numpy()
        print(tag, [[h["tokens"].tolist() for h in hl] for hl in hyps], [[round(float(h["score"]), 3) for h in hl] for hl in hyps])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **beams, **{"lm::" + k: v.numpy() for k, v in lm.state_dict().items()})


def lm_train_fixture(name="ref_lstm_lm_train_tiny"):
    """Language-model TRAINING step of the reference (lstm_lm_librispeech.yaml: model lstm_lm_espresso with tied embeddings,
    criterion cross_entropy = fairseq/criterions/cross_entropy.py:44-71): logits of a right-padded sentence batch
    (source = </s> w1 … w_{n-1}, target = w1 … w_{n-1} </s>, the `eos` sample-break mode of fairseq's MonolingualDataset),
    summed NLL over the non-pad targets, and all parameter gradients."""
    import argparse
    import torch.nn.functional as TF
    from espresso.data.asr_dictionary import AsrDictionary
    from espresso.models.lstm_lm import LSTMLanguageModelEspresso, base_lm_architecture
    from espresso.tasks.speech_recognition import SpeechRecognitionEspressoTask  # noqa: F401

    torch.manual_seed(4242)
    V = 40
    dic = AsrDictionary()
    for i in range(V - len(dic) - 1):
        dic.add_symbol(f"t{i}")
    dic.add_symbol("<space>")

    class T:
        pass
    T.target_dictionary = T.source_dictionary = dic
    out = {}
    for tag, share in (("tied", True), ("untied", False)):
        dim_out = 32
        args = argparse.Namespace(decoder_embed_dim=32 if share else 24, decoder_hidden_size=32, decoder_layers=2,
                                  decoder_out_embed_dim=dim_out, dropout=0.0, share_embed=share, is_wordlm=False,
                                  criterion_name="cross_entropy", tokens_per_sample=64)
        base_lm_architecture(args)
        lm = LSTMLanguageModelEspresso.build_model(args, T)
        with torch.no_grad():
            for p_ in lm.parameters():
                p_.mul_(2.0)
        lm.train()
        rng = np.random.default_rng(11)
        lens = [7, 5, 2, 6]
        B, U = len(lens), max(lens)
        target = torch.full((B, U), dic.pad(), dtype=torch.long)
        src = torch.full((B, U), dic.pad(), dtype=torch.long)
        for b, n in enumerate(lens):
            sent = torch.from_numpy(rng.integers(4, V, size=n - 1))
            target[b, : n - 1], target[b, n - 1] = sent, dic.eos()
            src[b, 0], src[b, 1:n] = dic.eos(), sent
        logits = lm(src, src_lengths=torch.tensor(lens))[0]
        lprobs = lm.get_normalized_probs((logits, None), log_probs=True)
        loss = TF.nll_loss(lprobs.view(-1, lprobs.size(-1)), target.view(-1), ignore_index=dic.pad(), reduction="sum")
        loss.backward()
        out.update({f"{tag}::sd::{k}": v.detach().numpy().copy() for k, v in lm.state_dict().items()})
        out.update({f"{tag}::grad::{k}": p_.grad.numpy().copy() for k, p_ in lm.named_parameters()})
        out[f"{tag}::logits"], out[f"{tag}::loss"] = logits.detach().numpy(), np.array(float(loss))
        out[f"{tag}::src"], out[f"{tag}::target"], out[f"{tag}::lens"] = src.numpy(), target.numpy(), np.array(lens)
        print(tag, "loss", float(loss), "params", sum(p_.numel() for p_ in lm.parameters()), [k for k, _ in lm.named_parameters()][:4])
    out["pad"], out["eos"], out["V"] = np.array(dic.pad()), np.array(dic.eos()), np.array(len(dic))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def lm_data_fixture(name="ref_lm_data_tiny"):
    """The language-model data path of the reference on token files written by THIS repo's writer: fairseq's
    MMapIndexedDataset reads them, TokenBlockDataset (`eos` and `none` break modes) + MonolingualDataset build the samples the
    way LanguageModelingTask.load_dataset does (fairseq/tasks/language_modeling.py:185-262), and the batch plan is the one of
    FairseqTask.get_batch_iterator (ordered_indices under numpy_seed(seed) -> filter_indices_by_size -> batch_by_size with the
    reference's own Cython planner, built by oracle/build_ref_cython.py)."""
    import tempfile

    sys.path.insert(2, ROOT)
    sys.path.insert(2, HERE)
    import build_ref_cython

    build_ref_cython.attach()
    import types

    if "compat" not in np.__dict__:  # removed in numpy 2; token_block_dataset.py:125 still spells np.compat.long
        np.compat = types.SimpleNamespace(long=int)
    from espresso.data.asr_dictionary import AsrDictionary
    from fairseq.data import MonolingualDataset, TokenBlockDataset, data_utils
    from fairseq.data.indexed_dataset import MMapIndexedDataset

    from espresso_amd.data.lm_dataset import MMapTokenFile

    dic = AsrDictionary()
    for i in range(30):
        dic.add_symbol(f"w{i}")
    rng = np.random.default_rng(21)
    sents = [np.concatenate((rng.integers(4, len(dic), size=int(rng.integers(1, 12))), [dic.eos()])).astype(np.int64) for _ in range(41)]
    tmp = tempfile.mkdtemp(dir=os.path.join(HERE, "_ref"))
    prefix = os.path.join(tmp, "train")
    MMapTokenFile.write(prefix, sents, dtype=np.int32)
    ds = MMapIndexedDataset(prefix)
    assert len(ds) == len(sents) and all(np.array_equal(ds[i].numpy(), sents[i]) for i in range(len(sents)))
    out = {"n_sent": np.array(len(sents)), "flat": np.concatenate(sents), "sizes": np.array([len(x) for x in sents]),
           "pad": np.array(dic.pad()), "eos": np.array(dic.eos()), "V": np.array(len(dic))}
    for mode, tps in (("eos", 8), ("none", 7)):
        blocks = TokenBlockDataset(ds, ds.sizes, tps, pad=dic.pad(), eos=dic.eos(), break_mode=mode, include_targets=True,
                                   use_plasma_view=False)
        add_eos = mode is not None and mode != "none"
        mono = MonolingualDataset(dataset=blocks, sizes=blocks.sizes, src_vocab=dic, tgt_vocab=dic, add_eos_for_other_targets=add_eos,
                                  shuffle=True, targets=["future"], add_bos_token=False)
        items = [mono[i] for i in range(len(mono))]
        out[f"{mode}::n"] = np.array(len(mono))
        out[f"{mode}::sizes"] = np.asarray(mono.sizes)
        out[f"{mode}::src"] = np.concatenate([it["source"].numpy() for it in items])
        out[f"{mode}::tgt"] = np.concatenate([it["target"].numpy() for it in items])
        out[f"{mode}::len"] = np.array([len(it["source"]) for it in items])
        for seed in (1, 5):
            with data_utils.numpy_seed(seed):
                idx = mono.ordered_indices()
            out[f"{mode}::order::{seed}"] = np.asarray(idx)
            idx, _ = mono.filter_indices_by_size(idx, tps)
            batches = mono.batch_by_size(idx, max_tokens=40, max_sentences=6, required_batch_size_multiple=4)  # fairseq_dataset.py:144-188
            out[f"{mode}::batches::{seed}"] = np.concatenate(batches)
            out[f"{mode}::batch_sizes::{seed}"] = np.array([len(b) for b in batches])
        b = mono.collater([mono[i] for i in (0, 3, 2)])
        out[f"{mode}::collate::src"], out[f"{mode}::collate::tgt"] = b["net_input"]["src_tokens"].numpy(), b["target"].numpy()
        out[f"{mode}::collate::lens"], out[f"{mode}::collate::ntokens"] = b["net_input"]["src_lengths"].numpy(), np.array(b["ntokens"])
        print(mode, len(mono), [len(x) for x in batches][:8])
    import shutil
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def lookahead_fixture(name="ref_lookahead_wordlm_tiny"):
    """Look-ahead word LM (espresso/models/tensorized_lookahead_language_model.py) over a tiny character lexicon: the
    tensorized prefix tree, the word LSTM LM weights, and the sub-word log-probs the reference emits along scripted
    hypotheses (including an OOV path, word ends and a beam reorder)."""
    import argparse
    from espresso.data.asr_dictionary import AsrDictionary
    from espresso.models.lstm_lm import LSTMLanguageModelEspresso, base_lm_architecture
    from espresso.models.tensorized_lookahead_language_model import TensorizedLookaheadLanguageModel

    torch.manual_seed(99)
    words = sorted(["A", "AB", "ABC", "ABD", "B", "BA", "BAD", "BADE", "CAB", "DAD", "DEAD", "DEED", "E", "EBB", "ACE", "BEAD"])
    chars = ["A", "B", "C", "D", "E", "F"]
    wd = AsrDictionary()
    for w in words:
        wd.add_symbol(w)
    sd_ = AsrDictionary()
    for c in chars:
        sd_.add_symbol(c)
    sd_.add_symbol("<space>")
    sd_.space_index = sd_.indices["<space>"]  # AsrDictionary.load sets it (asr_dictionary.py:86)
    wd.space_index = -1

    class T:
        pass
    T.source_dictionary = T.target_dictionary = T.word_dictionary = wd
    args = argparse.Namespace(decoder_embed_dim=16, decoder_hidden_size=24, decoder_layers=2, decoder_out_embed_dim=24, dropout=0.0,
                              share_embed=False, is_wordlm=True, criterion_name="cross_entropy", tokens_per_sample=64)
    base_lm_architecture(args)
    word_lm = LSTMLanguageModelEspresso.build_model(args, T)
    with torch.no_grad():
        for p_ in word_lm.parameters():
            p_.mul_(4.0)  # U(-0.4, 0.4): a word distribution that is far from uniform
    word_lm.eval()
    la = TensorizedLookaheadLanguageModel(word_lm, sd_, oov_penalty=1e-4, open_vocab=True)
    la.eval()
    tree = la.decoder.tree
    sp, eos = sd_.space(), sd_.eos()
    ci = {c: sd_.index(c) for c in chars}
    # four hypotheses, 9 steps after the initial <eos>
    script = [
        [ci["A"], ci["B"], sp, ci["B"], ci["A"], ci["D"], sp, ci["E"], sp],
        [ci["D"], ci["E"], ci["E"], ci["D"], sp, ci["A"], ci["C"], ci["E"], sp],
        [ci["F"], ci["F"], sp, ci["C"], ci["A"], ci["B"], sp, ci["B"], sp],      # OOV word first (F is in no word)
        [ci["B"], ci["E"], ci["A"], ci["D"], sp, ci["D"], ci["A"], ci["C"], sp],  # leaves the tree at the last letters
    ]
    B = len(script)
    inc = {}
    toks = torch.full((B, 1), eos, dtype=torch.long)
    outs, orders, last_tok = [], [], []
    with torch.no_grad():
        for step in range(len(script[0]) + 1):
            lp, _ = la.decoder(toks, incremental_state=inc)
            outs.append(lp.squeeze(1).numpy().copy())
            last_tok.append(toks[:, -1].numpy().copy())
            if step == len(script[0]):
                break
            order = torch.arange(B)
            if step == 4:  # a beam reorder in the middle: hypotheses 0 and 3 swap places, 1 is duplicated over 2
                order = torch.tensor([3, 1, 1, 0])
                script = [script[int(i)] for i in order]
                toks = toks.index_select(0, order)
                la.decoder.reorder_incremental_state(inc, order)
                word_lm.decoder.reorder_incremental_state(inc, order)
            orders.append(order.numpy())
            toks = torch.cat([toks, torch.tensor([[script[b][step]] for b in range(B)])], 1)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), words=np.array(words), chars=np.array(chars), tokens=toks.numpy(),
                        orders=np.stack(orders), lprobs=np.stack(outs), last_tok=np.stack(last_tok), tree_children=tree.children.numpy(),
                        tree_prev_subword_idx=tree.prev_subword_idx.numpy(), tree_word_idx=tree.word_idx.numpy(),
                        tree_word_set_idx=tree.word_set_idx.numpy(),
                        **{"sd::" + k: v.numpy() for k, v in word_lm.state_dict().items()})
    print(name, "nodes", tree.children.shape, "lprobs", np.stack(outs).shape, list(word_lm.state_dict().keys()))
    print(np.round(outs[1][0], 2))


def label_smoothing_fixture():
    from espresso.criterions.label_smoothed_cross_entropy_v2 import label_smoothed_nll_loss

    torch.manual_seed(7)
    V, M = 37, 23
    logits = torch.randn(M, V) * 2
    target = torch.randint(2, V, (M,))
    target[[3, 11]] = 1  # pad
    lprobs = torch.log_softmax(logits, -1)
    res = {}
    for eps in (0.0, 0.1):
        loss, nll = label_smoothed_nll_loss(lprobs, target.unsqueeze(-1), eps, ignore_index=1, reduce=True)
        res[f"loss_{eps}"] = loss.item()
        res[f"nll_{eps}"] = nll.item()
    lg = logits.clone().requires_grad_(True)
    loss, _ = label_smoothed_nll_loss(torch.log_softmax(lg, -1), target.unsqueeze(-1), 0.1, ignore_index=1, reduce=True)
    loss.backward()
    # unigram / temporal smoothing (:49-119): rows are (b, u) with B = 1 sentence of M tokens
    from espresso.criterions.label_smoothed_cross_entropy_v2 import temporal_label_smoothing_prob_mask
    extra = {}
    prior = torch.rand(V) + 0.1
    prior = (prior / prior.sum()).unsqueeze(-1)
    target_uniform = target.clone()
    target = target.clone()
    target[20] = target[18]  # equal neighbours: their weights must add up
    for kind, kw in (("unigram", dict(unigram_tensor=prior)),
                     ("temporal", dict(prob_mask=temporal_label_smoothing_prob_mask(lprobs.view(1, M, V), target.view(1, M), 1)))):
        lg2 = logits.clone().requires_grad_(True)
        loss2, nll2 = label_smoothed_nll_loss(torch.log_softmax(lg2, -1), target.unsqueeze(-1), 0.1, ignore_index=1, reduce=True,
                                              smoothing_type=kind, **kw)
        loss2.backward()
        extra[f"{kind}_loss"], extra[f"{kind}_nll"], extra[f"{kind}_dlogits"] = np.array(loss2.item()), np.array(nll2.item()), lg2.grad.numpy()
    np.savez(os.path.join(OUT, "label_smoothing.npz"), logits=logits.numpy(), target=target_uniform.numpy(), target2=target.numpy(),
             prior=prior.squeeze(-1).numpy(), dlogits=lg.grad.numpy(), **{k: np.array(v) for k, v in res.items()}, **extra)
    print("label smoothing", res)


def specaug_fixture():
    """AdaptiveSpecAugment mask parameters + result under numpy_seed(seed, epoch, index)."""
    from espresso.data.feature_transforms.adaptive_specaugment import AdaptiveSpecAugmentTransform
    from fairseq.data import data_utils

    tr = AdaptiveSpecAugmentTransform.from_config_dict(
        {"freq_mask_N": 2, "freq_mask_F": 27, "time_mask_pm": 0.04, "time_mask_ps": 0.04})
    rng = np.random.default_rng(3)
    outs = {}
    for k, (M, idx) in enumerate([(313, 5), (1000, 17), (40, 2)]):
        spec = rng.standard_normal((M, 80))
        with data_utils.numpy_seed(1, 2, idx):
            o = tr(spec)
        outs[f"in_{k}"] = spec
        outs[f"out_{k}"] = o
        outs[f"meta_{k}"] = np.array([M, idx])
    np.savez_compressed(os.path.join(OUT, "specaug.npz"), **outs)
    print("specaug ok")


def batch_by_size_fixture(name="ref_batch_by_size"):
    """The reference's own Cython planner (fairseq/data/data_utils_fast.pyx: batch_by_size_vec and batch_by_size_fn, reached
    through fairseq.data.data_utils.batch_by_size :297-360; built by oracle/build_ref_cython.py) on seeded size vectors: the
    recipe settings (26000 frames / 24 sentences / multiple 1), multiples of 8, no sentence cap, tiny budgets, samples larger
    than max_tokens are excluded by construction (the planner asserts)."""
    sys.path.insert(2, HERE)
    import build_ref_cython

    build_ref_cython.attach()
    from fairseq.data import data_utils

    rng = np.random.default_rng(77)
    out, case = {}, 0
    for n, lo, hi in ((400, 100, 3500), (257, 1, 40), (64, 5, 6), (1, 10, 11)):
        sizes = rng.integers(lo, hi, size=n).astype(np.int64)
        order = np.argsort(sizes, kind="mergesort")
        for max_tokens, max_sentences, mult in ((26000, 24, 1), (26000, 24, 8), (8000, None, 8), (None, 7, 1), (4000, 3, 2), (3600, 64, 4)):
            if max_tokens is not None and int(sizes.max()) > max_tokens:
                continue
            vec = data_utils.batch_by_size(order, None, num_tokens_vec=sizes[order], max_tokens=max_tokens, max_sentences=max_sentences,
                                           required_batch_size_multiple=mult)
            fn = data_utils.batch_by_size(order, lambda i: int(sizes[i]), num_tokens_vec=None, max_tokens=max_tokens,
                                          max_sentences=max_sentences, required_batch_size_multiple=mult)
            assert [b.tolist() for b in vec] == [b.tolist() for b in fn]
            out[f"{case}::sizes"], out[f"{case}::order"] = sizes, order
            out[f"{case}::args"] = np.array([-1 if max_tokens is None else max_tokens, -1 if max_sentences is None else max_sentences, mult])
            out[f"{case}::flat"], out[f"{case}::lens"] = np.concatenate(vec), np.array([len(b) for b in vec])
            case += 1
    out["n_cases"] = np.array(case)
    print("batch_by_size cases", case)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def checkpoint_rules_fixture(name="ref_checkpoint_rules"):
    """Directory listings after each call of the reference's `checkpoint_utils.save_checkpoint` (fairseq/checkpoint_utils.py:34-172)
    over a scripted run: mid-epoch saves every 2 updates, epoch ends, validation scores that improve / worsen / tie, with
    keep_interval_updates=2, keep_last_epochs=2, keep_best_checkpoints=2 (minimising `wer`), then a maximising metric."""
    import json
    import tempfile

    from fairseq import checkpoint_utils
    from fairseq.dataclass.configs import CheckpointConfig

    class Itr:
        def __init__(self):
            self.epoch, self._end = 1, False

        def end_of_epoch(self):
            return self._end

        def state_dict(self):
            return {"epoch": self.epoch}

    class Tr:
        data_parallel_rank = 0
        should_save_checkpoint_on_current_rank = True
        always_call_state_dict_during_save_checkpoint = False
        checkpoint_suffix = ""

        def __init__(self):
            self.n = 0

        def get_num_updates(self):
            return self.n

        def consolidate_optimizer(self):
            pass

        def save_checkpoint(self, filename, extra_state):
            with open(filename, "w") as f:
                f.write("x")

    script = [  # (epoch, end_of_epoch, num_updates, val_loss)
        (1, False, 2, 30.0), (1, False, 4, 35.0), (1, True, 5, 28.0), (2, False, 6, 28.0), (2, False, 8, 27.5), (2, True, 10, None),
        (3, False, 12, 40.0), (3, True, 15, 26.123), (4, True, 20, 26.5), (5, False, 22, 25.0), (5, True, 25, 25.0)]
    out = {}
    for tag, maximize in (("min", False), ("max", True)):
        tmp = tempfile.mkdtemp(dir=os.path.join(HERE, "_ref"))
        cfg = CheckpointConfig()
        cfg.save_dir, cfg.save_interval_updates, cfg.keep_interval_updates, cfg.keep_last_epochs = tmp, 2, 2, 2
        cfg.keep_best_checkpoints, cfg.best_checkpoint_metric, cfg.maximize_best_checkpoint_metric = 2, "wer", maximize
        if hasattr(checkpoint_utils.save_checkpoint, "best"):
            del checkpoint_utils.save_checkpoint.best
        tr, itr, listings = Tr(), Itr(), []
        for epoch, end, n, val in script:
            itr.epoch, itr._end, tr.n = epoch, end, n
            checkpoint_utils.save_checkpoint(cfg, tr, itr, val)
            listings.append(sorted(os.listdir(tmp)))
        out[tag] = {"listings": listings, "best": checkpoint_utils.save_checkpoint.best}
        import shutil
        shutil.rmtree(tmp)
    out["script"] = script
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out["min"]["listings"][-1], out["min"]["best"], out["max"]["best"])


def lr_schedule_fixture(name="ref_lr_schedules"):
    """Learning-rate trajectories of the reference's own scheduler classes with the recipes' settings: `noam`
    (espresso/optim/lr_scheduler/noam_lr_scheduler.py), `tri_stage` (fairseq), `polynomial_decay_v2` (espresso),
    `reduce_lr_on_plateau_v2` (espresso over fairseq + torch's ReduceLROnPlateau) and fairseq's `reduce_lr_on_plateau`, driven by
    scripted update counts and validation scores."""
    import json
    from types import SimpleNamespace as NS

    from espresso.optim.lr_scheduler.noam_lr_scheduler import NoamLRScheduler as NoamSchedule
    from espresso.optim.lr_scheduler.polynomial_decay_schedule import PolynomialDecayV2LRSchedule as PolynomialDecayLRScheduleV2
    from espresso.optim.lr_scheduler.reduce_lr_on_plateau_v2 import ReduceLROnPlateauLRScheduleV2
    from fairseq.optim.lr_scheduler.reduce_lr_on_plateau import ReduceLROnPlateauLRSchedule
    from fairseq.optim.lr_scheduler.tri_stage_lr_scheduler import TriStageLRSchedule

    from fairseq.optim import FairseqOptimizer

    class Opt(FairseqOptimizer):  # a real FairseqOptimizer over one dummy parameter (the schedulers insist on the type)
        def __init__(self):
            super().__init__(None)
            self._optimizer = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)

    updates = [0, 1, 2, 10, 100, 999, 1000, 1001, 5000, 24999, 25000, 25001, 60000, 180000, 181000, 400000, 541000, 600000]
    out = {"updates": updates}
    cases = {
        "noam": (NoamSchedule, NS(lr=[5.0], warmup_steps=25000, model_size=512, final_lr=1e-6)),
        "noam_small": (NoamSchedule, NS(lr=[2.0], warmup_steps=10, model_size=128, final_lr=5e-3)),
        "tri_stage": (TriStageLRSchedule, NS(lr=[0.001], warmup_steps=1000, hold_steps=180000, decay_steps=360000, phase_ratio=None,
                                             init_lr_scale=0.01, final_lr_scale=0.05, max_update=0)),
        "tri_stage_ratio": (TriStageLRSchedule, NS(lr=[0.002], warmup_steps=0, hold_steps=0, decay_steps=0, phase_ratio=(0.1, 0.4, 0.5),
                                                   init_lr_scale=0.01, final_lr_scale=0.01, max_update=100000)),
        "polynomial_decay_v2": (PolynomialDecayLRScheduleV2, NS(lr=[0.003], warmup_updates=1000, force_anneal=None, end_learning_rate=1e-5,
                                                                power=2.0, total_num_update=400000)),
    }
    for tag, (cls, cfg) in cases.items():
        opt = Opt()
        sch = cls(cfg, opt)
        traj = []
        for n in updates:
            sch.step_update(n)
            traj.append(opt.get_lr())
        out[tag] = {"cfg": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(cfg).items()}, "lr": traj}
    scores = [10.0, 9.0, 9.5, 9.4, 8.0, 8.0, 8.1, 7.0, 7.5, 7.6, 7.7, 6.0]
    plate = {
        "reduce_lr_on_plateau_v2": (ReduceLROnPlateauLRScheduleV2, NS(lr=[0.001], lr_shrink=0.5, lr_threshold=1e-4, lr_patience=0, warmup_updates=900,
                                                                      warmup_init_lr=1e-5, start_reduce_lr_epoch=4, final_lr_scale=0.1,
                                                                      maximize_best_checkpoint_metric=False)),
        "reduce_lr_on_plateau": (ReduceLROnPlateauLRSchedule, NS(lr=[0.001], lr_shrink=0.5, lr_threshold=1e-4, lr_patience=0, warmup_updates=0,
                                                                 warmup_init_lr=-1, maximize_best_checkpoint_metric=False)),
        "reduce_lr_on_plateau_v2_max": (ReduceLROnPlateauLRScheduleV2, NS(lr=[0.002], lr_shrink=0.3, lr_threshold=1e-4, lr_patience=1, warmup_updates=0,
                                                                          warmup_init_lr=-1, start_reduce_lr_epoch=0, final_lr_scale=0.01,
                                                                          maximize_best_checkpoint_metric=True)),
    }
    for tag, (cls, cfg) in plate.items():
        opt = Opt()
        sch = cls(cfg, opt)
        warm = []
        for n in (0, 1, 450, 899, 900, 901, 2000):
            sch.step_update(n)
            warm.append(opt.get_lr())
        epochs = []
        for e, v in enumerate(scores, start=1):
            sch.step(e, -v if cfg.maximize_best_checkpoint_metric else v)
            sch.step_update(2000 + e)
            epochs.append(opt.get_lr())
        out[tag] = {"cfg": vars(cfg), "warm_updates": [0, 1, 450, 899, 900, 901, 2000], "warm_lr": warm, "scores": scores, "epoch_lr": epochs}
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: (v["lr"][-3:] if "lr" in v else v["epoch_lr"][-3:]) for k, v in out.items() if isinstance(v, dict)})


def wer_scorer_fixture(name="ref_wer_scorer"):
    """The reference's Scorer (espresso/tools/wer.py:16-140) on scripted (reference, hypothesis) token strings in character units
    with <space> word boundaries, non-language symbols and a WER output filter: totals, WER / CER after every utterance."""
    import json
    import tempfile

    from espresso.data.asr_dictionary import AsrDictionary
    from espresso.tools.wer import Scorer

    tmp = tempfile.mkdtemp(dir=os.path.join(HERE, "_ref"))
    with open(os.path.join(tmp, "dict.txt"), "w") as f:
        f.write("".join(f"{c} 1\n" for c in "abcdefghijklmnopqrstuvwxyz'") + "<space> 1\n<noise> 1\n<laugh> 1\n")
    with open(os.path.join(tmp, "nlsyms.txt"), "w") as f:
        f.write("<noise>\n<laugh>\n")
    with open(os.path.join(tmp, "filter"), "w") as f:
        f.write("#!/bin/sed -f\ns/uh //g\ns: um::g\n")
    pairs = [
        ("u1", "t h e <space> c a t <space> s a t", "t h e <space> c a t <space> s a t"),
        ("u2", "a <space> b i g <space> d o g", "a <space> b a g <space> d o g s"),
        ("u3", "<noise> h e l l o <space> w o r l d", "h e l o <space> <laugh> w o r l d"),
        ("u4", "u h <space> y e s <space> u m", "y e s"),
        ("u5", "o n e <space> t w o <space> t h r e e", "o n e <space> t h r e e <space> f o u r <space> f i v e"),
        ("u6", "x", ""),
    ]
    out = {"pairs": pairs}
    for tag, filt in (("plain", None), ("filtered", os.path.join(tmp, "filter"))):
        import argparse
        dic = AsrDictionary.load(os.path.join(tmp, "dict.txt"), f_non_lang_syms=os.path.join(tmp, "nlsyms.txt"))
        dic.build_bpe(argparse.Namespace(bpe="characters_asr"))  # character units: words are the <space>-separated groups
        sc = Scorer(dic, wer_output_filter=filt)
        steps = []
        for utt, ref, hyp in pairs:
            sc.add_evaluation(utt, ref, hyp)
            steps.append({"wer": list(sc.wer()), "cer": list(sc.cer()), "word_error": sc.tot_word_error(), "word_count": sc.tot_word_count(),
                          "char_error": sc.tot_char_error(), "char_count": sc.tot_char_count()})
        out[tag] = steps
    import shutil
    shutil.rmtree(tmp)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out["plain"][-1], out["filtered"][-1])


def dictionary_fixture(name="ref_asr_dictionary"):
    """The reference's AsrDictionary (espresso/data/asr_dictionary.py) + `characters_asr` encoder + `tokenize`
    (espresso/tools/utils.py) on scripted sentences: symbol layout with and without <s>, text -> pieces -> ids -> string ->
    text, unknown characters, non-language symbols kept whole."""
    import argparse
    import json
    import tempfile

    from espresso.data.asr_dictionary import AsrDictionary
    from espresso.tools.utils import tokenize

    tmp = tempfile.mkdtemp(dir=os.path.join(HERE, "_ref"))
    with open(os.path.join(tmp, "dict.txt"), "w") as f:
        f.write("".join(f"{c} {i + 3}\n" for i, c in enumerate("abcdefghijklmnopqrstuvwxyz'")) + "<space> 9\n<noise> 2\n<laugh> 1\n")
    with open(os.path.join(tmp, "nlsyms.txt"), "w") as f:
        f.write("<noise>\n<laugh>\n")
    texts = ["hello world", "it's <noise> a dog", "  two  spaces ", "caf3 ok", "<laugh>", "", "a"]
    out = {"texts": texts}
    for tag, bos in (("bos", True), ("nobos", False)):
        d = AsrDictionary.load(os.path.join(tmp, "dict.txt"), enable_bos=bos, f_non_lang_syms=os.path.join(tmp, "nlsyms.txt"))
        d.build_bpe(argparse.Namespace(bpe="characters_asr"))
        rows = []
        for t in texts:
            pieces = d.wordpiece_encode(t)
            ids = d.encode_line(pieces, add_if_not_exist=False, append_eos=True).tolist()
            s_ = d.string(torch.tensor(ids))
            rows.append({"pieces": pieces, "ids": ids, "string": s_, "decoded": d.wordpiece_decode(s_),
                         "tokenize": tokenize(t, space=d.space_word, non_lang_syms=d.non_lang_syms)})
        out[tag] = {"len": len(d), "pad": d.pad(), "eos": d.eos(), "unk": d.unk(), "bos": d.bos() if bos else None, "space": d.space(),
                    "symbols": d.symbols, "count": list(d.count), "rows": rows}
    import shutil
    shutil.rmtree(tmp)
    with open(os.path.join(OUT, name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out["bos"]["symbols"][:6], out["nobos"]["symbols"][:5], out["bos"]["rows"][1])


def collate_fixture(name="ref_asr_collate"):
    """The reference's `collate` (espresso/data/asr_dataset.py:17-136) on scripted feature samples: length sort, frame / token
    padding, input feeding with </s> moved to the front or <s> prepended, pad_to_multiple, samples without targets."""
    from espresso.data.asr_dataset import collate

    rng = np.random.default_rng(9)
    pad, eos, bos = 1, 2, 0
    lens, tl = [7, 12, 5, 12, 9], [3, 5, 2, 4, 6]
    samples = []
    for i, (n, m) in enumerate(zip(lens, tl)):
        tgt = np.concatenate((rng.integers(4, 30, size=m), [eos]))
        samples.append({"id": i, "utt_id": f"utt{i}", "source": torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32)),
                        "target": torch.from_numpy(tgt), "text": f"text {i}"})
    out = {"n": np.array(len(samples)), "pad": np.array(pad), "eos": np.array(eos), "bos": np.array(bos)}
    for i, smp in enumerate(samples):
        out[f"in::{i}::source"], out[f"in::{i}::target"] = smp["source"].numpy(), smp["target"].numpy()
    cases = {"feed_eos": dict(input_feeding=True), "feed_bos": dict(input_feeding=True, maybe_bos_idx=bos),
             "no_feed": dict(input_feeding=False), "mult4": dict(input_feeding=True, pad_to_multiple=4)}
    for tag, kw in cases.items():
        b = collate(samples, pad_idx=pad, eos_idx=eos, left_pad_source=False, left_pad_target=False, **kw)
        out[f"{tag}::id"], out[f"{tag}::src"], out[f"{tag}::src_lengths"] = b["id"].numpy(), b["net_input"]["src_tokens"].numpy(), b["net_input"]["src_lengths"].numpy()
        out[f"{tag}::target"], out[f"{tag}::ntokens"] = b["target"].numpy(), np.array(b["ntokens"])
        out[f"{tag}::utt_id"] = np.array(b["utt_id"])
        if "prev_output_tokens" in b["net_input"]:
            out[f"{tag}::prev"] = b["net_input"]["prev_output_tokens"].numpy()
    nt = [{k: v for k, v in smp.items() if k not in ("target", "text")} for smp in samples]
    b = collate(nt, pad_idx=pad, eos_idx=eos)
    out["notgt::id"], out["notgt::src_lengths"], out["notgt::keys"] = b["id"].numpy(), b["net_input"]["src_lengths"].numpy(), np.array(sorted(b.keys()))
    print({k: v.shape for k, v in out.items() if k.endswith("::src")}, out["feed_eos::id"], out["notgt::keys"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def epoch_iterator_fixture(name="ref_epoch_batches"):
    """Per-rank batch order of the reference's EpochBatchIterator (fairseq/data/iterators.py:262-520: frozen batches shuffled with
    `seed + epoch`, then ShardedIterator with empty fill) for 2 epochs x 4 shards (and 1 shard), built on the reference's own
    batch plan (ordered by length, Cython batch_by_size)."""
    sys.path.insert(2, HERE)
    import build_ref_cython

    build_ref_cython.attach()
    from fairseq.data import data_utils, iterators

    rng = np.random.default_rng(5)
    sizes = rng.integers(100, 3000, size=150).astype(np.int64)
    order = np.argsort(sizes, kind="mergesort")
    batches = data_utils.batch_by_size(order, None, num_tokens_vec=sizes[order], max_tokens=12000, max_sentences=8, required_batch_size_multiple=1)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(sizes)

        def __getitem__(self, i):
            return int(i)

        def set_epoch(self, e):
            pass

    out = {"sizes": sizes, "order": order, "args": np.array([12000, 8, 1, 3])}
    for shards in (1, 4):
        for shard in range(shards):
            itr = iterators.EpochBatchIterator(DS(), collate_fn=lambda x: x, batch_sampler=batches, seed=3, num_shards=shards, shard_id=shard,
                                               num_workers=0, epoch=1)
            for epoch in (1, 2):
                got = [list(b) for b in itr.next_epoch_itr(shuffle=True)]
                out[f"{shards}::{shard}::{epoch}::flat"] = np.array([i for b in got for i in b], dtype=np.int64)
                out[f"{shards}::{shard}::{epoch}::lens"] = np.array([len(b) for b in got])
    print("batches", len(batches), "per shard of 4:", len(out["4::0::1::lens"]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "encdec":
        encdec_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dh64":  # head dim 64: the fused-attention / native-runtime shape class
        encoder_fixture("conformer", "ref_conformer_ctc_dh64", d=128, heads=2, ffn=256, frames=300)
        encoder_fixture("transformer", "ref_transformer_ctc_dh64", d=128, heads=2, ffn=256, frames=300)
        encdec_fixture("ref_transformer_encdec_dh64", dm=128, heads=2, ffn=256, frames=300)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "encdec_trained":
        encdec_trained_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ensemble":
        ensemble_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "legacy":  # encoder configurations of the argparse presets / streaming options
        encoder_fixture("transformer", "ref_transformer_ctc_legacy", d=128, heads=2, ffn=256, frames=150, legacy={})
        encoder_fixture("transformer", "ref_transformer_ctc_postln_chunk", legacy=dict(
            learned_pos=True, normalize_before=False, chunk_size=4, chunk_left_window=1, chunk_right_window=0))
        encoder_fixture("conformer", "ref_conformer_ctc_abspos", legacy={})
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "speechlstm":
        speech_lstm_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "learnedpos":
        encoder_fixture("transformer", "ref_transformer_learnedpos_ctc_tiny", learned_pos=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lmfusion":
        lm_fusion_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "collate":
        collate_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dictionary":
        dictionary_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wer":
        wer_scorer_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lrsched":
        lr_schedule_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ckptrules":
        checkpoint_rules_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "epochitr":
        epoch_iterator_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "batchbysize":
        batch_by_size_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lmdata":
        lm_data_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lmtrain":
        lm_train_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lookahead":
        lookahead_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "transducer":
        transducer_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dropout":
        dropout_fixtures()
        sys.exit(0)
    encoder_fixture("conformer", "ref_conformer_ctc_tiny")
    encoder_fixture("transformer", "ref_transformer_ctc_tiny")
    label_smoothing_fixture()
    specaug_fixture()
