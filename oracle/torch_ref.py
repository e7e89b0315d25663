"""TEST INFRASTRUCTURE ONLY — fp32 CPU restatement of the reference's encoder arithmetic, driven by a
reference-format state_dict.  Pinned against the golden fixtures produced by the reference's own
modules (tests/golden/ref_*_ctc_tiny.npz, see oracle/gen_golden.py) in tests/test_oracle.py.

Each function cites the reference code it follows (paths relative to the reference root)."""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F


# ---- bf16 emulation of the HIP path's storage points -------------------------------------------------------------------
# The HIP path computes in fp32 (MFMA accumulators, LayerNorm / softmax / BatchNorm statistics) and rounds to bf16 exactly where
# a tensor is STORED to HBM: the bf16 weight shadows, every GEMM output (after bias / activation / residual in the fused
# epilogue), LayerNorm / BatchNorm / GLU / depthwise-conv outputs, the scaled queries, the attention probabilities that multiply
# V, the attention output.  With `bf16_emulation(True)` this restatement rounds at the same points (forward value AND the
# gradient flowing back through the point — the HIP backward stores its gradients in bf16 at the same tensor boundaries), so a
# comparison HIP vs emulation isolates real arithmetic differences from the expected bf16 rounding of the reference's fp32 run.
_EMU = {"on": False, "flash": True, "resid_f32": False, "joint_logits_f32": False}


class bf16_emulation:
    """Context manager: `with torch_ref.bf16_emulation(flash=True): torch_ref.encoder(...)`.  `flash`: head dim 64 (fused
    kernels: un-normalised bf16 probabilities, fp32 normaliser) vs the unfused path (normalised probabilities rounded)."""

    def __init__(self, on=True, flash=True, resid_f32=False, joint_logits_f32=False):
        # resid_f32: the residual stream (every `x + block(x)` and the layer's final LayerNorm output) stays fp32 — what the
        # reference's AMP run does (fairseq/tasks/fairseq_task.py:516: LayerNorm and the residual adds run in fp32 under autocast)
        # joint_logits_f32: the transducer's lattice logits are not rounded — what the `transducer_loss` criterion sees on the HIP path
        # since round 6 (csrc/joint_rnnt.hip: output layer fused with the loss on the fp32 accumulators; the logits' GRADIENT is
        # still stored in bf16)
        self.new = {"on": on, "flash": flash, "resid_f32": resid_f32, "joint_logits_f32": joint_logits_f32}

    def __enter__(self):
        self.old = dict(_EMU)
        _EMU.update(self.new)

    def __exit__(self, *a):
        _EMU.update(self.old)


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _r(x):
    """Storage point: identity in the fp32 restatement, bf16 round trip under emulation."""
    return _RoundBF16.apply(x) if _EMU["on"] and x.is_floating_point() else x


def _rs(x):
    """Storage point of the RESIDUAL STREAM: bf16 under emulation unless the stream is kept in fp32 (`resid_f32`) AND has become
    fp32 — under the reference's autocast the stream is bf16 (Linear outputs) until the first LayerNorm output joins it, i.e.
    from the first Conformer layer's final LayerNorm on (`stream_hot`)."""
    return x if (_EMU.get("resid_f32") and _EMU.get("stream_hot")) else _r(x)


class _RoundGradBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _rg(x):
    """A value the HIP path keeps in fp32 whose GRADIENT it stores in bf16 (the encoder's vocabulary logits under EA_LOGITS_F32=1:
    the output GEMM writes fp32 rows, the loss gradient is cast to bf16 for the weight / data gradient GEMMs)."""
    return _RoundGradBF16.apply(x) if _EMU["on"] and x.is_floating_point() else x


# ---- training-mode dropout with the HIP path's masks ---------------------------------------------------------------------
# The reference applies FairseqDropout (x * bernoulli(1-p) / (1-p), torch's Philox stream) at the sites cited below.  The HIP path
# draws its keep decisions from a counter-based hash instead; with `dropout_masks(plan)` active this restatement applies the
# reference's dropout at the reference's sites using the HIP path's decisions (oracle/dropout_ref.py rebuilds every mask from
# the (site, seed) list the HIP forward reported).  Without a plan every site is the identity (dropout 0 / eval).
_DROP = {"plan": None}


class dropout_masks:
    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        self.old = _DROP["plan"]
        _DROP["plan"] = self.plan

    def __exit__(self, *a):
        _DROP["plan"] = self.old


def _drop(x, site, layout, **kw):
    """FairseqDropout at `site` on x.  layout: how x relates to the HIP path's dense [rows m = b*T + t][channels] activation:
    'TBC', 'BTC', 'BCT' (convolution module), 'ZTS' (attention probabilities, z = b*H + h; needs B=, H=), 'SUB' (sub-sampler
    output (B, T', C*F'), reference feature order c*F' + f; needs C=)."""
    plan = _DROP["plan"]
    if plan is None:
        return x
    from . import dropout_ref as D

    sp = plan.seed_for(site)
    if sp is None:
        return x
    seed, p = sp
    if layout == "TBC":
        m = D.rows_tbc(seed, p, *x.shape)
    elif layout == "BTC":
        m = D.rows_btc(seed, p, *x.shape)
    elif layout == "BCT":
        m = D.rows_btc(seed, p, x.shape[0], x.shape[2], x.shape[1]).transpose(1, 2)
    elif layout == "ZTS":
        m = D.probs_zts(seed, p, kw["B"], kw["H"], x.shape[1], x.shape[2])
    elif layout == "SUB":
        m = D.subsample_out(seed, p, x.shape[0], x.shape[1], x.shape[2] // kw["C"], kw["C"])
    else:
        raise ValueError(layout)
    return x * m


def _drop_mask(site, shape):
    """The multiplier (0 or 1/(1-p)) of one dropout call over a dense row-major tensor of `shape`, or None (no plan / p = 0 there)."""
    plan = _DROP["plan"]
    if plan is None:
        return None
    from . import dropout_ref as D

    sp = plan.seed_for(site)
    return None if sp is None else D.scale_mask(sp[0], tuple(shape), sp[1])


def _lin(x, w, b=None):
    """Linear on the (bf16 shadow of the) weight, fp32 accumulation; the caller rounds where the result is stored."""
    return F.linear(x, _r(w), b)


def sinusoidal_rel_pe(seq_len, dim):
    """espresso/modules/sinusoidal_relative_positional_embedding.py:46-71,112-124 with
    scale_embedding (relative_positional_embedding.py:28-34): rows = offsets -(L-1)..(L-1)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(seq_len, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    pos = torch.cat([torch.sin(e), torch.cos(e)], 1)
    neg = torch.cat([torch.sin(-e), torch.cos(-e)], 1)
    neg = torch.flip(neg, [0])
    return dim ** -0.5 * torch.cat([neg, pos[1:]], 0)


def relpos_mhsa(x, sd, prefix, H, key_padding_mask=None, attn_mask=None):
    """fairseq/modules/multihead_attention.py:650-907 (rel-pos branch).  x: (T, B, C).  Returns the out_proj output BEFORE the
    residual add (the HIP epilogue adds the residual in fp32 and stores once)."""
    T, B, C = x.shape
    dh = C // H
    scaling = dh ** -0.5
    q = _r(_lin(x, sd[prefix + "q_proj.weight"], sd[prefix + "q_proj.bias"]))
    k = _r(_lin(x, sd[prefix + "k_proj.weight"], sd[prefix + "k_proj.bias"]))
    v = _r(_lin(x, sd[prefix + "v_proj.weight"], sd[prefix + "v_proj.bias"]))
    relpos = (prefix + "pos_bias_u") in sd
    learned = (prefix + "positional_embedding.weight") in sd  # learned relative table: used as is, plain queries (:806-815)
    if learned:
        qv = _r(q * scaling).contiguous().view(T, B * H, dh).transpose(0, 1)
    if relpos:
        qv = _r((q + sd[prefix + "pos_bias_v"]) * scaling).contiguous().view(T, B * H, dh).transpose(0, 1)
        q = q + sd[prefix + "pos_bias_u"]
    q = _r(q * scaling).contiguous().view(T, B * H, dh).transpose(0, 1)
    k = k.contiguous().view(T, B * H, dh).transpose(0, 1)
    v = v.contiguous().view(T, B * H, dh).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))
    if relpos or learned:
        if learned:
            tab = sd[prefix + "positional_embedding.weight"]  # learned_relative_positional_embedding.py:71-80: centre slice
            start = tab.shape[0] // 2 - T + 1
            pe = _r(tab[start: start + 2 * T - 1])
            if pe.shape[1] != C:
                pe = pe.repeat(1, H)
        else:
            pe = _r(sinusoidal_rel_pe(T, C))
            pe = _r(_lin(pe, sd[prefix + "pos_proj.weight"]))  # (2T-1, C), same for every batch element
        pe = pe.view(1, 2 * T - 1, H, dh).expand(B, -1, -1, -1).transpose(1, 2).reshape(B * H, 2 * T - 1, dh)
        raw = torch.bmm(qv, pe.transpose(1, 2))  # (BH, T, 2T-1)
        i = torch.arange(T).unsqueeze(1)
        j = torch.arange(T).unsqueeze(0)
        w = w + raw.gather(2, ((T - 1) - i + j).unsqueeze(0).expand(B * H, -1, -1))
    if attn_mask is not None:
        w = w + attn_mask.unsqueeze(0)
    if key_padding_mask is not None:
        w = w.view(B, H, T, T).masked_fill(key_padding_mask[:, None, None, :], float("-inf")).view(B * H, T, T)
    # attention dropout on the softmax output (multihead_attention.py:874): the normaliser is the sum of ALL probabilities
    if _EMU["on"] and _EMU["flash"]:
        # fused kernels: bf16 un-normalised probabilities feed P.V, the normaliser is the fp32 sum of the un-rounded ones
        pu = torch.exp(w.float() - w.float().max(dim=-1, keepdim=True).values)
        a = torch.bmm(_r(_drop(pu, "attn.probs", "ZTS", B=B, H=H)), v) / pu.sum(-1, keepdim=True)
    else:
        p = _r(_drop(torch.softmax(w.float(), dim=-1), "attn.probs", "ZTS", B=B, H=H))
        a = torch.bmm(p, v)
    a = _r(a).transpose(0, 1).contiguous().view(T, B, C)
    return _lin(a, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def _ln(x, sd, prefix, stream=False):
    """`stream`: the output IS the residual stream of what follows (a Conformer layer's final LayerNorm)."""
    y = F.layer_norm(x, (x.shape[-1],), sd[prefix + "weight"], sd[prefix + "bias"], 1e-5)
    if stream and _EMU.get("resid_f32"):
        _EMU["stream_hot"] = True
    return _rs(y) if stream else _r(y)


def _ffn_conformer(x, sd, p):
    """fairseq/modules/conformer_layer.py:134-146 (swish)."""
    y = _ln(x, sd, p + "layer_norm.")
    y = _lin(y, sd[p + "w_1.weight"], sd[p + "w_1.bias"])
    # (the HIP epilogue applies the activation and the activation dropout (:144) to the fp32 accumulator and stores the result)
    y = _r(_drop(F.silu(y), "ffn.act", "TBC"))
    return _drop(_lin(y, sd[p + "w_2.weight"], sd[p + "w_2.bias"]), "ffn.out", "TBC")  # :146


def _bn(x, sd, p, training, dim_c=1, momentum=0.1, eps=1e-5, update=None):
    rm, rv = sd[p + "running_mean"].clone(), sd[p + "running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[p + "weight"], sd[p + "bias"], training, momentum, eps)
    if update is not None and training:
        update[p + "running_mean"], update[p + "running_var"] = rm, rv
    return y


def conv_module(x_btc, sd, p, training, update=None):
    """fairseq/modules/conformer_layer.py:79-101.  x: (B, T, C)."""
    y = _ln(x_btc, sd, p + "layer_norm.").transpose(1, 2)
    y = _r(F.conv1d(y, _r(sd[p + "pointwise_conv1.weight"])))
    y = _r(F.glu(y, dim=1))
    w = sd[p + "depthwise_conv.weight"]  # (fp32 in the HIP kernel)
    y = _r(F.conv1d(y, w, padding=(w.shape[-1] - 1) // 2, groups=w.shape[0]))
    y = _bn(y, sd, p + "batch_norm.", training, update=update)
    y = _r(F.silu(y))
    y = F.conv1d(y, _r(sd[p + "pointwise_conv2.weight"]))
    return _drop(y, "conv.out", "BCT").transpose(1, 2)  # :100


def conformer_layer(x, sd, p, H, key_padding_mask, training, update=None, attn_mask=None):
    """espresso/modules/conformer_with_relative_positional_embedding_encoder_layer.py:112-141.  x: (T,B,C)."""
    x = _rs(0.5 * _ffn_conformer(x, sd, p + "ffn1.") + x)
    # (…encoder_layer.py:125: dropout on the attention block's output, before the residual)
    x = _rs(_drop(relpos_mhsa(_ln(x, sd, p + "self_attn_layer_norm."), sd, p + "self_attn.", H, key_padding_mask, attn_mask),
                  "attn.out", "TBC") + x)
    x = _rs(conv_module(x.transpose(0, 1), sd, p + "conv_module.", training, update).transpose(0, 1) + x)
    x = _rs(0.5 * _ffn_conformer(x, sd, p + "ffn2.") + x)
    return _ln(x, sd, p + "final_layer_norm.", stream=True)


def transformer_layer(x, sd, p, H, key_padding_mask, activation="relu", attn_mask=None, normalize_before=True):
    """fairseq/modules/transformer_layer.py:163-226 (pre-LN; post-LN when normalize_before is False); dropout sites :196
    (attention block output), :212 (activation_dropout), :216 (FFN output)."""
    act = (lambda y: F.relu(y)) if activation == "relu" else (lambda y: F.silu(y))
    if not normalize_before:
        x = _ln(_r(_drop(relpos_mhsa(x, sd, p + "self_attn.", H, key_padding_mask, attn_mask), "attn.out", "TBC") + x), sd,
                p + "self_attn_layer_norm.")
        y = _lin(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"])
        y = _r(_drop(act(y), "ffn.act", "TBC"))
        y = _drop(_lin(y, sd[p + "fc2.weight"], sd[p + "fc2.bias"]), "ffn.out", "TBC")
        return _ln(_r(x + y), sd, p + "final_layer_norm.")
    x = _r(_drop(relpos_mhsa(_ln(x, sd, p + "self_attn_layer_norm."), sd, p + "self_attn.", H, key_padding_mask, attn_mask),
                 "attn.out", "TBC") + x)
    y = _ln(x, sd, p + "final_layer_norm.")
    y = _lin(y, sd[p + "fc1.weight"], sd[p + "fc1.bias"])
    y = _r(_drop(act(y), "ffn.act", "TBC"))
    y = _drop(_lin(y, sd[p + "fc2.weight"], sd[p + "fc2.bias"]), "ffn.out", "TBC")
    return _r(x + y)


def conv_bn_relu(feats, lengths, sd, p, strides, training, update=None):
    """espresso/modules/speech_convolutions.py:78-102."""
    B, T, Fd = feats.shape
    x = feats.view(B, T, 1, Fd).transpose(1, 2)
    out_len = lengths.clone()
    i = 0
    while (p + f"convolutions.{i}.weight") in sd:
        s = strides[i]
        w = sd[p + f"convolutions.{i}.weight"]
        if i > 0:
            w = _r(w)  # (the first layer's 1-channel 3x3 kernel runs on fp32 features with fp32 weights; the others on MFMA)
        x = _r(F.conv2d(x, w, sd[p + f"convolutions.{i}.bias"], stride=s, padding=1))
        x = _r(F.relu(_bn(x, sd, p + f"batchnorms.{i}.", training, update=update)))
        out_len = torch.div(out_len + s[0] - 1, s[0], rounding_mode="floor")
        i += 1
    x = x.transpose(1, 2).contiguous()
    x = x.view(x.size(0), x.size(1), -1)
    pad = torch.arange(x.size(1)).unsqueeze(0) >= out_len.unsqueeze(1)
    x = x.masked_fill(pad.unsqueeze(-1), 0.0)
    return x, out_len, pad


def chunk_attn_mask(out_lengths, chunk_size, left_window, right_window, training, num_updates=0):
    """Additive (T', T') mask of the chunk-streaming encoder: espresso/models/transformer/speech_transformer_encoder.py:240-248 +
    espresso/tools/utils.py:131-194, restated with plain loops.  Chunks of `chunk_size` encoder frames; a frame sees its chunk,
    `left_window` chunks to the left and `right_window` to the right; in training a coin flip (numpy RNG seeded with the update
    count, fairseq numpy_seed) decides whether the short chunk is the first instead of the last; masked entries get -1e8
    (fairseq transformer_layer.py:186-189)."""
    T = int(out_lengths.max())
    state = np.random.get_state()
    np.random.seed(num_updates)
    short_first = training and np.random.rand() > 0.5
    np.random.set_state(state)
    bounds = list(range(0, T, chunk_size))
    if short_first:
        bounds = [0] + sorted(T - b for b in bounds)[:-1]
    bounds.append(T)
    n = len(bounds) - 1
    m = torch.full((T, T), -1e8)
    for t in range(T):
        c = max(i for i in range(n) if bounds[i] <= t)
        lo, hi = bounds[max(c - left_window, 0)], bounds[min(c + right_window, n - 1) + 1]
        m[t, lo:hi] = 0.0
    return m


def legacy_encoder_kwargs(meta, lengths, training, strides=((1, 1), (2, 2), (1, 1), (2, 2))):
    """Keyword arguments of `encoder` for a fixture written with `legacy=...` (oracle/gen_golden.py): post-LN flag and the chunk mask."""
    kw = {"normalize_before": bool(meta.get("normalize_before", True))}
    if int(meta.get("chunk_size", 0)) > 0:
        ol = lengths.clone()
        for s_ in strides:
            ol = torch.div(ol + s_[0] - 1, s_[0], rounding_mode="floor")
        kw["attn_mask"] = chunk_attn_mask(ol, int(meta["chunk_size"]), int(meta.get("chunk_left_window", 0)),
                                          int(meta.get("chunk_right_window", 0)), training)
    return kw


def encoder(feats, lengths, sd, H, layer_type="conformer", training=False, activation="relu",
            strides=((1, 1), (2, 2), (1, 1), (2, 2)), update=None, normalize_before=True, attn_mask=None):
    """espresso/models/transformer/speech_transformer_encoder.py:298-409 + fc_out
    (speech_transformer_encoder_model.py:207-208).  Dropout: identity unless a `dropout_masks` plan is active AND training
    (then the reference's FairseqDropout sites with the HIP path's keep decisions).  Returns (logits (T',B,V), out_lengths)."""
    def _t(v):
        if not torch.is_tensor(v):
            v = torch.from_numpy(np.asarray(v))
        return v.float() if v.is_floating_point() and v.dtype != torch.float32 else v

    sd = {k: _t(v) for k, v in sd.items()}
    x, out_len, pad = conv_bn_relu(feats.float(), lengths, sd, "pre_encoder.", strides, training, update)
    if training:  # speech_transformer_encoder.py:342: dropout on fc0's input (the HIP path stores the dropped bf16 tensor)
        nconv = 0
        while f"pre_encoder.convolutions.{nconv}.weight" in sd:
            nconv += 1
        x = _r(_drop(x, "subsample.out", "SUB", C=sd[f"pre_encoder.convolutions.{nconv - 1}.weight"].shape[0]))
    x = _r(_lin(x, sd["fc0.weight"], sd["fc0.bias"]))
    if "embed_positions.weight" in sd or "embed_positions._float_tensor" in sd:
        # absolute positions of the legacy presets (speech_transformer_encoder.py:345-347; make_positions with padding_idx 0)
        valid = (~pad).long()
        pos = torch.cumsum(valid, 1) * valid
        tab = sd["embed_positions.weight"] if "embed_positions.weight" in sd else sinusoidal_abs_pe(int(pos.max()) + 1, x.shape[-1], 0)
        x = _r(x + tab[pos])
    # :348-350: layernorm_embedding, then dropout (the HIP LayerNorm kernel applies the mask before it stores)
    if "layernorm_embedding.weight" in sd:
        if training and _DROP["plan"] is not None:
            x = _r(_drop(F.layer_norm(x, (x.shape[-1],), sd["layernorm_embedding.weight"], sd["layernorm_embedding.bias"], 1e-5),
                         "ln.out", "BTC"))
        else:
            x = _ln(x, sd, "layernorm_embedding.")
    elif training:
        x = _r(_drop(x, "dropout", "BTC"))
    x = x * (1 - pad.unsqueeze(-1).float())
    x = x.transpose(0, 1)
    kpm = pad if bool(pad.any()) else None
    i = 0
    _EMU["stream_hot"] = False
    while f"layers.{i}.final_layer_norm.weight" in sd:
        p = f"layers.{i}."
        if layer_type == "conformer":
            x = conformer_layer(x, sd, p, H, kpm, training, update, attn_mask=attn_mask)
        else:
            x = transformer_layer(x, sd, p, H, kpm, activation, attn_mask=attn_mask, normalize_before=normalize_before)
        i += 1
    if "layer_norm.weight" in sd:
        x = _ln(x, sd, "layer_norm.")
    if "fc_out.weight" in sd:
        y = _lin(x, sd["fc_out.weight"], sd["fc_out.bias"])
        x = _rg(y) if os.environ.get("EA_LOGITS_F32", "0") == "1" else _r(y)
    return x, out_len


def ctc_loss_sum(logits_tbv, targets_padded, in_len, tgt_len, blank=0):
    """espresso/criterions/ctc_loss.py:85-94 (sum, zero_infinity) on fp32 log-softmax."""
    lp = torch.log_softmax(logits_tbv.float(), -1)
    flat = torch.cat([targets_padded[b, : int(tgt_len[b])] for b in range(targets_padded.shape[0])])
    with torch.backends.cudnn.flags(enabled=False):
        return F.ctc_loss(lp, flat, in_len, tgt_len, blank=blank, reduction="sum", zero_infinity=True)


def ctc_nll_numpy(lprobs_tv, target, blank=0):
    """Plain alpha recursion (Graves 2006) in float64 for one utterance — an independent check of
    the ATen kernel the reference calls.  lprobs_tv: (T, V) log-probabilities."""
    T = lprobs_tv.shape[0]
    ext = [blank]
    for t in target:
        ext += [int(t), blank]
    S = len(ext)
    a = np.full(S, -np.inf)
    a[0] = lprobs_tv[0, blank]
    if S > 1:
        a[1] = lprobs_tv[0, ext[1]]
    for t in range(1, T):
        n = np.full(S, -np.inf)
        for s in range(S):
            c = [a[s]]
            if s >= 1:
                c.append(a[s - 1])
            if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                c.append(a[s - 2])
            m = max(c)
            if m > -np.inf:
                n[s] = m + np.log(sum(np.exp(v - m) for v in c)) + lprobs_tv[t, ext[s]]
        a = n
    tail = [a[S - 1]] + ([a[S - 2]] if S > 1 else [])
    m = max(tail)
    return np.inf if m == -np.inf else -(m + np.log(sum(np.exp(v - m) for v in tail)))


def label_smoothed_nll(logits, target, eps, pad_idx, smoothing="uniform", prior=None, tgt_len=None):
    """espresso/criterions/label_smoothed_cross_entropy_v2.py:49-119: uniform, unigram (prior over the vocabulary) and
    temporal smoothing (previous / next two targets of the sentence, weights 2:5:5:2, pad neighbours dropped, normalised;
    rows are b*tgt_len + u)."""
    lp = torch.log_softmax(logits.float(), -1)
    nll = -lp.gather(-1, target.unsqueeze(-1)).squeeze(-1)
    if smoothing == "uniform":
        smooth = -lp.sum(-1)
    elif smoothing == "unigram":
        smooth = -(lp * prior.to(lp).view(1, -1)).sum(-1)
    else:
        M, V = lp.shape
        U = tgt_len
        w = torch.zeros(M, V)
        t = target.view(-1, U)
        for off, wt in ((-2, 2.0), (-1, 5.0), (1, 5.0), (2, 2.0)):
            for b in range(t.shape[0]):
                for u in range(U):
                    if 0 <= u + off < U and int(t[b, u + off]) != pad_idx:
                        w[b * U + u, int(t[b, u + off])] += wt
        z = w.sum(-1, keepdim=True)
        w = w / torch.where(z == 0, torch.ones_like(z), z)
        smooth = -(lp * w).sum(-1)
    m = target.eq(pad_idx)
    nll = nll.masked_fill(m, 0.0)
    smooth = smooth.masked_fill(m, 0.0)
    if smoothing == "uniform":
        eps_i = eps / (lp.size(-1) - 1)
        return (1.0 - eps - eps_i) * nll.sum() + eps_i * smooth.sum(), nll.sum()
    return (1.0 - eps) * nll.sum() + eps * smooth.sum(), nll.sum()


# ------------------------------------------------------------------------------------------------
# Attention encoder-decoder (speech_transformer_base): decoder restatement
def sinusoidal_abs_pe(num_embeddings, dim, padding_idx):
    """fairseq/modules/sinusoidal_positional_embedding.py:36-58 (get_embedding)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(num_embeddings, -1)
    if dim % 2 == 1:
        e = torch.cat([e, torch.zeros(num_embeddings, 1)], dim=1)
    if padding_idx is not None:
        e[padding_idx, :] = 0
    return e


def mha(q_in, kv_in, sd, p, H, key_padding_mask=None, causal=False):
    """fairseq/modules/multihead_attention.py (no positional embedding): q_in (U,B,C), kv_in (S,B,C).  Returns the out_proj
    output BEFORE the residual add (the HIP epilogue adds the residual in fp32 and stores once)."""
    U, B, C = q_in.shape
    S = kv_in.shape[0]
    dh = C // H
    q = _r(_lin(q_in, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]))
    q = _r(q * dh ** -0.5)
    k = _r(_lin(kv_in, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]))
    v = _r(_lin(kv_in, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]))
    q = q.contiguous().view(U, B * H, dh).transpose(0, 1)
    k = k.contiguous().view(S, B * H, dh).transpose(0, 1)
    v = v.contiguous().view(S, B * H, dh).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))
    if causal:
        w = w + torch.triu(torch.full((U, S), float("-inf")), 1).unsqueeze(0)
    if key_padding_mask is not None:
        w = w.view(B, H, U, S).masked_fill(key_padding_mask[:, None, None, :], float("-inf")).view(B * H, U, S)
    if _EMU["on"] and _EMU["flash"]:
        pu = torch.exp(w.float() - w.float().max(dim=-1, keepdim=True).values)
        a = torch.bmm(_r(_drop(pu, "attn.probs", "ZTS", B=B, H=H)), v) / pu.sum(-1, keepdim=True)
    else:
        a = torch.bmm(_r(_drop(torch.softmax(w.float(), -1), "attn.probs", "ZTS", B=B, H=H)), v)
    a = _r(a).transpose(0, 1).contiguous().view(U, B, C)
    return _lin(a, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def decoder(prev_tokens, enc_out, enc_pad, sd, H, pad_idx, p="decoder.", activation="relu"):
    """espresso/models/transformer/speech_transformer_decoder.py + fairseq transformer_decoder.py:254-370 and
    transformer_layer.py:384-529 (pre-LN, cross attention, no layerdrop; dropout only under `dropout_masks`).  Returns logits (B,U,V).
    `_r` marks the tensors the HIP decoder stores in bf16 (csrc/engine.hip ea_decoder_layer_*): identity outside bf16_emulation."""
    B, U = prev_tokens.shape
    W = sd[p + "embed_tokens.weight"]
    C = W.shape[1]
    x = math.sqrt(C) * F.embedding(prev_tokens, W)
    mask = prev_tokens.ne(pad_idx).int()
    positions = (torch.cumsum(mask, 1) * mask).long() + pad_idx
    x = _r(x + sinusoidal_abs_pe(pad_idx + 1 + U, C, pad_idx)[positions])
    # transformer_decoder.py:324-327: layernorm_embedding, then dropout (x is (B,U,C) here)
    if (p + "layernorm_embedding.weight") in sd:
        if _DROP["plan"] is not None:
            x = _r(_drop(F.layer_norm(x, (C,), sd[p + "layernorm_embedding.weight"], sd[p + "layernorm_embedding.bias"], 1e-5),
                         "ln.out", "BTC"))
        else:
            x = _ln(x, sd, p + "layernorm_embedding.")
    else:
        x = _r(_drop(x, "dropout", "BTC"))
    x = x.transpose(0, 1)
    act = (lambda y: F.relu(y)) if activation == "relu" else (lambda y: F.silu(y))
    i = 0
    while (p + f"layers.{i}.fc1.weight") in sd:
        lp = p + f"layers.{i}."
        # transformer_layer.py:384-529; dropout sites :456 (self-attention output), :481 (encoder attention output), :212-216 (FFN)
        y = _ln(x, sd, lp + "self_attn_layer_norm.")
        x = _r(_drop(mha(y, y, sd, lp + "self_attn.", H, causal=True), "attn.out", "TBC") + x)
        y = _ln(x, sd, lp + "encoder_attn_layer_norm.")
        x = _r(_drop(mha(y, enc_out, sd, lp + "encoder_attn.", H, key_padding_mask=enc_pad), "attn.out", "TBC") + x)
        y = _ln(x, sd, lp + "final_layer_norm.")
        y = _lin(y, sd[lp + "fc1.weight"], sd[lp + "fc1.bias"])
        y = _r(_drop(act(y), "ffn.act", "TBC"))
        x = _r(_drop(_lin(y, sd[lp + "fc2.weight"], sd[lp + "fc2.bias"]), "ffn.out", "TBC") + x)
        i += 1
    if (p + "layer_norm.weight") in sd:
        x = _ln(x, sd, p + "layer_norm.")
    x = x.transpose(0, 1)
    return _r(_lin(x, sd[p + "output_projection.weight"]))


def encdec(feats, lengths, prev_tokens, sd, H, pad_idx, training=False):
    """speech_transformer_base forward (espresso/models/transformer/speech_transformer_base.py:175-201)."""
    enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    x, out_len = encoder(feats, lengths, enc_sd, H, layer_type="transformer", training=training)
    pad = torch.arange(x.shape[0]).unsqueeze(0) >= out_len.unsqueeze(1)
    sdd = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}
    return decoder(prev_tokens, x, pad if bool(pad.any()) else None, sdd, H, pad_idx)


# ------------------------------------------------------------------------------------------------ transducer
def lstm_cell(x, h, c, sd, p):
    """torch.nn.LSTMCell (gate order i, f, g, o) as wrapped by fairseq/models/lstm.py:LSTMCell."""
    # HIP (csrc/lstm.hip, lstm_seq.hip): bf16 weights and bf16 x / h feed the MFMA products, gates and the cell state stay
    # fp32, the hidden state is stored in bf16 (`_r`: identity outside bf16_emulation)
    g = _lin(x, sd[p + "weight_ih"], sd[p + "bias_ih"]) + _lin(h, sd[p + "weight_hh"], sd[p + "bias_hh"])
    i, f, gg, o = g.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return _r(torch.sigmoid(o) * torch.tanh(c2)), c2


def lstm_predictor(prev_tokens, sd, p="decoder.", residual=False, pad_idx=1, state=None):
    """espresso/models/speech_lstm.py:766-919 with encoder_out None (no attention, no input feeding): the per-step loop over the
    LSTMCell stack.  Returns (features (B,U,H), final state [(h, c)] per layer).  Dropout (only under `dropout_masks`): on the
    embeddings (:811 dropout_in), on every layer's output at every step (:866 dropout_out; the recurrent state stays un-dropped)
    and after additional_fc (:909) — the HIP path draws one mask per layer over the time-major [U*B][H] sequence, so step j of
    layer i uses rows j*B .. (j+1)*B of that layer's mask."""
    emb = sd[p + "embed_tokens.weight"]
    x = _r(F.embedding(prev_tokens, emb, padding_idx=pad_idx)).transpose(0, 1)  # U x B x E
    nl = 0
    while (p + f"layers.{nl}.weight_ih") in sd:
        nl += 1
    B = prev_tokens.shape[0]
    U = x.shape[0]
    Hd = sd[p + "layers.0.weight_hh"].shape[1]
    m_in = _drop_mask("dropout", (U, B, x.shape[2]))
    if m_in is not None:
        x = _r(x * m_in)
    m_out = [_drop_mask("dropout", (U, B, Hd)) for _ in range(nl)]
    if state is None:
        state = [(x.new_zeros(B, Hd), x.new_zeros(B, Hd)) for _ in range(nl)]
    outs = []
    for j in range(U):
        inp = x[j]
        for i in range(nl):
            h, c = lstm_cell(inp, state[i][0], state[i][1], sd, p + f"layers.{i}.")
            prev_in = inp
            inp = h if m_out[i] is None else _r(h * m_out[i][j])
            if residual and i > 0:
                inp = _r(inp + prev_in)
            state[i] = (h, c)
        outs.append(inp)
    y = torch.stack(outs, 0).transpose(0, 1)
    if (p + "additional_fc.weight") in sd:
        y = _r(_lin(y, sd[p + "additional_fc.weight"], sd[p + "additional_fc.bias"]))
        m_fc = _drop_mask("dropout", tuple(y.shape))
        if m_fc is not None:
            y = _r(y * m_fc)
    return y, state


def lstm_lm(tokens, sd, pad_idx=0, residual=False):
    """espresso/models/lstm_lm.py:88-252 (`lstm_lm_espresso`): SpeechLSTMDecoder without attention over the whole padded token
    matrix, then the output projection (tied to the embedding or `fc_out`).  tokens (B,U) -> logits (B,U,V)."""
    sdd = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() and v.dtype != torch.float32 else v) for k, v in sd.items()}
    y, _ = lstm_predictor(tokens, sdd, p="decoder.", residual=residual, pad_idx=pad_idx)
    if "decoder.fc_out.weight" in sdd:
        return _lin(y, sdd["decoder.fc_out.weight"], sdd["decoder.fc_out.bias"])
    return _lin(y, sdd["decoder.embed_tokens.weight"])


def transducer_joint(enc_btc, dec_buh, sd):
    """speech_transformer_transducer_base.py:276-299 with the weight-normed fc_out (weight = g * v / ||v||_row)."""
    # HIP: both projections (Linear outputs), relu(E + D) as fc_out's operand, the effective (weight-normed) matrix and the logits are
    # stored in bf16; the two LayerNorm outputs E, D, their sum, the ReLU (and its derivative mask) and the gradients dE, dD are
    # fp32 — the fp32 island of the reference's autocast run (:292-294), round 6
    e = F.layer_norm(_r(_lin(enc_btc, sd["proj_encoder.weight"], sd["proj_encoder.bias"])), (sd["proj_encoder.weight"].shape[0],),
                     sd["laynorm_proj_encoder.weight"], sd["laynorm_proj_encoder.bias"])
    d = F.layer_norm(_r(_lin(dec_buh, sd["proj_decoder.weight"], sd["proj_decoder.bias"])), (sd["proj_decoder.weight"].shape[0],),
                     sd["laynorm_proj_decoder.weight"], sd["laynorm_proj_decoder.bias"])
    z = _r(F.relu(e.unsqueeze(2) + d.unsqueeze(1)))
    if "fc_out.weight_v" in sd:
        v = sd["fc_out.weight_v"]
        w = v * (sd["fc_out.weight_g"] / v.norm(dim=1, keepdim=True))
    else:
        w = sd["decoder.embed_tokens.weight"]
    y = _lin(z, w, sd["fc_out.bias"])
    # (fp32 logits: the fused criterion path, or the diagnostic switch EA_JOINT_LOGITS_F32=1 of the unfused one)
    return _rg(y) if (_EMU.get("joint_logits_f32") or os.environ.get("EA_JOINT_LOGITS_F32", "0") == "1") else _r(y)


def transducer(feats, lengths, prev_tokens, sd, H, pad_idx=1, residual=False, training=False, update=None):
    """speech_transformer_transducer_base forward (:221-274): (logits (B,T',U+1,V), encoder_out_lengths)."""
    enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    x, out_len = encoder(feats, lengths, enc_sd, H, layer_type="conformer", training=training, update=update)
    sdd = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}
    dec, _ = lstm_predictor(prev_tokens, sdd, residual=residual, pad_idx=pad_idx)
    return transducer_joint(x.transpose(0, 1), dec, sdd), out_len


# ------------------------------------------------------------------------------------------------ speech_lstm (config 1)
def packed_lstm_direction(x_tbd, lengths, sd, p, sfx, reverse):
    """One direction of a single-layer nn.LSTM over a packed batch (pack_padded_sequence / pad_packed_sequence, padding 0):
    every utterance runs over its own `length` steps from the zero state; padded steps emit zeros."""
    T, B, _ = x_tbd.shape
    H = sd[p + "weight_hh_l0" + sfx].shape[1]
    out = x_tbd.new_zeros(T, B, H)
    cell_sd = {"weight_ih": sd[p + "weight_ih_l0" + sfx], "weight_hh": sd[p + "weight_hh_l0" + sfx],
               "bias_ih": sd[p + "bias_ih_l0" + sfx], "bias_hh": sd[p + "bias_hh_l0" + sfx]}
    for b in range(B):
        L = int(lengths[b])
        h, c = x_tbd.new_zeros(1, H), x_tbd.new_zeros(1, H)
        steps = range(L - 1, -1, -1) if reverse else range(L)
        rows = {}
        for t in steps:
            h, c = lstm_cell(x_tbd[t, b:b + 1], h, c, cell_sd, "")
            rows[t] = h
        if L > 0:
            out[:L, b] = torch.cat([rows[t] for t in range(L)], 0)
    return out


def speech_lstm_encoder(feats, lengths, sd, p="encoder.", residual=False, training=False, update=None,
                        strides=((1, 1), (2, 2), (1, 1), (2, 2))):
    """espresso/models/speech_lstm.py:425-530 (dropout 0).  Returns (x (T',B,C), out_lengths)."""
    enc_sd = {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    x, out_len, pad = conv_bn_relu(feats.float(), lengths, enc_sd, "pre_encoder.", strides, training, update)
    x = x.transpose(0, 1)
    i = 0
    while f"lstm.{i}.weight_ih_l0" in enc_sd:
        q = f"lstm.{i}."
        outs = [packed_lstm_direction(x, out_len, enc_sd, q, "", False)]
        if (q + "weight_ih_l0_reverse") in enc_sd:
            outs.append(packed_lstm_direction(x, out_len, enc_sd, q, "_reverse", True))
        y = torch.cat(outs, -1)
        x = _r(y + x) if (residual and i > 0) else y  # HIP: bf16 hidden states, bf16 residual sum
        i += 1
    return x, out_len


def bahdanau_attention(query, value, sd, p, key_padding_mask=None):
    """espresso/modules/speech_attention.py:66-87 (normalize=True).  query (B,Hq); value (T,B,Cv); mask (T,B)."""
    # HIP (csrc/lstm.hip bahdanau kernels): the two projections are stored in bf16, scores / softmax / the weighted sum run in
    # fp32 on those, the context is stored in bf16 (`_r` / `_lin`: plain fp32 outside bf16_emulation)
    pq = _r(_lin(query, sd[p + "query_proj.weight"])).unsqueeze(0)
    key = _r(_lin(value, sd[p + "value_proj.weight"]))
    v = sd[p + "v"]
    nv = sd[p + "g"] * v / torch.norm(v)
    scores = (nv * torch.tanh(pq + key + sd[p + "b"])).sum(2)
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask, float("-inf"))
    a = torch.softmax(scores, 0)
    return _r((a.unsqueeze(2) * value).sum(0)), a


def speech_lstm_decoder(prev_tokens, enc_tbc, enc_lengths, sd, p="decoder.", residual=True, pad_idx=1, state=None):
    """espresso/models/speech_lstm.py:766-930 with attention + input feeding (dropout 0).  Returns (logits (B,U,V), state)."""
    T, B, Cv = enc_tbc.shape
    mask = torch.arange(T).unsqueeze(1) >= enc_lengths.unsqueeze(0)
    mask = mask if bool(mask.any()) else None
    x = _r(F.embedding(prev_tokens, sd[p + "embed_tokens.weight"], padding_idx=pad_idx)).transpose(0, 1)
    nl = 0
    while (p + f"layers.{nl}.weight_ih") in sd:
        nl += 1
    H = sd[p + "layers.0.weight_hh"].shape[1]
    if state is None:
        state = {"h": [x.new_zeros(B, H) for _ in range(nl)], "c": [x.new_zeros(B, H) for _ in range(nl)], "feed": x.new_zeros(B, Cv)}
    outs = []
    for j in range(x.shape[0]):
        inp = torch.cat((x[j], state["feed"]), 1)
        ctx = None
        for i in range(nl):
            h, c = lstm_cell(inp, state["h"][i], state["c"][i], sd, p + f"layers.{i}.")
            prev_in = inp[:, :H] if (residual and i > 0) else None
            if i == 0:
                ctx, _ = bahdanau_attention(h, enc_tbc, sd, p + "attention.", mask)
            inp = torch.cat((h, ctx), 1)
            if prev_in is not None:
                inp = torch.cat((_r(inp[:, :H] + prev_in), inp[:, H:]), 1)
            state["h"][i], state["c"][i] = h, c
        state["feed"] = ctx
        outs.append(inp)
    y = torch.stack(outs, 0).transpose(0, 1)
    if (p + "additional_fc.weight") in sd:
        y = _r(_lin(y, sd[p + "additional_fc.weight"], sd[p + "additional_fc.bias"]))
    if (p + "fc_out.weight") in sd:  # (HIP: fp32 logits from bf16 operands)
        y = _lin(y, sd[p + "fc_out.weight"], sd[p + "fc_out.bias"])
    else:
        y = _lin(y, sd[p + "embed_tokens.weight"])
    return y, state


def speech_lstm(feats, lengths, prev_tokens, sd, enc_residual=False, dec_residual=True, pad_idx=1, training=False, update=None):
    sdd = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() and v.dtype != torch.float32 else v) for k, v in sd.items()}
    x, out_len = speech_lstm_encoder(feats, lengths, sdd, residual=enc_residual, training=training, update=update)
    return speech_lstm_decoder(prev_tokens, x, out_len, sdd, residual=dec_residual, pad_idx=pad_idx)[0], x, out_len
