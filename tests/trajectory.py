"""Training-trajectory parity on a learnable synthetic task (stand-in for the north-star's "WER within 0.1 abs": there is no
corpus and no trained checkpoint in this environment).

Task: every target token owns a random 80-dimensional template; an utterance is its token sequence with each template held
for 8 frames plus Gaussian noise (sigma 2.5: neighbouring tokens are confusable), ragged lengths, no repeated neighbours.  A 2-layer
Conformer-CTC (the weights of tests/golden/ref_conformer_ctc_dh64.npz: head dim 64, i.e. the fused rel-pos attention kernels)
is trained from the same initial weights on the same batches in the same order, with dropout 0 or with dropout 0.1 and the
HIP path's own keep decisions handed to the oracle, by
  * the HIP path: model forward / CTC / backward through the C ABI, FlatAdam (csrc/optim.hip), and
  * the oracle: oracle/torch_ref.py (fp32, or rounding to bf16 at the HIP storage points) + a restatement of
    fairseq/utils.py:347-397 (clip) and fairseq/optim/adam.py:215-240 (Adam).
Compared: the loss of every update and the greedy-CTC token error rate on held-out batches at the end.
TEST INFRASTRUCTURE ONLY."""
import math

import numpy as np
import torch

FIXTURE = "ref_conformer_ctc_dh64"
HEADS = 2
LR, BETAS, EPS, CLIP = 2e-3, (0.9, 0.98), 1e-8, 2.0
STEPS, TRAIN_BATCHES, HELDOUT_BATCHES = 80, 30, 12  # 30 x 8 utterances, each batch seen 2-3 times; 675 held-out tokens


def make_batches(nbatch, seed, B=8, ntok=36, hold=8, noise=2.5, tmpl_seed=1234):
    tmpl = np.random.default_rng(tmpl_seed).standard_normal((40, 80)).astype(np.float32)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(nbatch):
        U = rng.integers(5, 10, size=B)
        feats = np.zeros((B, int(U.max()) * hold, 80), np.float32)
        tg = np.ones((B, int(U.max())), np.int64)  # pad = 1
        lens = np.zeros(B, np.int64)
        for b in range(B):
            toks, prev = [], -1
            for _u in range(U[b]):
                t = int(rng.integers(4, 4 + ntok))
                while t == prev:
                    t = int(rng.integers(4, 4 + ntok))
                toks.append(t)
                prev = t
            tg[b, : U[b]] = toks
            lens[b] = U[b] * hold
            feats[b, : lens[b]] = np.repeat(tmpl[toks], hold, axis=0) + noise * rng.standard_normal((U[b] * hold, 80)).astype(np.float32)
        order = np.argsort(-lens, kind="stable")
        out.append((torch.from_numpy(feats[order]), torch.from_numpy(lens[order]), torch.from_numpy(tg[order])))
    return out


def greedy_errors(logits_tbv, out_len, targets):
    """(edit-distance errors, reference tokens) of the greedy CTC hypotheses (collapse repeats, drop blank 0)"""
    err = tot = 0
    for b in range(targets.shape[0]):
        ids = logits_tbv[: int(out_len[b]), b].argmax(-1).tolist()
        hyp, prev = [], -1
        for i in ids:
            if i != prev and i != 0:
                hyp.append(i)
            prev = i
        ref = [int(t) for t in targets[b] if t != 1]
        d = list(range(len(ref) + 1))
        for i, h in enumerate(hyp, 1):
            nd = [i]
            for j, r in enumerate(ref, 1):
                nd.append(min(d[j] + 1, nd[j - 1] + 1, d[j - 1] + (h != r)))
            d = nd
        err += d[-1]
        tot += len(ref)
    return err, tot


def train_oracle(sd, train, heldout, steps, emulate, traces=None, layer_seed_fn=None):
    """`traces`: per update, the (site, seed, p) list the HIP path's forward drew (dropout on): the oracle then applies the
    reference's dropout sites with those keep decisions (oracle/dropout_ref.py); None = dropout 0."""
    from oracle import dropout_ref, torch_ref

    P = {k: (v.clone().float().requires_grad_(True) if v.is_floating_point() and "running" not in k and not k.endswith("_float_tensor")
             and k != "version" else v.clone()) for k, v in sd.items()}
    names = [k for k, v in P.items() if v.requires_grad]
    m = {k: torch.zeros_like(P[k]) for k in names}
    v2 = {k: torch.zeros_like(P[k]) for k in names}
    losses = []
    for step in range(steps):
        feats, lens, tg = train[step % len(train)]
        upd = {}
        plan = dropout_ref.MaskPlan(traces[step], layer_seed_fn) if traces is not None else None
        with torch_ref.bf16_emulation(emulate, flash=True), torch_ref.dropout_masks(plan):
            lt, ol = torch_ref.encoder(feats, lens, P, H=HEADS, layer_type="conformer", training=True, update=upd)
            loss = torch_ref.ctc_loss_sum(lt, tg, ol, (tg != 1).sum(-1))
        if plan is not None:
            plan.done()
        for k in names:
            P[k].grad = None
        loss.backward()
        B = feats.shape[0]
        with torch.no_grad():
            gn = math.sqrt(sum(float((P[k].grad / B).pow(2).sum()) for k in names if P[k].grad is not None))
            coef = min(1.0, CLIP / (gn + 1e-6)) / B
            t = step + 1
            ss = LR * math.sqrt(1 - BETAS[1] ** t) / (1 - BETAS[0] ** t)
            for k in names:
                if P[k].grad is None:
                    continue
                gk = P[k].grad * coef
                m[k].mul_(BETAS[0]).add_(gk, alpha=1 - BETAS[0])
                v2[k].mul_(BETAS[1]).addcmul_(gk, gk, value=1 - BETAS[1])
                P[k].addcdiv_(m[k], v2[k].sqrt().add_(EPS), value=-ss)
            for k, val in upd.items():
                P[k] = val
        losses.append(float(loss.detach()) / B)
    err = tot = 0
    with torch.no_grad(), torch_ref.bf16_emulation(emulate, flash=True):
        for feats, lens, tg in heldout:
            lt, ol = torch_ref.encoder(feats, lens, P, H=HEADS, layer_type="conformer", training=False)
            e, t_ = greedy_errors(lt, ol, tg)
            err += e
            tot += t_
    return losses, err, tot


# ---- encoder-decoder (label-smoothed CE) trajectory: the same synthetic task, teacher-forced ------------------------------
ENCDEC_FIXTURE = "ref_transformer_encdec_dh64"
ENCDEC_PAD, ENCDEC_EOS, LS_EPS = 0, 1, 0.1


def encdec_targets(tg):
    """CTC-style targets (pad 1) -> (target with EOS, prev_output_tokens) in the attention model's dictionary (pad 0, eos 1)."""
    B, U = tg.shape
    lens = (tg != 1).sum(1)
    target = torch.full((B, U + 1), ENCDEC_PAD, dtype=torch.long)
    prev = torch.full((B, U + 1), ENCDEC_PAD, dtype=torch.long)
    for b in range(B):
        n = int(lens[b])
        target[b, :n], target[b, n] = tg[b, :n], ENCDEC_EOS
        prev[b, 0], prev[b, 1:n + 1] = ENCDEC_EOS, tg[b, :n]
    return target, prev


def train_oracle_encdec(sd, train, heldout, steps, emulate):
    """oracle/torch_ref.py encdec + label_smoothed_nll (espresso/criterions/label_smoothed_cross_entropy_v2.py:94-119), clip and
    Adam as in train_oracle; gradients are normalised by the number of target tokens.  -> per-update loss per token, held-out
    (nll per token, teacher-forced token accuracy) with BatchNorm on batch statistics (training mode, dropout 0)."""
    from oracle import torch_ref

    P = {k: (v.clone().float().requires_grad_(True) if v.is_floating_point() and "running" not in k and not k.endswith("_float_tensor")
             and k != "version" else v.clone()) for k, v in sd.items()}
    names = [k for k, v in P.items() if v.requires_grad]
    m = {k: torch.zeros_like(P[k]) for k in names}
    v2 = {k: torch.zeros_like(P[k]) for k in names}
    losses = []
    for step in range(steps):
        feats, lens, tg = train[step % len(train)]
        target, prev = encdec_targets(tg)
        with torch_ref.bf16_emulation(emulate, flash=True):
            lo = torch_ref.encdec(feats, lens, prev, P, HEADS, ENCDEC_PAD, training=True)
            loss, _ = torch_ref.label_smoothed_nll(lo.reshape(-1, lo.shape[-1]), target.reshape(-1), LS_EPS, ENCDEC_PAD)
        for k in names:
            P[k].grad = None
        loss.backward()
        n = int((target != ENCDEC_PAD).sum())
        with torch.no_grad():
            gn = math.sqrt(sum(float((P[k].grad / n).pow(2).sum()) for k in names if P[k].grad is not None))
            coef = min(1.0, CLIP / (gn + 1e-6)) / n
            t = step + 1
            ss = LR * math.sqrt(1 - BETAS[1] ** t) / (1 - BETAS[0] ** t)
            for k in names:
                if P[k].grad is None:
                    continue
                gk = P[k].grad * coef
                m[k].mul_(BETAS[0]).add_(gk, alpha=1 - BETAS[0])
                v2[k].mul_(BETAS[1]).addcmul_(gk, gk, value=1 - BETAS[1])
                P[k].addcdiv_(m[k], v2[k].sqrt().add_(EPS), value=-ss)
        losses.append(float(loss.detach()) / n)
    nll = tok = hit = 0.0
    with torch.no_grad(), torch_ref.bf16_emulation(emulate, flash=True):
        for feats, lens, tg in heldout:
            target, prev = encdec_targets(tg)
            lo = torch_ref.encdec(feats, lens, prev, P, HEADS, ENCDEC_PAD, training=True)
            _, nl = torch_ref.label_smoothed_nll(lo.reshape(-1, lo.shape[-1]), target.reshape(-1), LS_EPS, ENCDEC_PAD)
            valid = target != ENCDEC_PAD
            nll += float(nl)
            tok += int(valid.sum())
            hit += int((lo.argmax(-1) == target)[valid].sum())
    return losses, nll / tok, hit / tok


# ---- transducer (RNN-T loss) trajectory -------------------------------------------------------------------------------------
TD_FIXTURE = "ref_conformer_transducer_tiny"
TD_HEADS, TD_BLANK, TD_PAD, TD_EOS = 4, 0, 1, 2


def transducer_targets(tg):
    """CTC-style targets (pad 1) -> (target = tokens + EOS, prev_output_tokens = EOS + tokens, token counts); the transducer
    criterion drops the EOS again (espresso/criterions/transducer_loss.py:73-90)."""
    B, U = tg.shape
    lens = (tg != 1).sum(1)
    target = torch.full((B, U + 1), TD_PAD, dtype=torch.long)
    prev = torch.full((B, U + 1), TD_PAD, dtype=torch.long)
    for b in range(B):
        n = int(lens[b])
        target[b, :n], target[b, n] = tg[b, :n], TD_EOS
        prev[b, 0], prev[b, 1:n + 1] = TD_EOS, tg[b, :n]
    return target, prev, lens


def train_oracle_transducer(sd, train, heldout, steps, emulate):
    """oracle/torch_ref.py transducer + oracle/rnnt_ref.py rnnt_loss_torch (sum over the batch), clip and Adam as above,
    normalised by the number of sentences.  -> per-update loss per sentence, held-out loss per sentence (training-mode BatchNorm)."""
    from oracle import rnnt_ref, torch_ref

    P = {k: (v.clone().float().requires_grad_(True) if v.is_floating_point() and "running" not in k and not k.endswith("_float_tensor")
             and k != "version" else v.clone()) for k, v in sd.items()}
    names = [k for k, v in P.items() if v.requires_grad]
    m = {k: torch.zeros_like(P[k]) for k in names}
    v2 = {k: torch.zeros_like(P[k]) for k in names}

    def batch_loss(batch, update):
        feats, lens, tg = batch
        target, prev, tl = transducer_targets(tg)
        # (joint_logits_f32: the criterion's fused output layer + loss works on the fp32 accumulators — csrc/joint_rnnt.hip)
        with torch_ref.bf16_emulation(emulate, flash=False, joint_logits_f32=True):
            lo, ol = torch_ref.transducer(feats, lens, prev, P, H=TD_HEADS, pad_idx=TD_PAD, residual=True, training=True, update=update)
        tot = 0.0
        for b in range(feats.shape[0]):
            n = int(tl[b])
            tot = tot + rnnt_ref.rnnt_loss_torch(lo[b, : int(ol[b]), : n + 1], target[b, :n].tolist(), blank=TD_BLANK)
        return tot, feats.shape[0]

    losses = []
    for step in range(steps):
        upd = {}
        loss, B = batch_loss(train[step % len(train)], upd)
        for k in names:
            P[k].grad = None
        loss.backward()
        with torch.no_grad():
            gn = math.sqrt(sum(float((P[k].grad / B).pow(2).sum()) for k in names if P[k].grad is not None))
            coef = min(1.0, CLIP / (gn + 1e-6)) / B
            t = step + 1
            ss = LR * math.sqrt(1 - BETAS[1] ** t) / (1 - BETAS[0] ** t)
            for k in names:
                if P[k].grad is None:
                    continue
                gk = P[k].grad.float() * coef
                m[k].mul_(BETAS[0]).add_(gk, alpha=1 - BETAS[0])
                v2[k].mul_(BETAS[1]).addcmul_(gk, gk, value=1 - BETAS[1])
                P[k].addcdiv_(m[k], v2[k].sqrt().add_(EPS), value=-ss)
            for k, val in upd.items():
                P[k] = val
        losses.append(float(loss.detach()) / B)
    tot = n = 0.0
    with torch.no_grad():
        for batch in heldout:
            l, B = batch_loss(batch, {})
            tot += float(l)
            n += B
    return losses, tot / n
