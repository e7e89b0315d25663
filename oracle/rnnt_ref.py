"""TEST INFRASTRUCTURE ONLY — float64 restatement of the RNN-T loss that the reference obtains from
torchaudio.functional.rnnt_loss(logits, targets, logit_lengths, target_lengths, blank, clamp=-1, reduction)
(espresso/criterions/transducer_loss.py:130-140).  torchaudio is not in the reference tree and not installable here:
**parity unpinned** against torchaudio itself; the recursion (Graves 2012, eq. 16-20) is instead pinned against a brute-force
enumeration of every alignment on tiny lattices (tests/test_oracle.py) and its gradient against finite differences."""
import itertools

import numpy as np


def log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def rnnt_loss_one(logits_tuv, target, blank=0, want_grad=False):
    """logits (T, U+1, V) for one utterance, target length U.  Returns nll (and d nll / d logits)."""
    lp = log_softmax(np.asarray(logits_tuv, dtype=np.float64))
    T, U1, V = lp.shape
    U = U1 - 1
    assert len(target) == U
    alpha = np.full((T, U1), -np.inf)
    alpha[0, 0] = 0.0
    for t in range(T):
        for u in range(U1):
            if t == 0 and u == 0:
                continue
            a = alpha[t - 1, u] + lp[t - 1, u, blank] if t > 0 else -np.inf
            c = alpha[t, u - 1] + lp[t, u - 1, target[u - 1]] if u > 0 else -np.inf
            alpha[t, u] = np.logaddexp(a, c)
    logp = alpha[T - 1, U] + lp[T - 1, U, blank]
    if not want_grad:
        return -logp
    beta = np.full((T, U1), -np.inf)
    beta[T - 1, U] = lp[T - 1, U, blank]
    for t in range(T - 1, -1, -1):
        for u in range(U, -1, -1):
            if t == T - 1 and u == U:
                continue
            a = beta[t + 1, u] + lp[t, u, blank] if t < T - 1 else -np.inf
            c = beta[t, u + 1] + lp[t, u, target[u]] if u < U else -np.inf
            beta[t, u] = np.logaddexp(a, c)
    grad = np.exp(lp + (alpha + beta - logp)[:, :, None])
    for t in range(T):
        for u in range(U1):
            if t == T - 1 and u == U:
                grad[t, u, blank] -= np.exp(alpha[t, u] + lp[t, u, blank] - logp)
            elif t < T - 1:
                grad[t, u, blank] -= np.exp(alpha[t, u] + lp[t, u, blank] + beta[t + 1, u] - logp)
            if u < U:
                grad[t, u, target[u]] -= np.exp(alpha[t, u] + lp[t, u, target[u]] + beta[t, u + 1] - logp)
    return -logp, grad


def rnnt_loss_bruteforce(logits_tuv, target, blank=0):
    """Sum over every monotonic alignment (T blanks interleaved with U labels, last symbol a blank at frame T-1)."""
    lp = log_softmax(np.asarray(logits_tuv, dtype=np.float64))
    T, U1, V = lp.shape
    U = U1 - 1
    total = -np.inf
    # a path = order of T-1 "blank moves" (t += 1) and U "label moves" (u += 1), followed by the final blank
    for labels_at in itertools.combinations(range(T - 1 + U), U):
        t = u = 0
        s = 0.0
        for step in range(T - 1 + U):
            if step in labels_at:
                s += lp[t, u, target[u]]
                u += 1
            else:
                s += lp[t, u, blank]
                t += 1
        s += lp[T - 1, U, blank]
        total = np.logaddexp(total, s)
    return -total


def rnnt_loss_torch(logits_tuv, target, blank=0):
    """The same negative log-likelihood as `rnnt_loss_one` as a differentiable torch expression (float64 inside; autograd gives
    d loss / d logits): alpha recursion over anti-diagonal-free (t, u) order — for the model-level gradient checks."""
    import torch

    lp = torch.log_softmax(logits_tuv.double(), -1)
    T, U1, _ = lp.shape
    tgt = torch.as_tensor(list(target), dtype=torch.long)
    lb = lp[:, :, blank]                                   # (T, U+1): emit blank at (t, u)
    ly = lp[:, : U1 - 1, :].gather(2, tgt.view(1, -1, 1).expand(T, -1, 1)).squeeze(2) if U1 > 1 else lp.new_zeros(T, 0)
    alpha = [[None] * U1 for _ in range(T)]
    alpha[0][0] = lp.new_zeros(())
    for t in range(T):
        for u in range(U1):
            if t == 0 and u == 0:
                continue
            terms = []
            if t > 0:
                terms.append(alpha[t - 1][u] + lb[t - 1, u])
            if u > 0:
                terms.append(alpha[t][u - 1] + ly[t, u - 1])
            alpha[t][u] = terms[0] if len(terms) == 1 else torch.logaddexp(terms[0], terms[1])
    return -(alpha[T - 1][U1 - 1] + lb[T - 1, U1 - 1])
