"""Length-bucketed batching and token collation with the reference's semantics.

batch_by_size: fairseq/data/data_utils.py:282-365 + the Cython kernel
fairseq/data/data_utils_fast.pyx:20-103 (`batch_by_size_vec`): scan indices in the given order,
close a batch when adding the next sample would exceed max_tokens (max_len_in_batch * n) or
max_sentences, keeping batch sizes a multiple of bsz_mult.  collate_tokens: data_utils.py:37-77."""
import contextlib
from typing import List

import numpy as np


@contextlib.contextmanager
def numpy_seed(seed, *addl_seeds):
    """Seed numpy's global RNG inside the block and restore the previous state afterwards
    (fairseq/data/data_utils.py:126-140; CPython's hash of a tuple of ints is deterministic across runs)."""
    if seed is None:
        yield
        return
    if len(addl_seeds) > 0:
        seed = int(hash((seed, *addl_seeds)) % 1e6)
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        yield
    finally:
        np.random.set_state(state)


def batch_by_size(indices, num_tokens_vec, max_tokens=None, max_sentences=None, bsz_mult=1) -> List[np.ndarray]:
    """`num_tokens_vec[i]` is the token count of `indices[i]` (aligned with `indices`, as the reference
    passes it).  Greedy scan that maintains a committed batch [start, committed) and a tail
    (committed, pos]; the tail is merged whenever the merged size is < bsz_mult or a multiple of it;
    on overflow the committed batch is closed and the tail (or, if the tail alone overflows, the tail
    without the current sample) starts the next one."""
    indices = np.asarray(indices, dtype=np.int64)
    ntok = np.asarray(num_tokens_vec, dtype=np.int64)
    max_tokens = int(max_tokens) if max_tokens is not None else -1
    max_sentences = int(max_sentences) if max_sentences is not None else -1
    n = len(indices)
    if n == 0:
        return []
    assert max_tokens <= 0 or ntok.max() <= max_tokens, f"Sentences lengths should not exceed max_tokens={max_tokens}"
    cuts = [0]            # cuts[-1] = end of the batch currently being grown ("committed" end)
    start = 0             # start of the batch currently being grown
    committed_max = 0     # longest sample in [start, cuts[-1])
    tail_max = 0          # longest sample in [cuts[-1], pos]
    closed: List[int] = []
    for pos in range(n):
        tail_max = max(tail_max, int(ntok[pos]))
        size = pos + 1 - start
        merged_max = max(committed_max, tail_max)
        overflow = (0 < max_sentences < size) or (0 < max_tokens < size * merged_max)
        fits_mult = size < bsz_mult or size % bsz_mult == 0
        if overflow:
            tail_tokens = tail_max * (pos + 1 - cuts[-1])
            if 0 < max_tokens < tail_tokens:
                # the tail alone overflows: close [cuts[-1], pos) as its own batch too
                closed.append(cuts[-1])
                cuts.append(pos)
                tail_max = int(ntok[pos])
            closed.append(cuts[-1])
            start = cuts[-1]
            merged_max = tail_max
        if overflow or fits_mult:
            cuts.append(pos + 1) if not overflow else cuts.append(pos + 1)
            committed_max = merged_max
            tail_max = 0
    bounds = sorted(set(closed + [cuts[-1]]) - {0})
    if not bounds or bounds[-1] != n:
        bounds.append(n)
    out, s0 = [], 0
    for e in bounds:
        if e > s0:
            out.append(indices[s0:e])
        s0 = e
    return out


def collate_tokens(values, pad_idx, eos_idx=None, left_pad=False, move_eos_to_beginning=False, pad_to_length=None,
                   pad_to_multiple=1):
    """Convert a list of 1d tensors into a padded 2d tensor."""
    size = max(v.size(0) for v in values)
    size = size if pad_to_length is None else max(size, pad_to_length)
    if pad_to_multiple != 1 and size % pad_to_multiple != 0:
        size = int(((size - 0.1) // pad_to_multiple + 1) * pad_to_multiple)
    res = values[0].new(len(values), size).fill_(pad_idx)
    for i, v in enumerate(values):
        dst = res[i][size - len(v):] if left_pad else res[i][: len(v)]
        if move_eos_to_beginning:
            if eos_idx is None:
                dst[0] = v[-1]
            else:
                dst[0] = eos_idx
            dst[1:] = v[:-1]
        else:
            dst.copy_(v)
    return res
