"""Host-side helpers with the reference's names and semantics (espresso/tools/utils.py):
collate_frames :97-113, sequence_mask :116-129, convert_padding_direction :197-221,
eval_str_nested_list_or_tuple :224-237, edit_distance :265-331, num_samples_to_num_frames :457-486."""
import ast
from typing import List

import numpy as np
import torch


def collate_frames(values, pad_value=0.0, left_pad=False, pad_to_length=None, pad_to_multiple=1):
    """List of (T_i, D) feature matrices -> one (N, T, D) batch padded with `pad_value` on the right (or on the left), with
    T = max T_i, raised to `pad_to_length` and rounded up to a multiple of `pad_to_multiple` — the contract of
    espresso/tools/utils.py:97-113.  (The training path pads on the GPU inside `ea_fbank_batch`; this host version serves
    pre-computed Kaldi features.)"""
    first = values[0]
    if first.dim() != 2:
        raise AssertionError(f"expected 2-d feature matrices, got {first.dim()}-d")
    lens = [int(v.shape[0]) for v in values]
    T = max(lens + ([int(pad_to_length)] if pad_to_length is not None else []))
    T = -(-T // pad_to_multiple) * pad_to_multiple
    out = torch.full((len(values), T, int(first.shape[1])), pad_value, dtype=first.dtype, device=first.device)
    for row, v, n in zip(out, values, lens):
        row.narrow(0, T - n if left_pad else 0, n).copy_(v)
    return out


def sequence_mask(sequence_length, max_len=None):
    if max_len is None:
        max_len = int(sequence_length.max())
    else:
        assert int(sequence_length.max()) <= int(max_len)
    seq_range = torch.arange(0, max_len, device=sequence_length.device, dtype=sequence_length.dtype)
    return seq_range.unsqueeze(0) < sequence_length.unsqueeze(1)


def convert_padding_direction(src_frames, src_lengths, right_to_left=False, left_to_right=False):
    """Move padding of a (B, T, C) batch between the left and the right side."""
    assert right_to_left ^ left_to_right
    assert src_frames.size(0) == src_lengths.size(0)
    max_len = src_frames.size(1)
    # the reference's early return (espresso/tools/utils.py:209): taken when NO row has the full length (its collated batches
    # always contain one), not when no row is padded
    if not bool(src_lengths.eq(max_len).any()):
        return src_frames
    index = torch.arange(max_len, device=src_frames.device).unsqueeze(0).expand(src_frames.size(0), -1)
    num_pads = (max_len - src_lengths).unsqueeze(1)
    index = torch.remainder(index - num_pads, max_len) if right_to_left else torch.remainder(index + num_pads, max_len)
    return src_frames.gather(1, index.unsqueeze(2).expand_as(src_frames))


def eval_str_nested_list_or_tuple(x, type=int):
    if x is None:
        return None
    if isinstance(x, str):
        x = ast.literal_eval(x)
    if isinstance(x, list):
        return [eval_str_nested_list_or_tuple(e, type) for e in x]
    if isinstance(x, tuple):
        return tuple(eval_str_nested_list_or_tuple(e, type) for e in x)
    try:
        return type(x)
    except TypeError:
        raise TypeError


def num_samples_to_num_frames(num_samples: List[int], sample_rate: int, frame_length: float = 25.0,
                              frame_shift: float = 10.0) -> List[int]:
    """Kaldi snip_edges frame count (espresso/tools/utils.py:457-486)."""
    flen = int(sample_rate * frame_length * 0.001)
    fshift = int(sample_rate * frame_shift * 0.001)
    return [0 if n < flen else 1 + (n - flen) // fshift for n in num_samples]


def edit_distance(ref, hyp):
    """Word-level Levenshtein distance with backtrace — same results as espresso/tools/utils.py:265-331:
    returns (dist matrix uint32 [len(ref)+1][len(hyp)+1], steps, Counter{words,corr,sub,ins,del}).
    Backtrace priority at ties: corr, then sub, then ins, then del (walking from the end)."""
    from collections import Counter

    assert isinstance(ref, list) and isinstance(hyp, list)
    R, Hn = len(ref), len(hyp)
    dist = np.zeros((R + 1, Hn + 1), dtype=np.uint32)
    dist[0, :] = np.arange(Hn + 1, dtype=np.uint32)
    dist[:, 0] = np.arange(R + 1, dtype=np.uint32)
    for i in range(1, R + 1):
        row, up = dist[i], dist[i - 1]
        r = ref[i - 1]
        for j in range(1, Hn + 1):
            row[j] = up[j - 1] if r == hyp[j - 1] else 1 + min(up[j - 1], row[j - 1], up[j])
    steps = []
    i, j = R, Hn
    while i or j:
        here = int(dist[i, j])
        diag = int(dist[i - 1, j - 1]) if (i and j) else None
        if diag is not None and here == diag and ref[i - 1] == hyp[j - 1]:
            steps.append("corr")
            i, j = i - 1, j - 1
        elif diag is not None and here == diag + 1:
            steps.append("sub")
            i, j = i - 1, j - 1
        elif j and here == int(dist[i, j - 1]) + 1:
            steps.append("ins")
            j -= 1
        else:
            steps.append("del")
            i -= 1
    steps.reverse()
    counter = Counter({"words": R, "corr": 0, "sub": 0, "ins": 0, "del": 0})
    counter.update(steps)
    return dist, steps, counter


def tokenize(sent, space="<space>", non_lang_syms=None):
    """espresso/tools/utils.py:29-56 lives here in the reference; the implementation sits next to its main user
    (tools/tensorized_prefix_tree.py) and is re-exported under the reference's location."""
    from .tensorized_prefix_tree import tokenize as _tokenize

    return _tokenize(sent, space=space, non_lang_syms=non_lang_syms)


def chunk_streaming_mask(sequence_length, chunk_size, left_window=0, right_window=0, always_partial_in_last=False):
    """Visibility mask (T x T bool, True = query row may attend key column) of the chunk-streaming encoder —
    espresso/tools/utils.py:131-194.  The time axis is cut into chunks of `chunk_size` frames; a frame sees its own chunk,
    `left_window` chunks before it and `right_window` chunks after it.  The short chunk is the last one, or — in training, on a
    coin flip from numpy's global RNG (drawn exactly when the reference draws it: only if `always_partial_in_last` is False) —
    the first one."""
    import numpy as np

    T = int(sequence_length.max())
    dev = sequence_length.device
    starts = torch.arange(0, T, chunk_size, device=dev)
    if not always_partial_in_last and np.random.rand() > 0.5:
        starts = torch.cat([starts.new_zeros(1), (T - starts).flip(0)[:-1]])  # short chunk first
    n = starts.numel()
    ends = torch.cat([starts[1:], starts.new_full((1,), T)])
    t = torch.arange(T, device=dev)
    chunk = torch.searchsorted(starts, t, right=True) - 1  # chunk index of every frame
    lo = starts[(chunk - left_window).clamp(min=0)]
    hi = ends[(chunk + right_window).clamp(max=n - 1)]
    return (t.unsqueeze(0) >= lo.unsqueeze(1)) & (t.unsqueeze(0) < hi.unsqueeze(1))
