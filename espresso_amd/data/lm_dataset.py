"""Token data for language-model training — what fairseq's LanguageModelingTask builds for `language_modeling_for_asr`
(fairseq/tasks/language_modeling.py:185-262): a binarised token file (`<split>.bin/.idx`, fairseq/data/indexed_dataset.py:
390-560 "mmap" implementation) or a raw text file, cut into samples by `sample_break_mode` (`eos`: one sentence per sample,
`none`: blocks of `tokens_per_sample`; fairseq/data/token_block_dataset.py:17-150), each sample paired with its "future"
target (fairseq/data/monolingual_dataset.py:13-253): source = the sample shifted right by one token, starting with the token
before it in the stream (`</s>` for the first), target = the sample."""
import os
import struct
from typing import List

import numpy as np
import torch

_MAGIC = b"MMIDIDX\x00\x00"
_DTYPES = {1: np.uint8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float64, 7: np.double, 8: np.uint16,
           9: np.uint32, 10: np.uint64}


class MMapTokenFile:
    """Read-only view of fairseq's mmap indexed dataset: `prefix.idx` = magic, version <Q, dtype code <B, count <Q,
    sizes int32[count], byte pointers int64[count]; `prefix.bin` = the token arrays back to back."""

    def __init__(self, prefix):
        with open(prefix + ".idx", "rb") as f:
            if f.read(9) != _MAGIC:
                raise ValueError(f"{prefix}.idx is not an mmap indexed dataset (binarise with --dataset-impl mmap)")
            (version,) = struct.unpack("<Q", f.read(8))
            if version != 1:
                raise ValueError(f"unsupported index version {version}")
            (code,) = struct.unpack("<B", f.read(1))
            (n,) = struct.unpack("<Q", f.read(8))
            offset = f.tell()
        self.dtype = _DTYPES[code]
        idx = np.memmap(prefix + ".idx", mode="r", order="C")
        self.sizes = np.frombuffer(idx, dtype=np.int32, count=n, offset=offset)
        self.pointers = np.frombuffer(idx, dtype=np.int64, count=n, offset=offset + self.sizes.nbytes)
        self._idx = idx
        self.data = np.memmap(prefix + ".bin", mode="r", order="C")

    def __len__(self):
        return len(self.sizes)

    def __getitem__(self, i):
        return np.frombuffer(self.data, dtype=self.dtype, count=int(self.sizes[i]), offset=int(self.pointers[i])).astype(np.int64)

    @staticmethod
    def exists(prefix):
        return os.path.exists(prefix + ".idx") and os.path.exists(prefix + ".bin")

    @staticmethod
    def write(prefix, items: List[np.ndarray], dtype=np.int32):
        """Writer of the same format (fairseq/data/indexed_dataset.py:395-431), for tools and tests."""
        code = {v: k for k, v in _DTYPES.items() if k != 7}[dtype]
        sizes = np.array([len(x) for x in items], dtype=np.int32)
        pointers = np.concatenate(([0], np.cumsum(sizes[:-1].astype(np.int64) * np.dtype(dtype).itemsize))).astype(np.int64)
        with open(prefix + ".bin", "wb") as f:
            for x in items:
                f.write(np.asarray(x, dtype=dtype).tobytes(order="C"))
        with open(prefix + ".idx", "wb") as f:
            f.write(_MAGIC)
            f.write(struct.pack("<Q", 1))
            f.write(struct.pack("<B", code))
            f.write(struct.pack("<Q", len(sizes)))
            f.write(sizes.tobytes(order="C"))
            f.write(pointers.tobytes(order="C"))


class RawTextTokenFile:
    """`--dataset-impl raw`: one sentence per line, encoded with the dictionary, `</s>` appended
    (fairseq/data/indexed_dataset.py:277-330)."""

    def __init__(self, path, dictionary):
        self.items = []
        with open(path, encoding="utf-8") as f:
            for line in f:
                self.items.append(dictionary.encode_line(line.strip("\n"), append_eos=True).long().numpy())
        self.sizes = np.array([len(x) for x in self.items], dtype=np.int32)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def load_token_file(prefix, dictionary):
    if MMapTokenFile.exists(prefix):
        return MMapTokenFile(prefix)
    for p in (prefix, prefix + ".txt"):
        if os.path.isfile(p):
            return RawTextTokenFile(p, dictionary)
    raise FileNotFoundError(f"Dataset not found: {prefix}(.bin/.idx | text file)")


class MonolingualDataset:
    """Samples of (source, future target) over a token stream.  `eos`: sample i is sentence i; `none`: consecutive blocks of
    `tokens_per_sample` tokens over the concatenated stream (the last one shorter)."""

    def __init__(self, tokens, dictionary, sample_break_mode="eos", tokens_per_sample=1024, shuffle=True):
        self.tokens, self.dictionary, self.shuffle = tokens, dictionary, shuffle
        self.pad, self.eos = dictionary.pad(), dictionary.eos()
        mode = sample_break_mode or "none"
        if mode == "eos":
            self._flat = None
            self.sizes = np.asarray(tokens.sizes, dtype=np.int64)
        elif mode == "none":
            self._flat = np.concatenate([tokens[i] for i in range(len(tokens))]) if len(tokens) else np.zeros(0, dtype=np.int64)
            n, bs = len(self._flat), int(tokens_per_sample)
            self.sizes = np.array([min(bs, n - s) for s in range(0, n, bs)], dtype=np.int64)
            self._block = bs
        else:
            raise NotImplementedError(f"sample_break_mode {sample_break_mode!r} (the ASR language-model recipes use eos)")

    def __len__(self):
        return len(self.sizes)

    def __getitem__(self, i):
        if self._flat is None:
            tgt = np.asarray(self.tokens[i], dtype=np.int64)
            prev = self.eos if i == 0 else int(self.tokens[i - 1][-1])
        else:
            s = i * self._block
            tgt = self._flat[s:s + int(self.sizes[i])]
            prev = self.eos if s == 0 else int(self._flat[s - 1])
        src = np.concatenate(([prev], tgt[:-1]))
        return {"id": i, "source": torch.from_numpy(src.astype(np.int64)), "target": torch.from_numpy(np.ascontiguousarray(tgt))}

    def num_tokens(self, i):
        return int(self.sizes[i])

    def num_tokens_vec(self, indices):
        return self.sizes[np.asarray(indices, dtype=np.int64)]

    def size(self, i):
        return int(self.sizes[i])

    def ordered_indices(self):
        """monolingual_dataset.py:238-246: random order, then stable by length."""
        order = [np.random.permutation(len(self))] if self.shuffle else [np.arange(len(self))]
        order.append(self.sizes)
        return np.lexsort(order)

    def filter_indices_by_size(self, indices, max_positions):
        limit = max_positions if isinstance(max_positions, int) else min(m for m in max_positions if m is not None)
        keep = self.sizes[indices] <= limit
        return indices[keep], indices[~keep].tolist()

    def collater(self, samples):
        """monolingual_dataset.py:13-60: right padding, no sorting."""
        if len(samples) == 0:
            return {}
        L = max(len(s["source"]) for s in samples)

        def merge(key):
            out = torch.full((len(samples), L), self.pad, dtype=torch.long)
            for r, s in enumerate(samples):
                out[r, : len(s[key])] = s[key]
            return out

        lens = torch.tensor([len(s["source"]) for s in samples], dtype=torch.long)
        return {"id": torch.tensor([s["id"] for s in samples], dtype=torch.long), "nsentences": len(samples),
                "ntokens": int(lens.sum()), "net_input": {"src_tokens": merge("source"), "src_lengths": lens},
                "target": merge("target")}
