"""Minimal reader/writer for Kaldi binary matrices addressed as `file.ark:offset` — the `feat` entries of the data
json (espresso/tasks/speech_recognition.py:143-147; the reference reads them through the third-party `kaldi_io`
package, `espresso/data/feat_text_dataset.py:128-131`, which is not vendored).  Format (Kaldi `matrix/kaldi-matrix.cc`,
`matrix/compressed-matrix.{h,cc}`): `\\0B` + token; `FM `/`DM ` = `\\4 rows \\4 cols` + row-major float32/float64;
`CM ` = global header (min f32, range f32, rows i32, cols i32) + per-column 4×u16 percentiles + column-major bytes,
piecewise-linear between the percentiles; `CM2 `/`CM3 ` = global header + row-major u16 / u8."""
import re
import struct
from typing import Tuple

import numpy as np


def parse_rxfile(rxfile: str) -> Tuple[str, int]:
    m = re.match(r"^(.*):(\d+)$", rxfile.strip())
    return (m.group(1), int(m.group(2))) if m else (rxfile.strip(), 0)


def _read_token(fd) -> str:
    tok = b""
    while True:
        c = fd.read(1)
        if c in (b" ", b""):
            return tok.decode()
        tok += c


def _read_int(fd) -> int:
    n = fd.read(1)
    assert n == b"\4", "expected a 4-byte integer marker"
    return struct.unpack("<i", fd.read(4))[0]


def _uncompress_cm(glob, col_hdr: np.ndarray, data: np.ndarray) -> np.ndarray:
    mn, rng, rows, cols = glob
    p = mn + rng * (1.0 / 65535.0) * col_hdr.astype(np.float32)  # [cols][4]: percentiles 0, 25, 75, 100
    v = data.astype(np.float32)  # [cols][rows]
    p0, p25, p75, p100 = (p[:, i:i + 1] for i in range(4))
    out = np.where(v <= 64, p0 + (p25 - p0) * v * (1.0 / 64.0),
                   np.where(v <= 192, p25 + (p75 - p25) * (v - 64.0) * (1.0 / 128.0), p75 + (p100 - p75) * (v - 192.0) * (1.0 / 63.0)))
    return np.ascontiguousarray(out.T.astype(np.float32))


def read_mat_fd(fd) -> np.ndarray:
    assert fd.read(2) == b"\0B", "only binary Kaldi matrices are supported"
    tok = _read_token(fd)
    if tok in ("FM", "DM"):
        rows, cols = _read_int(fd), _read_int(fd)
        dt = np.dtype("<f4") if tok == "FM" else np.dtype("<f8")
        buf = fd.read(rows * cols * dt.itemsize)
        return np.frombuffer(buf, dtype=dt).reshape(rows, cols).astype(np.float32)
    if tok in ("CM", "CM2", "CM3"):
        mn, rng, rows, cols = struct.unpack("<ffii", fd.read(16))
        if tok == "CM":
            hdr = np.frombuffer(fd.read(cols * 8), dtype="<u2").reshape(cols, 4)
            data = np.frombuffer(fd.read(rows * cols), dtype=np.uint8).reshape(cols, rows)
            return _uncompress_cm((mn, rng, rows, cols), hdr, data)
        if tok == "CM2":
            v = np.frombuffer(fd.read(rows * cols * 2), dtype="<u2").reshape(rows, cols).astype(np.float32)
            return (mn + rng * (1.0 / 65535.0) * v).astype(np.float32)
        v = np.frombuffer(fd.read(rows * cols), dtype=np.uint8).reshape(rows, cols).astype(np.float32)
        return (mn + rng * (1.0 / 255.0) * v).astype(np.float32)
    raise ValueError(f"unsupported Kaldi matrix token '{tok}'")


def read_mat(rxfile: str) -> np.ndarray:
    path, off = parse_rxfile(rxfile)
    with open(path, "rb") as fd:
        fd.seek(off)
        return read_mat_fd(fd)


def read_mat_shape(rxfile: str) -> Tuple[int, int]:
    """Rows/cols from the header only (the reference's utt2num_frames fallback reads the whole matrix)."""
    path, off = parse_rxfile(rxfile)
    with open(path, "rb") as fd:
        fd.seek(off)
        assert fd.read(2) == b"\0B"
        tok = _read_token(fd)
        if tok in ("FM", "DM"):
            return _read_int(fd), _read_int(fd)
        _, _, rows, cols = struct.unpack("<ffii", fd.read(16))
        return rows, cols


def write_ark(path: str, mats: dict) -> dict:
    """Write `{utt_id: float32 matrix}` as one uncompressed ark; returns `{utt_id: "path:offset"}` (the scp)."""
    scp = {}
    with open(path, "wb") as fd:
        for k, m in mats.items():
            fd.write(k.encode() + b" ")
            scp[k] = f"{path}:{fd.tell()}"
            m = np.ascontiguousarray(m, dtype="<f4")
            fd.write(b"\0BFM " + b"\4" + struct.pack("<i", m.shape[0]) + b"\4" + struct.pack("<i", m.shape[1]))
            fd.write(m.tobytes())
    return scp
