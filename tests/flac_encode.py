"""Test helper: a small FLAC ENCODER (RFC 9639) written only to produce test streams for the library's decoder
(espresso_amd/csrc/ingest.hip) — every subframe type, Rice partition orders with an escape partition, wasted bits, all four stereo
modes, explicit block sizes; STREAMINFO carries the MD5 of the audio so that `ea_audio_verify` checks the decoder against a second,
independent statement of the samples (hashlib).  Not part of the product."""
import hashlib

import numpy as np


class BitWriter:
    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def bits(self, value, k):
        if k == 0:
            return
        value &= (1 << k) - 1
        self.acc = (self.acc << k) | value
        self.n += k
        while self.n >= 8:
            self.n -= 8
            self.buf.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def unary(self, q):
        while q >= 32:
            self.bits(0, 32)
            q -= 32
        self.bits(1, q + 1)

    def align(self):
        if self.n:
            self.bits(0, 8 - self.n)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def _utf8(n):
    if n < 0x80:
        return bytes([n])
    out = []
    first_bits = 6
    while n >= (1 << first_bits):
        out.append(0x80 | (n & 0x3F))
        n >>= 6
        first_bits -= 1
    lead = (0xFF << (first_bits + 1)) & 0xFF
    out.append(lead | n)
    return bytes(reversed(out))


def _zigzag(v):
    return (v << 1) if v >= 0 else ((-v << 1) - 1)


def _residual(w, res, blocksize, order, po, escape_part=None, rice2=False):
    w.bits(1 if rice2 else 0, 2)
    w.bits(po, 4)
    parts = 1 << po
    i = 0
    for part in range(parts):
        cnt = (blocksize >> po) - (order if part == 0 else 0)
        seg = res[i:i + cnt]
        i += cnt
        pbits = 5 if rice2 else 4
        if escape_part is not None and part == escape_part:
            nb = max(1, max((int(abs(int(v))).bit_length() + 1 for v in seg), default=1))
            w.bits((1 << pbits) - 1, pbits)
            w.bits(nb, 5)
            for v in seg:
                w.bits(int(v), nb)
            continue
        mean = float(np.mean(np.abs(seg))) if len(seg) else 0.0
        k = max(0, min(14, int(np.ceil(np.log2(mean + 1)))))
        w.bits(k, pbits)
        for v in seg:
            u = _zigzag(int(v))
            w.unary(u >> k)
            w.bits(u & ((1 << k) - 1), k)


def _subframe(w, x, bps, kind, po=0, escape_part=None, wasted=0, rice2=False):
    """x: int array (already the channel's values, e.g. the side signal); bps: bits of this subframe's samples."""
    n = len(x)
    if wasted:
        assert all((int(v) & ((1 << wasted) - 1)) == 0 for v in x)
        x = x >> wasted
    eff = bps - wasted
    hdr_w = (1, wasted - 1) if wasted else (0, None)
    def head(type_bits):
        w.bits(0, 1)
        w.bits(type_bits, 6)
        w.bits(hdr_w[0], 1)
        if wasted:
            w.unary(wasted - 1)
    if kind == "constant":
        head(0)
        w.bits(int(x[0]), eff)
    elif kind == "verbatim":
        head(1)
        for v in x:
            w.bits(int(v), eff)
    elif kind.startswith("fixed"):
        order = int(kind[5:])
        head(8 + order)
        for v in x[:order]:
            w.bits(int(v), eff)
        xl = x.astype(np.int64)
        res = np.diff(xl, n=order) if order else xl  # the order-th finite difference = the fixed predictor's residual
        _residual(w, res, n, order, po, escape_part, rice2)
    elif kind.startswith("lpc"):
        order = int(kind[3:])
        prec, shift = 14, 10
        # a deliberately simple predictor (quantised taps of the order-2 fixed predictor + small extras): exercises the LPC path
        taps = [2, -1] + [0] * (order - 2) if order >= 2 else [1]
        coef = [int(round(t * (1 << shift))) for t in taps][:order]
        if order > 2:
            coef[-1] = 3
        head(32 + order - 1)
        for v in x[:order]:
            w.bits(int(v), eff)
        w.bits(prec - 1, 4)
        w.bits(shift, 5)
        for c in coef:
            w.bits(c, prec)
        xl = [int(v) for v in x]
        res = []
        for i in range(order, n):
            acc = sum(coef[j] * xl[i - 1 - j] for j in range(order))
            res.append(xl[i] - (acc >> shift))
        _residual(w, np.array(res, dtype=np.int64), n, order, po, escape_part, rice2)
    else:
        raise ValueError(kind)


def encode(samples, sample_rate=16000, bits=16, blocksize=1152, stereo_mode="independent", plan=None, with_md5=True):
    """samples: int array [n] (mono) or [n][2].  plan(block_index, channel) -> dict(kind=..., po=..., escape_part=..., wasted=...,
    rice2=...) chooses the subframe coding per block and channel (default: fixed2, partition order 2)."""
    x = np.asarray(samples)
    if x.ndim == 1:
        x = x[:, None]
    n, nch = x.shape
    bytes_ps = (bits + 7) // 8
    raw = bytearray()
    for row in x:
        for v in row:
            raw += int(v).to_bytes(bytes_ps, "little", signed=True)
    md5 = hashlib.md5(bytes(raw)).digest() if with_md5 else bytes(16)
    frames = bytearray()
    nblocks = (n + blocksize - 1) // blocksize
    bs_codes = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
    min_frame, max_frame = 1 << 24, 0
    for bi in range(nblocks):
        blk = x[bi * blocksize:(bi + 1) * blocksize].astype(np.int64)
        m = len(blk)
        w = BitWriter()
        w.bits(0b11111111111110, 14)
        w.bits(0, 1)
        w.bits(0, 1)  # fixed block size: the coded number is the frame index
        if m in bs_codes:
            bs_code, extra = bs_codes[m], None
        elif m <= 256:
            bs_code, extra = 6, (m - 1, 8)
        else:
            bs_code, extra = 7, (m - 1, 16)
        w.bits(bs_code, 4)
        w.bits(0, 4)  # sample rate from STREAMINFO
        if nch == 1:
            ca = 0
        else:
            ca = {"independent": 1, "left_side": 8, "side_right": 9, "mid_side": 10}[stereo_mode]
        w.bits(ca, 4)
        w.bits({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}.get(bits, 0) if bi % 2 else 0, 3)  # explicit and from-STREAMINFO alternate
        w.bits(0, 1)
        for b in _utf8(bi):
            w.bits(b, 8)
        if extra:
            w.bits(*extra)
        w.bits(crc8(bytes(w.buf)), 8)
        chans = [blk[:, c] for c in range(nch)]
        sub_bps = [bits] * nch
        if nch == 2 and ca == 8:
            chans = [chans[0], chans[0] - chans[1]]
            sub_bps = [bits, bits + 1]
        elif nch == 2 and ca == 9:
            chans = [chans[0] - chans[1], chans[1]]
            sub_bps = [bits + 1, bits]
        elif nch == 2 and ca == 10:
            side = chans[0] - chans[1]
            mid = (chans[0] + chans[1]) >> 1
            chans = [mid, side]
            sub_bps = [bits, bits + 1]
        for c in range(nch):
            p = dict(kind="fixed2", po=2, escape_part=None, wasted=0, rice2=False)
            if plan is not None:
                p.update(plan(bi, c) or {})
            po = p["po"]
            order = int(p["kind"][5:]) if p["kind"].startswith("fixed") else int(p["kind"][3:]) if p["kind"].startswith("lpc") else 0
            while po > 0 and (m % (1 << po) or (m >> po) < order):
                po -= 1
            esc = p["escape_part"] if p["escape_part"] is not None and p["escape_part"] < (1 << po) else None
            kind = p["kind"]
            if kind == "constant" and len(set(int(v) for v in chans[c])) != 1:
                kind = "verbatim"
            if order > m:  # a last block shorter than the predictor order (RFC 9639 9.2.6: warm-up samples must fit the block)
                kind, po, esc = "verbatim", 0, None
            _subframe(w, chans[c], sub_bps[c], kind, po, esc, p["wasted"], p["rice2"])
        w.align()
        body = bytes(w.buf)
        frame = body + crc16(body).to_bytes(2, "big")
        frames += frame
        min_frame, max_frame = min(min_frame, len(frame)), max(max_frame, len(frame))
    si = BitWriter()
    si.bits(blocksize, 16)
    si.bits(blocksize, 16)
    si.bits(min_frame if nblocks else 0, 24)
    si.bits(max_frame, 24)
    si.bits(sample_rate, 20)
    si.bits(nch - 1, 3)
    si.bits(bits - 1, 5)
    si.bits(n, 36)
    out = bytearray(b"fLaC")
    out += bytes([0x00, 0, 0, 34]) + bytes(si.buf) + md5           # STREAMINFO (not last)
    out += bytes([0x84, 0, 0, 8]) + (4).to_bytes(4, "little") + b"test"  # a VORBIS_COMMENT-shaped last block the decoder skips
    out += frames
    return bytes(out)
