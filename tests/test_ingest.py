"""Waveform ingestion (csrc/ingest.hip behind include/espresso_amd.h, espresso_amd/data/audio_utils.py): PCM WAV and FLAC files
decoded to int16 by the library — replaces fairseq/data/audio/audio_utils.py:74-118 get_waveform (soundfile) as called by
espresso/data/feat_text_dataset.py:128-155.  Host code only: these tests run without a GPU."""
import hashlib
import os
import struct
import wave

import numpy as np
import pytest

from espresso_amd.data import audio_utils as AU
from tests import flac_encode as FE


def _signal(n, seed=0, amp=9000):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = amp * np.sin(2 * np.pi * 220 * t / 16000) + 0.3 * amp * np.sin(2 * np.pi * 1733 * t / 16000 + 1.0) + rng.normal(0, amp * 0.05, n)
    return np.clip(np.round(x), -32768, 32767).astype(np.int64)


def _write_wav(path, x, ch=1, width=2, rate=16000, extra_chunk=False, extensible=False):
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    if width == 1:
        raw = (x + 128).astype(np.uint8).tobytes()
    elif width == 2:
        raw = x.astype("<i2").tobytes()
    elif width == 3:
        raw = b"".join(int(v).to_bytes(3, "little", signed=True) for v in x.reshape(-1))
    else:
        raw = x.astype("<i4").tobytes()
    nch = x.shape[1]
    fmt_body = struct.pack("<HHIIHH", 0xFFFE if extensible else 1, nch, rate, rate * nch * width, nch * width, 8 * width)
    if extensible:
        fmt_body += struct.pack("<HHI", 22, 8 * width, 0) + struct.pack("<H", 1) + bytes(14)
    chunks = b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body
    if extra_chunk:
        chunks += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"  # odd length: padded to even
    chunks += b"data" + struct.pack("<I", len(raw)) + raw
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def test_wav_variants(tmp_path):
    x = _signal(5000)
    p = str(tmp_path / "a.wav")
    _write_wav(p, x)
    assert AU.probe(p) == (5000, 16000, 1, 16)
    y, sr = AU.read_i16(p)
    assert sr == 16000 and np.array_equal(y, x)
    # the standard library agrees (the reader the round 1 - 4 path used)
    with wave.open(p, "rb") as w:
        assert np.array_equal(np.frombuffer(w.readframes(w.getnframes()), dtype="<i2"), y)
    st = np.stack([x, -x // 2], axis=1)
    _write_wav(p, st, extra_chunk=True)
    assert AU.probe(p) == (5000, 16000, 2, 16)
    assert np.array_equal(AU.read_i16(p)[0], x)  # mono = channel 0 (espresso/tools/utils.py:438-440)
    _write_wav(p, x, extensible=True)
    assert np.array_equal(AU.read_i16(p)[0], x)
    _write_wav(p, x >> 8, width=1)
    assert np.array_equal(AU.read_i16(p)[0], (x >> 8) << 8)
    _write_wav(p, x * 256 + 17, width=3)
    assert AU.probe(p)[3] == 24 and np.array_equal(AU.read_i16(p)[0], x)
    _write_wav(p, x * 65536 + 4242, width=4)
    assert np.array_equal(AU.read_i16(p)[0], x)
    # get_waveform keeps the reference's contract: fp32 at int16 scale; wider PCM keeps its fraction
    f, _ = AU.get_waveform(p)
    assert f.dtype == np.float32 and np.allclose(f, x + 4242 / 65536.0, atol=1e-2)
    with pytest.raises(OSError):
        AU.probe(str(tmp_path / "missing.wav"))
    open(p, "wb").write(b"not audio at all")
    with pytest.raises(OSError):
        AU.probe(p)


PLANS = {
    "fixed2_po2": None,
    "all_kinds": lambda b, c: [dict(kind="constant"), dict(kind="verbatim"), dict(kind="fixed0", po=0), dict(kind="fixed1", po=1),
                               dict(kind="fixed3", po=3), dict(kind="fixed4", po=2, escape_part=1), dict(kind="lpc2", po=2),
                               dict(kind="lpc8", po=1, rice2=True), dict(kind="lpc1", po=0), dict(kind="lpc32", po=0)][(b + 3 * c) % 10],
}


@pytest.mark.parametrize("plan", list(PLANS))
@pytest.mark.parametrize("stereo", [None, "independent", "left_side", "side_right", "mid_side"])
def test_flac_decoder_against_test_encoder(tmp_path, plan, stereo):
    n = 1152 * 9 + 333  # a short last block (explicit 16-bit block size)
    x = _signal(n, seed=1)
    x[1152 * 2:1152 * 3] = 77  # a constant block
    src = x if stereo is None else np.stack([x, np.roll(x, 5) // 2 + _signal(n, seed=2, amp=300)], axis=1)
    data = FE.encode(src, stereo_mode=stereo or "independent", plan=PLANS[plan])
    p = str(tmp_path / "t.flac")
    open(p, "wb").write(data)
    assert AU.probe(p) == (n, 16000, 1 if stereo is None else 2, 16)
    y, sr = AU.read_i16(p)
    assert sr == 16000 and np.array_equal(y, x)
    from espresso_amd import _lib

    assert _lib.lib().ea_audio_verify(p.encode()) == 1  # MD5 of ALL channels == hashlib's over the source samples
    bad = bytearray(data)
    bad[len(bad) // 2] ^= 0x10
    open(p, "wb").write(bytes(bad))
    out = np.empty(n, dtype=np.int16)
    import ctypes

    rc = _lib.lib().ea_audio_read_i16(p.encode(), out.ctypes.data_as(ctypes.c_void_p), n, None)
    assert rc < 0, rc  # a flipped bit is caught by the frame checksums (or breaks the stream), never returned as audio


def test_flac_other_sample_sizes_and_wasted_bits(tmp_path):
    from espresso_amd import _lib

    x = _signal(4096 * 2 + 100, seed=3)
    p = str(tmp_path / "w.flac")
    q = _signal(len(x), seed=4, amp=2000) * 4  # two wasted (always-zero) low bits
    open(p, "wb").write(FE.encode(q, blocksize=4096, plan=lambda b, c: dict(kind="fixed2", wasted=2)))
    assert np.array_equal(AU.read_i16(p)[0], q) and _lib.lib().ea_audio_verify(p.encode()) == 1
    open(p, "wb").write(FE.encode(x * 256 + 3, bits=24, blocksize=4096))
    assert AU.probe(p)[3] == 24 and np.array_equal(AU.read_i16(p)[0], x) and _lib.lib().ea_audio_verify(p.encode()) == 1
    open(p, "wb").write(FE.encode(x >> 8, bits=8, blocksize=256))
    assert np.array_equal(AU.read_i16(p)[0], (x >> 8) << 8) and _lib.lib().ea_audio_verify(p.encode()) == 1
    open(p, "wb").write(FE.encode(x, blocksize=4096, with_md5=False))  # no signature recorded: verify passes on the checksums
    assert _lib.lib().ea_audio_verify(p.encode()) == 1


REF_FLAC = "/root/reference/examples/hubert/tests/6313-76958-0021.flac"


@pytest.mark.skipif(not os.path.exists(REF_FLAC), reason="the reference tree (its LibriSpeech test utterance) is only in the build container")
def test_flac_decoder_on_a_real_librispeech_file():
    """libFLAC-encoded LibriSpeech audio from the reference's own test data: the MD5 of what this decoder produces must equal the
    signature libFLAC stored in STREAMINFO, every frame CRC-16 must hold."""
    from espresso_amd import _lib

    n, sr, ch, bits = AU.probe(REF_FLAC)
    assert (sr, ch, bits) == (16000, 1, 16) and n > 16000
    assert _lib.lib().ea_audio_verify(REF_FLAC.encode()) == 1
    y, _ = AU.read_i16(REF_FLAC)
    sig = open(REF_FLAC, "rb").read()[8 + 18:8 + 34]
    assert hashlib.md5(y.astype("<i2").tobytes()).digest() == sig and np.abs(y).max() > 500


def test_batch_reader_and_lazy_collate(tmp_path):
    """ea_audio_read_batch_i16 on several threads == file-by-file reads; the lazy collate path (files decoded inside the collater,
    int16 staging buffer) == the eager path (fp32 arrays) value for value, with the same sort order and offsets."""
    import torch

    from espresso_amd.data.asr_dataset import AudioWaveDataset, collate

    paths, sigs = [], []
    for i in range(23):
        x = _signal(3000 + 517 * i, seed=10 + i)
        p = str(tmp_path / (f"u{i}.flac" if i % 3 == 0 else f"u{i}.wav"))
        if i % 3 == 0:
            open(p, "wb").write(FE.encode(x))
        else:
            _write_wav(p, x)
        paths.append(p)
        sigs.append(x)
    offsets = np.zeros(len(paths) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(s) for s in sigs])
    for nt in (1, 4, 64):
        buf = np.full(int(offsets[-1]), -1, dtype=np.int16)
        rates = AU.read_batch_i16(paths, buf, offsets, nt)
        assert rates == [16000] * len(paths) and np.array_equal(buf, np.concatenate(sigs))
    short = offsets.copy()
    short[-1] -= 1
    with pytest.raises(OSError):
        AU.read_batch_i16(paths, np.zeros(int(offsets[-1]), dtype=np.int16), short, 4)

    ds = AudioWaveDataset([f"utt{i}" for i in range(len(paths))], paths)
    def items(lazy):
        ds.lazy = lazy
        return [{"id": i, "utt_id": ds.utt_ids[i], "source": ds[i], "target": torch.tensor([4 + i % 5, 7, 2])} for i in range(len(paths))]
    a = collate(items(False), pad_idx=1, eos_idx=2)
    b = collate(items(True), pad_idx=1, eos_idx=2, wave_workers=4)
    assert a["wav"].dtype == torch.float32 and b["wav"].dtype == torch.int16
    assert torch.equal(a["wav"], b["wav"].float()) and torch.equal(a["wav_offsets"], b["wav_offsets"])
    assert a["utt_id"] == b["utt_id"] and torch.equal(a["net_input"]["src_lengths"], b["net_input"]["src_lengths"])
    assert torch.equal(a["target"], b["target"]) and a["num_samples"] == b["num_samples"]


def test_readers_survive_corrupt_files_under_sanitizers(tmp_path):
    """The WAV / FLAC readers parse files from disk: whatever the bytes are they must RETURN (samples or an error code).
    tests/fuzz_ingest_main.cpp includes csrc/ingest.hip as it lies, is built here with g++ -fsanitize=address,undefined
    -fno-sanitize-recover and runs 300 mutants (bit flips, header garbage, truncation, runs of 0x00 / 0xff, duplicated slices,
    huge length fields) of each of nine seed files — every PCM width, stereo, extra chunks, every FLAC subframe type / Rice
    mode / stereo decorrelation / wasted bits — through probe / read (also into a deliberately short buffer) / verify / the
    threaded batch entry.  (Round 5: the first 300 000 mutants found three undefined-behaviour sites — signed overflow in the
    stereo decorrelation and the LPC accumulator on corrupt streams, a negative left shift, memcpy(NULL, …, 0) — fixed; the
    following 430 000 were clean.)"""
    import shutil
    import subprocess

    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuzz_ingest")
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-x", "c++",
                            "-I", os.path.join(root, "include"), "-pthread", os.path.join(root, "tests", "fuzz_ingest_main.cpp"), "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and ("asan" in build.stderr.lower() or "sanitize" in build.stderr.lower()):
        pytest.skip("sanitizer runtime not installed: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    x = _signal(6000, seed=3, amp=3000).astype(np.int64)
    seeds = tmp_path / "seeds"
    seeds.mkdir()
    _write_wav(str(seeds / "seed_a.wav"), x, extra_chunk=True)
    _write_wav(str(seeds / "seed_b.wav"), x[:2000], width=3)
    _write_wav(str(seeds / "seed_c.wav"), np.stack([x[:3000], -x[:3000]], 1), ch=2, extensible=True)
    _write_wav(str(seeds / "seed_d.wav"), x[:1500], width=1)
    _write_wav(str(seeds / "seed_e.wav"), x[:1500], width=4)
    kinds = ["constant", "verbatim", "fixed0", "fixed1", "fixed3", "fixed4", "lpc2", "lpc5", "lpc8"]
    (seeds / "seed_f.flac").write_bytes(FE.encode(x, blocksize=1152, plan=lambda b, c: dict(
        kind=kinds[b % 9], po=b % 4, escape_part=(0 if b % 3 == 0 else None), rice2=bool(b % 2))))
    (seeds / "seed_g.flac").write_bytes(FE.encode(np.stack([x[:4000], (x[:4000] * 0.7).astype(np.int64)], 1), blocksize=576,
                                                  stereo_mode="mid_side", plan=lambda b, c: dict(kind="lpc4", po=3)))
    (seeds / "seed_h.flac").write_bytes(FE.encode(np.stack([x[:3000], x[:3000] + 5], 1), blocksize=4096, stereo_mode="left_side"))
    (seeds / "seed_i.flac").write_bytes(FE.encode((x[:2500] // 4) * 4, blocksize=1000, plan=lambda b, c: dict(kind="fixed2", wasted=2)))
    run = subprocess.run([exe, str(seeds), "300", "11"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", LD_PRELOAD=""))
    assert run.returncode == 0 and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, (run.stdout[-500:], run.stderr[-3000:])
    assert "2700 mutants" in run.stdout, run.stdout


def test_advice_r5_wav_header_edge_cases(tmp_path):
    """(1) a WAVE_FORMAT_EXTENSIBLE `fmt ` chunk that is declared long but truncated after its first 16 bytes
    must be rejected (-3), not read past the buffer (found under ASan); (2) ea_audio_verify of a WAV is 1 ("WAV: always 1"),
    not the -5 of a capacity check that ran before the header-only early-out; (3) a FLAC frame whose sample size disagrees with
    STREAMINFO fails the stream (-6) instead of being silently truncated."""
    import ctypes

    from espresso_amd import _lib

    lib = _lib.lib()
    p = str(tmp_path / "trunc.wav")
    fmt16 = struct.pack("<HHIIHH", 0xFFFE, 1, 16000, 32000, 2, 16)
    open(p, "wb").write(b"RIFF" + struct.pack("<I", 28) + b"WAVE" + b"fmt " + struct.pack("<I", 40) + fmt16)  # 36 bytes
    n, sr, ch, bits = ctypes.c_long(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    assert lib.ea_audio_probe(p.encode(), ctypes.byref(n), ctypes.byref(sr), ctypes.byref(ch), ctypes.byref(bits)) != 0
    buf = np.zeros(16, dtype=np.int16)
    assert lib.ea_audio_read_i16(p.encode(), buf.ctypes.data_as(ctypes.c_void_p), 16, ctypes.byref(sr)) < 0
    # a complete EXTENSIBLE header still reads
    x = _signal(100)
    q = str(tmp_path / "ext.wav")
    _write_wav(q, x, extensible=True)
    assert np.array_equal(AU.read_i16(q)[0], x)
    assert lib.ea_audio_verify(q.encode()) == 1
    w = str(tmp_path / "plain.wav")
    _write_wav(w, x)
    assert lib.ea_audio_verify(w.encode()) == 1
    # FLAC: flip the frame header's sample-size code of a 16-bit stream to "24 bits" and repair the header CRC-8
    f = bytearray(FE.encode(_signal(4096, seed=3), blocksize=4096))
    i = f.index(b"\xff\xf8", 42)
    hdr_len = None
    for L in range(5, 17):  # the header ends where its CRC-8 matches
        c = 0
        for byte in f[i:i + L]:
            c ^= byte
            for _ in range(8):
                c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
        if c == f[i + L]:
            hdr_len = L
            break
    assert hdr_len is not None
    assert (f[i + 3] >> 1) & 7 in (0, 4)
    f[i + 3] = (f[i + 3] & 0xF1) | (6 << 1)
    c = 0
    for byte in f[i:i + hdr_len]:
        c ^= byte
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    f[i + hdr_len] = c
    g = str(tmp_path / "bps.flac")
    open(g, "wb").write(bytes(f))
    out = np.zeros(4096, dtype=np.int16)
    assert lib.ea_audio_read_i16(g.encode(), out.ctypes.data_as(ctypes.c_void_p), 4096, ctypes.byref(sr)) == -6
