"""Lane-level CPU model of csrc/wgrad_w8.hip `wgrad_w8_kernel` (8 wavefronts, 256 x 256 output tile; both operand tiles are
[64 reduction rows][256 columns] as they lie in memory, moved by global_load_lds and read as MFMA fragments by
ds_read_b64_tr_b16): the kernel's index arithmetic transcribed — load-instruction rows / slots with the source-side swizzle, the
column clamp at the row pitch, zero-page rows past M, fragment addresses and their +4-row / +32-row immediates, the SWAPPED MFMA
operand order (a lane ends up with four consecutive output columns of one row), the bias sums routed to wavefront wn for row
tiles wn and 4 + wn — checked against dy^T x in float64, together with the LDS bank rules (tests/test_kernel_models.py).
    python tools/emu_wgrad_w8.py"""
import numpy as np

from emu_wgrad_tr import Lds, f256, mfma32
from lds_layout_check import HALF_GROUPS, worst

PITCH, A_BYTES = 512, 64 * 512


def tile(dy, x, m0, n0, R, Cn):
    """one workgroup -> (C [256][256] (rows >= R / columns >= Cn garbage-free zeros), bias [256], worst bank-conflict way)"""
    M, lda, ldb = dy.shape[0], dy.shape[1], x.shape[1]
    lds = Lds(2 * A_BYTES)
    acc = [[[np.zeros((64, 4)) for _ in range(4)] for _ in range(8)] for _ in range(8)]  # [wave][i][j]
    accb = [[np.zeros((64, 4)) for _ in range(2)] for _ in range(8)]
    ways = 1
    zero = np.zeros(8)
    lane = np.arange(64)
    for kt in range((M + 63) // 64):
        for w in range(8):  # issue(): instruction i of wave w fills image rows (w + 8 i) * 2 + (lane >> 5), slot lane & 31
            for i in range(4):
                rows_a, rows_b = [], []
                for l in range(64):
                    row = (w + 8 * i) * 2 + (l >> 5)
                    slot = l & 31
                    ca = min(m0 + 8 * (slot ^ f256(row)), lda - 8)  # a column past the pitch re-reads the row's last 16 bytes
                    cb = min(n0 + 8 * (slot ^ f256(row)), ldb - 8)
                    m = kt * 64 + row
                    rows_a.append(dy[m, ca:ca + 8] if m < M else zero)
                    rows_b.append(x[m, cb:cb + 8] if m < M else zero)
                lds.glds((w + 8 * i) * 1024, rows_a)
                lds.glds(A_BYTES + (w + 8 * i) * 1024, rows_b)
        for w in range(8):
            wm, wn = w >> 2, w & 3
            g, j = lane >> 4, lane & 15
            e, q = j >> 2, j & 3
            row0 = 8 * g + e
            fz = np.array([f256(r) for r in row0])
            adA = [row0 * PITCH + ((((wm * 128 + 16 * i) // 8 + (q >> 1)) ^ fz) << 4) + (q & 1) * 8 for i in range(8)]
            adB = [A_BYTES + row0 * PITCH + ((((wn * 64 + 16 * jj) // 8 + (q >> 1)) ^ fz) << 4) + (q & 1) * 8 for jj in range(4)]
            for ks in range(2):
                def frag(ad):
                    nonlocal ways
                    o = []
                    for h in range(2):
                        a = ad + h * 4 * PITCH + ks * 32 * PITCH
                        ways = max(ways, worst(HALF_GROUPS, lambda l: int(a[l]), 8))
                        o.append(lds.trr(a))
                    return np.concatenate(o, 1)
                af = [frag(adA[i]) for i in range(8)]
                bf = [frag(adB[jj]) for jj in range(4)]
                for i in range(8):
                    for jj in range(4):
                        acc[w][i][jj] = mfma32(bf[jj], af[i], acc[w][i][jj])  # operands swapped: x fragment first
                for t in range(2):  # bias sums of row tiles wn and 4 + wn: ones x dy fragment
                    accb[w][t] = mfma32(np.ones((64, 8)), af[4 * t + wn], accb[w][t])
    C = np.zeros((256, 256))
    bias = np.zeros(256)
    for w in range(8):
        wm, wn = w >> 2, w & 3
        for l in range(64):
            g4, lj = l >> 4, l & 15
            for i in range(8):
                m = m0 + wm * 128 + 16 * i + lj
                if m >= R:
                    continue
                for jj in range(4):
                    for r in range(4):
                        n = n0 + wn * 64 + 16 * jj + 4 * g4 + r
                        if n < Cn:
                            C[m - m0, n - n0] = acc[w][i][jj][l, r]
            if g4 == 0:
                for t in range(2):
                    n = m0 + wm * 128 + (4 * t + wn) * 16 + lj
                    if n < R:
                        bias[n - m0] = accb[w][t][l, 0]
    return C, bias, ways


def run(M=150, R=300, Cn=264, pad_dy=4, pad_x=0, seed=0):
    """R output rows (columns of dy, pitch R + pad_dy rounded to 8), Cn output columns (columns of x), M reduction rows."""
    rng = np.random.default_rng(seed)
    lda = (R + pad_dy + 7) // 8 * 8
    ldb = (Cn + pad_x + 7) // 8 * 8
    dy, x = rng.standard_normal((M, lda)), rng.standard_normal((M, ldb))
    ref = dy[:, :R].T @ x[:, :Cn]
    refb = dy[:, :R].sum(0)
    err, ways = 0.0, 1
    for m0 in range(0, R, 256):
        for n0 in range(0, Cn, 256):
            C, bias, w = tile(dy, x, m0, n0, R, Cn)
            ways = max(ways, w)
            rr, cc = min(256, R - m0), min(256, Cn - n0)
            err = max(err, float(np.abs(C[:rr, :cc] - ref[m0:m0 + rr, n0:n0 + cc]).max()))
            if n0 == 0:
                err = max(err, float(np.abs(bias[:rr] - refb[m0:m0 + rr]).max()))
    return err, ways


if __name__ == "__main__":
    print(run())
    print(run(M=64, R=256, Cn=256, pad_dy=0))
