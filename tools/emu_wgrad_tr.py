"""Lane-level CPU model of csrc/gemm.hip `wgrad_group_tr_kernel` (weight gradients dW[n][k] += sum_m dy[m][n] x[m][k] with both
operand tiles brought global -> LDS by global_load_lds as they lie in memory — reduction index m along the ROWS — and read
as MFMA fragments by ds_read_b64_tr_b16): the kernel's index arithmetic transcribed, so that the 16-byte-slot swizzles, the
fragment addresses / immediate offsets and the operand order are checked on the CPU (tests/test_kernel_models.py) against
dy^T x in float64, together with the LDS bank rules.   python tools/emu_wgrad_tr.py"""
import numpy as np

from lds_layout_check import HALF_GROUPS, worst


def f256(r):  # slot swizzle of a 256-byte-pitch image (16 slots per row)
    return ((r & 3) | ((r >> 1) & 4)) << 1


def f128(r):  # 128-byte pitch (8 slots per row)
    return (((r >> 1) & 1) | ((r >> 2) & 2)) << 1


class Lds:
    def __init__(self, nbytes):
        self.c = np.full(nbytes // 2, np.nan)

    def glds(self, dst, rows):  # one wave instruction: lane l lands at dst + 16 l
        for l in range(64):
            self.c[(dst + 16 * l) // 2:(dst + 16 * l) // 2 + 8] = rows[l]

    def trr(self, addr):  # ds_read_b64_tr_b16 (hardware probe: profiles/r02_ds_read_tr_probe.txt)
        piece = np.stack([self.c[a // 2:a // 2 + 4] for a in addr])
        out = np.empty((64, 4))
        for l in range(64):
            for e in range(4):
                out[l, e] = piece[16 * (l >> 4) + 4 * e + ((l & 15) >> 2), l & 3]
        return out


def mfma32(a, b, c):
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a[l]
        B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b[l]
    D = A @ B
    out = c.copy()
    for l in range(64):
        out[l] += D[4 * (l >> 4):4 * (l >> 4) + 4, l & 15]
    return out


def tile(dy, x, n0, k0, BM):
    """one workgroup: -> (C [BM][128], bias [BM], worst bank-conflict way of the fragment reads)"""
    M = dy.shape[0]
    PA, PB = BM * 2, 256
    fA = f256 if BM == 128 else f128
    A_BYTES = 64 * PA
    lds = Lds(A_BYTES + 64 * PB)
    NA, NB = BM // 32, 4
    NJ = 4 if BM == 128 else 2
    acc = [[[np.zeros((64, 4)) for _ in range(NJ)] for _ in range(4)] for _ in range(4)]  # [wave][i][j]
    accb = [[np.zeros((64, 4)) for _ in range(4)] for _ in range(4)]
    ways = 1
    zero = np.zeros(8)
    for kt in range((M + 63) // 64):
        for w in range(4):  # issue(): every wave's share of the stage
            for i in range(NA):
                t = w + 4 * i
                rows = []
                for l in range(64):
                    row = t * (1024 // PA) + l // (PA // 16)
                    slot = l % (PA // 16)
                    m = kt * 64 + row
                    c = n0 + 8 * (slot ^ fA(row))
                    rows.append(dy[m, c:c + 8] if m < M else zero)  # rows past M come from the zero page
                lds.glds(t * 1024, rows)
            for i in range(NB):
                t = w + 4 * i
                rows = []
                for l in range(64):
                    row = t * 4 + (l >> 4)
                    slot = l & 15
                    m = kt * 64 + row
                    c = k0 + 8 * (slot ^ f256(row))
                    rows.append(x[m, c:c + 8] if m < M else zero)
                lds.glds(A_BYTES + t * 1024, rows)
        for w in range(4):
            wm = (w >> 1) if BM == 128 else 0
            wcol = (w & 1) * 64 if BM == 128 else w * 32
            lane = np.arange(64)
            g, j = lane >> 4, lane & 15
            e, q = j >> 2, j & 3
            row0 = 8 * g + e
            adA = [row0 * PA + ((((wm * 64 + 16 * i) // 8 + (q >> 1)) ^ np.array([fA(r) for r in row0])) << 4) + (q & 1) * 8 for i in range(4)]
            adB = [A_BYTES + row0 * PB + ((((wcol + 16 * jj) // 8 + (q >> 1)) ^ np.array([f256(r) for r in row0])) << 4) + (q & 1) * 8
                   for jj in range(NJ)]
            for ks in range(2):
                af, bf = [], []
                for i in range(4):
                    o = []
                    for h in range(2):
                        a = adA[i] + h * 4 * PA + ks * 32 * PA
                        ways = max(ways, worst(HALF_GROUPS, lambda l: int(a[l]), 8))
                        o.append(lds.trr(a))
                    af.append(np.concatenate(o, 1))
                for jj in range(NJ):
                    o = []
                    for h in range(2):
                        a = adB[jj] + h * 4 * PB + ks * 32 * PB
                        ways = max(ways, worst(HALF_GROUPS, lambda l: int(a[l]), 8))
                        o.append(lds.trr(a))
                    bf.append(np.concatenate(o, 1))
                for i in range(4):
                    for jj in range(NJ):
                        acc[w][i][jj] = mfma32(af[i], bf[jj], acc[w][i][jj])
                    accb[w][i] = mfma32(af[i], np.ones((64, 8)), accb[w][i])
    C = np.zeros((BM, 128))
    bias = np.zeros(BM)
    for w in range(4):
        wm = (w >> 1) if BM == 128 else 0
        wcol = (w & 1) * 64 if BM == 128 else w * 32
        for i in range(4):
            for l in range(64):
                for r in range(4):
                    n = wm * 64 + i * 16 + (l >> 4) * 4 + r
                    for jj in range(NJ):
                        C[n, wcol + jj * 16 + (l & 15)] = acc[w][i][jj][l, r]
                    if (l & 15) == 0 and wcol == 0:
                        bias[n] = accb[w][i][l, r]
    return C, bias, ways


def run(M=150, BM=64, seed=0):
    rng = np.random.default_rng(seed)
    N, K = 2 * BM, 256
    dy, x = rng.standard_normal((M, N)), rng.standard_normal((M, K))
    err = 0.0
    ways = 1
    for n0 in range(0, N, BM):
        for k0 in range(0, K, 128):
            C, bias, w = tile(dy, x, n0, k0, BM)
            ways = max(ways, w)
            err = max(err, float(np.abs(C - dy[:, n0:n0 + BM].T @ x[:, k0:k0 + 128]).max()))
            err = max(err, float(np.abs(bias - dy[:, n0:n0 + BM].sum(0)).max()))
    return err, ways


if __name__ == "__main__":
    for BM in (64, 128):
        print(BM, run(BM=BM))
