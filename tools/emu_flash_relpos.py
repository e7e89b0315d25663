#!/usr/bin/env python3
"""Lane-level model of csrc/flash_relpos.hip on the CPU: the index arithmetic of the three kernels (LDS image swizzle,
global_load_lds placement, ds_read_b128 / ds_read_b64_tr_b16 fragment addresses, MFMA operand / accumulator layouts, the
positional-block ring, the skew / un-skew buffers, the exchange tiles, the keep-bit pieces) transcribed statement by
statement and executed on a software model of the wavefront, then compared with the direct formulas in float64.

The hardware semantics are modelled independently of the kernels:
  * global_load_lds (16 B): lane l's 16 bytes land at dst + 16 l;
  * ds_read_b64_tr_b16: lane l receives element (l&3) of the pieces of lanes 16 (l>>4) + 4 e + ((l&15)>>2), e = 0..3
    (probed on hardware: profiles/r02_ds_read_tr_probe.txt);
  * v_mfma_f32_16x16x32_bf16: A[m = l&15][k = 8 (l>>4) + e], B[k = 8 (l>>4) + e][n = l&15], D[m = 4 (l>>4) + r][n = l&15];
    v_mfma_f32_16x16x16_bf16: the same with 4 k values per lane.
Values are float64 (no bf16 rounding): any mismatch is an indexing error.  Run: python tools/emu_flash_relpos.py [T] [B]
(also imported by tests/test_host_logic.py).
"""
import sys
import numpy as np

IMG, BDP, OBP = 8192, 18, 88
BNC = 80 * BDP * 4
PP_OFF, BNC_OFF = 4 * IMG, 7 * IMG
LDS_BYTES = 7 * IMG + 4 * BNC
LANES = np.arange(64)
LI, G4 = LANES & 15, LANES >> 4


def swz(r):
    return ((r >> 2) & 1) | (((r >> 1) & 1) << 1) | ((((r >> 2) ^ (r >> 3)) & 1) << 2)


class Lds:
    """2-byte cells; an fp32 store occupies two cells (value, NaN marker)"""

    def __init__(self):
        self.c = np.full(LDS_BYTES // 2, np.nan)

    def glds(self, dst, rows):  # rows: [64 lanes][8 values]
        assert dst % 16 == 0
        for l in range(64):
            self.c[(dst + 16 * l) // 2:(dst + 16 * l) // 2 + 8] = rows[l]

    def ldf(self, addr):  # per-lane byte address -> [64][8]
        assert np.all(addr % 16 == 0)
        return np.stack([self.c[a // 2:a // 2 + 8] for a in addr])

    def ld64(self, addr):
        assert np.all(addr % 8 == 0)
        return np.stack([self.c[a // 2:a // 2 + 4] for a in addr])

    def trr(self, addr):
        piece = self.ld64(addr)
        out = np.empty((64, 4))
        for l in range(64):
            for e in range(4):
                out[l, e] = piece[16 * (l >> 4) + 4 * e + ((l & 15) >> 2), l & 3]
        return out

    def st32(self, addr, v):
        assert np.all(addr % 4 == 0)
        self.c[addr // 2] = v
        self.c[addr // 2 + 1] = np.nan

    def ld32(self, addr):
        return self.c[addr // 2]

    def st16(self, addr, v):
        self.c[addr // 2] = v

    def ld16(self, addr):
        return self.c[addr // 2]

    def st64(self, addr, v4):
        for l in range(64):
            self.c[addr[l] // 2:addr[l] // 2 + 4] = v4[l]


def mfma(a, b, c):
    """a, b: [64 lanes][kk] (kk = 8: 16x16x32, kk = 4: 16x16x16); c: [64][4]"""
    kk = a.shape[1]
    A = np.zeros((16, 4 * kk))
    Bm = np.zeros((4 * kk, 16))
    for l in range(64):
        A[l & 15, kk * (l >> 4):kk * (l >> 4) + kk] = a[l]
        Bm[kk * (l >> 4):kk * (l >> 4) + kk, l & 15] = b[l]
    D = A @ Bm
    out = c.copy()
    for l in range(64):
        out[l] += D[4 * (l >> 4):4 * (l >> 4) + 4, l & 15]
    return out


class LaneK:
    def __init__(self, w):
        self.w = w
        self.offk = [LI * 128 + (((ks * 4 + G4) ^ swz(LI)) << 4) for ks in range(2)]
        rowl = 4 * G4 + (LI >> 2)
        self.tro = [rowl * 128 + (((dt * 2 + ((LI & 3) >> 1)) ^ swz(rowl)) << 4) + (LI & 1) * 8 for dt in range(4)]
        self.bd_w = (G4 * 4) * BDP + LI
        self.bd_r = (15 - LI + G4 * 4) * BDP + LI


def issue_img(lds, dst, mat, row0, rmax, w):
    """mat: [rows][64] (one head's columns)"""
    for n in range(2):
        grp = w + 4 * n
        r = grp * 8 + (LANES >> 3)
        ch = (LANES & 7) ^ swz(r)
        g = np.minimum(np.maximum(row0 + r, 0), rmax)
        rows = np.stack([mat[g[l], ch[l] * 8:ch[l] * 8 + 8] for l in range(64)])
        lds.glds(dst + grp * 1024, rows)


def add_band(lds, acc_s, qv, blk_lo, blk_hi, bd, L):
    for ct in range(5):
        q = ct - L.w + 3
        p = (blk_hi if (q >> 2) else blk_lo) + (q & 3) * 2048
        t = np.zeros((64, 4))
        for ks in range(2):
            t = mfma(lds.ldf(p + L.offk[ks]), qv[ks], t)
        for r in range(4):
            lds.st32(bd + 4 * (L.bd_w + (ct * 16 + r) * BDP), t[:, r])
    for jt in range(4):
        for r in range(4):
            acc_s[jt][:, r] += lds.ld32(bd + 4 * (L.bd_r + (jt * 16 + r) * BDP))


def frag_rows(mat, rows, T):
    """per-lane 16-byte global loads of the prologue: rows[lane] clamped, columns ks*32 + g4*8 .."""
    rc = np.minimum(rows, T - 1)
    return [np.stack([mat[rc[l], ks * 32 + G4[l] * 8:ks * 32 + G4[l] * 8 + 8] for l in range(64)]) for ks in range(2)]


def keep_pieces(keep, T):
    """keep: [T][T] 0/1 -> pieces[kt][i][g] as in keep_bits_kernel"""
    nkt = (T + 63) // 64
    pc = np.zeros((nkt, nkt * 64, 4), dtype=np.int64)
    for kt in range(nkt):
        for i in range(T):
            for g in range(4):
                v = 0
                for jt in range(4):
                    for r in range(4):
                        j = kt * 64 + jt * 16 + g * 4 + r
                        if j < T and keep[i, j]:
                            v |= 1 << (jt * 4 + r)
                pc[kt, i, g] = v
    return pc


def kbit(piece, k):
    return ((piece >> k) & 1).astype(np.float64)


def fwd_wg(qu, qv, K, V, PP, kl, T, qt, pieces, inv_keep):
    """one workgroup of rp_fwd_kernel -> out rows [i0, i0+64), lse"""
    lds = Lds()
    i0 = qt * 64
    nt = (kl + 63) // 64
    R = 2 * T - 1
    pbase = (T - 1) - (i0 + 63)
    st = []
    for w in range(4):
        L = LaneK(w)
        i = i0 + 16 * w + LI
        st.append(dict(L=L, i=i, qu=frag_rows(qu, i, T), qv=frag_rows(qv, i, T), acc_o=[np.zeros((64, 4)) for _ in range(4)],
                       m=np.full(64, -np.inf), l=np.zeros(64)))
    for w in range(4):
        issue_img(lds, 0, K, 0, T - 1, w)
        issue_img(lds, IMG, V, 0, T - 1, w)
        issue_img(lds, PP_OFF, PP, pbase, R - 1, w)
        issue_img(lds, PP_OFF + IMG, PP, pbase + 64, R - 1, w)
    slot_lo = 0
    for t in range(nt):
        j0 = t * 64
        slot_hi = 0 if slot_lo == 2 else slot_lo + 1
        slot_nx = 0 if slot_hi == 2 else slot_hi + 1
        if t + 1 < nt:
            for w in range(4):
                stg = ((t + 1) & 1) * 2 * IMG
                issue_img(lds, stg, K, j0 + 64, T - 1, w)
                issue_img(lds, stg + IMG, V, j0 + 64, T - 1, w)
                issue_img(lds, PP_OFF + slot_nx * IMG, PP, pbase + 64 * (t + 2), R - 1, w)
        sK = (t & 1) * 2 * IMG
        sV = sK + IMG
        # (the real kernel issues the next tile BEFORE computing; the model must not let it clobber live data: checked by
        #  construction here because the next stage / ring slot are different buffers)
        for w in range(4):
            s = st[w]
            L = s["L"]
            acc_s = [np.zeros((64, 4)) for _ in range(4)]
            for ks in range(2):
                for jt in range(4):
                    acc_s[jt] = mfma(lds.ldf(sK + jt * 2048 + L.offk[ks]), s["qu"][ks], acc_s[jt])
            add_band(lds, acc_s, s["qv"], PP_OFF + slot_lo * IMG, PP_OFF + slot_hi * IMG, BNC_OFF + w * BNC, L)
            if j0 + 64 > kl:
                for jt in range(4):
                    for r in range(4):
                        acc_s[jt][:, r] = np.where(j0 + jt * 16 + G4 * 4 + r >= kl, -np.inf, acc_s[jt][:, r])
            tmax = np.max(np.stack([a.max(axis=1) for a in acc_s]), axis=0)
            tmax = np.maximum(tmax, tmax[LANES ^ 16])
            tmax = np.maximum(tmax, tmax[LANES ^ 32])
            m_new = np.maximum(s["m"], tmax)
            alpha = np.exp(s["m"] - m_new)
            psum = np.zeros(64)
            for jt in range(4):
                for r in range(4):
                    p = np.exp(acc_s[jt][:, r] - m_new)
                    psum += p
                    if pieces is not None:
                        pc = pieces[t, np.minimum(s["i"], pieces.shape[1] - 1), G4]
                        p = p * kbit(pc, jt * 4 + r)
                    acc_s[jt][:, r] = p
            s["l"] = s["l"] * alpha + psum
            s["m"] = m_new
            for dt in range(4):
                s["acc_o"][dt] *= alpha[:, None]
            for kb in range(2):
                pb = np.concatenate([acc_s[2 * kb], acc_s[2 * kb + 1]], axis=1)
                for dt in range(4):
                    vf = np.concatenate([lds.trr(sV + kb * 4096 + L.tro[dt]), lds.trr(sV + kb * 4096 + 2048 + L.tro[dt])], axis=1)
                    s["acc_o"][dt] = mfma(vf, pb, s["acc_o"][dt])
        slot_lo = slot_hi
    out = np.zeros((64, 64))
    lse = np.zeros(64)
    for w in range(4):
        s = st[w]
        l = s["l"] + s["l"][LANES ^ 16]
        l = l + l[LANES ^ 32]
        inv = (inv_keep if pieces is not None else 1.0) / l
        for dt in range(4):
            for r in range(4):
                out[16 * w + LI, dt * 16 + G4 * 4 + r] = s["acc_o"][dt][:, r] * inv
        lse[16 * w + LI] = s["m"] + np.log(l)
    return out, lse


def softmax_bwd_tile(acc_s, acc_dp, lse, Di, inv_keep, pc, jrel_end, want_pd):
    dsb, pdb = [], []
    for jt in range(4):
        ds = np.zeros((64, 4))
        pd = np.zeros((64, 4))
        for r in range(4):
            p = np.exp(acc_s[jt][:, r] - lse)
            p = np.where((jrel_end < 64) & (jt * 16 + G4 * 4 + r >= jrel_end), 0.0, p)
            dp = acc_dp[jt][:, r]
            if pc is not None:
                m = kbit(pc, jt * 4 + r)
                dp = dp * inv_keep * m
                pd[:, r] = p * inv_keep * m
            else:
                pd[:, r] = p
            ds[:, r] = p * (dp - Di)
        dsb.append(ds)
        pdb.append(pd)
    return dsb, pdb


def bwd_q_wg(qu, qv, K, V, PP, O, dO, lse_all, kl, T, qt, pieces, inv_keep, scaling, ld_bd):
    lds = Lds()
    i0 = qt * 64
    nt = (kl + 63) // 64
    R = 2 * T - 1
    pbase = (T - 1) - (i0 + 63)
    dBD = np.full((64, ld_bd), np.nan)  # rows i0.., only the band of the visited tiles is written here
    st = []
    for w in range(4):
        L = LaneK(w)
        i = i0 + 16 * w + LI
        ic = np.minimum(i, T - 1)
        dOf = frag_rows(dO, i, T)
        Of = frag_rows(O, i, T)
        Di = sum((dOf[ks] * Of[ks]).sum(axis=1) for ks in range(2))
        Di = Di + Di[LANES ^ 16]
        Di = Di + Di[LANES ^ 32]
        lse = np.where(i < T, lse_all[ic], np.inf)
        mlo = np.stack([(4 * G4 + e >= 15 - LI) for e in range(4)], axis=1)
        mhi = np.stack([(64 + 4 * G4 + e <= 78 - LI) for e in range(4)], axis=1)
        st.append(dict(L=L, i=i, qu=frag_rows(qu, i, T), qv=frag_rows(qv, i, T), dO=dOf, Di=Di, lse=lse, mlo=mlo, mhi=mhi,
                       t1=[np.zeros((64, 4)) for _ in range(4)], t2=[np.zeros((64, 4)) for _ in range(4)]))
    for w in range(4):
        issue_img(lds, 0, K, 0, T - 1, w)
        issue_img(lds, IMG, V, 0, T - 1, w)
        issue_img(lds, PP_OFF, PP, pbase, R - 1, w)
        issue_img(lds, PP_OFF + IMG, PP, pbase + 64, R - 1, w)
    slot_lo = 0
    for t in range(nt):
        j0 = t * 64
        slot_hi = 0 if slot_lo == 2 else slot_lo + 1
        slot_nx = 0 if slot_hi == 2 else slot_hi + 1
        if t + 1 < nt:
            for w in range(4):
                stg = ((t + 1) & 1) * 2 * IMG
                issue_img(lds, stg, K, j0 + 64, T - 1, w)
                issue_img(lds, stg + IMG, V, j0 + 64, T - 1, w)
                issue_img(lds, PP_OFF + slot_nx * IMG, PP, pbase + 64 * (t + 2), R - 1, w)
        sK = (t & 1) * 2 * IMG
        sV = sK + IMG
        blk_lo, blk_hi = PP_OFF + slot_lo * IMG, PP_OFF + slot_hi * IMG
        for w in range(4):
            s = st[w]
            L = s["L"]
            bd = BNC_OFF + w * BNC
            acc_s = [np.zeros((64, 4)) for _ in range(4)]
            acc_dp = [np.zeros((64, 4)) for _ in range(4)]
            for ks in range(2):
                for jt in range(4):
                    acc_s[jt] = mfma(lds.ldf(sK + jt * 2048 + L.offk[ks]), s["qu"][ks], acc_s[jt])
                    acc_dp[jt] = mfma(lds.ldf(sV + jt * 2048 + L.offk[ks]), s["dO"][ks], acc_dp[jt])
            add_band(lds, acc_s, s["qv"], blk_lo, blk_hi, bd, L)
            pc = None if pieces is None else pieces[t, np.minimum(s["i"], pieces.shape[1] - 1), G4]
            dsb, _ = softmax_bwd_tile(acc_s, acc_dp, s["lse"], s["Di"], inv_keep, pc, kl - j0, False)
            for kb in range(2):
                db = np.concatenate([dsb[2 * kb], dsb[2 * kb + 1]], axis=1)
                for dt in range(4):
                    kf = np.concatenate([lds.trr(sK + kb * 4096 + L.tro[dt]), lds.trr(sK + kb * 4096 + 2048 + L.tro[dt])], axis=1)
                    s["t1"][dt] = mfma(kf, db, s["t1"][dt])
            ob_w = LI * (OBP - 1) + 15 + G4 * 4
            for jt in range(4):
                for r in range(4):
                    lds.st16(bd + 2 * (ob_w + jt * 16 + r), dsb[jt][:, r])
            ob_rd = bd + LI * (OBP * 2) + G4 * 8
            for cb in range(2):
                blo = lds.ld64(ob_rd + cb * 64)
                bhi = lds.ld64(ob_rd + cb * 64 + 32)
                if cb == 0:
                    blo = np.where(s["mlo"], blo, 0.0)
                db = np.concatenate([blo, bhi], axis=1)
                assert not np.isnan(db).any(), "stale un-skew entry reached the t2 operand"
                qa = 2 * cb - w + 3
                qb = qa + 1
                pa = (blk_hi if (qa >> 2) else blk_lo) + (qa & 3) * 2048
                pb = (blk_hi if (qb >> 2) else blk_lo) + (qb & 3) * 2048
                for dt in range(4):
                    af = np.concatenate([lds.trr(pa + L.tro[dt]), lds.trr(pb + L.tro[dt])], axis=1)
                    s["t2"][dt] = mfma(af, db, s["t2"][dt])
            bl = np.where(s["mhi"], lds.ld64(ob_rd + 128), 0.0)
            qa = 4 - w + 3
            pa = (blk_hi if (qa >> 2) else blk_lo) + (qa & 3) * 2048
            for dt in range(4):
                s["t2"][dt] = mfma(lds.trr(pa + L.tro[dt]), bl, s["t2"][dt])
            row_w = i0 + 16 * w
            nrow = T - row_w
            for lane in range(64):
                j = j0 + lane
                if j < T:
                    for iw in range(16):
                        if iw < nrow:
                            dBD[16 * w + iw, (T - 1 - (row_w + iw)) + j] = lds.ld16(bd + 2 * (15 + lane + iw * (OBP - 1)))
        slot_lo = slot_hi
    t1 = np.zeros((64, 64))
    t2 = np.zeros((64, 64))
    D = np.zeros(64)
    for w in range(4):
        s = st[w]
        for dt in range(4):
            for r in range(4):
                t1[16 * w + LI, dt * 16 + G4 * 4 + r] = s["t1"][dt][:, r] * scaling
                t2[16 * w + LI, dt * 16 + G4 * 4 + r] = s["t2"][dt][:, r] * scaling
        D[16 * w + LI] = s["Di"]
    return t1, t2, dBD, D


def bwd_kv_wg(qu, qv, K, V, PP, dO, lse_all, D_all, kl, T, kt, pieces, inv_keep):
    lds = Lds()
    j0 = kt * 64
    nqt = (T + 63) // 64
    nq = nqt if j0 < kl else 0
    R = 2 * T - 1
    pb0 = (T - 1) - 63 + j0
    st = []
    for w in range(4):
        L = LaneK(w)
        kf = [frag_rows(K, j0 + jt * 16 + LI, T) for jt in range(4)]
        vf = [frag_rows(V, j0 + jt * 16 + LI, T) for jt in range(4)]
        exo = LI * 128 + (((G4 >> 1) ^ swz(LI)) << 4) + (G4 & 1) * 8
        exr, trk = [], []
        for hh in range(2):
            rl = 8 * (G4 & 1) + 4 * hh + (LI >> 2)
            exr.append((G4 >> 1) * BNC + rl * 128 + (((2 * w + ((LI & 3) >> 1)) ^ swz(rl)) << 4) + (LI & 1) * 8)
            rr = 8 * G4 + 4 * hh + (LI >> 2)
            trk.append(rr * 128 + ((((LI & 3) >> 1) ^ swz(rr)) << 4) + (LI & 1) * 8)
        st.append(dict(L=L, kf=kf, vf=vf, exo=exo, exr=exr, trk=trk, dk=[np.zeros((64, 4)) for _ in range(4)],
                       dv=[np.zeros((64, 4)) for _ in range(4)]))
    if nq > 0:
        for w in range(4):
            issue_img(lds, 0, qu, 0, T - 1, w)
            issue_img(lds, IMG, dO, 0, T - 1, w)
            issue_img(lds, PP_OFF, PP, pb0, R - 1, w)
            issue_img(lds, PP_OFF + 2 * IMG, PP, pb0 + 64, R - 1, w)
    slot_lo = 0
    for it in range(nq):
        i0 = it * 64
        slot_hi = 2 if slot_lo == 0 else slot_lo - 1
        slot_nx = 0 if slot_lo == 2 else slot_lo + 1
        if it + 1 < nq:
            for w in range(4):
                stg = ((it + 1) & 1) * 2 * IMG
                issue_img(lds, stg, qu, i0 + 64, T - 1, w)
                issue_img(lds, stg + IMG, dO, i0 + 64, T - 1, w)
                issue_img(lds, PP_OFF + slot_nx * IMG, PP, pb0 - 64 * (it + 1), R - 1, w)
        sQ = (it & 1) * 2 * IMG
        sG = sQ + IMG
        for w in range(4):  # phase 1
            s = st[w]
            L = s["L"]
            bd = BNC_OFF + w * BNC
            i = i0 + 16 * w + LI
            ic = np.minimum(i, T - 1)
            qvf = frag_rows(qv, i, T)
            lse = np.where(i < T, lse_all[ic], np.inf)
            Di = D_all[ic]
            quf = [lds.ldf(sQ + w * 2048 + L.offk[ks]) for ks in range(2)]
            dOf = [lds.ldf(sG + w * 2048 + L.offk[ks]) for ks in range(2)]
            acc_s = [np.zeros((64, 4)) for _ in range(4)]
            acc_dp = [np.zeros((64, 4)) for _ in range(4)]
            for ks in range(2):
                for jt in range(4):
                    acc_s[jt] = mfma(s["kf"][jt][ks], quf[ks], acc_s[jt])
                    acc_dp[jt] = mfma(s["vf"][jt][ks], dOf[ks], acc_dp[jt])
            add_band(lds, acc_s, qvf, PP_OFF + slot_lo * IMG, PP_OFF + slot_hi * IMG, bd, L)
            pc = None if pieces is None else pieces[kt, np.minimum(i, pieces.shape[1] - 1), G4]
            dsb, pdb = softmax_bwd_tile(acc_s, acc_dp, lse, Di, inv_keep, pc, kl - j0, True)
            for jt in range(4):
                lds.st64(bd + (s["exo"] ^ (jt << 5)), dsb[jt])
                lds.st64(bd + 2048 + (s["exo"] ^ (jt << 5)), pdb[jt])
        for w in range(4):  # phase 2 (after the workgroup barrier)
            s = st[w]
            for ks2 in range(2):
                ex = BNC_OFF + 2 * ks2 * BNC
                bs = np.concatenate([lds.trr(ex + s["exr"][0]), lds.trr(ex + s["exr"][1])], axis=1)
                bp = np.concatenate([lds.trr(ex + 2048 + s["exr"][0]), lds.trr(ex + 2048 + s["exr"][1])], axis=1)
                for dt in range(4):
                    aq = np.concatenate([lds.trr(sQ + ks2 * 4096 + (s["trk"][0] ^ (dt << 5))), lds.trr(sQ + ks2 * 4096 + (s["trk"][1] ^ (dt << 5)))], axis=1)
                    s["dk"][dt] = mfma(aq, bs, s["dk"][dt])
                    ag = np.concatenate([lds.trr(sG + ks2 * 4096 + (s["trk"][0] ^ (dt << 5))), lds.trr(sG + ks2 * 4096 + (s["trk"][1] ^ (dt << 5)))], axis=1)
                    s["dv"][dt] = mfma(ag, bp, s["dv"][dt])
        slot_lo = slot_nx
    dk = np.zeros((64, 64))
    dv = np.zeros((64, 64))
    for w in range(4):
        for dt in range(4):
            for r in range(4):
                dk[16 * w + LI, dt * 16 + G4 * 4 + r] = st[w]["dk"][dt][:, r]
                dv[16 * w + LI, dt * 16 + G4 * 4 + r] = st[w]["dv"][dt][:, r]
    return dk, dv


def reference(qu, qv, K, V, PP, dO, kl, T, keep, inv_keep, scaling):
    ii, jj = np.arange(T)[:, None], np.arange(T)[None, :]
    raw = qv @ PP.T  # [T][2T-1]
    S = qu @ K.T + raw[ii, (T - 1) - ii + jj]
    S = np.where(jj >= kl, -np.inf, S)
    m = S.max(axis=1, keepdims=True)
    P = np.exp(S - m)
    l = P.sum(axis=1, keepdims=True)
    P = P / l
    lse = (m + np.log(l))[:, 0]
    km = keep * inv_keep if keep is not None else np.ones((T, T))
    Pd = P * km
    O = Pd @ V
    dPd = dO @ V.T
    dP = dPd * km
    Dv = (dO * O).sum(axis=1)
    dS = P * (dP - Dv[:, None])
    t1 = scaling * dS @ K
    dBD = np.zeros((T, 2 * T - 1))
    dBD[ii, (T - 1) - ii + jj] = dS
    t2 = scaling * dBD @ PP
    dk = dS.T @ qu
    dv = Pd.T @ dO
    return O, lse, t1, t2, dBD, dk, dv, Dv


def run(T=150, kl=None, drop=True, seed=0):
    rng = np.random.default_rng(seed)
    kl = T if kl is None else kl
    qu, qv = rng.standard_normal((T, 64)) * 0.3, rng.standard_normal((T, 64)) * 0.3
    K, V = rng.standard_normal((T, 64)), rng.standard_normal((T, 64))
    PP = rng.standard_normal((2 * T - 1, 64))
    dO = rng.standard_normal((T, 64))
    keep = (rng.random((T, T)) >= 0.1).astype(np.float64) if drop else None
    inv_keep = 1 / 0.9 if drop else 1.0
    pieces = keep_pieces(keep, T) if drop else None
    scaling = 0.125
    O, lse, t1, t2, dBD, dk, dv, Dv = reference(qu, qv, K, V, PP, dO, kl, T, keep, inv_keep, scaling)
    nq = (T + 63) // 64
    ld_bd = (2 * T - 1 + 7) // 8 * 8
    err = {}

    def upd(name, got, ref):
        err[name] = max(err.get(name, 0.0), float(np.abs(got - ref).max()))

    Dk = np.zeros(T)
    for qt in range(nq):
        n = min(64, T - qt * 64)
        o, l = fwd_wg(qu, qv, K, V, PP, kl, T, qt, pieces, inv_keep)
        upd("out", o[:n], O[qt * 64:qt * 64 + n])
        upd("lse", l[:n], lse[qt * 64:qt * 64 + n])
        a1, a2, dbd, D = bwd_q_wg(qu, qv, K, V, PP, O, dO, lse, kl, T, qt, pieces, inv_keep, scaling, ld_bd)
        upd("t1", a1[:n], t1[qt * 64:qt * 64 + n])
        upd("t2", a2[:n], t2[qt * 64:qt * 64 + n])
        upd("D", D[:n], Dv[qt * 64:qt * 64 + n])
        Dk[qt * 64:qt * 64 + n] = D[:n]
        jcov = min(T, ((kl + 63) // 64) * 64)
        for r in range(n):
            row = qt * 64 + r
            lo, hi = T - 1 - row, T - 1 - row + jcov
            upd("dBD", dbd[r, lo:hi], dBD[row, lo:hi])
            assert np.isnan(dbd[r, :lo]).all() and np.isnan(dbd[r, hi:]).all(), "dBD written outside the band"
    for kt in range(nq):
        n = min(64, T - kt * 64)
        gk, gv = bwd_kv_wg(qu, qv, K, V, PP, dO, lse, Dk, kl, T, kt, pieces, inv_keep)
        upd("dk", gk[:n], dk[kt * 64:kt * 64 + n])
        upd("dv", gv[:n], dv[kt * 64:kt * 64 + n])
    return err


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    for kw in (dict(T=T), dict(T=T, kl=max(1, T - 37)), dict(T=T, drop=False, kl=max(1, T // 2))):
        print(kw, run(**kw))
