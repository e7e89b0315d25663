#!/usr/bin/env python3
"""LDS bank-conflict model for the 128-byte-row bf16 tile images of the fused attention kernels (gfx950 rules from
/opt/skills/guides/MI355X_MICROARCH.md, LDS section): ds_read_b128 is served in four fixed 16-lane groups, ds_read_b64 and
ds_read_b64_tr_b16 in two 32-lane groups, bank = (byte/4) mod 64.  A 16-byte slot s of row r is stored at slot s ^ f(r);
the search below finds the linear f (3 output bits, each an XOR of row bits) for which every fragment-read pattern the
kernels use is conflict free.  Run with no arguments to re-derive csrc/flash_relpos.hip's `swz` (bit0 = r>>2, bit1 = r>>1, bit2 = (r>>2)^(r>>3)).
"""
import itertools
import sys

B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]
HALF_GROUPS = [list(range(32)), list(range(32, 64))]


def worst(groups, addr, width):
    """max over lane groups of the max number of distinct addresses per bank (1 = conflict free)."""
    w = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addr(l)
            for k in range(width // 4):
                banks.setdefault(((a // 4) + k) % 64, set()).add(a)
        w = max(w, max(len(v) for v in banks.values()))
    return w


def make_f(masks):
    def f(r):
        v = 0
        for bit, m in enumerate(masks):
            v |= (bin(r & m).count("1") & 1) << bit
        return v
    return f


def patterns(f):
    """yield (name, worst-way) for every read pattern of the kernels under slot swizzle f"""
    def slot_addr(row, slot, off=0):
        return row * 128 + ((slot ^ f(row)) << 4) + off
    out = []
    # regular MFMA operand fragments: lane (li, g4) reads row R0 + li, slot ks*4 + g4
    w = 0
    for R0 in (0, 16, 32, 48):
        for ks in (0, 1):
            w = max(w, worst(B128_GROUPS, lambda l: slot_addr(R0 + (l & 15), ks * 4 + (l >> 4)), 16))
    out.append(("b128 rows", w))
    # transposing reads, MFMA 16x16x32 fragments (k = 8 rows per 16-lane group): rows Rb + 8g + 4h + (i>>2)
    w = 0
    for Rb in (0, 32):
        for h in (0, 1):
            for dt in range(4):
                def a(l, Rb=Rb, h=h, dt=dt):
                    g, i = l >> 4, l & 15
                    p = dt * 4 + (i & 3)
                    return slot_addr(Rb + 8 * g + 4 * h + (i >> 2), p >> 1, (p & 1) * 8)
                w = max(w, worst(HALF_GROUPS, a, 8))
    out.append(("tr k32", w))
    # transposing reads, 16-row blocks (MFMA 16x16x16, and the two halves of the permuted-k 16x16x32): rows Rb + 4g + (i>>2)
    w = 0
    for Rb in (0, 16, 32, 48):
        for dt in range(4):
            def a(l, Rb=Rb, dt=dt):
                g, i = l >> 4, l & 15
                p = dt * 4 + (i & 3)
                return slot_addr(Rb + 4 * g + (i >> 2), p >> 1, (p & 1) * 8)
            w = max(w, worst(HALF_GROUPS, a, 8))
    out.append(("tr k16", w))
    return out


def search():
    best = []
    cands = [m << 1 for m in range(1, 8)]  # XORs of row bits 1..3 (the full 6-bit search takes ~30 min and finds the same family)
    for masks in itertools.product(cands, repeat=3):
        f = make_f(masks)
        res = patterns(f)
        if all(w == 1 for _, w in res):
            best.append(masks)
    return best


if __name__ == "__main__":
    if len(sys.argv) > 1:
        masks = tuple(int(x, 0) for x in sys.argv[1:4])
        print(masks, patterns(make_f(masks)))
    else:
        sols = search()
        print(len(sols), "conflict-free linear swizzles; simplest:")
        sols.sort(key=lambda m: sum(bin(x).count("1") for x in m))
        for m in sols[:8]:
            print("  bit0 = parity(r & %#x), bit1 = parity(r & %#x), bit2 = parity(r & %#x)" % m)
