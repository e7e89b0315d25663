"""CPU checks of kernel-side index arithmetic against software models of the gfx950 instructions (no GPU needed)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("kw", [dict(T=100), dict(T=130, kl=70), dict(T=70, drop=False, kl=64)])
def test_flash_relpos_lane_model(kw):
    """csrc/flash_relpos.hip transcribed onto a lane-level model of global_load_lds / ds_read_b64_tr_b16 / MFMA layouts
    (tools/emu_flash_relpos.py): forward, backward-Q and backward-KV reproduce the direct float64 formulas of
    fairseq/modules/multihead_attention.py:788-907 and their gradients — every LDS address, ring slot, skew / un-skew index,
    exchange-tile offset and keep-bit position is exercised."""
    import emu_flash_relpos as E

    err = E.run(**kw)
    assert max(err.values()) < 1e-11, err


def test_flash_relpos_swizzle_is_conflict_free():
    """the 16-byte-slot swizzle of the LDS images: no bank conflict for any fragment-read pattern the kernels use
    (bank rules of MI355X_MICROARCH.md, LDS section)"""
    import lds_layout_check as L

    f = L.make_f((0x4, 0x2, 0xC))
    assert all(w == 1 for _, w in L.patterns(f)), L.patterns(f)
    import emu_flash_relpos as E

    for r in range(64):
        assert E.swz(r) == f(r)


@pytest.mark.parametrize("BM", [64, 128])
def test_wgrad_transposing_read_kernel_lane_model(BM):
    """csrc/gemm.hip wgrad_group_tr_kernel on the lane model (tools/emu_wgrad_tr.py): global_load_lds slot placement with the
    source-side swizzle, ds_read_b64_tr_b16 fragment addresses / immediate offsets, MFMA operand order, the ones-fragment bias
    sums and the zero-page rows past M reproduce dy^T x (float64), and every fragment read is bank-conflict free."""
    import emu_wgrad_tr as W

    err, ways = W.run(M=150, BM=BM)
    assert err < 1e-11 and ways == 1, (err, ways)


@pytest.mark.parametrize("kw", [dict(M=150, R=300, Cn=264, pad_dy=4), dict(M=64, R=256, Cn=256, pad_dy=0), dict(M=65, R=40, Cn=512, pad_dy=0)])
def test_wgrad_8wave_kernel_lane_model(kw):
    """csrc/wgrad_w8.hip wgrad_w8_kernel on the lane model (tools/emu_wgrad_w8.py): 512-byte-pitch images with the source-side slot
    swizzle, the column clamp at the row pitch (the joint's 5004 columns at pitch 5056 in small), zero-page rows past M, fragment
    addresses and immediates, swapped MFMA operands, bias sums routed to wavefront wn — reproduce dy^T x and the column sums
    (float64), and every transposing fragment read is bank-conflict free."""
    import emu_wgrad_w8 as W

    err, ways = W.run(**kw)
    assert err < 1e-11 and ways == 1, (err, ways)
