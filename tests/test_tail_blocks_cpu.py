"""CPU check of the index algebra behind the stage kernels' tail tiles: the last (up to four) pixels of a wave's image go
through v_mfma_f32_4x4x1 -- sixteen independent 4x1 by 1x4 blocks per instruction, lane l = block l >> 2, row / column l & 3, result
register r of lane l = A[4 (l >> 2) + r] * B[l] (measured on the GPU by scripts/probe_mfma4x4.hip) -- and take their weights from the
ORDINARY fragment buffers by a lane permutation.  The test emulates the instruction and the kernels' address formulas in NumPy and compares
with the dense convolution sum; the GPU parity tests (tests/test_gpu_nnet.py) check the kernels themselves."""
import numpy as np


def mfma_4x4x1(a, b, c):
    """c[l][r] += a[4 * (l // 4) + r] * b[l] for the 64 lanes (the layout scripts/probe_mfma4x4.hip prints)."""
    out = c.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += a[4 * (l // 4) + r] * b[l]
    return out


def test_resstage16_tail_blocks_equal_the_dense_sum():
    """k_resstage16<NT, true> / rs_conv: 16 -> 16 channels, block (cg, kk) = (l >> 4, (l >> 2) & 3), fragment of lane (4 cg + i) + 16 kk."""
    rng = np.random.default_rng(0)
    W = rng.standard_normal((16, 16, 9))          # [co][ci][tap]
    X = rng.standard_normal((4, 9, 16))           # [tail pixel][tap][ci]: the pixel's 3x3 window, channels-last
    frag = np.zeros((9, 64, 4))                   # k_pack_conv16: frag[tap][l][j] = W[co = l & 15][ci = 4 (l >> 4) + j][tap]
    for tap in range(9):
        for l in range(64):
            for j in range(4):
                frag[tap, l, j] = W[l & 15, 4 * (l >> 4) + j, tap]
    acc = np.zeros((64, 4))
    for tap in range(9):
        for q in range(4):                        # four instructions per tap
            a = np.zeros(64); b = np.zeros(64)
            for l in range(64):
                cg, kk, i = l >> 4, (l >> 2) & 3, l & 3
                a[l] = frag[tap, (4 * cg + i) + 16 * kk, q]   # RsTail::voff
                b[l] = X[i, tap, 4 * kk + q]                  # one ds_read_b128 at channel quad kk, component q
            acc = mfma_4x4x1(a, b, acc)
    dense = np.einsum("oct,ptc->po", W, X)        # [pixel][co]
    for l in range(64):
        cg, i = l >> 4, l & 3
        row = 16 * (l >> 4)                       # rs_tail_sum: the four kk blocks of a channel group are lanes l, l + 4, l + 8, l + 12 of a row
        total = sum(acc[row + 4 * k + i] for k in range(4))
        np.testing.assert_allclose(total, dense[i, 4 * cg:4 * cg + 4], rtol=1e-12, atol=1e-12)


def test_conv32_tail_blocks_equal_the_dense_sum():
    """r32_conv_t: CIN -> 32 channels, block (cg, kk) = (l >> 3, (l >> 2) & 1); per half-tap (tap, h) the pieces w = 0, 1 are the ordinary
    fragments of lane (4 (cg & 3) + i) + 16 (2 kk + w) in M tile cg >> 2, channels (CIN / 4)(2 kk + w) + 4 h + j."""
    rng = np.random.default_rng(1)
    for CIN in (16, 32):
        HQ = CIN // 16
        W = rng.standard_normal((32, CIN, 9))
        X = rng.standard_normal((4, 9, CIN))
        frag = np.zeros((9, HQ, 2, 64, 4))        # k_pack_conv32: frag[tap][h][mt][l][j] = W[16 mt + (l & 15)][(CIN / 4)(l >> 4) + 4 h + j][tap]
        for tap in range(9):
            for h in range(HQ):
                for mt in range(2):
                    for l in range(64):
                        for j in range(4):
                            frag[tap, h, mt, l, j] = W[16 * mt + (l & 15), (CIN // 4) * (l >> 4) + 4 * h + j, tap]
        acc = np.zeros((64, 4))
        for tap in range(9):
            for h in range(HQ):
                for w in range(2):
                    for j in range(4):
                        a = np.zeros(64); b = np.zeros(64)
                        for l in range(64):
                            cg, kk, i = l >> 3, (l >> 2) & 1, l & 3
                            a[l] = frag[tap, h, cg >> 2, 4 * (cg & 3) + i + 16 * (2 * kk + w), j]
                            b[l] = X[i, tap, (CIN // 2) * kk + (CIN // 4) * w + 4 * h + j]
                        acc = mfma_4x4x1(a, b, acc)
        dense = np.einsum("oct,ptc->po", W, X)
        for l in range(64):
            cg, i = l >> 3, l & 3
            total = acc[l] + acc[l ^ 4]           # the two input-channel slices of a channel group: lanes l and l ^ 4
            np.testing.assert_allclose(total, dense[i, 4 * cg:4 * cg + 4], rtol=1e-12, atol=1e-12)
