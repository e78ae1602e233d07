"""Lane-by-lane NumPy emulation of v_mfma_f64_16x16x4_f64 and of the register layouts csrc/leaf.hpp builds on it (no GPU needed).

MFMA f64 16×16×4 on a 64-lane wave, lane l = (li, lg) = (l & 15, l >> 4):
    A operand (16×4):  lane holds A[i = li][k = lg]
    B operand (4×16):  lane holds B[k = lg][j = li]
    accumulator (16×16, 4 registers per lane): register r of lane (li, lg) is element (row lg + 4r, column li)
Layouts of a 16×16 block T in leaf.hpp:
    natural    n[s] = T[li][4·lg + s]   (one 32-byte piece of row li per lane)
    symmetric  a[r] = A[lg + 4r][li]    (the accumulator layout itself)
π(i) = 4·(i mod 4) + i div 4.

Checked here against plain matrix arithmetic:
    P1   acc = Σ_s mfma(±M[π(li)][4lg + s], n_T[s], acc)  with acc natural(T')  gives  natural(T' ± T·Mᵀ)
    P2   acc = Σ_s mfma(−n_L[s], n_L[s], acc)            with acc symmetric(A)  gives  symmetric(A − L·Lᵀ)
    P3   the 16×16 factorisation on the accumulator with the inverse riding (see check_p3)
    UPD  panel_updk_kernel: acc[c] = Σ_{q, s} mfma(−Q[16c + π(li)][16q + 4lg + s], P[row li][16q + 4lg + s], acc[c])  with acc[c][r] = C[row li][16c + 4lg + r]
         gives C − P·Qᵀ on the wave's 16 rows.
Run:  python tools/leaf_emu.py   (prints the three maximum deviations; tests/test_leaf_emu.py asserts them)."""
import numpy as np

LANES = np.arange(64)
LI, LG = LANES & 15, LANES >> 4


def pi(i):
    return 4 * (i % 4) + i // 4


def mfma(a, b, acc):
    """a, b: per-lane operand values (64,), acc: per-lane accumulator registers (64, 4) -> new accumulator registers"""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[LI, LG] = a          # lane (li, lg) -> A[i = li][k = lg]
    B[LG, LI] = b          # lane (li, lg) -> B[k = lg][j = li]
    D = A @ B
    out = acc.copy()
    for r in range(4):
        out[:, r] += D[LG + 4 * r, LI]   # register r of lane (li, lg) = element (lg + 4r, li)
    return out


def natural(T):
    """(64, 4): n[s] of every lane"""
    return np.stack([T[LI, 4 * LG + s] for s in range(4)], axis=1)


def from_natural(n):
    T = np.zeros((16, 16))
    for s in range(4):
        T[LI, 4 * LG + s] = n[:, s]
    return T


def symmetric(A):
    return np.stack([A[LG + 4 * r, LI] for r in range(4)], axis=1)


def from_symmetric(a):
    A = np.zeros((16, 16))
    for r in range(4):
        A[LG + 4 * r, LI] = a[:, r]
    return A


def check_p1(rng, sign=-1.0):
    T, M, Tp = rng.standard_normal((3, 16, 16))
    acc = natural(Tp)
    nT = natural(T)
    for s in range(4):
        acc = mfma(sign * M[pi(LI), 4 * LG + s], nT[:, s], acc)
    return np.max(np.abs(from_natural(acc) - (Tp + sign * T @ M.T)))


def check_p2(rng):
    L = np.tril(rng.standard_normal((16, 16)))
    A = rng.standard_normal((16, 16))
    acc = symmetric(A)
    nL = natural(L)
    for s in range(4):
        acc = mfma(-nL[:, s], nL[:, s], acc)
    return np.max(np.abs(from_symmetric(acc) - (A - L @ L.T)))


def check_updk(rng, K=64, N=128):
    """one wave's 16 rows of C[16 × N] −= P[16 × K] · Q[N × K]ᵀ as the kernel issues it (chunks of 32 columns = 2 slices of 16)"""
    P = rng.standard_normal((16, K))
    Q = rng.standard_normal((N, K))
    C = rng.standard_normal((16, N))
    acc = [np.stack([C[LI, 16 * c + 4 * LG + r] for r in range(4)], axis=1) for c in range(N // 16)]
    for q in range(K // 16):
        for s in range(4):
            a = P[LI, 16 * q + 4 * LG + s]
            for c in range(N // 16):
                acc[c] = mfma(-Q[16 * c + pi(LI), 16 * q + 4 * LG + s], a, acc[c])
    out = np.zeros_like(C)
    for c in range(N // 16):
        for r in range(4):
            out[LI, 16 * c + 4 * LG + r] = acc[c][:, r]
    return np.max(np.abs(out - (C - P @ Q.T)))


def check_p3(rng):
    """P3 of leaf.hpp: the 16×16 Cholesky column by column on the accumulator (symmetric layout), every column's rank-1 update ONE MFMA whose
    K-slot operand is the scaled column masked to its lane group, the pivot of column c+1 formed early from the values before the update
    (A[c+1][c+1] − L[c+1][c]²), and the identity riding transposed in a second accumulator so that L⁻¹ falls out without extra work.
    Returns the deviations of L and of Inv = L⁻¹ from numpy."""
    G = rng.standard_normal((16, 16))
    A = G @ G.T + 16 * np.eye(16)
    accA = symmetric(A)
    accW = symmetric(np.eye(16))
    Ls = np.zeros((64, 4))
    Ws = np.zeros((64, 4))
    bcast = lambda v, lane: v[lane]
    sel = np.where(LG == 0, 1.0 / np.sqrt(bcast(accA[:, 0], 0)), 0.0)
    for c in range(16):
        k, p = c & 3, c >> 2
        pa = accA[:, p] * sel            # lane (i, k): L[i][c]; zero outside lane group k
        if c < 15:
            c1 = c + 1
            k1, p1 = c1 & 3, c1 >> 2
            l = bcast(pa, 16 * k + c1)                    # L[c+1][c]
            dnext = bcast(accA[:, p1], 16 * k1 + c1)      # A[c+1][c+1] before this column's update
            accA = mfma(-pa, pa, accA)
        pw = accW[:, p] * sel            # lane (i, k): (L⁻ᵀ)[i][c]
        Ls[:, p] = np.where(LG == k, pa, Ls[:, p])
        Ws[:, p] = np.where(LG == k, pw, Ws[:, p])
        if c < 15:
            sel = np.where(LG == k1, 1.0 / np.sqrt(dnext - l * l), 0.0)
            accW = mfma(-pa, pw, accW)
    L = np.zeros((16, 16))
    Inv = np.zeros((16, 16))
    for r in range(4):
        col = LG + 4 * r
        L[LI, col] = np.where(LI >= col, Ls[:, r], 0.0)   # Ls[r] of lane (li, lg) = L[li][lg + 4r]
        Inv[col, LI] = Ws[:, r]                           # Ws[r] = (L⁻ᵀ)[li][lg + 4r] = Inv[lg + 4r][li]
    Lref = np.linalg.cholesky(A)
    return np.max(np.abs(L - Lref)), np.max(np.abs(np.tril(Inv) - np.linalg.inv(Lref)))


def check_p3_rank4(rng):
    """NOT in the kernel: built and measured in round 5 (`leaf_rank4`, history at 2ba1531: same values, 49 000 instead of 46 800 cycles per 128-column
    leaf — the two row moves and the per-group multipliers of every column cost what the 24 saved MFMAs gave; profiles/NOTES_r5.md) and removed again.
    The same factorisation with ONE MFMA per FOUR columns.  The rows 4p..4p+3 of a sub-block are register p
    of the four lane groups, i.e. exactly a K-slot operand, so after the three intra-sub-block recurrences (row k minus its projections on rows
    j < k — each needs row j of ANOTHER lane group: a 16-lane row move, counted here, and one scalar per group) a single MFMA applies the rank-4
    update to the block; the inverse rides the same way.  Returns (|L − chol|, |Inv − L⁻¹|, row moves per block, MFMAs per block)."""
    G = rng.standard_normal((16, 16))
    A = G @ G.T + 16 * np.eye(16)
    accA = symmetric(A)
    accW = symmetric(np.eye(16))
    Ls = np.zeros((64, 4))
    Ws = np.zeros((64, 4))
    moves = mfmas = 0

    def row_to_all_groups(v, k):
        """per-lane vector whose lane group k holds a 16-lane row -> that row in every lane group (v_permlane16_swap + v_permlane32_swap on gfx950)"""
        nonlocal moves
        moves += 1
        return v[16 * k + LI]

    for p in range(4):
        ra = accA[:, p].copy()   # lane (li, k): A[4p + k][li]   (row 4p + k of the Schur complement so far)
        rw = accW[:, p].copy()   # lane (li, k): W[4p + k][li]   (the transposed identity carried along)
        for j in range(4):
            c = 4 * p + j
            d = ra[16 * j + c]                                   # pivot: lane (li = c, lg = j)
            ri = 1.0 / np.sqrt(d)
            ra = np.where(LG == j, ra * ri, ra)                  # row j scaled: lane (i, j) = L[i][c]
            rw = np.where(LG == j, rw * ri, rw)
            if j < 3:
                la = row_to_all_groups(ra, j)                    # L[:, c] in every group
                lw = row_to_all_groups(rw, j)
                mult = la[16 * LG + 4 * p + LG]                  # group k's own scalar L[4p + k][c]  (one v_readlane per group)
                later = LG > j
                ra = np.where(later, ra - mult * la, ra)
                rw = np.where(later, rw - mult * lw, rw)
        Ls[:, p] = ra
        Ws[:, p] = rw
        if p < 3:
            accA = mfma(-ra, ra, accA)   # rank-4: all four K slots carry a column
            accW = mfma(-ra, rw, accW)
            mfmas += 2
    L = np.zeros((16, 16))
    Inv = np.zeros((16, 16))
    for r in range(4):
        col = LG + 4 * r
        L[LI, col] = np.where(LI >= col, Ls[:, r], 0.0)
        Inv[col, LI] = Ws[:, r]
    Lref = np.linalg.cholesky(A)
    return np.max(np.abs(L - Lref)), np.max(np.abs(np.tril(Inv) - np.linalg.inv(Lref))), moves, mfmas


def main():
    rng = np.random.default_rng(7)
    res = {"P1 (T' - T M^T, natural in / natural out)": check_p1(rng, -1.0), "P1 (+)": check_p1(rng, 1.0),
           "P2 (A - L L^T, symmetric accumulator)": check_p2(rng), "UPD (panel_updk wave tile)": check_updk(rng)}
    dl, di = check_p3(rng)
    res["P3 (16x16 Cholesky by rank-1 MFMA updates): L"] = dl
    res["P3: Inv = L^-1 riding transposed"] = di
    d4l, d4i, moves, mfmas = check_p3_rank4(rng)
    res["P3 rank-4 variant (not in the kernel): L"] = d4l
    res["P3 rank-4 variant: Inv"] = d4i
    print(f"(rank-4 variant: {mfmas} MFMAs and {moves} 16-lane row moves per 16x16 block, against 30 MFMAs and none)")
    for k, v in res.items():
        print(f"{k}: max |deviation| = {v:.2e}")
    return res


if __name__ == "__main__":
    main()
