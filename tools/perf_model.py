"""Launch-level cost model of the single-GPU factorisation, fed with the per-launch costs measured in round 2
(profiles/r2/traces/*_summary.txt, tools/panel_stamps.hip): what the three classic panel orders would cost at the BASELINE sizes.

  leaf(H)            one fused 64-column leaf over H rows below it: 22.5 µs per round of 256 workgroups (1 per CU, 128 rows each)
  gemm(M, N, K, tri) 128×128 tiles; hardware dispatch: 70 TF/s × the fill of the last round of 512 resident workgroups, but never
                     faster than one tile's own pipeline (2.3 µs per 16-deep k step when a tile is alone on its CU); launches of
                     ≤ 4 096 tiles run the stream-K variant: the k steps of all tiles spread over 512 workgroups (1.7 µs per step)
                     + 6 µs of fix-up / launch.
The model has no overlap between kernels (the timelines show none on one GPU) and no solve / assembly time.

  python tools/perf_model.py
"""
import math

PEAK = 70e12          # the trailing-update kernel on large launches (measured 68–70 TF/s)
STEP_ALONE = 2.3e-6   # one 16-deep k step of a tile that is alone on its CU
STEP_SK = 1.7e-6      # one k step inside the persistent stream-K kernel (two workgroups per CU)
LEAF = 22.5e-6
LAUNCH = 6e-6


def leaf(H):
    return LEAF * max(1, math.ceil(max(H, 1) / 128 / 256))


def gemm(M, N, K, tri=False):
    if M <= 0 or N <= 0 or K <= 0:
        return 0.0
    tm, tn = math.ceil(M / 128), math.ceil(N / 128)
    tiles = tn * (tn + 1) // 2 + (tm - tn) * tn if tri and tm >= tn else tm * tn
    flops = 2.0 * tiles * 128 * 128 * K
    steps = K / 16
    if tiles <= 4096:  # stream-K
        return LAUNCH + max(flops / PEAK, tiles * steps * STEP_SK / 512, min(steps, 16) * STEP_SK)
    fill = tiles / (math.ceil(tiles / 512) * 512)
    return LAUNCH + max(flops / (PEAK * fill), steps * STEP_ALONE)


def panel(H, n, group=128):
    """recursive factorisation of n columns including the H rows below the diagonal block's first row"""
    if n <= group:
        return sum(leaf(H - 64 * (t + 1)) + (3e-6 * t) for t in range(n // 64))  # in-leaf update by the tiles to the left
    h = n // 2
    return panel(H, h, group) + gemm(H - h, n - h, h, tri=True) + panel(H - h, n - h, group)


def right_looking(N, nb):
    t = 0.0
    for k in range(0, N, nb):
        H = N - k
        t += panel(H, min(nb, H))
        if H > nb:
            t += gemm(H - nb, min(nb, H - nb), nb, tri=True)                 # U1: the next panel's columns
            t += gemm(H - 2 * nb, H - 2 * nb, nb, tri=True) if H > 2 * nb else 0.0   # U2: the rest
    return t


def left_looking(N, nb):
    t = 0.0
    for k in range(0, N, nb):
        H = N - k
        t += gemm(H, min(nb, H), k, tri=True) if k else 0.0   # all earlier panels at once, K = k
        t += panel(H, min(nb, H))
    return t


def recursive(N):
    return panel(N, N)


if __name__ == "__main__":
    measured = {16384: 33.8, 32768: 195.0, 65536: 1415.7}   # potrf_ms of the final tree (sweep_nb2.jsonl / bench)
    print(f"{'N':>6} {'measured':>9} | {'right nb=1024':>13} {'2048':>8} {'4096':>8} | {'left nb=512':>11} {'1024':>8} | {'recursive':>9}   (ms, factorisation only)")
    for N in (4096, 16384, 32768, 65536):
        row = [right_looking(N, 1024), right_looking(N, 2048), right_looking(N, 4096), left_looking(N, 512), left_looking(N, 1024), recursive(N)]
        m = measured.get(N)
        print(f"{N:>6} {m if m else float('nan'):>9} | " + " ".join(f"{x * 1e3:>8.1f}" if i not in (0, 3) else f"{x * 1e3:>13.1f}" if i == 0 else f"{x * 1e3:>11.1f}"
                                                                 for i, x in enumerate(row[:3])) + " | "
              + f"{row[3] * 1e3:>11.1f} {row[4] * 1e3:>8.1f} | {row[5] * 1e3:>9.1f}")
    n = 16384
    leaves = sum(leaf(n - 64 * (i + 1)) for i in range(n // 64))
    print(f"\nN = {n}: {n // 64} leaves = {leaves * 1e3:.1f} ms; flops at {PEAK / 1e12:.0f} TF/s = {n**3 / 3 / PEAK * 1e3:.1f} ms")
