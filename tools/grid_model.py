"""Cost model of the multi-device schedule (csrc/multi.hip) over process grids P×Q and distribution blocks NB — UNMEASURED ON
HARDWARE (no multi-GPU box was available to any round); it prices the schedule with the single-GPU per-launch costs of
tools/perf_model.py (measured in round 2) and a per-link xGMI rate, to pick the default grid by argument that can be checked
line by line instead of by assertion.

Per block step k of the right-looking factorisation (nblk = N / NB block columns, look-ahead: panel k+1 and its exchange run beside
the bulk update of step k):
  panel(k)     diagonal owner: recursive Cholesky of the NB×NB block (perf_model.panel) -> L_kk to its P−1 column peers over P−1
               links at once (8·NB² B each) -> every owner-column rank: X ← X L_kk⁻ᵀ on ITS rows (MFMA TRSM, rows/P·NB² flops)
  exchange(k)  rank (p, q) receives  A part: its process row's piece, rows_p·NB·8 B over ONE link (from (p, q_k)) unless q == q_k
                                     B part: the blocks of its process column, cols_q·NB·8 B from the P owner-column ranks, 1/P each
               grouped send/recv: all links of a rank run at once -> time = the busiest link's bytes / LINK + per-message latency
  bulk(k)      rank's local share of the trailing update: 2·NB·(its lower-triangle elements right of the window) flops at the
               MFMA GEMM rate with the tile fill of ITS local launch (perf_model.gemm)
  step time    max( max_ranks bulk(k) + look-ahead updates , look-ahead updates + panel(k+1) + exchange(k+1) )
               (the look-ahead GEMMs run on the same device as the bulk update: their work adds to it; the panel chain overlaps)
Block-cyclic imbalance (ranks own different numbers of lower blocks), per-rank receive volume and the busiest link are reported.

  python tools/grid_model.py                 # table for N = 65 536 on 2 / 4 / 8 devices, argmin per device count
"""
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import perf_model as pm  # noqa: E402

LINK = 64e9         # B/s per xGMI link and direction that a large transfer sustains (MI355X: 153.6 GB/s per link bidirectional peak)
MSG_LAT = 8e-6      # per point-to-point message (group launch + link latency)
TRSM_RATE = 45e12   # MFMA TRSM of a tall block against an NB×NB factor (recursion of GEMMs + 64-wide leaves)
# Round 5: the diagonal block's chain MEASURED through the library's own entry points on one MI355X (tools/cumask_chain_probe.py,
# profiles/r5/cumask_chain_probe.jsonl; ms per NB×NB block, the bulk update = one rank's share of a C4 step on an 8×1 grid running beside it):
#   standalone  nothing else on the device (64-column leaves of the rank contexts)
#   unmasked    on a high-priority stream beside the update, no CU mask — what the driver does today: 64-column leaves, 6–10× slower
#               (the 128-column leaf waits for the END of the update: 12.8 / 6.2 ms)
#   masked16    chain on a stream masked to 16 CUs (2 per XCD) with 128-column leaves, the update on the other 240: the update runs 4 % slower
# `python tools/grid_model.py chain=unmasked` / `chain=masked16` price the fit with these instead of the model's stand-alone chain × CORES.
CHAIN_MS = {"standalone": {512: 0.179, 1024: 0.358}, "unmasked": {512: 1.023, 1024: 3.590}, "masked16": {512: 0.174, 1024: 0.456}}
BULK_SLOW = {"standalone": 1.0, "unmasked": 1.0, "masked16": 1.04}
CHAIN = None        # None: pm.panel(NB, NB) × CORES (the rounds 2–4 model)
# Round 6: the panel step's pieces measured STANDALONE on one idle device through gpd_potrf / gpd_inv_lower / gpd_trsm_inv / gpd_trsm with the rank-context settings
# (tools/panel_step_probe.py, profiles/r6/panel_step_probe.jsonl; ms): Cholesky of the diagonal block, its inverse level by level, and the rows-below solve of m rows
# by one GEMM with the inverse ("inv", what the driver ships) / by the substitution recursion ("subst"), each as a + b·m through the two measured row counts.
# `python tools/grid_model.py step=inv|subst` prices panel(k) with these instead of CHAIN_MS / TRSM_RATE (an idle chain stream: the optimistic end; beside an unmasked
# bulk update every launch of the chain waits for workgroup slots, and the inverse form has 46 launches per step where the substitution form has 62).
STEP = None
STEP_MS = {512: {"potrf": 0.1688, "inv": 0.2182, "solve_inv": (8192, 0.0866, 32768, 0.3080), "solve_subst": (8192, 0.2347, 32768, 0.3296)},
           1024: {"potrf": 0.3537, "inv": 0.4166, "solve_inv": (8192, 0.2788, 32768, 0.9784), "solve_subst": (8192, 0.5401, 32768, 0.9497)},
           2048: {"potrf": 0.8532, "inv": 0.7906, "solve_inv": (8192, 0.8304, 32768, 2.7166), "solve_subst": (8192, 1.3504, 32768, 2.7897)}}


def _lin(tab, m):
    m0, t0, m1, t1 = tab
    b = (t1 - t0) / (m1 - m0)
    return max(0.0, t0 + b * (m - m0)) if m > 0 else 0.0
CORES = 1.0         # slow-down of the diagonal block's leaf chain when it runs BESIDE the bulk update (round 4, one GPU: a 64-column leaf that shares CUs
                    # with the tile GEMM runs ≈ 5× slower, profiles/r4/traces/c3_c64_summary.txt; `python tools/grid_model.py cores=5` prices that)


def local_lower_blocks(nblk, P, Q, p, q, r0, c0):
    """number of (full, diagonal) lower blocks (i, j), i >= r0, j >= c0, i >= j, owned by rank (p, q)"""
    full = diag = 0
    for j in range(c0, nblk):
        if j % Q != q:
            continue
        for i in range(max(j, r0), nblk):
            if i % P != p:
                continue
            if i == j:
                diag += 1
            else:
                full += 1
    return full, diag


def model(N, P, Q, NB, depth=2, forward=False):
    nblk = math.ceil(N / NB)
    lcm = P * Q // math.gcd(P, Q)
    nblk = math.ceil(nblk / lcm) * lcm
    R = P * Q
    ranks = [(p, q) for p in range(P) for q in range(Q)]
    t_total = 0.0
    recv = {r: 0.0 for r in ranks}
    link = {}
    flops = {r: 0.0 for r in ranks}
    bulk_t = []
    crit_t = []

    def panel_time(k):
        m = nblk - k - 1
        if STEP is not None and NB in STEP_MS:                 # measured standalone pieces of the shipped panel step
            tab = STEP_MS[NB]
            rows = math.ceil(m / P) * NB
            t = tab["potrf"] * 1e-3
            if P > 1:
                t += (tab["inv"] * 1e-3 if STEP == "inv" else 0.0) + 8.0 * NB * NB / LINK + MSG_LAT
                t += _lin(tab["solve_inv" if STEP == "inv" else "solve_subst"], rows) * 1e-3
            else:
                t += _lin(tab["solve_subst"], rows) * 1e-3     # P = 1: the column is factored in place (eng_potrf over all rows)
            return t
        if CHAIN is None:
            t = pm.panel(NB, NB) * CORES                       # diagonal block on its owner (beside the bulk update: CORES)
        else:                                                  # measured (NB = 2 048 was not: scaled from 1 024 by the model's own ratio)
            tab = CHAIN_MS[CHAIN]
            t = tab[NB] * 1e-3 if NB in tab else tab[1024] * 1e-3 * pm.panel(NB, NB) / pm.panel(1024, 1024)
        if P > 1:
            t += 8.0 * NB * NB / LINK + MSG_LAT                # L_kk to the column peers (P−1 links at once)
        rows = math.ceil(m / P) * NB
        t += rows * NB * NB / TRSM_RATE + (30e-6 if rows else 0.0)
        return t

    def exchange_time(k):
        m = nblk - k - 1
        qk = k % Q
        worst = 0.0
        for (p, q) in ranks:
            per_link = {}
            msgs = 0
            rows_p = sum(1 for i in range(k + 1, nblk) if i % P == p) * NB
            cols_q = [j for j in range(k + 1, nblk) if j % Q == q]
            a_fwd = 0.0
            if q != qk and rows_p:
                if forward and Q > 2:
                    # A-operand forwarding ("up to Q−1 links at once"; NOT implemented in csrc/multi.hip): the owner column
                    # sends a disjoint 1/(Q−1) slice of the row piece to each of the Q−1 peers of its process row, which then pass their
                    # slice on to the other Q−2 peers — two dependent phases, each moving rows_p·NB·8/(Q−1) bytes per link
                    sl = rows_p * NB * 8.0 / (Q - 1)
                    per_link[(p, qk)] = per_link.get((p, qk), 0.0) + sl
                    a_fwd = sl / LINK + MSG_LAT          # the second phase, behind the first
                    for q2 in range(Q):
                        if q2 != qk and q2 != q:
                            link[((p, q2), (p, q))] = link.get(((p, q2), (p, q)), 0.0) + sl
                            recv[(p, q)] += sl
                    msgs += Q - 1
                else:
                    per_link[(p, qk)] = per_link.get((p, qk), 0.0) + rows_p * NB * 8.0
                    msgs += 1
            for j in cols_q:
                src = (j % P, qk)
                if src == (p, q):
                    continue
                per_link[src] = per_link.get(src, 0.0) + NB * NB * 8.0
                msgs += 1
            for src, b in per_link.items():
                recv[(p, q)] += b
                link[(src, (p, q))] = link.get((src, (p, q)), 0.0) + b
            t = (max(per_link.values()) / LINK if per_link else 0.0) + (MSG_LAT if msgs else 0.0) + 1e-6 * msgs + a_fwd
            worst = max(worst, t)
        return worst

    def bulk_time(k):
        gfirst = k + depth + 1
        worst = 0.0
        for (p, q) in ranks:
            full, diag = local_lower_blocks(nblk, P, Q, p, q, gfirst, gfirst)
            if full + diag == 0:
                continue
            fl = 2.0 * NB * (full * NB * NB + diag * NB * (NB + 1) / 2)
            flops[(p, q)] += fl
            tiles = (full + diag) * (NB // 128) ** 2
            fill = tiles / (math.ceil(tiles / 512) * 512)
            t = pm.LAUNCH + max(fl / (pm.PEAK * fill), NB / 16 * pm.STEP_ALONE)
            if CHAIN is not None:
                t *= BULK_SLOW[CHAIN]
            worst = max(worst, t)
        return worst

    def la_time(k):  # look-ahead updates of the window columns by panel k (small GEMMs on the panel stream), the critical one first
        m = nblk - k - 1
        rows = math.ceil(m / P) * NB
        return pm.gemm(rows, NB, NB) * min(depth, max(m, 0))

    t_total += panel_time(0) + exchange_time(0)
    for k in range(nblk):
        b = bulk_time(k)
        c = (la_time(k) + panel_time(k + 1) + exchange_time(k + 1)) if k + 1 < nblk else 0.0
        bulk_t.append(b)
        crit_t.append(c)
        t_total += max(b + la_time(k), c)
    fl = list(flops.values())
    return {"N": N, "grid": f"{P}x{Q}", "NB": NB, "t_ms": t_total * 1e3, "bulk_ms": sum(bulk_t) * 1e3, "chain_ms": sum(crit_t) * 1e3,
            "chain_bound_steps": sum(1 for b, c in zip(bulk_t, crit_t) if c > b), "nblk": nblk,
            "flop_imbalance": max(fl) / (sum(fl) / len(fl)) if sum(fl) else 1.0,
            "recv_gb_max": max(recv.values()) / 1e9, "link_gb_max": (max(link.values()) / 1e9 if link else 0.0),
            "link_bytes": dict(link),   # ((p, q) source, (p, q) destination) -> panel bytes shipped over that link during the whole fit
                                        # (tests/test_multi_schedule.py compares it with the transfers of the traced schedule)
            "tflops": (N**3 / 3) / t_total / 1e12, "frac_of_peak": (N**3 / 3) / t_total / (78.6e12 * R)}


def grids(R):
    return [(P, R // P) for P in range(1, R + 1) if R % P == 0]


def best(N, R, nbs=(512, 1024, 2048), forward=False):
    rows = [model(N, P, Q, nb, forward=forward) for (P, Q) in grids(R) for nb in nbs]
    return min(rows, key=lambda r: r["t_ms"]), rows


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
    FWD = "forward" in sys.argv
    for a in sys.argv[1:]:
        if a.startswith("cores="):
            CORES = float(a.split("=")[1])
            print(f"diagonal-block chain priced {CORES:g}x slower (co-resident with the bulk update)")
        if a.startswith("step="):
            STEP = a.split("=")[1]
            print(f"panel step: MEASURED standalone pieces '{STEP}' (tools/panel_step_probe.py, profiles/r6/panel_step_probe.jsonl)")
        if a.startswith("chain="):
            CHAIN = a.split("=")[1]
            print(f"diagonal-block chain: MEASURED durations '{CHAIN}' {CHAIN_MS[CHAIN]} ms (tools/cumask_chain_probe.py), bulk update x{BULK_SLOW[CHAIN]}")
    if FWD:
        print("A-operand forwarding PRICED (two-phase slice exchange inside a process row; not implemented in the driver)")
    one = model(N, 1, 1, 2048)
    print(f"calibration: 1x1, NB = 2048 -> {one['t_ms']:.0f} ms (measured through the in-library driver with one rank: 1451 ms, profiles/r2/multi_virtual_bench.txt)")
    print(f"N = {N}; single-GPU costs from tools/perf_model.py, link {LINK / 1e9:.0f} GB/s per direction — UNMEASURED ON HARDWARE")
    print(f"{'devices':>7} {'grid':>5} {'NB':>5} {'t ms':>8} {'bulk':>8} {'chain':>8} {'chain-bound':>11} {'imbalance':>9} {'recv GB':>8} {'link GB':>8} {'% peak':>7}")
    for R in (1, 2, 4, 8):
        b, rows = best(N, R, forward=FWD)
        for r in sorted(rows, key=lambda r: r["t_ms"]):
            mark = " <- argmin" if r is b else ""
            print(f"{R:>7} {r['grid']:>5} {r['NB']:>5} {r['t_ms']:>8.1f} {r['bulk_ms']:>8.1f} {r['chain_ms']:>8.1f} {r['chain_bound_steps']:>5}/{r['nblk']:<5} "
                  f"{r['flop_imbalance']:>9.3f} {r['recv_gb_max']:>8.2f} {r['link_gb_max']:>8.2f} {100 * r['frac_of_peak']:>7.1f}{mark}")
