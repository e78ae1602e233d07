"""CPU: happens-before check of the multi-device schedule ON THE SCHEDULE THE LIBRARY EMITS.

gp_multi_schedule_trace runs the rank threads of abstractgps.jl_amd/csrc/multi.hip::fit_rank without a device — the same control
flow that drives the GPUs — and writes every stream operation (with its block footprint), event record / wait and send / receive
as JSON lines; tools/multi_schedule_check.py rebuilds the dependency graph from that and looks for unordered conflicting
accesses and for protocol errors (send / receive sequences of a rank pair that do not match, waits on unrecorded events).
The rank threads really run (generation-numbered events, host spins): a schedule whose threads could block each other fails the
trace call itself after its time-out.  The checker is shown to FIND the defects it is there for by editing recorded traces."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
import multi_schedule_check as M  # noqa: E402

COPIES, SENDRECV = 2, 1
SUBST = 16   # comm + 16: the schedule with the substitution solve of the rows below a diagonal block ("multi_trsm_inv" = 0: L_kk itself travels);
             # without it (the default) the diagonal owner's −inv(L_kk) travels and every owner's solve is one triangular-k GEMM
CHAIN = 32   # comm + 32: the diagonal block's chain (Cholesky + inverse) on a stream of its own ("multi_chain_cus" > 0)
ALL_COMMS = [COPIES, SENDRECV, COPIES + SUBST, SENDRECV + SUBST, COPIES + CHAIN, SENDRECV + CHAIN, COPIES + SUBST + CHAIN]
ALL_IDS = ["copies", "sendrecv", "copies_subst", "sendrecv_subst", "copies_chain", "sendrecv_chain", "copies_subst_chain"]


@pytest.fixture(scope="module", autouse=True)
def _built(agp):  # the .so (cross-compiled here when missing); no GPU is needed to load it or to trace schedules
    return agp


@pytest.mark.parametrize("comm", ALL_COMMS, ids=ALL_IDS)
@pytest.mark.parametrize("grid", M.GRIDS)
def test_no_unordered_conflicts(grid, comm):
    P, Q = grid
    for nblk in (1, 2, 3, 5, 9, 17):
        for depth in (1, 2, 3):
            problems, rs = M.check_config(P, Q, nblk, depth, comm)
            assert not problems, (grid, nblk, depth, problems[:4])
            assert not rs, (grid, nblk, depth, rs[:4])


def test_trace_covers_the_backward_sweep_transfers():
    """the send/recv transport of the backward sweep (partial sums to the diagonal owner, alpha blocks along the process row) is in
    the trace: round 2's branch sent alpha blocks that no rank received — the pair sequences must now match exactly"""
    with __import__("tempfile").TemporaryDirectory() as td:
        path = Path(td) / "t.jsonl"
        M.emit_trace(2, 4, 9, 2, SENDRECV, path)
        hdr, lines = M.load(path)
    sm_sends = [ln for ln in lines if ln["t"] == "send" and ln["s"] == "sm"]
    sm_recvs = [ln for ln in lines if ln["t"] == "recv" and ln["s"] == "sm"]
    assert sm_sends and len(sm_sends) == len(sm_recvs)
    g, problems = M.build(hdr, lines)
    assert not problems and not M.races(g)
    # an unmatched alpha send (what round 2's code issued for ranks without columns left of k) is a protocol finding
    extra = dict(sm_sends[-1])
    g, problems = M.build(hdr, lines + [extra])
    assert any("sends but" in p for p in problems)


@pytest.mark.parametrize("grid", M.GRIDS)
def test_forward_solve_on_the_distributed_factor_is_ordered(grid):
    """the schedule of the predictive-variance solve (csrc/multi.hip: solve_rank — X_k published by the diagonal owner, fetched by its
    process column, partial sums gathered along its process row), traced from the real rank threads without a device"""
    P, Q = grid
    for nblk in (1, 2, 3, 5, 9, 17):
        problems, rs = M.check_config(P, Q, nblk, 0, COPIES, solve=True)
        assert not problems and not rs, (grid, nblk, problems[:3], rs[:3])


def test_checker_finds_unwaited_solution_blocks_and_partial_sums():
    for tag in ("sx", "sa"):   # X_k fetched before it is final / a partial sum fetched before the peer's last update
        edit = lambda lines, tag=tag: [ln for ln in lines if not (ln["t"] == "wait" and ln.get("tag") == tag)]  # noqa: E731
        assert M.check_config(2, 3, 6, 0, COPIES, mutate=edit, solve=True)[1]


@pytest.mark.parametrize("grid", M.GRIDS)
def test_solve_update_and_backward_sweeps_on_the_pieces_are_ordered(grid):
    """the other passes over the distributed factor, traced from the real rank threads without a device: `C \\ B` (given right-hand
    sides, forward pass + two backward block sweeps separated by the rank barrier), the sequential update's forward pass (every rank
    keeps the rows of the new blocks it owns in its piece of the extended factor), and both together"""
    P, Q = grid
    for flags in (1 | 4, 2, 1 | 2 | 4):
        for nblk in (1, 2, 5, 9):
            problems, rs = M.check_config(P, Q, nblk, 0, COPIES, solve=flags)
            assert not problems and not rs, (grid, flags, nblk, problems[:3], rs[:3])


def test_checker_needs_the_barrier_between_backward_sweeps_and_the_sweep_events():
    """without the waits on the peers' barrier events, sweep 2 resets partial sums a peer may still be fetching for sweep 1; without
    the alr / accr waits a solution block or a partial sum is fetched before it is final"""
    for tag in ("bar", "alr", "accr"):
        edit = lambda lines, tag=tag: [ln for ln in lines if not (ln["t"] == "wait" and ln.get("tag") == tag)]  # noqa: E731
        assert M.check_config(2, 2, 4, 0, COPIES, mutate=edit, solve=1 | 4)[1], tag
    with __import__("tempfile").TemporaryDirectory() as td:   # the update pass writes the new block rows of every process column
        path = Path(td) / "t.jsonl"
        M.emit_trace(2, 3, 6, 0, COPIES, path, solve=2)
        hdr, lines = M.load(path)
    sinks = [ln for ln in lines if ln["t"] == "op" and ln["n"] == "sink"]
    assert len(sinks) == 6 * 6 and all(ln["W"][0][0] == "A2" for ln in sinks)   # 6 new block rows (lcm) × 6 block columns, one owner each


@pytest.mark.parametrize("grid", M.GRIDS)
def test_update_pass_fills_every_new_block_exactly_once(grid):
    """completeness of the sequential update's forward pass: block (nblk + t, k) of the extended factor — for every new block row t and
    every old block column k — is written by exactly one rank, the block-cyclic owner ((nblk + t) mod P, k mod Q), into the right local
    block of its new piece"""
    import math

    P, Q = grid
    lcm = P * Q // math.gcd(P, Q)
    for nblk_in in (1, 3, 7):
        with __import__("tempfile").TemporaryDirectory() as td:
            path = Path(td) / "t.jsonl"
            M.emit_trace(P, Q, nblk_in, 0, COPIES, path, solve=2)
            hdr, lines = M.load(path)
        nblk = hdr["nblk"]
        seen = {}
        for ln in lines:
            if ln["t"] == "op" and ln["n"] == "sink":
                (name, r, a0, a1, b0, b1, fl), = ln["W"]
                assert name == "A2" and a1 == a0 + 1 and b1 == b0 + 1
                p, q = r // Q, r % Q
                key = (a0 * P + p, b0 * Q + q)          # global (block row, block column) of the local block written
                assert key not in seen, (grid, key)
                seen[key] = r
                k, t = ln["k"]
                assert key == (nblk + t, k) and p == (nblk + t) % P and q == k % Q, (grid, ln)
        assert set(seen) == {(nblk + t, k) for t in range(lcm) for k in range(nblk)}, grid


@pytest.mark.parametrize("grid", M.GRIDS)
def test_solve_passes_visit_every_block_column_once_per_sweep(grid):
    """completeness of `C \\ B` on the pieces: the forward pass forms every X_k once on the diagonal owner (k mod P, k mod Q) and
    every rank of process column k mod Q applies it to its block rows below k; each of the two backward sweeps solves every
    diagonal block once, on its owner, in descending order"""
    P, Q = grid
    for nblk_in in (2, 5):
        with __import__("tempfile").TemporaryDirectory() as td:
            path = Path(td) / "t.jsonl"
            M.emit_trace(P, Q, nblk_in, 0, COPIES, path, solve=1 | 4)
            hdr, lines = M.load(path)
        nblk = hdr["nblk"]
        ops = [ln for ln in lines if ln["t"] == "op"]
        fwd = [(ln["k"][0], ln["r"]) for ln in ops if ln["n"] == "trsm"]
        assert sorted(fwd) == [(k, (k % P) * Q + k % Q) for k in range(nblk)], grid
        gemms = {}
        for ln in ops:
            if ln["n"] == "gemm":
                gemms.setdefault(ln["k"][0], set()).add(ln["r"])
        for k in range(nblk):   # the ranks of process column k mod Q that own a block row below k
            want = {p * Q + k % Q for p in range(P) if any(i % P == p for i in range(k + 1, nblk))}
            assert gemms.get(k, set()) == want, (grid, k)
        for sweep in (0, 1):
            bw = [(ln["k"][0], ln["r"]) for ln in ops if ln["n"] == "trsv" and ln["k"][1] == sweep]
            assert sorted(bw) == [(k, (k % P) * Q + k % Q) for k in range(nblk)], (grid, sweep)
            per_rank = {}
            for k, r in bw:                      # (file order = issue order of a rank's thread)
                per_rank.setdefault(r, []).append(k)
            assert all(v == sorted(v, reverse=True) for v in per_rank.values()), (grid, sweep)


@pytest.mark.parametrize("comm", ALL_COMMS, ids=ALL_IDS)
@pytest.mark.parametrize("grid", M.GRIDS)
def test_fit_updates_every_block_by_every_earlier_panel_exactly_once(grid, comm):
    """completeness of the factorisation schedule (right-looking): block (i, j) of the lower triangle — and the right-hand-side
    block row — is updated by every panel k < j exactly once, whichever stream (look-ahead or bulk) and look-ahead depth carries the
    update, and is then finalised exactly once (Cholesky of the diagonal block / triangular solve of the rows below it)"""
    P, Q = grid
    for nblk_in, depth in ((1, 2), (4, 1), (5, 2), (9, 3)):
        with __import__("tempfile").TemporaryDirectory() as td:
            path = Path(td) / "t.jsonl"
            M.emit_trace(P, Q, nblk_in, depth, comm, path)
            hdr, lines = M.load(path)
        nblk = hdr["nblk"]
        upd, fin = {}, {}
        for ln in lines:
            if ln["t"] != "op" or ln["n"] not in ("la", "bulk", "potrf", "potrf_diag", "trsm"):
                continue
            for f in ln["W"]:
                for key in M.expand(f, hdr):
                    if key[0] != "A":
                        continue
                    _, r, li, lj = key
                    gi = "rhs" if li == "rhs" else li * P + r // Q
                    gj = lj * Q + r % Q
                    if ln["n"] in ("la", "bulk"):
                        upd.setdefault((gi, gj), []).append(ln["k"][1])       # k = [first target column, panel]
                    else:
                        fin.setdefault((gi, gj), []).append(ln["k"][0])
        for j in range(nblk):
            for i in list(range(j, nblk)) + ["rhs"]:
                assert sorted(upd.get((i, j), [])) == list(range(j)), (grid, depth, (i, j), upd.get((i, j)))
                assert fin.get((i, j)) == [j], (grid, depth, (i, j), fin.get((i, j)))
        # backward sweep: every diagonal block solved once, on its owner; the block row k of Lᵀ applied by exactly the ranks of
        # process row k mod P that hold block columns left of k
        ops = [ln for ln in lines if ln["t"] == "op"]
        assert sorted((ln["k"][0], ln["r"]) for ln in ops if ln["n"] == "trsv") == [(k, (k % P) * Q + k % Q) for k in range(nblk)]
        gemv = {}
        for ln in ops:
            if ln["n"] == "gemv":
                gemv.setdefault(ln["k"][0], set()).add(ln["r"])
        for k in range(nblk):
            want = {(k % P) * Q + q for q in range(Q) if any(j % Q == q for j in range(k))}
            assert gemv.get(k, set()) == want, (grid, depth, k)


def _operand_dataflow(hdr, lines):
    """-> (reads checked, findings): provenance of the operand buffers.  Every write into an operand-type buffer (A / B operand
    slots, staging images, the L_kk image) gets the PANEL its data comes from: a copy out of the factor matrix carries the block
    column it read; a copy between buffers, a send and its matched receive carry the panel of what they read (the LATEST write to
    that block in happens-before order — race-freedom makes the writes to one block totally ordered).  Then every update of panel k
    and every rows-below solve of panel k must read blocks that carry panel k."""
    import re

    Q = hdr["Q"]
    g, problems = M.build(hdr, lines)
    assert not problems and not M.races(g)
    n = len(g.label)
    succ, indeg = [[] for _ in range(n)], [0] * n
    for i, ps in enumerate(g.preds):
        for q in ps:
            succ[q].append(i)
            indeg[i] += 1
    order, stack = [], [i for i in range(n) if indeg[i] == 0]
    while stack:
        i = stack.pop()
        order.append(i)
        for j in succ[i]:
            indeg[j] -= 1
            if indeg[j] == 0:
                stack.append(j)
    reach = [0] * n
    for i in order:
        b = 0
        for q in g.preds[i]:
            b |= reach[q] | (1 << q)
        reach[i] = b
    writers = {}
    for i in range(n):
        for loc in g.wr[i]:
            writers.setdefault(loc, []).append(i)
    OPERAND = ("Ab", "Bb", "St", "Lkk", "Wi")

    def latest_writer(i, loc):
        cands = [w for w in writers.get(loc, []) if (reach[i] >> w) & 1]
        if not cands:
            return None
        last = max(cands, key=lambda w: bin(reach[w]).count("1"))
        assert all(w == last or (reach[last] >> w) & 1 for w in cands)
        return last

    tag = [None] * n          # panel carried by what node i writes into operand-type buffers
    for i in order:
        if not any(loc[0] in OPERAND for loc in g.wr[i]) or ":init(" in g.label[i]:
            continue
        if ":recv<-" in g.label[i]:
            sends = [q for q in g.preds[i] if ":send->" in g.label[q]]
            assert len(sends) == 1
            tag[i] = tag[sends[0]]
            continue
        tags = set()
        for loc in g.rd[i]:
            if loc[0] == "A":
                tags.add(loc[3] * Q + loc[1] % Q)          # global block column of a block of the factor matrix
            elif loc[0] in OPERAND:
                w = latest_writer(i, loc)
                tags.add(None if w is None else tag[w])
        tag[i] = tags.pop() if len(tags) == 1 else ("mixed", sorted(map(str, tags)))
    for i in order:            # sends carry the panel of what they read (they write nothing themselves)
        if ":send->" in g.label[i]:
            tags = set()
            for loc in g.rd[i]:
                if loc[0] == "A":
                    tags.add(loc[3] * Q + loc[1] % Q)
                elif loc[0] in OPERAND:
                    w = latest_writer(i, loc)
                    tags.add(None if w is None else tag[w])
            if tags:
                tag[i] = tags.pop() if len(tags) == 1 else ("mixed", sorted(map(str, tags)))
    # (receives were tagged before their sends in the loop above when the send reads operand buffers: one more pass settles them)
    for i in order:
        if ":recv<-" in g.label[i] and any(loc[0] in OPERAND for loc in g.wr[i]):
            sends = [q for q in g.preds[i] if ":send->" in g.label[q]]
            tag[i] = tag[sends[0]]
    pat = re.compile(r"r(\d+)\.(\w+):(\w+)\((-?\d+),(-?\d+)\)")
    checked, bad = 0, []
    for i in range(n):
        m = pat.match(g.label[i])
        if not m or m.group(3) not in ("la", "bulk", "trsm"):
            continue
        panel = int(m.group(5)) if m.group(3) in ("la", "bulk") else int(m.group(4))
        for loc in g.rd[i]:
            if loc[0] not in ("Ab", "Bb", "Lkk", "Wi"):
                continue
            w = latest_writer(i, loc)
            checked += 1
            if w is None or tag[w] != panel:
                bad.append((g.label[i], loc, "never written before" if w is None else f"{g.label[w]} carrying panel {tag[w]}"))
    return checked, bad


@pytest.mark.parametrize("comm", ALL_COMMS, ids=ALL_IDS)
@pytest.mark.parametrize("grid", M.GRIDS)
def test_updates_consume_the_blocks_of_their_own_panel(grid, comm):
    """data flow of the traced schedule, both transports: what an update of panel k reads from the operand slots — and what the
    rows-below solve of panel k reads from the L_kk image — came out of PANEL k of the factor matrix (through the peer copy, or
    through the owner's staging image, the send and its matched receive), never a neighbour's (the slots are reused every depth + 1
    steps); ordering alone (the race check) would not notice a stale-but-ordered slot"""
    P, Q = grid
    for nblk_in, depth in ((4, 1), (7, 2), (9, 3)):
        with __import__("tempfile").TemporaryDirectory() as td:
            path = Path(td) / "t.jsonl"
            M.emit_trace(P, Q, nblk_in, depth, comm, path)
            hdr, lines = M.load(path)
        checked, bad = _operand_dataflow(hdr, lines)
        assert not bad, (grid, depth, bad[:3])
        assert checked > 0 or P * Q == 1


@pytest.mark.parametrize("comm", [COPIES, SENDRECV], ids=["copies", "sendrecv"])
@pytest.mark.parametrize("grid", [g for g in M.GRIDS if g[0] > 1])
def test_inverse_block_solve_schedule(grid, comm):
    """"multi_trsm_inv" (default): per block column k exactly ONE inv_lkk, on the diagonal owner, written to a slot of its own that no other
    operation of the fit writes (peers may read it at any later time); the owner's rows-below solve reads that slot, every other owner of the
    column reads its L_kk image — filled from the owner's slot by the peer copy or by the matched send / receive; nothing reads L_kk out of the
    factor matrix any more except the inverse itself.  With comm + 16 no inv_lkk appears and the images are filled from the matrix."""
    P, Q = grid
    with __import__("tempfile").TemporaryDirectory() as td:
        path = Path(td) / "t.jsonl"
        M.emit_trace(P, Q, 9, 2, comm, path)
        hdr, lines = M.load(path)
        M.emit_trace(P, Q, 9, 2, comm + SUBST, path)
        hdr0, lines0 = M.load(path)
    assert hdr["inv"] == 1 and hdr0["inv"] == 0
    assert not [ln for ln in lines0 if ln["t"] == "op" and ln["n"] == "inv_lkk"]
    nblk = hdr["nblk"]
    inv = [ln for ln in lines if ln["t"] == "op" and ln["n"] == "inv_lkk"]
    assert sorted((ln["k"][0], ln["r"]) for ln in inv) == [(k, (k % P) * Q + k % Q) for k in range(nblk)]
    slots = {}
    for ln in lines:
        if ln["t"] != "op" or ln["n"] == "init":
            continue
        for f in ln["W"]:
            for key in M.expand(f, hdr):
                if key[0] == "Wi":
                    slots.setdefault(key, []).append(ln["n"])
    assert len(slots) == nblk and all(v == ["inv_lkk"] for v in slots.values()), slots
    for ln in lines:
        if ln["t"] == "op" and ln["n"] == "trsm":
            names = {f[0] for f in ln["R"]}
            owner = ln["r"] == (ln["k"][0] % P) * Q + ln["k"][0] % Q
            assert ("Wi" in names) == owner and ("Lkk" in names) == (not owner), ln
        if ln["t"] == "op" and ln["n"] == "pull_lkk":
            assert {f[0] for f in ln["R"]} == {"Wi"}, ln


def test_chain_stream_schedule():
    """"multi_chain_cus" > 0 (comm + 32): the Cholesky of every diagonal block and its inverse are the ONLY operations on the chain stream "sd", each preceded
    by a wait for the panel stream (the block's last look-ahead update) and followed by a record the panel stream waits for; without the flag no
    operation names that stream."""
    with __import__("tempfile").TemporaryDirectory() as td:
        path = Path(td) / "t.jsonl"
        M.emit_trace(4, 2, 9, 2, COPIES + CHAIN, path)
        hdr, lines = M.load(path)
        M.emit_trace(4, 2, 9, 2, COPIES, path)
        hdr0, lines0 = M.load(path)
    assert hdr["chain"] == 1 and hdr0["chain"] == 0
    assert not [ln for ln in lines0 if ln.get("s") == "sd"]
    on_sd = [ln for ln in lines if ln["t"] == "op" and ln["s"] == "sd"]
    assert {ln["n"] for ln in on_sd} == {"potrf_diag", "inv_lkk"}
    assert not [ln for ln in lines if ln["t"] == "op" and ln["n"] in ("potrf_diag", "inv_lkk") and ln["s"] != "sd"]
    nblk = hdr["nblk"]
    assert len(on_sd) == 2 * nblk
    waits = [ln for ln in lines if ln["t"] == "wait" and ln["s"] == "sd"]
    recs = [ln for ln in lines if ln["t"] == "rec" and ln["s"] == "sd"]
    assert len(waits) == nblk and {ln["tag"] for ln in waits} == {"to_chain"}
    assert len(recs) == nblk and {ln["tag"] for ln in recs} == {"from_chain"}
    back = [ln for ln in lines if ln["t"] == "wait" and ln["tag"] == "from_chain"]
    assert len(back) == nblk and {ln["s"] for ln in back} == {"sp"}


def test_dataflow_check_notices_a_stale_slot():
    """a fetch that never happens (rank 3's B-operand pulls of panel 5 removed from the trace): every later access is still ordered —
    the events are all there — so the race check is silent, but the update of panel 5 now reads what panel 2 left in the slot"""
    with __import__("tempfile").TemporaryDirectory() as td:
        path = Path(td) / "t.jsonl"
        M.emit_trace(2, 2, 9, 2, COPIES, path)
        hdr, lines = M.load(path)
    kept = [ln for ln in lines if not (ln["t"] == "op" and ln["n"] == "pullB" and ln["r"] == 3 and ln["k"][0] == 5)]
    assert len(kept) < len(lines)
    g, problems = M.build(hdr, kept)
    assert not problems and not M.races(g)
    checked, bad = _operand_dataflow(hdr, kept)
    assert bad and all("(5," in b[0] or ",5)" in b[0] for b in bad) and any("pullB(2," in b[2] and b[2].endswith("panel 2") for b in bad), bad[:4]


def test_dataflow_check_follows_staging_sends_and_receives():
    """send/recv transport: the owners' staging copy of panel 8 removed — their sends then ship what panel 2 left in the staging image
    (same slot: 8 ≡ 2 mod depth + 1; same owners: 8 ≡ 2 mod Q); every transfer is still matched and ordered, the receivers' updates
    of panel 8 are flagged with the panel the data really came from"""
    with __import__("tempfile").TemporaryDirectory() as td:
        path = Path(td) / "t.jsonl"
        M.emit_trace(2, 2, 12, 2, SENDRECV, path)
        hdr, lines = M.load(path)
    owners = {ln["r"] for ln in lines if ln["t"] == "op" and ln["n"] == "stage" and ln["k"][0] == 8}
    assert owners
    kept = [ln for ln in lines if not (ln["t"] == "op" and ln["n"] == "stage" and ln["k"][0] == 8)]
    g, problems = M.build(hdr, kept)
    assert not problems and not M.races(g)
    checked, bad = _operand_dataflow(hdr, kept)
    assert bad and all(",8)" in b[0] for b in bad), bad[:4]
    assert any("recv<-" in b[2] and b[2].endswith("panel 2") for b in bad), bad[:4]
    assert {int(b[0].split(".")[0][1:]) for b in bad} - owners, "ranks other than the owners see the stale data"


@pytest.mark.parametrize("grid", M.GRIDS)
def test_cost_model_ships_what_the_schedule_ships(grid):
    """tools/grid_model.py prices the exchange of the multi-device fit from its own formula for who receives which
    panel blocks; here that formula meets the schedule the library really issues (send/recv transport, traced without a device): per
    ordered rank pair, the panel blocks the model ships over the link == the panel blocks the traced sends carry out of the owners'
    staging images (the right-hand-side block row that rides along and the 32-column row padding are not in the model)"""
    sys.path.insert(0, str(ROOT / "tools"))
    import grid_model as G

    P, Q = grid
    for nblk_in in (3, 6, 9):
        with __import__("tempfile").TemporaryDirectory() as td:
            path = Path(td) / "t.jsonl"
            M.emit_trace(P, Q, nblk_in, 2, SENDRECV, path)
            hdr, lines = M.load(path)
        nblk = hdr["nblk"]
        traced = {}
        for ln in lines:
            if ln["t"] == "send" and ln["R"][0][0] == "St":
                blocks = sum(f[5] - f[4] for f in ln["R"])                      # local block rows [b0, b1) of the staging image
                assert ln["n"] == (blocks + sum(1 for f in ln["R"] if f[6] & 2)) * 128 * 160   # (+ the RHS block row; rows of NB + 32)
                traced[(ln["r"], ln["to"])] = traced.get((ln["r"], ln["to"]), 0) + blocks
        m = G.model(nblk * 128, P, Q, 128)
        assert m["nblk"] == nblk
        want = {}
        for (src, dst), b in m["link_bytes"].items():
            want[(src[0] * Q + src[1], dst[0] * Q + dst[1])] = round(b / (128 * 128 * 8.0))
        want = {k: v for k, v in want.items() if v}
        traced = {k: v for k, v in traced.items() if v}
        assert traced == want, (grid, nblk, sorted(set(traced.items()) ^ set(want.items()))[:6])


def _drop(pred, first_per=None):
    """trace edit: remove the lines pred selects (first_per: only the first one per key)"""
    def f(lines):
        out, seen, dropped = [], set(), 0
        for ln in lines:
            if pred(ln):
                key = first_per(ln) if first_per else None
                if first_per is None or key not in seen:
                    seen.add(key)
                    dropped += 1
                    continue
            out.append(ln)
        assert dropped, "the edit did not match any trace line"
        return out
    return f


def test_checker_finds_the_round2_bug():
    """the dependency that was missing during development (a rank in the owner column whose updates read its own matrix did
    not wait for its own panel when it pulled nothing from itself): grids with gcd(P, Q) > 1 must be flagged without it"""
    edit = _drop(lambda ln: ln["t"] == "wait" and ln["s"] == "sc" and ln.get("tag") == "ready" and ln["e"].startswith(f"r{ln['r']}e"),
                 first_per=lambda ln: (ln["r"], ln["k"]))
    assert M.check_config(2, 2, 9, 2, COPIES, mutate=edit)[1]
    assert M.check_config(4, 2, 9, 2, COPIES, mutate=edit)[1]
    assert not M.check_config(2, 1, 9, 2, COPIES, mutate=edit)[1]  # there every rank pulls from itself and waits again


def test_checker_finds_missing_bulk_wait_and_buffer_reuse():
    edit = _drop(lambda ln: ln["t"] == "wait" and ln["s"] == "sp" and ln.get("tag") == "bulk_done")
    assert M.check_config(2, 2, 9, 2, COPIES, mutate=edit)[1]
    edit = _drop(lambda ln: ln["t"] == "wait" and ln["s"] == "sc" and ln.get("tag") in ("bulk_done", "la_done"))
    assert M.check_config(2, 2, 9, 2, COPIES, mutate=edit)[1]


def test_checker_finds_unstaged_sends_and_unfinished_lkk_image():
    """send/recv transport: without the wait for the rank's own panel in front of the group the sends read the staging image
    before the panel stream has written it; without the lkk_image / lkk_recv events the comm stream sends an L_kk image that is
    still being copied, or the panel stream solves with one that has not arrived.  (The lkk_free waits are implied by the
    arrived -> look-ahead chain: dropping them alone leaves the schedule ordered — they stay as a guard against reordering.)"""
    edit = _drop(lambda ln: ln["t"] == "wait" and ln["s"] == "sc" and ln.get("tag") == "ready")
    assert M.check_config(2, 2, 9, 2, SENDRECV, mutate=edit)[1]
    edit = _drop(lambda ln: ln["t"] == "wait" and ln.get("tag") == "lkk_image")
    assert M.check_config(2, 2, 9, 2, SENDRECV, mutate=edit)[1]
    edit = _drop(lambda ln: ln["t"] == "wait" and ln.get("tag") == "lkk_recv")
    assert M.check_config(2, 2, 9, 2, SENDRECV, mutate=edit)[1]
    edit = _drop(lambda ln: ln["t"] == "wait" and ln.get("tag") == "lkk_free")
    assert not M.check_config(2, 2, 9, 2, SENDRECV, mutate=edit)[1]


def test_checker_flags_unrecorded_events_and_size_mismatch():
    def unrecord(lines):
        return [ln for ln in lines if not (ln["t"] == "rec" and ln.get("tag") == "arrived" and ln["k"] == 3 and ln["r"] == 1)]
    problems, _ = M.check_config(2, 2, 9, 2, COPIES, mutate=unrecord)
    assert any("has not been recorded" in p for p in problems)

    def shrink(lines):
        out, done = [], False
        for ln in lines:
            if not done and ln["t"] == "recv":
                ln = dict(ln, n=ln["n"] - 1)
                done = True
            out.append(ln)
        return out
    problems, _ = M.check_config(2, 2, 5, 2, SENDRECV, mutate=shrink)
    assert any("elements" in p for p in problems)


def test_trace_arguments_are_validated(agp):
    lib = agp._lib.load()
    assert lib.gp_multi_schedule_trace(0, 1, 4, 2, 2, b"/tmp/x") < 0
    assert lib.gp_multi_schedule_trace(2, 2, 4, 5, 2, b"/tmp/x") < 0
    assert lib.gp_multi_schedule_trace(2, 2, 4, 2, 3, b"/tmp/x") < 0
    assert lib.gp_multi_schedule_trace(2, 2, 4, 2, 2, b"/nonexistent-dir/x") < 0
