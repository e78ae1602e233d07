"""Happens-before check of the multi-device schedule — on the schedule the LIBRARY emits, not on a re-model of it.

abstractgps.jl_amd/csrc/multi.hip routes every stream operation, event record, event wait and point-to-point transfer of a
rank thread through one layer (RankRun) that can write them as JSON lines with their block footprints:

  gp_multi_schedule_trace(P, Q, nblk, depth, comm, path)   the rank threads run fit_rank's control flow WITHOUT a device
                                                           (this is what runs on a machine without a GPU)
  GPMI_TRACE_SCHEDULE=<file>                               the same lines from a real fit on the GPU

This tool rebuilds the dependency graph from such a trace — stream order, event record -> wait edges, i-th send of a rank
pair -> i-th receive of the peer — and reports
  * every pair of operations that touches the same block (local matrix block, operand-buffer slot, L_kk image, staging image,
    partial-sum / alpha block), at least one writing, with no path between them                     -> "race"
  * waits on events that were never recorded, send / receive sequences of a rank pair that differ in length or size
                                                                                                      -> "protocol"
The host-level progress of the rank threads (generation-numbered events) is checked by the trace run itself: a thread that
could block forever fails gp_multi_schedule_trace after its time-out.

  python tools/multi_schedule_check.py                 # all grids of the test-suite × block counts × depths × transports
  python tools/multi_schedule_check.py trace.jsonl     # one recorded trace (e.g. from a GPU run)
"""
import itertools
import json
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
GRIDS = [(1, 1), (2, 1), (1, 2), (2, 2), (3, 1), (4, 1), (2, 3), (4, 2), (2, 4), (8, 1), (1, 3)]


def nlb_before(k, p, P):  # number of global blocks i <= k with i ≡ p (mod P)
    return (k - p) // P + 1 if k >= p else 0


def emit_trace(P, Q, nblk, depth, comm, path, solve=False):
    """run the library's rank threads without a device (no GPU needed; the .so must be built); solve: a pass over the distributed
    factor instead of the fit — True / 0: the forward solve of the predictive variances, else the flags of gp_multi_solve_trace_ex
    (1 given right-hand sides, 2 rows kept for an extended factor (sequential update), 4 followed by two backward sweeps)"""
    sys.path.insert(0, str(ROOT))
    import abstractgps_jl_amd as agp

    lib = agp._lib.load()
    if solve is True:
        agp._lib.check(lib.gp_multi_solve_trace(P, Q, nblk, str(path).encode()))
    elif solve is not False and solve is not None:
        agp._lib.check(lib.gp_multi_solve_trace_ex(P, Q, nblk, int(solve), str(path).encode()))
    else:
        agp._lib.check(lib.gp_multi_schedule_trace(P, Q, nblk, depth, comm, str(path).encode()))


def load(path):
    lines = [json.loads(x) for x in Path(path).read_text().splitlines() if x.strip()]
    assert lines and lines[0]["t"] == "hdr", "trace without a header line"
    return lines[0], lines[1:]


def expand(fp, hdr):
    """footprint -> set of block keys (see struct Fp in multi.hip)"""
    name, r, a0, a1, b0, b1, fl = fp
    P, Q = hdr["P"], hdr["Q"]
    p, q = r // Q, r % Q
    out = set()
    if name == "A":
        rows = [] if fl & 4 else list(range(a0, a1))
        if fl & 6:
            rows.append("rhs")
        for lj in range(b0, b1):
            gj = lj * Q + q
            for li in rows:
                gi = 10**9 if li == "rhs" else li * P + p
                if (fl & 1) and gi < gj:
                    continue  # lower predicate: blocks above the global diagonal are not touched
                out.add(("A", r, li, lj))
    elif name in ("Ab", "St"):
        rows = list(range(b0, b1)) + (["rhs"] if fl & 2 else [])
        for s in range(a0, a1):
            for li in rows:
                out.add((name, r, s, li))
    elif name == "Bb":
        for s in range(a0, a1):
            for lj in range(b0, b1):
                out.add(("Bb", r, s, lj))
    elif name == "A2":  # the rank's piece of an extended factor (sequential update): local block rows [a0, a1) × local block columns [b0, b1)
        for li in range(a0, a1):
            for lj in range(b0, b1):
                out.add(("A2", r, li, lj))
    elif name == "Lkk":
        out.add(("Lkk", r))
    elif name in ("acc", "alb", "tmp", "Wi", "ACC", "X", "Xb", "T", "Vs", "Cv"):  # (Wi: the −inv(L_kk) slots of a diagonal owner; the last six: buffers of the forward solve on the distributed factor)
        for i in range(b0, b1):
            out.add((name, r, i))
    else:
        raise ValueError(f"unknown buffer {name}")
    return out


class Graph:
    def __init__(self):
        self.label, self.rd, self.wr, self.preds = [], [], [], []

    def add(self, label, rd, wr, preds):
        self.label.append(label)
        self.rd.append(frozenset(rd))
        self.wr.append(frozenset(wr))
        self.preds.append(list(preds))
        return len(self.label) - 1


def build(hdr, lines):
    """-> (graph, protocol findings)"""
    g = Graph()
    problems = []
    last, pending, group = {}, {}, {}
    events = {}
    sends, recvs = {}, {}

    def new_node(key, label, rd=(), wr=(), in_group=False):
        preds = list(pending.get(key, []))
        if in_group and key in group:  # members of a group are unordered among themselves: they all start from the group's entry state
            preds += group[key]["entry"]
            idx = g.add(label, rd, wr, preds)
            group[key]["members"].append(idx)
            return idx
        if key in last:
            preds.append(last[key])
        pending[key] = []
        idx = g.add(label, rd, wr, preds)
        last[key] = idx
        return idx

    for ln in lines:
        t = ln["t"]
        key = (ln["r"], ln["s"])
        name = f"r{ln['r']}.{ln['s']}"
        if t == "op":
            rd = set().union(*[expand(f, hdr) for f in ln["R"]]) if ln["R"] else set()
            wr = set().union(*[expand(f, hdr) for f in ln["W"]]) if ln["W"] else set()
            new_node(key, f"{name}:{ln['n']}({ln['k'][0]},{ln['k'][1]})", rd, wr)
        elif t == "rec":
            if pending.get(key) or key not in last:
                new_node(key, f"{name}:marker")
            events[ln["e"]] = last[key]
        elif t == "wait":
            if ln["e"] not in events:
                problems.append(f"{name} waits for event {ln['e']} that has not been recorded")
            else:
                pending.setdefault(key, []).append(events[ln["e"]])
        elif t == "grp":
            if ln["b"]:  # everything before the group (stream order and pending waits) precedes every member
                begin = new_node(key, f"{name}:group_begin")
                group[key] = {"entry": [begin], "members": []}
            else:
                gr = group.pop(key)
                preds = gr["members"] + [last[key]] + list(pending.get(key, []))
                pending[key] = []
                last[key] = g.add(f"{name}:group_end", (), (), preds)
        elif t == "send":
            rd = set().union(*[expand(f, hdr) for f in ln["R"]])
            idx = new_node(key, f"{name}:send->{ln['to']}", rd, (), in_group=True)
            sends.setdefault((ln["r"], ln["to"]), []).append((idx, ln["n"]))
        elif t == "recv":
            wr = set().union(*[expand(f, hdr) for f in ln["W"]])
            idx = new_node(key, f"{name}:recv<-{ln['from']}", (), wr, in_group=True)
            recvs.setdefault((ln["from"], ln["r"]), []).append((idx, ln["n"]))
        else:
            raise ValueError(f"unknown trace line {ln}")
    for k in group:
        problems.append(f"group on {k} never closed")
    for pair in sorted(set(sends) | set(recvs)):
        ss, rs = sends.get(pair, []), recvs.get(pair, [])
        if len(ss) != len(rs):
            problems.append(f"rank pair {pair[0]}->{pair[1]}: {len(ss)} sends but {len(rs)} receives")
        for i, ((si, sn), (ri, rn)) in enumerate(zip(ss, rs)):
            if sn != rn:
                problems.append(f"rank pair {pair[0]}->{pair[1]}: transfer {i} sends {sn} elements, receives {rn}")
            g.preds[ri].append(si)  # the data is there after the send has started; the receive also follows its own stream order
    return g, problems


def races(g):
    n = len(g.label)
    succ = [[] for _ in range(n)]
    indeg = [0] * n
    for i, ps in enumerate(g.preds):
        for p in ps:
            succ[p].append(i)
            indeg[i] += 1
    order, stack = [], [i for i in range(n) if indeg[i] == 0]
    while stack:
        i = stack.pop()
        order.append(i)
        for j in succ[i]:
            indeg[j] -= 1
            if indeg[j] == 0:
                stack.append(j)
    if len(order) != n:
        return [("cycle", "the dependency graph has a cycle (send/receive order of two ranks is inconsistent)", "")]
    reach = [0] * n
    for i in order:
        b = 0
        for p in g.preds[i]:
            b |= reach[p] | (1 << p)
        reach[i] = b
    by_loc = {}
    for i in range(n):
        for loc in g.rd[i] | g.wr[i]:
            by_loc.setdefault(loc, []).append(i)
    out = []
    for loc, idxs in by_loc.items():
        for a, b in itertools.combinations(idxs, 2):
            if not (loc in g.wr[a] or loc in g.wr[b]):
                continue
            if not ((reach[b] >> a) & 1 or (reach[a] >> b) & 1):
                out.append((loc, g.label[a], g.label[b]))
    return out


def check_trace(path):
    hdr, lines = load(path)
    g, problems = build(hdr, lines)
    return hdr, problems, races(g)


def check_config(P, Q, nblk, depth, comm, mutate=None, solve=False):
    """trace the library's schedule for one configuration and check it; mutate(lines) -> lines edits the trace first (tests)"""
    with tempfile.TemporaryDirectory() as td:
        path = Path(td) / "trace.jsonl"
        emit_trace(P, Q, nblk, depth, comm, path, solve=solve)
        hdr, lines = load(path)
    if mutate:
        lines = mutate(lines)
    g, problems = build(hdr, lines)
    return problems, races(g)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        hdr, problems, rs = check_trace(sys.argv[1])
        print(f"trace {sys.argv[1]}: grid {hdr['P']}x{hdr['Q']} nblk {hdr['nblk']} depth {hdr['depth']} comm {hdr['comm']} "
              f"({'dry run' if hdr.get('dry') else 'device run'}): {len(problems)} protocol findings, {len(rs)} unordered conflicting pairs")
        for p in problems[:10]:
            print("   ", p)
        for r in rs[:10]:
            print("   ", r)
        sys.exit(1 if problems or rs else 0)
    bad = 0
    for comm, (P, Q) in itertools.product((2, 1), GRIDS):
        for nblk in (1, 2, 3, 5, 9, 17):
            for depth in (1, 2, 3):
                problems, rs = check_config(P, Q, nblk, depth, comm)
                if problems or rs:
                    bad += 1
                    print(f"{'send/recv' if comm == 1 else 'copies'} grid {P}x{Q} nblk {nblk} depth {depth}: {len(problems)} protocol findings, "
                          f"{len(rs)} unordered conflicting pairs, e.g.")
                    for x in (problems + rs)[:6]:
                        print("   ", x)
    # passes over the distributed factor: predictive variances (forward solve), C \\ B (given right-hand sides, forward + backward
    # sweeps), the sequential update (rows kept for the extended factor)
    for (P, Q), flags in itertools.product(GRIDS, (True, 1 | 4, 2, 1 | 2 | 4)):
        for nblk in (1, 2, 3, 5, 9, 17):
            problems, rs = check_config(P, Q, nblk, 0, 2, solve=flags)
            if problems or rs:
                bad += 1
                print(f"solve (flags {flags}) grid {P}x{Q} nblk {nblk}: {len(problems)} protocol findings, {len(rs)} unordered conflicting pairs, e.g.")
                for x in (problems + rs)[:6]:
                    print("   ", x)
    print("configurations with findings:", bad)
    sys.exit(1 if bad else 0)
