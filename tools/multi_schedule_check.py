"""Happens-before check of the multi-device schedule (abstractgps.jl_amd/csrc/multi.hip, fit_rank; both transports).

fit_rank (factorisation + backward block sweep) is replayed symbolically: every stream operation of every rank becomes a node with the
set of BLOCKS it reads and writes (local matrix blocks, operand-buffer slots, the L_kk image), stream order and event waits
become edges (own events and the cross-thread generation-numbered events alike).  Two operations that touch the same block,
at least one writing, must be ordered by a path in that graph; anything else is reported as a race.  This checks the
DEPENDENCY STRUCTURE the host code builds — not the kernels and not the HIP runtime.

  python tools/multi_schedule_check.py            # all grids of the test-suite × block counts × look-ahead depths
"""
import itertools
import sys


def nlb_before(k, p, P):  # number of global blocks i <= k with i ≡ p (mod P)
    return (k - p) // P + 1 if k >= p else 0


class Graph:
    def __init__(self):
        self.nodes = []      # (label, reads, writes)
        self.preds = []      # list of predecessor index lists

    def add(self, label, reads, writes, preds):
        self.nodes.append((label, frozenset(reads), frozenset(writes)))
        self.preds.append(list(preds))
        return len(self.nodes) - 1


class Stream:
    """in-order queue: every new node depends on the previous one and on the pending waits"""

    def __init__(self, g, name):
        self.g, self.name, self.last, self.pending = g, name, None, []

    def wait(self, node):  # hipStreamWaitEvent on an event recorded after `node`
        if node is not None:
            self.pending.append(node)

    def op(self, label, reads=(), writes=()):
        preds = list(self.pending)
        if self.last is not None:
            preds.append(self.last)
        self.pending = []
        self.last = self.g.add(f"{self.name}:{label}", reads, writes, preds)
        return self.last

    def record(self):  # hipEventRecord: the event completes after everything enqueued so far (including pending waits)
        if self.pending:
            self.op("marker")
        return self.last


def build(P, Q, nblk, depth, rccl=False):
    g = Graph()
    R = P * Q
    NBUF = depth + 1
    ranks = [(r // Q, r % Q) for r in range(R)]
    sm = [Stream(g, f"r{r}.sm") for r in range(R)]
    sp = [Stream(g, f"r{r}.sp") for r in range(R)]
    sc = [Stream(g, f"r{r}.sc") for r in range(R)]
    nlb_r = [nlb_before(nblk - 1, p, P) for (p, q) in ranks]
    nlb_c = [nlb_before(nblk - 1, q, Q) for (p, q) in ranks]

    def rank_of(pp, qq):
        return pp * Q + qq

    def rows_ge(r, gblk):  # local row blocks of rank r with global block >= gblk, + the RHS block on process row 0
        p = ranks[r][0]
        first = nlb_before(gblk - 1, p, P)
        out = list(range(first, nlb_r[r]))
        if p == 0:
            out.append("rhs")
        return out

    def A(r, li, lj):
        return ("A", r, li, lj)

    ready = [[None] * nblk for _ in range(R)]
    lkk = [[None] * nblk for _ in range(R)]
    arrived = [[None] * nblk for _ in range(R)]
    bulk_done = [[None] * nblk for _ in range(R)]
    la_done = [[None] * nblk for _ in range(R)]
    lkk_free = [None] * R      # RCCL: my L_kk image may be overwritten again after this event
    pending_send = {}          # RCCL: (src rank, dst rank, tag) -> send node, consumed by the matching receive

    # assembly + buffer clears on the main stream, then sp / sc wait for it
    for r in range(R):
        w = [A(r, li, lj) for li in list(range(nlb_r[r])) + (["rhs"] if ranks[r][0] == 0 else []) for lj in range(nlb_c[r])]
        w += [("Ab", r, s, li) for s in range(NBUF) for li in list(range(nlb_r[r])) + ["rhs"]]
        w += [("Bb", r, s, lj) for s in range(NBUF) for lj in range(nlb_c[r])]
        w += [("Lkk", r)] + [("acc", r, lj) for lj in range(nlb_c[r])]
        w += [("St", r, s, li) for s in range(NBUF) for li in list(range(nlb_r[r])) + ["rhs"]]
        sm[r].op("assemble", writes=w)
        ev = sm[r].record()
        sp[r].wait(ev)
        sc[r].wait(ev)

    def a_operand(r, i, rowblocks):
        p, q = ranks[r]
        if q == i % Q:
            return [A(r, li, i // Q) for li in rowblocks]
        return [("Ab", r, i % NBUF, li) for li in rowblocks]

    def update(stream, r, i, gr0, lcols, label):
        p, q = ranks[r]
        rb = rows_ge(r, gr0)
        if not rb or not lcols:
            return
        reads = a_operand(r, i, rb) + [("Bb", r, i % NBUF, lj) for lj in lcols]
        cw = []
        for lj in lcols:
            gj = lj * Q + q
            for li in rb:
                gi = 10**9 if li == "rhs" else li * P + p
                if gi >= gj:  # lower predicate (tiles above the global diagonal are skipped)
                    cw.append(A(r, li, lj))
        stream.op(label, reads=reads + cw, writes=cw)

    def panel(r, k):
        p, q = ranks[r]
        pk, qk = k % P, k % Q
        if q != qk:
            return
        c0 = k // Q
        if P == 1:
            blocks = [A(r, li, c0) for li in rows_ge(r, k)]
            sp[r].op(f"potrf({k})", reads=blocks, writes=blocks)
        else:
            if p == pk:
                d = A(r, k // P, c0)
                sp[r].op(f"potrf_diag({k})", reads=[d], writes=[d])
                lread = [d]
                if rccl:  # contiguous image for the sends, all RCCL calls of a rank on its comm stream
                    sp[r].wait(lkk_free[r])
                    sp[r].op(f"lkk_image({k})", reads=[d], writes=[("Lkk", r)])
                    sc[r].wait(sp[r].record())
                    for pp in range(P):
                        if pp != pk:
                            pending_send[(r, rank_of(pp, qk), ("lkk", k))] = sc[r].op(f"send('lkk', {k})->{rank_of(pp, qk)}", reads=[("Lkk", r)])
                    lkk_free[r] = sc[r].record()
                lkk[r][k] = sp[r].record()
            else:
                own = rank_of(pk, qk)
                if rccl:
                    sc[r].wait(lkk_free[r])
                    sc[r].wait(pending_send[(own, r, ("lkk", k))])
                    sc[r].op(f"recv('lkk', {k})<-{own}", writes=[("Lkk", r)])
                    sp[r].wait(sc[r].record())
                else:
                    sp[r].wait(lkk[own][k])
                    sp[r].op(f"pull_lkk({k})", reads=[A(own, k // P, c0)], writes=[("Lkk", r)])
                lread = [("Lkk", r)]
            rb = rows_ge(r, k + 1)
            if rb:
                blocks = [A(r, li, c0) for li in rb]
                sp[r].op(f"trsm({k})", reads=blocks + lread, writes=blocks)
            if rccl and p != pk:
                lkk_free[r] = sp[r].record()
        if rccl:  # contiguous image of my piece of the panel for the sends of exchange(k)
            rb = rows_ge(r, k + 1)
            if rb:
                sp[r].op(f"stage({k})", reads=[A(r, li, c0) for li in rb], writes=[("St", r, k % NBUF, li) for li in rb])
        ready[r][k] = sp[r].record()

    def exchange(r, k):
        p, q = ranks[r]
        qk = k % Q
        s = k % NBUF
        if k - NBUF >= 0:
            sc[r].wait(bulk_done[r][k - NBUF])
            sc[r].wait(la_done[r][k - NBUF])
        if q == qk:
            sc[r].wait(ready[r][k])
        if not rccl:
            if q != qk:
                src = rank_of(p, qk)
                rb = rows_ge(r, k + 1)
                if rb:
                    sc[r].wait(ready[src][k])
                    sc[r].op(f"pullA({k})", reads=[A(src, li, k // Q) for li in rb], writes=[("Ab", r, s, li) for li in rb])
            for pp in range(P):
                src = rank_of(pp, qk)
                waited = False
                for lj in range(nlb_before(k, q, Q), nlb_c[r]):
                    gj = lj * Q + q
                    if gj % P != pp:
                        continue
                    if not waited:
                        sc[r].wait(ready[src][k])
                        waited = True
                    sc[r].op(f"pullB({k},{lj})", reads=[A(src, gj // P, k // Q)], writes=[("Bb", r, s, lj)])
        else:
            return ("rccl", r, k, s)   # sends / receives are generated for all ranks together (matched pairs)
        arrived[r][k] = sc[r].record()

    def la_update(r, j, i):
        p, q = ranks[r]
        if q != j % Q:
            return
        sp[r].wait(arrived[r][i])
        first = max(0, j - depth)
        if i == first and first - 1 >= 0 and bulk_done[r][first - 1] is not None:
            sp[r].wait(bulk_done[r][first - 1])
        update(sp[r], r, i, j, [j // Q], f"la({j},{i})")

    def exchange_all(k):
        """exchange(k) of every rank.  RCCL: the same transfers as matched send / receive pairs, one group per rank, both sides
        enumerating (source process row, destination rank, block) in the same order; a receive completes after its send started."""
        if not rccl:
            for r in range(R):
                exchange(r, k)
            return
        qk, s = k % Q, k % NBUF
        for r in range(R):   # the waits in front of the group
            p, q = ranks[r]
            if k - NBUF >= 0:
                sc[r].wait(bulk_done[r][k - NBUF])
                sc[r].wait(la_done[r][k - NBUF])
            if q == qk:
                sc[r].wait(ready[r][k])
        sends, recvs = [], []
        for pp in range(P):
            src = rank_of(pp, qk)
            rbs = rows_ge(src, k + 1)
            for dp in range(P):
                for dq in range(Q):
                    dst = rank_of(dp, dq)
                    if dp == pp and dq != qk and rbs:   # A part
                        sends.append((src, dst, ("A", k), [("St", src, s, li) for li in rbs]))
                        recvs.append((src, dst, ("A", k), [("Ab", dst, s, li) for li in rbs]))
                    for lj in range(nlb_before(k, dq, Q), nlb_c[dst]):   # B part
                        gj = lj * Q + dq
                        if gj % P != pp:
                            continue
                        if src == dst:
                            sc[src].op(f"selfB({k},{lj})", reads=[("St", src, s, gj // P)], writes=[("Bb", src, s, lj)])
                        else:
                            sends.append((src, dst, ("B", k, lj), [("St", src, s, gj // P)]))
                            recvs.append((src, dst, ("B", k, lj), [("Bb", dst, s, lj)]))
        for src, dst, tag, rd in sends:
            pending_send[(src, dst, tag)] = sc[src].op(f"send{tag}->{dst}", reads=rd)
        for src, dst, tag, wr in recvs:
            sc[dst].wait(pending_send[(src, dst, tag)])
            sc[dst].op(f"recv{tag}<-{src}", writes=wr)
        for r in range(R):
            arrived[r][k] = sc[r].record()

    # the host loops of every rank interleave arbitrarily; the graph only needs each rank's own program order, plus the
    # cross-rank events, which must exist before they are waited for -> build in rounds of k, owners before consumers
    def step(fn, *a):
        for r in range(R):
            fn(r, *a)

    for r in sorted(range(R), key=lambda r: 0 if ranks[r][0] == 0 else 1):
        panel(r, 0)
    exchange_all(0)
    for k in range(nblk):
        if k + 1 < nblk:
            step(la_update, k + 1, k)
            # diagonal owners publish lkk before their column peers wait for it
            for r in sorted(range(R), key=lambda r: 0 if ranks[r][0] == (k + 1) % P else 1):
                panel(r, k + 1)
            exchange_all(k + 1)
            for j in range(k + 2, min(k + depth, nblk - 1) + 1):
                step(la_update, j, k)
        for r in range(R):
            la_done[r][k] = sp[r].record()
            sm[r].wait(arrived[r][k])
            gfirst = k + depth + 1
            if gfirst < nblk:
                q = ranks[r][1]
                update(sm[r], r, k, gfirst, list(range(nlb_before(gfirst - 1, q, Q), nlb_c[r])), f"bulk({k})")
            bulk_done[r][k] = sm[r].record()
    # join: the main stream continues after everything on the panel and comm streams
    for r in range(R):
        sm[r].wait(sp[r].record())
        sm[r].wait(sc[r].record())
        if ranks[r][0] == 0:
            sm[r].op("rowsumsq", reads=[A(r, "rhs", lj) for lj in range(nlb_c[r])])
    # backward block sweep alpha = L^-T z on the main streams (cross-thread events accr / alr)
    accr = [[None] * nblk for _ in range(R)]
    alr = [[None] * nblk for _ in range(R)]
    for k in range(nblk - 1, -1, -1):
        pk, qk = k % P, k % Q
        # publication order inside one k: column peers publish accr, then the diagonal owner consumes them and publishes alr
        order = sorted(range(R), key=lambda r: (ranks[r][1] != qk, ranks[r][0] == pk))
        for r in order:
            p, q = ranks[r]
            c0 = k // Q
            if q == qk:
                if p == 0:
                    sm[r].op(f"addz({k})", reads=[A(r, "rhs", c0), ("acc", r, c0)], writes=[("acc", r, c0)])
                if p == pk:
                    sm[r].op(f"ak({k})", reads=[("acc", r, c0)], writes=[("alb", r, k)])
                    for pp in range(P):
                        if pp == pk:
                            continue
                        src = rank_of(pp, qk)
                        sm[r].wait(accr[src][k])
                        sm[r].op(f"pull_acc({k},{pp})", reads=[("acc", src, c0)], writes=[("tmp", r, pp)])
                        sm[r].op(f"add_acc({k},{pp})", reads=[("tmp", r, pp), ("alb", r, k)], writes=[("alb", r, k)])
                    sm[r].op(f"trsv({k})", reads=[A(r, k // P, c0), ("alb", r, k)], writes=[("alb", r, k)])
                    alr[r][k] = sm[r].record()
                else:
                    accr[r][k] = sm[r].record()
        for r in range(R):
            p, q = ranks[r]
            if p == pk and k > 0:
                ncb = nlb_before(k - 1, q, Q)
                if ncb > 0:
                    if q == qk:
                        ak = ("alb", r, k)
                    else:
                        own = rank_of(pk, qk)
                        sm[r].wait(alr[own][k])
                        sm[r].op(f"pull_alpha({k})", reads=[("alb", own, k)], writes=[("tmp", r, P)])
                        ak = ("tmp", r, P)
                    accs = [("acc", r, lj) for lj in range(ncb)]
                    sm[r].op(f"gemv({k})", reads=[A(r, k // P, lj) for lj in range(ncb)] + [ak] + accs, writes=accs)
    return g


def races(g):
    n = len(g.nodes)
    reach = [0] * n  # bitset of ancestors (nodes are created in a topological order: preds have smaller indices)
    for i in range(n):
        b = 0
        for pidx in g.preds[i]:
            b |= reach[pidx] | (1 << pidx)
        reach[i] = b
    by_loc = {}
    for i, (lab, rd, wr) in enumerate(g.nodes):
        for loc in rd | wr:
            by_loc.setdefault(loc, []).append(i)
    out = []
    for loc, idxs in by_loc.items():
        for a, b in itertools.combinations(idxs, 2):
            wa, wb = loc in g.nodes[a][2], loc in g.nodes[b][2]
            if not (wa or wb):
                continue
            if not ((reach[b] >> a) & 1 or (reach[a] >> b) & 1):
                out.append((loc, g.nodes[a][0], g.nodes[b][0]))
    return out


def rccl_pair_order(g, R):
    """RCCL matching rule: between any two ranks the sends of one and the receives of the other must be posted in the same
    order (point-to-point operations of a pair match by posting order).  Returns the pairs whose sequences differ."""
    import re

    seq_s, seq_r = {}, {}
    for lab, _, _ in g.nodes:
        m = re.match(r"r(\d+)\.sc:send(.*)->(\d+)$", lab)
        if m:
            seq_s.setdefault((int(m.group(1)), int(m.group(3))), []).append(m.group(2))
        m = re.match(r"r(\d+)\.sc:recv(.*)<-(\d+)$", lab)
        if m:
            seq_r.setdefault((int(m.group(3)), int(m.group(1))), []).append(m.group(2))
    return [k for k in set(seq_s) | set(seq_r) if seq_s.get(k) != seq_r.get(k)]


def host_program(P, Q, nblk, depth, r):
    """The cross-thread events one rank thread publishes / awaits on the HOST, in the program order of fit_rank (copy transport):
    ('pub', kind, rank, k) after the hipEventRecord, ('await', kind, rank, k) = spin until that rank has published."""
    p, q = r // Q, r % Q
    nlb_c = nlb_before(nblk - 1, q, Q)
    ev = []

    def panel(k):
        pk, qk = k % P, k % Q
        if q != qk:
            return
        if P > 1:
            if p == pk:
                ev.append(("pub", "lkk", r, k))
            else:
                ev.append(("await", "lkk", pk * Q + qk, k))
        ev.append(("pub", "ready", r, k))

    def exchange(k):
        qk = k % Q
        if q != qk:
            ev.append(("await", "ready", p * Q + qk, k))
        for pp in range(P):
            if any((lj * Q + q) % P == pp for lj in range(nlb_before(k, q, Q), nlb_c)):
                ev.append(("await", "ready", pp * Q + qk, k))

    panel(0)
    exchange(0)
    for k in range(nblk):
        if k + 1 < nblk:
            panel(k + 1)
            exchange(k + 1)
    for k in range(nblk - 1, -1, -1):
        pk, qk = k % P, k % Q
        if q == qk:
            if p == pk:
                for pp in range(P):
                    if pp != pk:
                        ev.append(("await", "accr", pp * Q + qk, k))
                ev.append(("pub", "alr", r, k))
            else:
                ev.append(("pub", "accr", r, k))
        if p == pk and k > 0 and nlb_before(k - 1, q, Q) > 0 and q != qk:
            ev.append(("await", "alr", pk * Q + qk, k))
    return ev


def host_deadlock(P, Q, nblk, depth):
    """run the rank threads' host programs to completion: returns the blocked (rank, event) list if they cannot all finish"""
    R = P * Q
    prog = [host_program(P, Q, nblk, depth, r) for r in range(R)]
    pc = [0] * R
    done = set()
    progress = True
    while progress:
        progress = False
        for r in range(R):
            while pc[r] < len(prog[r]):
                kind, what, who, k = prog[r][pc[r]]
                if kind == "pub":
                    done.add((what, who, k))
                elif (what, who, k) not in done:
                    break
                pc[r] += 1
                progress = True
    return [(r, prog[r][pc[r]]) for r in range(R) if pc[r] < len(prog[r])]


GRIDS = [(1, 1), (2, 1), (1, 2), (2, 2), (3, 1), (4, 1), (2, 3), (4, 2), (2, 4), (8, 1), (1, 3)]

if __name__ == "__main__":
    bad = 0
    for rccl, (P, Q) in itertools.product((False, True), GRIDS):
        for nblk in (1, 2, 3, 5, 9, 17):
            for depth in (1, 2, 3):
                gr = build(P, Q, nblk, depth, rccl=rccl)
                rs = races(gr)
                if rccl and rccl_pair_order(gr, P * Q):
                    bad += 1
                    print(f"rccl grid {P}x{Q} nblk {nblk} depth {depth}: send / receive posting order differs for", rccl_pair_order(gr, P * Q)[:4])
                if not rccl and host_deadlock(P, Q, nblk, depth):
                    bad += 1
                    print(f"grid {P}x{Q} nblk {nblk} depth {depth}: rank threads block on", host_deadlock(P, Q, nblk, depth)[:4])
                if rs:
                    bad += 1
                    print(f"{'rccl' if rccl else 'copies'} grid {P}x{Q} nblk {nblk} depth {depth}: {len(rs)} unordered conflicting pairs, e.g.")
                    for r in rs[:6]:
                        print("   ", r)
    print("configurations with races:", bad)
    sys.exit(1 if bad else 0)
