"""The ordering rules of the in-process slab transport, checked as a happens-before problem (no GPU).

`wayverb_amd/csrc/comm.cpp` orders a chain of slabs on one process with events between each slab's compute stream and halo
stream: `wait_ghosts`, `exchange_faces`, `step_done`, `bulk_begin / bulk_end`.  Whether those waits are ENOUGH is a question about
a partial order, and a timing-dependent test answers it only when the timing happens to go wrong (round 3: a push that ran late
let a soft source on a slab face be added twice on the neighbour -- one random chain in 1 500).  This file restates the protocol
as data -- every launch and copy a chain enqueues for a step (`enqueue_step`, engine_single.hip.h) or a two-step pass
(`enqueue_pair_a / _b`, engine_pair.hip.h), in `wv_run_group`'s host order, with the planes of the field buffers it reads and
writes -- and checks that any two operations that touch the same plane of the same buffer, one of them writing, are ordered by
stream order and event waits.  `hipStreamWaitEvent` semantics as the runtime has them: a wait refers to the event's latest record
at the time the wait is enqueued.

It is a model, kept next to the code it restates (the rules are few); what it is good for: each rule in `Rules` can be switched
off, and the tests show which hazards that rule, and no other, closes.

Round 4: a two-step pass of a slab steps its face planes AND the planes next to them ahead of the march, and the faces' second
step runs on the HALO stream between the two exchanges (engine_pair.hip.h, `slab_early_now`): both exchanges are then under the
march.  The model has the planes that takes (`next_lo` / `next_hi`), the launches on the halo stream, the per-slab fallback to
the older order when a source lies within two planes of a cut -- neighbours need not agree -- and `Rules.early` switches
the whole chain back to the older order.
"""
import itertools
from dataclasses import dataclass, field

import pytest


@dataclass
class Rules:
    per_buffer_ghost_events: bool = True   # wait_ghosts: the neighbours' pushes into the buffer about to be read (else: their latest pushes)
    previous_step_by_parity: bool = True   # exchange_faces: the neighbour's PREVIOUS step is over (else: its latest enqueued step)
    own_pushes: bool = True                # wait_ghosts: this slab's own pushes have read its face planes
    war_on_ghosts: bool = True             # exchange_faces waits for the neighbour's step end at all
    early: bool = True                     # two-step passes: faces + next planes first, the faces' t+2 on the halo stream (round 4)
    halo_waits_for_ghosts: bool = True     # ... whose launches wait for the neighbours' pushes of t+1 (wait_ghosts on the halo stream)


@dataclass
class Op:
    stream: tuple
    name: str
    reads: frozenset = frozenset()
    writes: frozenset = frozenset()
    waits: list = field(default_factory=list)      # indices of ops (event records) this op waits for
    index: int = -1


class Chain:
    """Streams ("S", k) and ("H", k) of slab k; events are names -> index of the op that last recorded them."""

    def __init__(self, n, rules, source=None):
        self.n, self.rules, self.source = n, rules, source     # source: (slab, plane name) of the OWNER, or None
        self.ops = []
        self.events = {}
        self.pending_waits = {}
        self.steps_done = [0] * n
        self.last_own_push = [None] * n
        self.last_bulk = None
        self.early_now = [False] * n

    # --- stream primitives -----------------------------------------------------------------------------------
    def op(self, stream, name, reads=(), writes=()):
        o = Op(stream, name, frozenset(reads), frozenset(writes), self.pending_waits.pop(stream, []), len(self.ops))
        self.ops.append(o)
        return o.index

    def record(self, stream, event):
        self.events[event] = self.op(stream, "record " + str(event))

    def wait(self, stream, event):
        if event in self.events:                       # the latest record at enqueue time
            self.pending_waits.setdefault(stream, []).append(self.events[event])

    def wait_op(self, stream, op_index):
        if op_index is not None:
            self.pending_waits.setdefault(stream, []).append(op_index)

    # --- what a slab holds -------------------------------------------------------------------------------------
    def planes(self, k, buf, names):
        return {(k, buf, p) for p in names}

    def lo(self, k):
        return k - 1 if k > 0 else None

    def hi(self, k):
        return k + 1 if k + 1 < self.n else None

    def source_planes(self, k):
        """Planes of slab k that hold the source node: the owner's plane, and the ghost copy next door (slab.py)."""
        if self.source is None:
            return []
        owner, plane = self.source
        if k == owner:
            return [plane]
        if plane == "face_lo" and k == owner - 1:
            return ["ghost_hi"]
        if plane == "face_hi" and k == owner + 1:
            return ["ghost_lo"]
        return []

    def early_pass(self, k):
        """slab_early_now(): the round-4 order unless a source (or its ghost copy) lies in planes g, f, n of this slab."""
        if not self.rules.early:
            return False
        return not any(p != "inner" for p in self.source_planes(k))

    # --- comm.cpp ----------------------------------------------------------------------------------------------
    def wait_ghosts(self, k, buf, stream=None):
        S = stream or ("S", k)
        for nb, ev in ((self.lo(k), "pushed_hi"), (self.hi(k), "pushed_lo")):
            if nb is None:
                continue
            if self.rules.per_buffer_ghost_events:
                self.wait(S, (ev, nb, buf))
            else:
                self.wait(S, (ev + "_latest", nb))
        if self.rules.own_pushes:
            self.wait_op(S, self.last_own_push[k])

    def exchange_faces(self, k, buf, on_halo=False):
        S, H = ("S", k), ("H", k)
        if not on_halo:                                    # (faces produced on the halo stream itself: stream order)
            self.record(S, ("faces_ready", k))
            self.wait(H, ("faces_ready", k))
        c = self.steps_done[k]
        for nb, mine, theirs, ev in ((self.lo(k), "face_lo", "ghost_hi", "pushed_lo"), (self.hi(k), "face_hi", "ghost_lo", "pushed_hi")):
            if nb is None:
                continue
            if self.rules.war_on_ghosts:
                if self.rules.previous_step_by_parity:
                    if c > 0:
                        self.wait(H, ("step_done", nb, (c - 1) & 1))
                else:
                    self.wait(H, ("step_done_latest", nb))
            i = self.op(H, "push %s of buffer %d to slab %d" % (mine, buf, nb), reads=self.planes(k, buf, [mine]), writes=self.planes(nb, buf, [theirs]))
            self.events[(ev, k, buf)] = i
            self.events[(ev + "_latest", k)] = i
            self.last_own_push[k] = i

    def step_done(self, k):
        i = self.op(("S", k), "record step_done")
        self.events[("step_done", k, self.steps_done[k] & 1)] = i
        self.events[("step_done_latest", k)] = i
        self.steps_done[k] += 1

    def bulk(self, k, name, reads, writes):
        S = ("S", k)
        self.wait_op(S, self.last_bulk if self.last_bulk is not None and self.ops[self.last_bulk].stream != S else None)
        self.last_bulk = self.op(S, name, reads, writes)

    # --- engine_single.hip.h: enqueue_step ---------------------------------------------------------------------------
    OWNED = ["face_lo", "next_lo", "inner", "next_hi", "face_hi"]
    ALL = ["ghost_lo"] + OWNED + ["ghost_hi"]

    def enqueue_step(self, k, cur, nxt):
        S = ("S", k)
        self.wait_ghosts(k, cur)
        src = self.source_planes(k)
        if src:
            self.op(S, "source sample into `current`", reads=self.planes(k, cur, src), writes=self.planes(k, cur, src))
        faces = [p for p, nb in (("face_lo", self.lo(k)), ("face_hi", self.hi(k))) if nb is not None]
        rest = [p for p in self.OWNED if p not in faces]
        self.op(S, "faces: sweep + boundary nodes", reads=self.planes(k, cur, self.ALL) | self.planes(k, nxt, faces), writes=self.planes(k, nxt, faces))
        yield nxt, False                                    # exchange_faces(nxt): the driver below (transports differ)
        self.bulk(k, "interior sweep", reads=self.planes(k, cur, self.OWNED) | self.planes(k, nxt, rest), writes=self.planes(k, nxt, rest))
        self.op(S, "interior boundary nodes", reads=self.planes(k, cur, self.OWNED) | self.planes(k, nxt, rest), writes=self.planes(k, nxt, rest))
        self.step_done(k)

    # --- engine_pair.hip.h: enqueue_pair_a / _b ----------------------------------------------------------------------
    def sides(self, k):
        return [side for side, nb in (("lo", self.lo(k)), ("hi", self.hi(k))) if nb is not None]

    def enqueue_pair_a(self, k, a, b, o1, o2):
        S = ("S", k)
        early = self.early_pass(k)
        self.early_now[k] = early
        self.wait_ghosts(k, b)
        src = self.source_planes(k)
        if src:
            self.op(S, "source sample into t", reads=self.planes(k, b, src), writes=self.planes(k, b, src))
        faces = ["face_" + side for side in self.sides(k)]
        first = faces + (["next_" + side for side in self.sides(k)] if early else [])       # stepped ahead of the march
        rest = [p for p in self.OWNED if p not in first]                                     # t+1 by the march + the big boundary launch
        marched = [p for p in self.OWNED if p not in faces]                                  # t+2 by the march + the big boundary launch
        self.op(S, "faces%s to t+1" % (" and the planes next to them" if early else ""),
                reads=self.planes(k, b, self.ALL) | self.planes(k, a, first), writes=self.planes(k, o1, first))
        yield o1, False
        self.bulk(k, "march", reads=self.planes(k, b, self.ALL) | self.planes(k, a, self.OWNED), writes=self.planes(k, o1, rest) | self.planes(k, o2, marched))
        self.op(S, "boundary nodes to t+1", reads=self.planes(k, b, self.OWNED) | self.planes(k, a, rest), writes=self.planes(k, o1, rest))

    def enqueue_pair_b(self, k, a, b, o1, o2):
        S, H = ("S", k), ("H", k)
        src = self.source_planes(k)
        faces = ["face_" + side for side in self.sides(k)]
        marched = [p for p in self.OWNED if p not in faces]
        if self.early_now[k]:
            # halo stream, behind exchange #1: t+1 ghosts in place -> the faces' second step -> exchange #2
            near = faces + ["ghost_" + side for side in self.sides(k)] + ["next_" + side for side in self.sides(k)]
            if self.rules.halo_waits_for_ghosts:
                self.wait_ghosts(k, o1, stream=H)
            self.op(H, "faces to t+2 (halo stream)", reads=self.planes(k, o1, near) | self.planes(k, b, faces), writes=self.planes(k, o2, faces))
            yield o2, True
            if src:                                         # (a slab with a source or receivers looks at the t+1 ghosts)
                self.wait_ghosts(k, o1)
                self.op(S, "source sample into t+1", reads=self.planes(k, o1, src), writes=self.planes(k, o1, src))
        else:
            self.wait_ghosts(k, o1)
            if src:
                self.op(S, "source sample into t+1", reads=self.planes(k, o1, src), writes=self.planes(k, o1, src))
            self.op(S, "faces to t+2", reads=self.planes(k, o1, self.ALL) | self.planes(k, b, faces), writes=self.planes(k, o2, faces))
            yield o2, False
        self.op(S, "fix-up list + boundary nodes to t+2", reads=self.planes(k, o1, self.OWNED) | self.planes(k, b, marched), writes=self.planes(k, o2, marched))
        self.step_done(k)

    # --- engine_triple.hip.h: enqueue_triple_slab (round 6) -------------------------------------------------------------
    # Three steps per pass, three exchanges.  o1 is the engine's FIFTH field (the t+1 field; the communicator does not know it): its
    # faces travel in the face / ghost planes of the t+3 field o3, one plane-sized copy either side of exchange 1.  The march covers
    # "inner" (t+1 at its shell nodes, t+3) and stores t+2 on the planes next to the faces as well; faces and next planes take
    # plain steps.  Every level's source sample comes behind the wait for that level's ghosts.
    def enqueue_triple_part(self, k, part, a, b, o1, o2, o3):
        S = ("S", k)
        src = self.source_planes(k)
        faces = ["face_" + side for side in self.sides(k)]
        ghosts = ["ghost_" + side for side in self.sides(k)]
        nexts = ["next_" + side for side in self.sides(k)]
        first = faces + nexts
        inner = [p for p in self.OWNED if p not in first]
        if part == 0:
            self.wait_ghosts(k, b)
            if src:
                self.op(S, "source sample into t", reads=self.planes(k, b, src), writes=self.planes(k, b, src))
            self.op(S, "faces and the planes next to them to t+1", reads=self.planes(k, b, self.ALL) | self.planes(k, a, first), writes=self.planes(k, o1, first))
            if faces:
                self.op(S, "t+1 faces into the t+3 field's face planes", reads=self.planes(k, o1, faces), writes=self.planes(k, o3, faces))
            self.bulk(k, "three-step march", reads=self.planes(k, b, self.ALL) | self.planes(k, a, self.OWNED),
                      writes=self.planes(k, o1, inner) | self.planes(k, o2, inner + nexts) | self.planes(k, o3, inner))
            yield o3, False                                  # exchange 1, behind the march
            self.op(S, "boundary nodes of the march's planes to t+1", reads=self.planes(k, b, self.OWNED) | self.planes(k, a, inner), writes=self.planes(k, o1, inner))
        elif part == 1:
            self.wait_ghosts(k, o3)
            if ghosts:
                self.op(S, "the neighbours' t+1 faces out of the t+3 field's ghost planes", reads=self.planes(k, o3, ghosts), writes=self.planes(k, o1, ghosts))
            if src:
                self.op(S, "source sample into t+1", reads=self.planes(k, o1, src), writes=self.planes(k, o1, src))
            self.op(S, "faces to t+2", reads=self.planes(k, o1, ghosts + faces + nexts) | self.planes(k, b, faces), writes=self.planes(k, o2, faces))
            yield o2, False                                  # exchange 2
            self.op(S, "second level's list + boundary nodes from the next planes on to t+2", reads=self.planes(k, o1, self.OWNED) | self.planes(k, b, inner + nexts),
                    writes=self.planes(k, o2, inner + nexts))
        else:
            self.wait_ghosts(k, o2)
            if src:
                self.op(S, "source sample into t+2", reads=self.planes(k, o2, src), writes=self.planes(k, o2, src))
            self.op(S, "faces and the planes next to them to t+3", reads=self.planes(k, o2, self.ALL) | self.planes(k, o1, first), writes=self.planes(k, o3, first))
            yield o3, False                                  # exchange 3
            self.op(S, "third level's list + boundary nodes of the march's planes to t+3", reads=self.planes(k, o2, self.OWNED) | self.planes(k, o1, inner),
                    writes=self.planes(k, o3, inner))
            self.step_done(k)

    # --- engine_slab.hip.h: group_run ----------------------------------------------------------------------------------
    def enqueue_all(self, make):
        """One step (or half a pass) of every slab, lockstep as wv_run_group enqueues them: slab after slab."""
        for k in range(self.n):
            for buf, on_halo in make(k):
                self.exchange_faces(k, buf, on_halo)

    def run(self, kinds):
        """kinds: a sequence of "step" / "pass" / "triple" (a three-step pass); every slab takes the same ones."""
        prv, cur, spare = 0, 1, [2, 3]
        for kind in kinds:
            if kind == "triple":
                for part in range(3):
                    self.enqueue_all(lambda k: self.enqueue_triple_part(k, part, prv, cur, 4, spare[0], spare[1]))
                prv, cur, spare = spare[0], spare[1], [prv, cur]
            elif kind == "step":
                self.enqueue_all(lambda k: self.enqueue_step(k, cur, prv))          # in place: the next field goes where `previous` was
                prv, cur = cur, prv
            else:
                self.enqueue_all(lambda k: self.enqueue_pair_a(k, prv, cur, spare[0], spare[1]))
                self.enqueue_all(lambda k: self.enqueue_pair_b(k, prv, cur, spare[0], spare[1]))
                prv, cur, spare = spare[0], spare[1], [prv, cur]
        return self


class RcclChain(Chain):
    """The same engine code over the RCCL transport: one process per slab, grouped ncclSend / ncclRecv on the halo stream
    (comm.cpp, the non-local branch of exchange_faces), "ghosts ready" recorded behind them.  Ranks enqueue independently;
    any interleaving that keeps each rank's own order is a valid host order, this one goes phase by phase.  A send / receive
    pair is a rendezvous: what follows either side's group follows what preceded the other side's."""

    def __init__(self, n, source=None, wait_for_ghosts=True, early=True):
        super().__init__(n, Rules(early=early), source)
        self.wait_for_ghosts = wait_for_ghosts
        self.exchanges = 0

    def wait_ghosts(self, k, buf, stream=None):
        if self.wait_for_ghosts:
            self.wait(stream or ("S", k), ("ghosts_ready", k))

    def step_done(self, k):
        pass

    def bulk(self, k, name, reads, writes):
        self.op(("S", k), name, reads, writes)              # (one GPU per rank: nobody to take turns with)

    def enqueue_all(self, make):
        runs = [make(k) for k in range(self.n)]
        yielded = [next(r) for r in runs]                   # every rank up to its exchange
        bufs = [y[0] for y in yielded]
        starts = []
        for k in range(self.n):
            if not yielded[k][1]:                           # (faces stepped on the halo stream itself need no event)
                self.record(("S", k), ("faces_ready", k))
                self.wait(("H", k), ("faces_ready", k))
            starts.append(self.op(("H", k), "ncclGroupStart"))
        self.exchanges += 1
        meet = {}
        for k in range(self.n - 1):
            self.pending_waits[("R", k, self.exchanges)] = [starts[k], starts[k + 1]]
            meet[k] = self.op(("R", k, self.exchanges), "send / receive pair of ranks %d and %d meet" % (k, k + 1))
        for k in range(self.n):
            faces = [p for p, nb in (("face_lo", self.lo(k)), ("face_hi", self.hi(k))) if nb is not None]
            ghosts = [p for p, nb in (("ghost_lo", self.lo(k)), ("ghost_hi", self.hi(k))) if nb is not None]
            for pair in (k - 1, k):
                if pair in meet:
                    self.wait_op(("H", k), meet[pair])
            self.op(("H", k), "sends and receives of buffer %d" % bufs[k], reads=self.planes(k, bufs[k], faces), writes=self.planes(k, bufs[k], ghosts))
            self.record(("H", k), ("ghosts_ready", k))
        for r in runs:
            for _ in r:
                raise AssertionError("one exchange per step or half pass")


class IpcChain(Chain):
    """The engine code over the IPC transport (wv_options::transport = WV_TRANSPORT_IPC, comm.cpp): one process per slab like
    RCCL, but the planes travel by copies into the neighbour's own (IPC-mapped) fields and the ranks order themselves with
    COUNTERS in per-rank mailboxes instead of events -- a wait names the number it waits for, so it may be enqueued before the
    thing it waits for exists (events cannot say that across processes):
      wait_ghosts(buf)     the neighbours' copy number `pushes[buf]` of that buffer has landed (each rank takes the same steps,
                           so a neighbour's latest exchange of a buffer has the number of this rank's own), + this rank's own
                           latest copies have read its face planes (a local event);
      exchange_faces(buf)  the neighbours have finished step number `steps_done` (they no longer read the ghost planes about to
                           be overwritten); copy; post the copy's number into the neighbour's mailbox;
      step_done            post the step's number into both neighbours' mailboxes.
    Ranks enqueue independently; this model goes phase by phase like RcclChain, which keeps every waited-for operation ahead of
    its waiter in the list (the check below needs that), and `lag` lets one rank's host run whole phases behind the others."""

    def __init__(self, n, source=None, early=True, counters=True, own_pushes=True, war=True):
        super().__init__(n, Rules(early=early, own_pushes=own_pushes), source)
        self.counters, self.war = counters, war
        self.pushes = [[0] * 4 for _ in range(n)]             # exchanges of buffer b issued by rank k
        self.push_op = {}                                       # (rank, side, buffer, number) -> op
        self.done_op = {}                                       # (rank, number) -> op

    def wait_ghosts(self, k, buf, stream=None):
        S = stream or ("S", k)
        number = self.pushes[k][buf]
        if number:
            for nb, side in ((self.lo(k), "hi"), (self.hi(k), "lo")):
                if nb is None:
                    continue
                # with counters: exactly that copy; without (what an event would give across processes): whatever copy of the
                # buffer the neighbour happens to have enqueued so far, here: the one before
                want = number if self.counters else number - 1
                if want:
                    self.wait_op(S, self.push_op[(nb, side, buf, want)])
        if self.rules.own_pushes:
            self.wait_op(S, self.last_own_push[k])

    def exchange_faces(self, k, buf, on_halo=False):
        S, H = ("S", k), ("H", k)
        if not on_halo:
            self.record(S, ("faces_ready", k))
            self.wait(H, ("faces_ready", k))
        self.pushes[k][buf] += 1
        number = self.pushes[k][buf]
        c = self.steps_done[k]
        for nb, mine, theirs, side in ((self.lo(k), "face_lo", "ghost_hi", "lo"), (self.hi(k), "face_hi", "ghost_lo", "hi")):
            if nb is None:
                continue
            if self.war and c > 0:
                self.wait_op(H, self.done_op[(nb, c)])
            i = self.op(H, "push %s of buffer %d to slab %d" % (mine, buf, nb), reads=self.planes(k, buf, [mine]), writes=self.planes(nb, buf, [theirs]))
            self.push_op[(k, side, buf, number)] = i
            self.last_own_push[k] = i

    def step_done(self, k):
        self.steps_done[k] += 1
        self.done_op[(k, self.steps_done[k])] = self.op(("S", k), "post step %d done" % self.steps_done[k])

    def bulk(self, k, name, reads, writes):
        self.op(("S", k), name, reads, writes)

    def enqueue_all(self, make):
        runs = [make(k) for k in range(self.n)]
        yielded = [next(r) for r in runs]                   # every rank up to its exchange
        for k in range(self.n):
            self.exchange_faces(k, yielded[k][0], yielded[k][1])
        for r in runs:
            for _ in r:
                raise AssertionError("one exchange per step or half pass")


def unordered_conflicts(chain):
    """Pairs of operations that touch the same plane of the same buffer, at least one writing, with no happens-before path
    between them (stream order + event waits, transitively)."""
    ops = chain.ops
    n = len(ops)
    before = [0] * n                       # bitset of the ops that happen before op i
    last_in_stream = {}
    for o in ops:                          # ops are in host enqueue order: every edge points backwards
        mask = 0
        prev = last_in_stream.get(o.stream)
        for p in ([prev] if prev is not None else []) + o.waits:
            mask |= before[p] | (1 << p)
        before[o.index] = mask
        last_in_stream[o.stream] = o.index
    touched = {}
    for o in ops:
        for r in o.reads | o.writes:
            touched.setdefault(r, []).append(o.index)
    found = []
    for r, users in touched.items():
        for i, j in itertools.combinations(users, 2):
            if r not in ops[i].writes and r not in ops[j].writes:
                continue
            if not (before[j] >> i) & 1:                   # (i < j in host order: only i -> j is possible)
                found.append((r, ops[i].name, ops[i].stream, ops[j].name, ops[j].stream))
    return found


SEQUENCES = [["step"] * 6, ["pass"] * 5, ["step", "step", "pass", "pass", "step", "pass", "pass", "pass", "step", "step"],
             ["pass", "step", "pass", "step", "step", "pass", "pass"]]
SOURCES = [None, (1, "face_lo"), (1, "face_hi"), (0, "face_hi"), (1, "inner"), (1, "next_lo"), (0, "next_hi")]


CHAINS = [(n, src) for n in (2, 3, 4) for src in SOURCES if not (src and src[1].endswith("_hi") and src[0] + 1 >= n)]   # (the last slab has no upper face)


@pytest.mark.parametrize("early", [True, False], ids=["both-exchanges-under-the-march", "round-3-order"])
@pytest.mark.parametrize("n,source", CHAINS, ids=str)
@pytest.mark.parametrize("kinds", SEQUENCES, ids=lambda s: "".join(k[0] for k in s))
def test_the_transport_orders_every_conflicting_access(n, kinds, source, early):
    chain = Chain(n, Rules(early=early), source).run(kinds)
    bad = unordered_conflicts(chain)
    assert not bad, bad[:3]
    if early and "pass" in kinds:                          # (the order under test is the one that ran)
        near_cut = source is not None and source[1] != "inner"
        assert any("halo stream" in o.name for o in chain.ops) or (near_cut and n == 2)


TRIPLES = [["triple"] * 5, ["step", "step", "triple", "triple", "pass", "step", "triple", "pass", "triple", "triple", "step"],
           ["triple", "pass", "triple", "step", "step", "triple", "triple", "pass", "pass", "triple"]]


@pytest.mark.parametrize("transport", ["in-process", "rccl", "ipc"])
@pytest.mark.parametrize("n,source", CHAINS, ids=str)
@pytest.mark.parametrize("kinds", TRIPLES, ids=lambda s: "".join(k[0] for k in s))
def test_three_step_passes_of_slabs_order_every_conflicting_access(n, kinds, source, transport):
    """Round 6 (engine_triple.hip.h, enqueue_triple_slab): three exchanges per pass -- the t+1 faces by way of the t+3 field's face and
    ghost planes, which the SAME pass's third exchange overwrites -- mixed with two-step passes and single steps, the source anywhere
    (on a face, in the neighbour's ghost copy, next to them), over all three transports, whose code is unchanged."""
    for early in (True, False):
        chain = {"in-process": lambda: Chain(n, Rules(early=early), source), "rccl": lambda: RcclChain(n, source, early=early),
                 "ipc": lambda: IpcChain(n, source, early=early)}[transport]().run(kinds)
        bad = unordered_conflicts(chain)
        assert not bad, bad[:3]
        assert any("three-step march" in o.name for o in chain.ops)


def test_what_orders_the_t1_faces_detour_through_the_t3_field():
    """The ghost plane of the t+3 field holds the neighbour's t+1 face from exchange 1 until this slab has copied it out (part 1), and
    the neighbour's t+3 face from exchange 3 on: nothing but causality orders the two -- the neighbour's exchange 3 follows its wait for
    THIS slab's exchange 2, which follows the copy in stream order.  Without waits for ghosts at all the detour races (and so does
    everything else); without the wait for its own pushes a source on a face still races with the push of the plane it goes into."""
    assert not unordered_conflicts(Chain(3, Rules(), None).run(["triple"] * 4))
    assert unordered_conflicts(RcclChain(3, None, wait_for_ghosts=False).run(["triple"] * 4))
    late = unordered_conflicts(IpcChain(3, None, counters=False).run(["triple"] * 4))
    assert late and any("t+3 field's ghost planes" in a or "t+3 field's ghost planes" in b for _, a, _, b, _ in late), late[:3]
    assert unordered_conflicts(Chain(3, Rules(own_pushes=False), (1, "face_lo")).run(["triple"] * 4))


@pytest.mark.parametrize("kinds", [k for k in SEQUENCES if "pass" in k], ids=lambda s: "".join(k[0] for k in s))
def test_a_source_near_a_cut_sends_only_the_slabs_that_see_it_back_to_the_older_order(kinds):
    """The fallback is per slab: with the source on the upper face of slab 1 of 4, slabs 1 and 2 (owner and holder of the ghost
    copy) keep round 3's order, slabs 0 and 3 step their faces on the halo stream -- and nothing is unordered."""
    chain = Chain(4, Rules(), (1, "face_hi")).run(kinds)
    assert not unordered_conflicts(chain)
    on_halo = {o.stream[1] for o in chain.ops if "halo stream" in o.name}
    assert on_halo == {0, 3}
    chain = Chain(4, Rules(), (1, "next_lo")).run(kinds)     # one plane in from the face: the owner alone
    assert not unordered_conflicts(chain)
    assert {o.stream[1] for o in chain.ops if "halo stream" in o.name} == {0, 2, 3}


@pytest.mark.parametrize("kinds", [k for k in SEQUENCES if "pass" in k], ids=lambda s: "".join(k[0] for k in s))
def test_the_faces_second_step_on_the_halo_stream_must_wait_for_the_neighbours_pushes(kinds):
    """What the halo stream's own wait_ghosts is for: without it a slab's face would be stepped to t+2 from a ghost plane its
    neighbour has not pushed yet (the push runs on the NEIGHBOUR's halo stream).  Every hazard found is that one."""
    bad = unordered_conflicts(Chain(3, Rules(halo_waits_for_ghosts=False), None).run(kinds))
    assert bad and all(("push" in a and "halo stream" in b) or ("halo stream" in a and "push" in b) for _, a, _, b, _ in bad), bad[:3]


@pytest.mark.parametrize("kinds", SEQUENCES, ids=lambda s: "".join(k[0] for k in s))
def test_without_the_wait_for_its_own_pushes_a_source_on_a_face_races_with_the_push(kinds):
    """Round 3's race, found by its symptom on the GPU first: the only hazards this rule closes are between a slab's push of a
    face plane and the next source sample into that plane -- and with no source on a face there is none (the neighbours'
    waits order everything else)."""
    bad = unordered_conflicts(Chain(3, Rules(own_pushes=False, early=False), (1, "face_hi")).run(kinds))
    assert bad and all("push face_hi" in a and "source sample" in b for _, a, _, b, _ in bad), bad[:3]
    assert not unordered_conflicts(Chain(3, Rules(own_pushes=False, early=False), None).run(kinds))
    assert not unordered_conflicts(Chain(3, Rules(own_pushes=False, early=False), (1, "inner")).run(kinds))


@pytest.mark.parametrize("kinds", [k for k in SEQUENCES if "pass" in k], ids=lambda s: "".join(k[0] for k in s))
def test_with_the_faces_stepped_on_the_halo_stream_that_wait_is_what_joins_the_two_streams(kinds):
    """Round 4: the halo stream now carries launches that write this slab's own planes (the faces' t+2), not only copies out of
    them.  The compute stream's wait for the slab's latest push -- the last thing on the halo stream in a pass -- is then what
    orders the NEXT pass behind them, source or no source: without it the faces' second step races with everything that reads or
    rewrites those planes afterwards, and every hazard involves the halo stream's launch or the push behind it."""
    bad = unordered_conflicts(Chain(3, Rules(own_pushes=False), None).run(kinds))
    assert bad and all("halo stream" in a or "halo stream" in b or "push" in a or "push" in b for _, a, _, b, _ in bad), bad[:3]


@pytest.mark.parametrize("kinds", SEQUENCES, ids=lambda s: "".join(k[0] for k in s))
def test_the_wider_waits_of_round_2_were_safe_too(kinds):
    """Waiting for the neighbours' LATEST pushes and latest step ends (what the transport did until round 3) orders at least as
    much -- it put the slabs of a chain one behind the other, that was all that was wrong with it."""
    for source in (None, (1, "face_lo")):
        rules = Rules(per_buffer_ghost_events=False, previous_step_by_parity=False)
        assert not unordered_conflicts(Chain(3, rules, source).run(kinds))


@pytest.mark.parametrize("kinds", SEQUENCES, ids=lambda s: "".join(k[0] for k in s))
def test_the_wait_for_the_neighbours_previous_step_is_implied(kinds):
    """exchange_faces makes a push wait for the end of the neighbour's previous step (it overwrites a ghost plane the neighbour
    read then).  The other waits already imply it: a slab pushes only after it has waited for its neighbours' pushes of the same
    step or half pass, and those follow everything the neighbour enqueued before.  The explicit wait stays in comm.cpp as a
    safety net that costs nothing; this test records that it is one."""
    for source in SOURCES[:3]:
        assert not unordered_conflicts(Chain(3, Rules(war_on_ghosts=False), source).run(kinds))


@pytest.mark.parametrize("n,source", CHAINS, ids=str)
@pytest.mark.parametrize("kinds", SEQUENCES, ids=lambda s: "".join(k[0] for k in s))
def test_the_rccl_transport_orders_every_conflicting_access(n, kinds, source):
    """The path no box here could run between GPUs: ghosts_ready, recorded on the halo stream behind a rank's sends and
    receives, is the one event its compute stream waits for -- it covers the receives having landed AND the sends having read
    the face planes (the race of the in-process transport does not exist here)."""
    for early in (True, False):
        assert not unordered_conflicts(RcclChain(n, source, early=early).run(kinds))
        assert unordered_conflicts(RcclChain(n, source, wait_for_ghosts=False, early=early).run(kinds))


@pytest.mark.parametrize("n,source", CHAINS, ids=str)
@pytest.mark.parametrize("kinds", SEQUENCES, ids=lambda s: "".join(k[0] for k in s))
def test_the_ipc_transport_orders_every_conflicting_access(n, kinds, source):
    """Round 5: copies into the neighbours' IPC-mapped fields ordered by mailbox counters (comm.cpp, the ipc_ branches).  Every
    conflicting pair of accesses is ordered, in both orders of a pass; what each ingredient is for shows when it is left out:
    a wait that cannot name its number (an event's "latest record" seen from another process) lets a rank read ghost planes the
    neighbour has not filled yet; without the wait for its own copies the faces' second step / a source on a face races with them."""
    for early in (True, False):
        assert not unordered_conflicts(IpcChain(n, source, early=early).run(kinds))
        late = unordered_conflicts(IpcChain(n, source, early=early, counters=False).run(kinds))
        assert late and any("push" in a or "push" in b for _, a, _, b, _ in late), late[:3]
    if "pass" in kinds:
        assert unordered_conflicts(IpcChain(n, source, own_pushes=False).run(kinds))


@pytest.mark.parametrize("kinds", SEQUENCES, ids=lambda s: "".join(k[0] for k in s))
def test_the_ipc_transports_wait_for_the_neighbours_step_end_is_a_safety_net_too(kinds):
    """As in the in-process transport: a rank copies into a neighbour's ghost plane only after it has seen that neighbour's copy
    of the same exchange, which follows everything the neighbour read before."""
    for source in SOURCES[:3]:
        assert not unordered_conflicts(IpcChain(3, source, war=False).run(kinds))
