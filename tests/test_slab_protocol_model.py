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

    # --- comm.cpp ----------------------------------------------------------------------------------------------
    def wait_ghosts(self, k, buf):
        S = ("S", k)
        for nb, ev in ((self.lo(k), "pushed_hi"), (self.hi(k), "pushed_lo")):
            if nb is None:
                continue
            if self.rules.per_buffer_ghost_events:
                self.wait(S, (ev, nb, buf))
            else:
                self.wait(S, (ev + "_latest", nb))
        if self.rules.own_pushes:
            self.wait_op(S, self.last_own_push[k])

    def exchange_faces(self, k, buf):
        S, H = ("S", k), ("H", k)
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
    OWNED = ["face_lo", "inner", "face_hi"]
    ALL = ["ghost_lo", "face_lo", "inner", "face_hi", "ghost_hi"]

    def enqueue_step(self, k, cur, nxt):
        S = ("S", k)
        self.wait_ghosts(k, cur)
        src = self.source_planes(k)
        if src:
            self.op(S, "source sample into `current`", reads=self.planes(k, cur, src), writes=self.planes(k, cur, src))
        faces = [p for p, nb in (("face_lo", self.lo(k)), ("face_hi", self.hi(k))) if nb is not None]
        rest = [p for p in self.OWNED if p not in faces]
        self.op(S, "faces: sweep + boundary nodes", reads=self.planes(k, cur, self.ALL) | self.planes(k, nxt, faces), writes=self.planes(k, nxt, faces))
        yield nxt                                           # exchange_faces(nxt): the driver below (transports differ)
        self.bulk(k, "interior sweep", reads=self.planes(k, cur, self.OWNED) | self.planes(k, nxt, rest), writes=self.planes(k, nxt, rest))
        self.op(S, "interior boundary nodes", reads=self.planes(k, cur, self.OWNED) | self.planes(k, nxt, rest), writes=self.planes(k, nxt, rest))
        self.step_done(k)

    # --- engine_pair.hip.h: enqueue_pair_a / _b ----------------------------------------------------------------------
    def enqueue_pair_a(self, k, a, b, o1, o2):
        S = ("S", k)
        self.wait_ghosts(k, b)
        src = self.source_planes(k)
        if src:
            self.op(S, "source sample into t", reads=self.planes(k, b, src), writes=self.planes(k, b, src))
        faces = [p for p, nb in (("face_lo", self.lo(k)), ("face_hi", self.hi(k))) if nb is not None]
        rest = [p for p in self.OWNED if p not in faces]
        self.op(S, "faces to t+1", reads=self.planes(k, b, self.ALL) | self.planes(k, a, faces), writes=self.planes(k, o1, faces))
        yield o1
        self.bulk(k, "march", reads=self.planes(k, b, self.ALL) | self.planes(k, a, self.OWNED), writes=self.planes(k, o1, rest) | self.planes(k, o2, rest))
        self.op(S, "boundary nodes to t+1", reads=self.planes(k, b, self.OWNED) | self.planes(k, a, rest), writes=self.planes(k, o1, rest))

    def enqueue_pair_b(self, k, a, b, o1, o2):
        S = ("S", k)
        self.wait_ghosts(k, o1)
        src = self.source_planes(k)
        if src:
            self.op(S, "source sample into t+1", reads=self.planes(k, o1, src), writes=self.planes(k, o1, src))
        faces = [p for p, nb in (("face_lo", self.lo(k)), ("face_hi", self.hi(k))) if nb is not None]
        rest = [p for p in self.OWNED if p not in faces]
        self.op(S, "faces to t+2", reads=self.planes(k, o1, self.ALL) | self.planes(k, b, faces), writes=self.planes(k, o2, faces))
        yield o2
        self.op(S, "fix-up list + boundary nodes to t+2", reads=self.planes(k, o1, self.OWNED) | self.planes(k, b, rest), writes=self.planes(k, o2, rest))
        self.step_done(k)

    # --- engine_slab.hip.h: group_run ----------------------------------------------------------------------------------
    def enqueue_all(self, make):
        """One step (or half a pass) of every slab, lockstep as wv_run_group enqueues them: slab after slab."""
        for k in range(self.n):
            for buf in make(k):
                self.exchange_faces(k, buf)

    def run(self, kinds):
        """kinds: a sequence of "step" / "pass"; every slab takes the same ones."""
        prv, cur, spare = 0, 1, [2, 3]
        for kind in kinds:
            if kind == "step":
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

    def __init__(self, n, source=None, wait_for_ghosts=True):
        super().__init__(n, Rules(), source)
        self.wait_for_ghosts = wait_for_ghosts
        self.exchanges = 0

    def wait_ghosts(self, k, buf):
        if self.wait_for_ghosts:
            self.wait(("S", k), ("ghosts_ready", k))

    def step_done(self, k):
        pass

    def bulk(self, k, name, reads, writes):
        self.op(("S", k), name, reads, writes)              # (one GPU per rank: nobody to take turns with)

    def enqueue_all(self, make):
        runs = [make(k) for k in range(self.n)]
        bufs = [next(r) for r in runs]                      # every rank up to its exchange
        starts = []
        for k in range(self.n):
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
SOURCES = [None, (1, "face_lo"), (1, "face_hi"), (0, "face_hi"), (1, "inner")]


CHAINS = [(n, src) for n in (2, 3, 4) for src in SOURCES if not (src and src[1] == "face_hi" and src[0] + 1 >= n)]   # (the last slab has no upper face)


@pytest.mark.parametrize("n,source", CHAINS, ids=str)
@pytest.mark.parametrize("kinds", SEQUENCES, ids=lambda s: "".join(k[0] for k in s))
def test_the_transport_orders_every_conflicting_access(n, kinds, source):
    bad = unordered_conflicts(Chain(n, Rules(), source).run(kinds))
    assert not bad, bad[:3]


@pytest.mark.parametrize("kinds", SEQUENCES, ids=lambda s: "".join(k[0] for k in s))
def test_without_the_wait_for_its_own_pushes_a_source_on_a_face_races_with_the_push(kinds):
    """Round 3's race, found by its symptom on the GPU first: the only hazards this rule closes are between a slab's push of a
    face plane and the next source sample into that plane -- and with no source on a face there is none (the neighbours'
    waits order everything else)."""
    bad = unordered_conflicts(Chain(3, Rules(own_pushes=False), (1, "face_hi")).run(kinds))
    assert bad and all("push face_hi" in a and "source sample" in b for _, a, _, b, _ in bad), bad[:3]
    assert not unordered_conflicts(Chain(3, Rules(own_pushes=False), None).run(kinds))
    assert not unordered_conflicts(Chain(3, Rules(own_pushes=False), (1, "inner")).run(kinds))


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
    assert not unordered_conflicts(RcclChain(n, source).run(kinds))
    assert unordered_conflicts(RcclChain(n, source, wait_for_ghosts=False).run(kinds))
