"""A model of the peer-window push PROTOCOL (csrc/halo_wait.h, csrc/window.hip) run under random interleavings on the CPU.

What hardware the build can reach has ONE GPU, so the transport has only ever run between ranks that share a device; whether a
store over xGMI becomes visible when the recipe says it does is a hardware question no CPU test can answer.  What CAN be checked
without hardware is the protocol's LOGIC at the target's rank count: epochs, flag lines, ack lines and the double-buffered ghost
segment, with 8 ranks arriving at arbitrary times.  The model keeps exactly the rules of the device code:

  step e of rank r (one fused launch; steps of a rank follow each other in stream order):
    * push workgroup 0 first publishes r's ACKS: to every producer p of r, ack[p <- r] = e - 1            (halo_push_block)
    * for every consumer c of r (in any order, concurrently): wait until ack[r <- c] >= e - nbuf (only when e > nbuf),
      store r's values of step e into c's ghost buffer e % nbuf, then publish flag[c <- r] = e
    * the boundary workgroups wait until flag[r <- p] >= e for every producer p, then read buffer e % nbuf  (halo_wait_block)
  and a push whose ack wait has not been satisfied stores NOTHING.

A scheduler picks one enabled micro-action at a time, uniformly at random (seeded): ranks drift apart by as many steps as the
protocol allows.  Checked: every read returns exactly the values its producer stored for THAT epoch (no buffer is overwritten
before its reader is done, no reader sees a stale buffer), and every rank finishes all its steps (no deadlock) -- for symmetric
stencil neighbourhoods, all-to-all (7 neighbours each way at 8 ranks), one-directional chains and ranks without neighbours, with
double-buffered (vectors) and single-buffered (dense ghost rows) plans.  And the model is itself checked: with the ack wait
removed (one-directional chain) or the acks published late (single buffer), the corruption / the deadlock is found.
"""
import random

import pytest


def _graphs(n):
    every = {r: [q for q in range(n) if q != r] for r in range(n)}
    slab = {r: [q for q in (r - 1, r + 1) if 0 <= q < n] for r in range(n)}
    chain = {r: ([r - 1] if r > 0 else []) for r in range(n)}            # r sends to r - 1 only (one-directional band)
    lonely = {r: ([q for q in (r - 1, r + 1) if 0 <= q < n - 1] if r < n - 1 else []) for r in range(n)}   # last rank: no neighbours
    return {"alltoall": every, "slab": slab, "chain": chain, "lonely": lonely}


class Model:
    """consumers[r] = ranks r pushes to; producers[r] = ranks that push to r."""

    def __init__(self, consumers, nbuf, steps, rng, ack_wait=True, acks_first=True):
        self.n = len(consumers)
        self.consumers = consumers
        self.producers = {r: [p for p in range(self.n) if r in consumers[p]] for r in range(self.n)}
        self.nbuf, self.steps, self.rng = nbuf, steps, rng
        self.ack_wait, self.acks_first = ack_wait, acks_first
        self.flag = {(c, p): 0 for p in range(self.n) for c in consumers[p]}       # flag[c <- p]: epoch p has completed in c's ghost
        self.ack = {(p, c): 0 for p in range(self.n) for c in consumers[p]}        # ack[p <- c]: last epoch c has finished reading
        self.ghost = {(c, p, b): None for p in range(self.n) for c in consumers[p] for b in range(nbuf)}
        self.epoch = [1] * self.n                                                   # the step a rank is in
        self.pending = [self._actions(r) for r in range(self.n)]
        self.errors = []

    def _actions(self, r):
        e = self.epoch[r]
        if e > self.steps:
            return []
        acts = [("push", c) for c in self.consumers[r]] + [("read", p) for p in self.producers[r]]
        self.rng.shuffle(acts)
        return ([("acks", None)] if self.acks_first else []) + acts + ([] if self.acks_first else [("acks", None)])

    def _enabled(self, r, act):
        kind, other = act
        e = self.epoch[r]
        if kind == "acks":
            return self.acks_first or len(self.pending[r]) == 1     # (mutation: published only once the step's work is done)
        if self.acks_first and ("acks", None) in self.pending[r]:
            return False                                  # workgroup 0 publishes the acks before any wait of the step begins
        if kind == "push":
            return (not self.ack_wait) or e <= self.nbuf or self.ack[(r, other)] >= e - self.nbuf
        return self.flag[(r, other)] >= e                 # read

    def _run(self, r, act):
        kind, other = act
        e = self.epoch[r]
        if kind == "acks":
            for p in self.producers[r]:
                self.ack[(p, r)] = e - 1
        elif kind == "push":
            self.ghost[(other, r, e % self.nbuf)] = (r, e)
            self.flag[(other, r)] = e
        else:
            got = self.ghost[(r, other, e % self.nbuf)]
            if got != (other, e):
                self.errors.append(f"rank {r} step {e}: ghost segment of rank {other} holds {got}")

    def run(self, max_ticks=2_000_000):
        for _ in range(max_ticks):
            ready = [(r, a) for r in range(self.n) for a in self.pending[r] if self._enabled(r, a)]
            if not ready:
                break
            r, a = self.rng.choice(ready)
            self._run(r, a)
            self.pending[r].remove(a)
            if not self.pending[r]:                       # the launch is over: the next step of this rank may begin
                self.epoch[r] += 1
                self.pending[r] = self._actions(r)
        return all(e > self.steps for e in self.epoch), self.errors


@pytest.mark.parametrize("graph", ["alltoall", "slab", "chain", "lonely"])
@pytest.mark.parametrize("nbuf", [2, 1])
@pytest.mark.parametrize("n", [8, 3])
def test_protocol_delivers_every_epoch_and_never_deadlocks(graph, nbuf, n):
    for seed in range(6):
        m = Model(_graphs(n)[graph], nbuf, steps=40, rng=random.Random(1000 * seed + 17 * n + nbuf))
        finished, errors = m.run()
        assert not errors, errors[:3]
        assert finished, f"deadlock: ranks stopped in steps {m.epoch} ({graph}, nbuf {nbuf}, seed {seed})"


def test_ranks_may_drift_apart_but_only_as_far_as_the_buffers_allow():
    """A producer can be ahead of its consumer by at most nbuf steps (it may fill every buffer the consumer has released)."""
    for nbuf in (1, 2):
        rng = random.Random(5 + nbuf)
        m = Model(_graphs(8)["slab"], nbuf, steps=60, rng=rng)
        worst = 0
        for _ in range(200_000):
            ready = [(r, a) for r in range(m.n) for a in m.pending[r] if m._enabled(r, a)]
            if not ready:
                break
            r, a = rng.choice(ready)
            m._run(r, a)
            m.pending[r].remove(a)
            if not m.pending[r]:
                m.epoch[r] += 1
                m.pending[r] = m._actions(r)
            worst = max(worst, max(m.epoch[p] - m.epoch[c] for p in range(m.n) for c in m.consumers[p]))
        assert not m.errors and all(e > 60 for e in m.epoch)
        assert 1 <= worst <= nbuf + 1, worst              # (+1: the producer's counter moves on when its launch ends)


@pytest.mark.parametrize("broken", ["no_ack_wait", "acks_last"])
def test_the_model_finds_the_fault_when_a_rule_is_removed(broken):
    """Mutation check of the model itself.  Without the ack wait a producer that receives nothing back (the one-directional
    chain: nothing else holds it) runs ahead and overwrites a buffer its consumer has not read.  With the acks published at
    the END of a step instead of first, single-buffered neighbours wait for each other forever (a push of step e needs the
    ack e - 1, which its consumer would only publish after reading the very data the push has not stored yet)."""
    found = False
    for seed in range(40):
        if broken == "no_ack_wait":
            m = Model(_graphs(8)["chain"], 2, steps=30, rng=random.Random(seed), ack_wait=False)
        else:
            m = Model(_graphs(8)["alltoall"], 1, steps=30, rng=random.Random(seed), acks_first=False)
        finished, errors = m.run()
        if errors or not finished:
            found = True
            break
    assert found, f"the model did not notice the removed rule ({broken})"


# ---- the scalar all-reduce through the communicator windows (csrc/window.hip window_allreduce_kernel) ---------------------------
# all-reduce e of rank r: store {value, tag = e} into slot [e % 2][r] of EVERY rank's window, then poll the nranks slots of its own
# window of that parity until every tag reads e, and sum them in rank order.  Two parities: a rank can be at most one all-reduce ahead.
def _allreduce_model(n, count, rng, parities=2):
    slots = {(w, par, r): (None, 0) for w in range(n) for par in range(parities) for r in range(n)}   # window w, parity, writer r
    epoch = [1] * n
    pending = [[("store", w) for w in range(n)] + [("read", None)] for _ in range(n)]
    errors = []
    for _ in range(2_000_000):
        ready = []
        for r in range(n):
            e = epoch[r]
            if e > count:
                continue
            stores = [a for a in pending[r] if a[0] == "store"]
            if stores:
                ready += [(r, a) for a in stores]
            elif all(slots[(r, e % parities, q)][1] >= e for q in range(n)):         # polled with >= like spin_until_ge; the value check
                                                                                     # below catches a slot overwritten by a later epoch
                ready.append((r, ("read", None)))
        if not ready:
            break
        r, a = rng.choice(ready)
        e = epoch[r]
        if a[0] == "store":
            slots[(a[1], e % parities, r)] = ((r, e), e)
            pending[r].remove(a)
        else:
            for q in range(n):
                if slots[(r, e % parities, q)][0] != (q, e):
                    errors.append(f"rank {r} all-reduce {e}: slot of rank {q} holds {slots[(r, e % parities, q)][0]}")
            epoch[r] += 1
            pending[r] = [("store", w) for w in range(n)] + [("read", None)]
    return all(e > count for e in epoch), errors


@pytest.mark.parametrize("n", [8, 5, 2])
def test_window_allreduce_two_parities_suffice(n):
    for seed in range(8):
        finished, errors = _allreduce_model(n, 50, random.Random(seed * 31 + n))
        assert finished and not errors, errors[:3]


def test_window_allreduce_one_parity_is_not_enough():
    """Mutation check: with ONE slot per writer a rank that is one all-reduce ahead overwrites a partial its peer has not read."""
    assert any(_allreduce_model(8, 50, random.Random(seed), parities=1)[1] for seed in range(20))


# ---- the plan's step counter: two-level release counting (csrc/halo_wait.h epoch_release) -------------------------------------
def test_two_level_release_counter_fires_exactly_once_on_the_last_release():
    """epoch_release: reader i bumps shard i % 64; the release that completes a shard (in_shard = readers with that residue)
    resets it and bumps the top counter; the release that completes the top (min(n_readers, 64) active shards) resets it and
    stores done = e.  For every reader count and any release order exactly ONE release may store `done`, and it must be the
    LAST one -- nobody may still need the epoch number when it moves on.  The arithmetic below is the kernel's, line by line."""
    SHARDS = 64
    rng = random.Random(3)
    for n_readers in list(range(1, 200)) + [255, 256, 257, 2048, 2049, 65536 + 7]:
        shards, top, fired = [0] * SHARDS, 0, []
        order = list(range(n_readers))
        rng.shuffle(order)
        for k, reader in enumerate(order):
            sh = reader % SHARDS
            in_shard = (n_readers - sh + SHARDS - 1) // SHARDS
            shards[sh] += 1
            if shards[sh] != in_shard:
                continue
            shards[sh] = 0
            active = n_readers if n_readers < SHARDS else SHARDS
            top += 1
            if top != active:
                continue
            top = 0
            fired.append(k)
        assert fired == [n_readers - 1], (n_readers, fired)
        assert top == 0 and not any(shards)               # the counters are back at zero for the next step
